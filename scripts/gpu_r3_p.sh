#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 240 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_parity.py -x -q -m gpu -k "filed or occlusion_dicts or stream or synthetic_frame or float64 or L5_literal" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 90 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/calib_f -o c -- $R/scripts/probe/pmc_calib > /dev/null 2>&1
timeout 90 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/calib_w -o c -- $R/scripts/probe/pmc_calib > /dev/null 2>&1
cd $R
python scripts/pmc_calibration.py gpurun_out/calib_f gpurun_out/calib_w gpurun_out/pmc_calibration.json
run() { name=$1; shift; echo -n "$name: "; env "$@" timeout 90 python scripts/pcie_bench.py --reps 3 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('src %.4g nosrc %.4g single %.3f ms py %.3f ms'%(d['points_per_s'],d['points_per_s_without_src'],d['single_frame_c_abi_ms'],d['single_frame_python_ms']))"; }
run base A=1
run serial SNOWGPU_SERIAL=1
timeout 200 python bench.py --workload C5 --frames 4096 2>/dev/null | cut -c1-900
