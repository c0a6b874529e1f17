#!/bin/bash
cd "$(dirname "$0")/.."
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3
run() { name=$1; shift; echo -n "$name: "; env "$@" python bench.py --no-pmc --no-cpu-baseline --no-pcie --steps 10 $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%.4g pts/s %.3f ms/step region %.3f ms'%(d['value'],d['ms_per_step'],d['roofline']['avg_launch_ms']))"; }
run f256 A=1
EXTRA="--frames 24" run f24 A=1
EXTRA="--frames 16" run f16 A=1
EXTRA="--frames 1" run f1 A=1
EXTRA="--workload C2far" run far A=1
python scripts/pcie_bench.py --reps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('pcie src %.4g nosrc %.4g single %.3f ms python %.3f ms'%(d['points_per_s'],d['points_per_s_without_src'],d['single_frame_c_abi_ms'],d['single_frame_python_ms']))"
