#!/bin/bash
cd "$(dirname "$0")/.."
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pipelined" 2>&1 | tail -2
run() { name=$1; shift; echo -n "$name: "; env "$@" python scripts/pcie_bench.py --reps 3 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('src %.4g (best %.4g) nosrc %.4g (best %.4g) single %.3f ms'%(d['points_per_s'],d['points_per_s_best'],d['points_per_s_without_src'],d['points_per_s_without_src_best'],d['single_frame_c_abi_ms']))"; }
run "R2M new" SNOWGPU_PIPE_ROWS=2097152
run "R2M old(16)" SNOWGPU_PIPE_KICK=16 SNOWGPU_PIPE_ROWS=2097152
run "R2M old+sync(24)" SNOWGPU_PIPE_KICK=24 SNOWGPU_PIPE_ROWS=2097152
run "R1M new" SNOWGPU_PIPE_ROWS=1048576
run "R1.5M new" SNOWGPU_PIPE_ROWS=1572864
run "R3M new" SNOWGPU_PIPE_ROWS=3145728
run "R4M new" SNOWGPU_PIPE_ROWS=4194304
