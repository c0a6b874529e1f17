#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 120 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pipeline" 2>&1 | tail -3
run() { name=$1; shift; echo -n "$name: "; env "$@" timeout 90 python scripts/pcie_bench.py --reps 3 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('src %.4g nosrc %.4g'%(d['points_per_s'],d['points_per_s_without_src']))"; }
run "L1" A=1
run "L2 R3M" SNOWGPU_PIPE_LANES=2
run "L2 R1.5M" SNOWGPU_PIPE_LANES=2 SNOWGPU_PIPE_ROWS=1572864
run "L3 R1.5M" SNOWGPU_PIPE_LANES=3 SNOWGPU_PIPE_ROWS=1572864
run "L2 R2M" SNOWGPU_PIPE_LANES=2 SNOWGPU_PIPE_ROWS=2097152
SNOWGPU_PIPE_LANES=2 SNOWGPU_PIPE_TRACE=1 timeout 60 python scripts/pcie_bench.py --reps 1 2>&1 | grep "^pipe chunk" | tail -11
