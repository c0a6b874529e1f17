#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -3
run() { name=$1; shift; echo -n "$name: "; env "$@" timeout 90 python scripts/pcie_bench.py --reps 3 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('src %.4g nosrc %.4g single %.3f ms py %.3f ms'%(d['points_per_s'],d['points_per_s_without_src'],d['single_frame_c_abi_ms'],d['single_frame_python_ms']))"; }
run R3M A=1
run R2M SNOWGPU_PIPE_ROWS=2097152
run R4M SNOWGPU_PIPE_ROWS=4194304
