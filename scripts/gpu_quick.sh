#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; echo -n "$name: "; env "$@" timeout 90 python scripts/pcie_bench.py --reps 3 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('src %.4g nosrc %.4g'%(d['points_per_s'],d['points_per_s_without_src']))"; }
for K in 0 8 1 2 4 7; do run "kick $K" SNOWGPU_PIPE_KICK=$K; done
run "trace" SNOWGPU_PIPE_TRACE=1
