#!/bin/bash
# GPU side: the two orders of the received-power phase (SNOWGPU_HEAVY_TAIL=0 / 1) and the adaptive default, per workload, on one box.
#   usage: bash scripts/probe/tail_ab.sh [rounds] [workloads]
cd $GRAFT_REPO_ROOT
R=${1:-2}; WL=${2:-"C1 C2far C2"}
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], sys.argv[2], round(d["value"]/1e9,3), round(d["ms_per_step"],3))'
for w in $WL; do for i in $(seq $R); do for v in 0 1 auto; do
  if [ $v = auto ]; then unset SNOWGPU_HEAVY_TAIL; else export SNOWGPU_HEAVY_TAIL=$v; fi
  timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-pcie --workload $w 2>/dev/null | python -c "$P" $w tail=$v
done; done; done
