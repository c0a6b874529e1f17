#!/bin/bash
# GPU side (through gpurun): A/B of environment switches on one box.  usage: scripts/ab_bench.sh "<workload args>" "ENV1=.. ENV2=.." "ENV=.." ...
# each remaining argument is one variant: a space-separated list of VAR=value (use "-" for the plain default)
export R=$GRAFT_REPO_ROOT; cd $R
WARGS=$1; shift
for v in "$@"; do
  if [ "$v" = "-" ]; then envs=""; else envs="$v"; fi
  for rep in 1 2; do
    env $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-pcie $WARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('BENCH [$v]', round(d['value']/1e9,3), 'Gpts/s', round(d['ms_per_step'],3), 'ms/step region', round(d['roofline']['avg_launch_ms'],3))"
  done
done
