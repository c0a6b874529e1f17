#!/bin/bash
# GPU side (through gpurun): the whole GPU suite, the smoke check, then C2 / C2fire bench lines.  usage: scripts/gpu_check.sh [pytest args]
export R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q "$@" 2>&1 | tail -15
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -2
for w in C2 C2fire; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-pcie --workload $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('BENCH [$w]', round(d['value']/1e9,3), 'Gpts/s', round(d['ms_per_step'],3), 'ms/step region', round(d['roofline']['avg_launch_ms'],3))"
done
