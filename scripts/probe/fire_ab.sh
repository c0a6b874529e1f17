#!/bin/bash
# GPU side: C2 against C2fire (rows in firing order) on one box, then a one-step timeline of C2fire
export R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/fire; mkdir -p $O
for w in C2 C2fire C2 C2fire; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-pcie --workload $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('BENCH [$w]', round(d['value']/1e9,3), 'Gpts/s', round(d['ms_per_step'],3), 'ms/step region', round(d['roofline']['avg_launch_ms'],3))"
done
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/stats -o b --output-format csv -- python $R/bench.py --no-cpu-baseline --no-pmc --no-pcie --steps 10 --warmup 2 --workload C2fire > $O/stats.log 2>&1
cd $R; python scripts/trace_timeline.py $O/stats > $O/timeline_C2fire.txt; cat $O/timeline_C2fire.txt | tail -40
