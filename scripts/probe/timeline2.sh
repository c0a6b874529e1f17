#!/bin/bash
# GPU side: one-step timelines of the workloads given as arguments (default C2 C2fire)
export R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/tl; mkdir -p $O
for w in ${@:-C2 C2fire}; do
  rm -rf $O/stats_$w
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/stats_$w -o b --output-format csv -- python $R/bench.py --no-cpu-baseline --no-pmc --no-pcie --steps 10 --warmup 2 --workload $w $EXTRA > $O/stats_$w.log 2>&1
  python $R/scripts/trace_timeline.py $O/stats_$w > $O/timeline_$w.txt; echo "== $w"; tail -34 $O/timeline_$w.txt
done
