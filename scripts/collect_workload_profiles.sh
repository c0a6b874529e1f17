#!/bin/bash
# GPU side (run through gpurun): kernel stats, one-step timeline and counters of the workloads other than the default one --
# the same passes scripts/collect_profiles.sh makes for C2.
#   usage: scripts/collect_workload_profiles.sh "C2far C1 C4" [extra bench args]
#   result: gpurun_out/wl/<W>/{stats,timeline.txt,fetch,write,sq}; scripts/make_workload_profiles.py turns it into profiles/<tag>_<W>_*
export R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
WL=${1:-"C2far C1 C4"}; shift
for w in $WL; do
  O=$R/gpurun_out/wl/$w; rm -rf $O; mkdir -p $O
  FR=$( [ $w = C4 ] && echo "--frames 128" )
  B="python $R/bench.py --no-cpu-baseline --no-pmc --no-pcie --workload $w $FR $@"
  timeout 200 $B --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/stats -o b --output-format csv -- $B --steps 6 --warmup 2 > $O/stats.log 2>&1
  timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o b --output-format csv -- $B --steps 2 --warmup 1 > $O/fetch.log 2>&1
  timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o b --output-format csv -- $B --steps 2 --warmup 1 > $O/write.log 2>&1
  timeout 150 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY --kernel-trace -d $O/sq -o b --output-format csv -- $B --steps 2 --warmup 1 > $O/sq.log 2>&1
  python $R/scripts/trace_timeline.py $O/stats > $O/timeline.txt 2>&1
  # the traces themselves are large: keep the summaries
  find $O -name "*kernel_trace.csv" -path "*fetch*" -delete; find $O -name "*kernel_trace.csv" -path "*write*" -delete; find $O -name "*kernel_trace.csv" -path "*sq*" -delete
  find $O/stats -name "*kernel_trace.csv" -delete
  echo "== $w"; tail -c 300 $O/bench.json; tail -2 $O/timeline.txt
done
