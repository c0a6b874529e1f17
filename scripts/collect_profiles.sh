#!/bin/bash
# GPU side (run through gpurun): the evidence profiles/ is built from.  Every rocprofv3 pass is separate (kernel trace +
# stats; FETCH_SIZE; WRITE_SIZE; two SQ passes) and runs under its own timeout.   usage: scripts/collect_profiles.sh [tag]
export R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
TAG=${1:-r02}
O=$R/gpurun_out/evidence; rm -rf $O; mkdir -p $O
B="python $R/bench.py --no-cpu-baseline --no-pmc --no-pcie"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o b --output-format csv -- $B --steps 10 --warmup 2 > $O/stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o b --output-format csv -- $B --steps 2 --warmup 1 > $O/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o b --output-format csv -- $B --steps 2 --warmup 1 > $O/write.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace -d $O/sq_a -o b --output-format csv -- $B --steps 2 --warmup 1 > $O/sq_a.log 2>&1
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR --kernel-trace -d $O/sq_b -o b --output-format csv -- $B --steps 2 --warmup 1 > $O/sq_b.log 2>&1
cd $R
python scripts/trace_timeline.py $O/stats > $O/timeline.txt
SNOWGPU_SERIAL=1 timeout 300 rocprofv3 --kernel-trace -d $O/serial -o b --output-format csv -- $B --steps 6 --warmup 2 > $O/serial.log 2>&1
python scripts/trace_timeline.py $O/serial > $O/timeline_serial.txt
timeout 600 python bench.py > $O/bench_C2.json 2> $O/bench_C2.err
for w in C2far C1 C4 C3; do timeout 400 python bench.py --workload $w --no-pmc $( [ $w = C4 ] && echo --frames 128 ) > $O/bench_$w.json 2> $O/bench_$w.err; done
timeout 900 python scripts/gpu_stream_c5.py --frames 10000 --batch 64 > $O/stream_c5.log 2>&1; cp gpurun_out/stream_c5.json $O/ 2>/dev/null
tail -c 600 $O/bench_C2.json; tail -3 $O/timeline.txt; tail -1 $O/stream_c5.log | cut -c1-400
