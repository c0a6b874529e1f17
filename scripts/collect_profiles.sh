#!/bin/bash
# GPU side (run through gpurun): the evidence profiles/ is built from.  Every pass runs under its own timeout.
#   usage: scripts/collect_profiles.sh [tag]          (scripts/make_profiles.py turns gpurun_out/evidence into profiles/<tag>_*)
export R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
TAG=${1:-r05}
O=$R/gpurun_out/evidence; rm -rf $O; mkdir -p $O
B="python $R/bench.py --no-cpu-baseline --no-pmc --no-pcie"
# 1. per-kernel time summary + one-step timeline of the default command's timed loop
timeout 200 rocprofv3 --kernel-trace --stats -d $O/stats -o b --output-format csv -- $B --steps 10 --warmup 2 > $O/stats.log 2>&1
# 1b. the same with every kernel on one stream (pure kernel durations), and one sweep end to end (kernels and copies on one axis)
SNOWGPU_SERIAL=1 timeout 200 rocprofv3 --kernel-trace --stats -d $O/stats_serial -o b --output-format csv -- $B --steps 6 --warmup 2 > $O/stats_serial.log 2>&1
timeout 120 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/single -o t -- python $R/scripts/probe/single_trace.py > $O/single.log 2>&1
# 2. counter calibration: kernels of known byte counts (scripts/probe/pmc_calib.hip), one pass per counter
timeout 90 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/calib_f -o c -- $R/scripts/probe/pmc_calib > /dev/null 2>&1
timeout 90 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/calib_w -o c -- $R/scripts/probe/pmc_calib > /dev/null 2>&1
# 3. SQ counters of the per-beam kernels
timeout 150 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY --kernel-trace -d $O/sq -o b --output-format csv -- $B --steps 2 --warmup 1 > $O/sq.log 2>&1
cd $R
python scripts/pmc_calibration.py $O/calib_f $O/calib_w $O/pmc_calibration.json > $O/pmc_calibration.txt
python scripts/trace_timeline.py $O/stats > $O/timeline.txt
python scripts/trace_timeline.py $O/stats_serial > $O/timeline_serial.txt
python scripts/probe/single_timeline.py $O/single > $O/single_sweep_timeline.txt 2>/dev/null; tail -1 $O/single.log >> $O/single_sweep_timeline.txt
# 4. the bench line itself (its own FETCH_SIZE / WRITE_SIZE / SQ child passes; per-kernel byte table dumped on the way)
SNOWGPU_BENCH_PMC_DUMP=$O/pmc_fetch_write_per_kernel.csv timeout 400 python bench.py > $O/bench_C2.json 2> $O/bench_C2.err
# 5. the other workloads of BASELINE.json
for w in ${WORKLOADS:-C4 C3 C2fire}; do timeout 200 python bench.py --workload $w --no-pmc --no-cpu-baseline $( [ $w = C4 ] && echo --frames 128 ) > $O/bench_$w.json 2> $O/bench_$w.err; done
timeout 200 python bench.py --tables device --no-pmc --no-pcie > $O/bench_C2_device_tables.json 2> $O/bench_C2_device_tables.err
timeout 300 python bench.py --workload C5 --frames ${C5_FRAMES:-10000} > $O/bench_C5.json 2> $O/bench_C5.err
# 6. the pipeline's own event trace (upload / compute / download per chunk)
SNOWGPU_PIPE_TRACE=1 timeout 120 python scripts/pcie_bench.py --reps 1 --fast 2>&1 | grep "^pipe" > $O/pipeline_trace_all.txt
head -30 $O/pipeline_trace_all.txt > $O/pipeline_trace.txt; tail -30 $O/pipeline_trace_all.txt > $O/pipeline_trace_packed.txt
timeout 200 python scripts/probe/packed_trace.py 8 4 > $O/packed_threads.json 2>/dev/null
# 7. copies and kernels of one pipelined call on one time axis (the profiler slows the host: read the structure, not the times)
cd /tmp
timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/pipe_trace -o t -- python $R/scripts/pcie_bench.py --reps 1 --frames 96 > /dev/null 2>&1
cd $R
python scripts/trace_pipe.py $O/pipe_trace 8 > $O/pipeline_timeline.txt 2>/dev/null
tail -c 400 $O/bench_C2.json; tail -3 $O/timeline.txt; cut -c1-300 $O/bench_C5.json
