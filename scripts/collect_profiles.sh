#!/bin/bash
# GPU side (run through gpurun): the evidence profiles/ is built from.  Every rocprofv3 pass is separate (kernel
# trace + stats; FETCH_SIZE; WRITE_SIZE) and runs under its own timeout.
export R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/evidence; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o b --output-format csv -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o b --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o b --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/write.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace -d $O/sq_a -o b --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/sq_a.log 2>&1
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_BRANCH --kernel-trace -d $O/sq_b -o b --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/sq_b.log 2>&1
cd $R && python scripts/make_profiles.py --traffic-only && timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -1 $O/bench_default.json | cut -c1-400; head -6 $O/stats/b_kernel_stats.csv | cut -c1-120
