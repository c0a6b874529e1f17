#!/bin/bash
# GPU side (run through gpurun): the evidence profiles/ is built from.  Every rocprofv3 pass is separate (kernel
# trace + stats; FETCH_SIZE; WRITE_SIZE) and runs under its own timeout.
export R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/evidence; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o b --output-format csv -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o b --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o b --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/write.log 2>&1
cd $R && python scripts/make_profiles.py --traffic-only && timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -1 $O/bench_default.json | cut -c1-400; head -6 $O/stats/b_kernel_stats.csv | cut -c1-120
