export R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for F in 64 128; do
O=$R/gpurun_out/trace_$F; rm -rf $O; mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --stats -d $O -o b --output-format csv -- python $R/bench.py --frames $F --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --no-pcie > $O/log.txt 2>&1
echo "=== frames $F"; python $R/scripts/trace_timeline.py $O
done
