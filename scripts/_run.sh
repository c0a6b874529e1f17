export R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sampler.py -m gpu -q -x 2>&1 | tail -3
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-pcie"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]/1e9,3), round(d["ms_per_step"],3), round(d["roofline"]["avg_launch_ms"],3))'
for i in 1 2; do timeout 200 $B 2>/dev/null | python -c "$P" default; SNOWGPU_SPILL=1 timeout 200 $B 2>/dev/null | python -c "$P" spill; done
SNOWGPU_SERIAL=1 timeout 200 $B 2>/dev/null | python -c "$P" serial
cd /tmp && export TMPDIR=/tmp; O=$R/gpurun_out/wr; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o b --output-format csv -- python $R/bench.py --no-cpu-baseline --no-pmc --no-pcie --steps 2 --warmup 1 > $O/write.log 2>&1
python $R/scripts/pmc_summary.py $O/write --filter "k_beams<float, 4"
