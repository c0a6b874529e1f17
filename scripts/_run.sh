cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sampler.py -m gpu -q -x 2>&1 | tail -3
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-pcie"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]/1e9,3), round(d["ms_per_step"],3), round(d["roofline"]["avg_launch_ms"],3))'
for i in 1 2; do
timeout 200 $B 2>/dev/null | python -c "$P" default
SNOWGPU_KP_QUARTERS=1 timeout 200 $B 2>/dev/null | python -c "$P" kpq1
SNOWGPU_KP_QUARTERS=3 timeout 200 $B 2>/dev/null | python -c "$P" kpq3
SNOWGPU_KP_QUARTERS=4 timeout 200 $B 2>/dev/null | python -c "$P" kpq4
SNOWGPU_PREPASS_EARLY=1 timeout 200 $B 2>/dev/null | python -c "$P" pre_early
done
