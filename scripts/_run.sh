export R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -2
B="timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-pcie"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]/1e9,3), round(d["ms_per_step"],3), round(d["roofline"]["avg_launch_ms"],3))'
for i in 1 2; do SNOWGPU_LISTS_LATE=1 $B 2>/dev/null | python -c "$P" late; $B 2>/dev/null | python -c "$P" first; done
for w in C2far C1; do SNOWGPU_LISTS_LATE=1 $B --workload $w 2>/dev/null | python -c "$P" late_$w; $B --workload $w 2>/dev/null | python -c "$P" first_$w; done
