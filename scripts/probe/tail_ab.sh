# GPU side: the 63-entry / global-list tiers behind k_power on its stream (default) or behind the 16-entry tier (SNOWGPU_TIER_TAIL_AUX=0)
export R=$GRAFT_REPO_ROOT; cd $R
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -2
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]/1e9,3), round(d["ms_per_step"],3))'
for w in "" "--workload C2far"; do for v in 0 1 0 1; do
SNOWGPU_TIER_TAIL_AUX=$v timeout 100 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-pmc --no-pcie $w 2>/dev/null | python -c "$P" "tail_aux=$v $w"
done; done
