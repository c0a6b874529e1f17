# GPU side: SNOWGPU_ONE_FLAKE = 0 / 1 on the heavy workloads and on the single-sweep latency
export R=$GRAFT_REPO_ROOT; cd $R
bash scripts/ab_bench.sh "--workload C2far" "SNOWGPU_ONE_FLAKE=0" "SNOWGPU_ONE_FLAKE=1"
bash scripts/ab_bench.sh "--workload C4 --frames 128" "SNOWGPU_ONE_FLAKE=0" "SNOWGPU_ONE_FLAKE=1"
for v in 0 1 0 1; do
SNOWGPU_ONE_FLAKE=$v timeout 200 python scripts/pcie_bench.py --frames 64 --reps 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ONE_FLAKE=$v single c_abi', round(d['single_frame_c_abi_ms'],4), 'min', round(d['single_frame_c_abi_min_ms'],4), 'python', round(d['single_frame_python_ms'],4), 'pcie G', round(d['points_per_s']/1e9,3))"
done
