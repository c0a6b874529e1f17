# usage (GPU side): bash scripts/ab_variants.sh <rounds> <variant> [<variant> ...]   -- same-box A/B of kernel variants (scripts/build_variant.sh)
cd $GRAFT_REPO_ROOT
R=$1; shift
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-pcie $BENCH_ARGS"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]/1e9,3), round(d["ms_per_step"],3), round(d["roofline"]["avg_launch_ms"],3))'
for i in $(seq $R); do for v in "$@"; do
cp lidar_snow_sim_amd/_variants/libsnowgpu_$v.so lidar_snow_sim_amd/libsnowgpu.so
timeout 200 $B 2>/dev/null | python -c "$P" $v
done; done
if [ -n "$SERIAL" ]; then for v in "$@"; do
cp lidar_snow_sim_amd/_variants/libsnowgpu_$v.so lidar_snow_sim_amd/libsnowgpu.so
SNOWGPU_SERIAL=1 timeout 200 $B 2>/dev/null | python -c "$P" ${v}_serial
done; fi
