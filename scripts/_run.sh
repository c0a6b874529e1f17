export R=$GRAFT_REPO_ROOT; cd $R
B="timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-pcie"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]/1e9,3), round(d["ms_per_step"],3), round(d["roofline"]["avg_launch_ms"],3))'
cp lidar_snow_sim_amd/_variants/libsnowgpu_kpw3.so lidar_snow_sim_amd/libsnowgpu.so
SNOWGPU_PREPASS_SPLIT=1 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "L5 or prepass or noise" 2>&1 | tail -2
for i in 1 2; do
cp lidar_snow_sim_amd/_variants/libsnowgpu_kpw2.so lidar_snow_sim_amd/libsnowgpu.so; $B 2>/dev/null | python -c "$P" kpw2
SNOWGPU_PREPASS_SPLIT=1 $B 2>/dev/null | python -c "$P" kpw2_split
cp lidar_snow_sim_amd/_variants/libsnowgpu_kpw3.so lidar_snow_sim_amd/libsnowgpu.so; $B 2>/dev/null | python -c "$P" kpw3
SNOWGPU_PREPASS_SPLIT=1 $B 2>/dev/null | python -c "$P" kpw3_split
done
