# GPU side: single-sweep latency of two library builds on one box (lidar_snow_sim_amd/_variants/libsnowgpu_{old,new}.so)
export R=$GRAFT_REPO_ROOT; cd $R
for v in old new old new; do
cp lidar_snow_sim_amd/_variants/libsnowgpu_$v.so lidar_snow_sim_amd/libsnowgpu.so
timeout 200 python scripts/pcie_bench.py --frames 32 --reps 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v single c_abi', round(d['single_frame_c_abi_ms'],4), 'min', round(d['single_frame_c_abi_min_ms'],4), 'python', round(d['single_frame_python_ms'],4), 'default', round(d['single_frame_python_default_ms'],4))"
done
cp lidar_snow_sim_amd/_variants/libsnowgpu_new.so lidar_snow_sim_amd/libsnowgpu.so
for e in "SNOWGPU_FEW=0" "SNOWGPU_FEW=2"; do env $e timeout 100 python scripts/gpu_graph.py 2>&1 | grep "F=1" | sed "s/^/[$e] /"; done
