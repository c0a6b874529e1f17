export R=$GRAFT_REPO_ROOT; cd $R
cp lidar_snow_sim_amd/_variants/libsnowgpu_f3o4.so lidar_snow_sim_amd/libsnowgpu.so
bash scripts/ab_bench.sh "" "SNOWGPU_FEW=2" "SNOWGPU_FEW=3"
bash scripts/ab_bench.sh "--workload C2far" "SNOWGPU_FEW=2" "SNOWGPU_FEW=3"
bash scripts/ab_bench.sh "--workload C4 --frames 128" "SNOWGPU_FEW=2" "SNOWGPU_FEW=3"
bash scripts/ab_bench.sh "--workload C1" "SNOWGPU_FEW=0" "SNOWGPU_FEW=2" "SNOWGPU_FEW=3"
