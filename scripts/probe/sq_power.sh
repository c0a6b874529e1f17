# GPU side: SQ counters of the k_power* kernels of the current build (one pmc pass), then A/B of SNOWGPU_FEW = 0 .. 3
export R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/sqp; rm -rf $O; mkdir -p $O
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY --kernel-trace -d $O/sq -o b --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-pcie > $O/sq.log 2>&1
python $R/scripts/pmc_summary.py $O/sq --filter k_power | grep -v "63, 16\|k_power_plan" 
cd $R; bash scripts/ab_bench.sh "" "SNOWGPU_FEW=0" "SNOWGPU_FEW=1" "SNOWGPU_FEW=2" "SNOWGPU_FEW=3"
