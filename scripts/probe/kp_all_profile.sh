#!/bin/bash
# GPU side: one-step timeline and SQ counters of the received-power phase for the configuration in the environment
# (SNOWGPU_KP_ALL, SNOWGPU_KP_ALL_WAVES, ...).  usage: [env] scripts/probe/kp_all_profile.sh <tag> [workload]
export R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
TAG=${1:-kpall}; WL=${2:-C2}
O=$R/gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O
B="python $R/bench.py --workload $WL --no-cpu-baseline --no-pmc --no-pcie"
timeout 200 rocprofv3 --kernel-trace --stats -d $O/stats -o b --output-format csv -- $B --steps 6 --warmup 2 > $O/stats.log 2>&1
python $R/scripts/trace_timeline.py $O/stats > $O/timeline.txt
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_BUSY_CYCLES --kernel-trace -d $O/sq -o b --output-format csv -- $B --steps 2 --warmup 1 > $O/sq.log 2>&1
python $R/scripts/pmc_summary.py $O/sq --filter k_power > $O/sq_power.txt
timeout 200 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS --kernel-trace -d $O/sq2 -o b --output-format csv -- $B --steps 2 --warmup 1 > $O/sq2.log 2>&1
python $R/scripts/pmc_summary.py $O/sq2 --filter k_power > $O/sq2_power.txt
timeout 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES --kernel-trace -d $O/sq3 -o b --output-format csv -- $B --steps 2 --warmup 1 > $O/sq3.log 2>&1
python $R/scripts/pmc_summary.py $O/sq3 --filter k_power > $O/sq3_power.txt
rm -rf $O/stats $O/sq $O/sq2 $O/sq3
cat $O/timeline.txt | tail -30
