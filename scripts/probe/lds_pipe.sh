#!/bin/bash
# GPU side: how busy the LDS pipe is in the per-beam kernels (cross-lane reads are LDS instructions).   usage: bash scripts/probe/lds_pipe.sh [bench args]
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/lds; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-pcie $*"
i=0
for set in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o b -- $B > $O/p$i.log 2>&1
done
python scripts/pmc_summary.py $O/p* > $O/summary.txt 2>&1
grep -A17 "${KERNELS:-k_beams<float, 4, 256, false, 1>\|k_power_few\|k_power<float, 4\|k_power<float, 8\|k_sort_hist}" $O/summary.txt | head -120
