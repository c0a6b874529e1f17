#!/bin/bash
# GPU side: is the vector-memory front end (texture addresser / L1) what the per-beam kernels wait for?   usage: bash scripts/probe/ta_busy.sh [bench args]
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/ta; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "\bTA_[A-Z0-9_]*\|\bTCP_[A-Z0-9_]*\|\bTD_[A-Z0-9_]*" | sort -u > $O/avail.txt
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-pcie $*"
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum" "TD_TD_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o b -- $B > $O/p$i.log 2>&1
done
python scripts/pmc_summary.py $O/p* > $O/summary.txt 2>&1
grep -A20 "${KERNELS:-k_beams<float, 4, 256, false, 1>\|k_power_few\|k_sort_hist\|k_compact_scatter}" $O/summary.txt | head -100
