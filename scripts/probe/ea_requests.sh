#!/bin/bash
# GPU side: sizes of the L2 <-> fabric requests per kernel (how much of FETCH_SIZE / WRITE_SIZE is partial-line traffic).
# usage: bash scripts/probe/ea_requests.sh [bench args]
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/ea; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_EA[A-Z0-9_]*\|TCC_REQ[A-Z_]*\|TCC_HIT[A-Z_]*\|TCC_MISS[A-Z_]*\|TCC_WRITE[A-Z_]*\|TCC_ATOMIC[A-Z_]*" | sort -u > $O/avail.txt
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-pcie $*"
i=0
rm -rf $O/p*
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum" ${FULL:+"TCC_ATOMIC_sum TCC_WRITE_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum"}; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o b -- $B > $O/p$i.log 2>&1
done
python scripts/pmc_summary.py $O/p* > $O/summary.txt 2>&1
grep -A${LINES:-5} "${KERNELS:-k_beams<float, 4, 256, false, 1>\|k_power_few\|k_compact_s\|k_compact_c\|k_sort_hist\|k_lean_hist\|k_power<float, 4\|k_power<float, 8}" $O/summary.txt | head -150
