#!/bin/bash
# Instruction-cache counters of the per-beam kernel (own pass, own timeout).
export R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
OUT=${1:-pmc_icache}
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
timeout 150 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace -d $R/gpurun_out/$OUT/a -o b --output-format csv -- $B > $R/gpurun_out/${OUT}_a.log 2>&1
echo "rc=$?"
cd $R; python scripts/pmc_summary.py gpurun_out/$OUT --filter "k_beams<float, 4" | tee gpurun_out/${OUT}_summary.txt
