#!/bin/bash
# SQ counters of the per-beam kernel on the bench workload (two passes; TA/TCP counters abort rocprofv3 on this pool -- do not add them).
export R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
OUT=${1:-pmc_sq}
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace -d $R/gpurun_out/$OUT/a -o b --output-format csv -- $B > $R/gpurun_out/${OUT}_a.log 2>&1
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_BRANCH --kernel-trace -d $R/gpurun_out/$OUT/b -o b --output-format csv -- $B > $R/gpurun_out/${OUT}_b.log 2>&1
cd $R; python scripts/pmc_summary.py gpurun_out/$OUT --filter "${2:-k_beams<float, 4}" | tee gpurun_out/${OUT}_summary.txt
