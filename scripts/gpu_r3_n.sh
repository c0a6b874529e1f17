#!/bin/bash
cd "$(dirname "$0")/.."
run() { name=$1; shift; echo -n "$name: "; env "$@" python scripts/pcie_bench.py --reps 3 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('src %.4g nosrc %.4g single %.3f ms'%(d['points_per_s'],d['points_per_s_without_src'],d['single_frame_c_abi_ms']))"; }
run base A=1
run sysscope0 ROC_SYSTEM_SCOPE_SIGNAL=0
run serial SNOWGPU_SERIAL=1
run serial_R2M SNOWGPU_SERIAL=1 SNOWGPU_PIPE_ROWS=2097152
run serial_sysscope0 SNOWGPU_SERIAL=1 ROC_SYSTEM_SCOPE_SIGNAL=0
run activewait ROC_ACTIVE_WAIT_TIMEOUT=1000
run R2M SNOWGPU_PIPE_ROWS=2097152
run R4M SNOWGPU_PIPE_ROWS=4194304
SNOWGPU_PIPE_TRACE=1 python scripts/pcie_bench.py --reps 1 2>&1 | grep "^pipe chunk" | tail -11
