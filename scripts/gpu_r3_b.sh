#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pipelined" 2>&1 | tail -15 > gpurun_out/r3b_pytest.log
tail -3 gpurun_out/r3b_pytest.log
for Q in 4 8; do
  GPU_MAX_HW_QUEUES=$Q SNOWGPU_PIPE_ROWS=2097152 python bench.py --no-pmc --no-cpu-baseline --steps 4 > gpurun_out/r3b_bench_q$Q.json 2> gpurun_out/r3b_bench_q$Q.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r3b_bench_q$Q.json"))
print("queues", $Q, d["value"], d["ms_per_step"], d.get("value_pcie_inclusive"), d["pcie_inclusive"]["value_without_src"], d["single_frame"]["c_abi_pinned"]["ms"])
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3b_trace -o t -- python $GRAFT_REPO_ROOT/bench.py --no-pmc --no-cpu-baseline --steps 1 --warmup 1 --frames 64 > $GRAFT_REPO_ROOT/gpurun_out/r3b_trace.log 2>&1
ls -la $GRAFT_REPO_ROOT/gpurun_out/r3b_trace/* | head
