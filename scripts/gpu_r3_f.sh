#!/bin/bash
cd "$(dirname "$0")/.."
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --no-pmc --no-cpu-baseline --steps 3 > gpurun_out/r3f_$name.json 2> gpurun_out/r3f_$name.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r3f_$name.json"))
    print("$name", "%.3g"%d["value"], "pcie %.4g"%d.get("value_pcie_inclusive"), "nosrc %.4g"%d["pcie_inclusive"]["value_without_src"], d["pcie_inclusive"]["matches_device_entry"], "single %.3f"%d["single_frame"]["c_abi_pinned"]["ms"])
except Exception as e:
    print("$name failed", e); print(open("gpurun_out/r3f_$name.err").read()[-600:])
PY
}
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pipelined" 2>&1 | tail -3
for L in 1 2 3; do for R in 524288 1048576 2097152; do
run L${L}_R$R SNOWGPU_PIPE_LANES=$L SNOWGPU_PIPE_ROWS=$R
done; done
run L2_R1M_b16 SNOWGPU_PIPE_LANES=2 SNOWGPU_PIPE_ROWS=1048576 SNOWGPU_LINK_BLOCKS=16
run L2_R1M_b256 SNOWGPU_PIPE_LANES=2 SNOWGPU_PIPE_ROWS=1048576 SNOWGPU_LINK_BLOCKS=256
run L2_R1M_b0 SNOWGPU_PIPE_LANES=2 SNOWGPU_PIPE_ROWS=1048576 SNOWGPU_LINK_BLOCKS=0
run L2_R1M_q8 SNOWGPU_PIPE_LANES=2 SNOWGPU_PIPE_ROWS=1048576 GPU_MAX_HW_QUEUES=8
