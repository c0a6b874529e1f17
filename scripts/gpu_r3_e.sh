#!/bin/bash
cd "$(dirname "$0")/.."
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --no-pmc --no-cpu-baseline --steps 3 > gpurun_out/r3e_$name.json 2> gpurun_out/r3e_$name.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r3e_$name.json"))
print("$name", "%.3g"%d["value"], "pcie %.4g"%d.get("value_pcie_inclusive"), "nosrc %.4g"%d["pcie_inclusive"]["value_without_src"], "single %.3f"%d["single_frame"]["c_abi_pinned"]["ms"])
PY
}
for L in 1 2 3; do for R in 1048576 2097152 4194304; do
run q16_L${L}_R$R GPU_MAX_HW_QUEUES=16 SNOWGPU_PIPE_LANES=$L SNOWGPU_PIPE_ROWS=$R
done; done
run q32_L2_R1048576 GPU_MAX_HW_QUEUES=32 SNOWGPU_PIPE_LANES=2 SNOWGPU_PIPE_ROWS=1048576
