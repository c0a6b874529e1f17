#!/bin/bash
cd "$(dirname "$0")/.."
run() { # name, env...
  name=$1; shift
  env "$@" SNOWGPU_PIPE_LANES=1 SNOWGPU_PIPE_ROWS=2097152 python bench.py --no-pmc --no-cpu-baseline --steps 2 --frames 128 > gpurun_out/r3d_$name.json 2> gpurun_out/r3d_$name.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r3d_$name.json"))
print("$name", "%.3g"%d["value"], "pcie %.4g"%d.get("value_pcie_inclusive"), "nosrc %.4g"%d["pcie_inclusive"]["value_without_src"], "single %.3f"%d["single_frame"]["c_abi_pinned"]["ms"])
PY
}
run base A=1
run q8 GPU_MAX_HW_QUEUES=8
run q16 GPU_MAX_HW_QUEUES=16
run nosdma HSA_ENABLE_SDMA=0
run nosdma_q8 HSA_ENABLE_SDMA=0 GPU_MAX_HW_QUEUES=8
run serial SNOWGPU_SERIAL=1
run serial_q8 SNOWGPU_SERIAL=1 GPU_MAX_HW_QUEUES=8
