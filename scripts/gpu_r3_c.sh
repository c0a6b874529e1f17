#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pipelined" 2>&1 | tail -15 > gpurun_out/r3c_pytest.log
tail -3 gpurun_out/r3c_pytest.log
for cfg in "1 4194304" "2 2097152" "2 1048576" "3 1048576" "3 2097152" "4 1048576" "3 524288"; do
  set -- $cfg
  SNOWGPU_PIPE_LANES=$1 SNOWGPU_PIPE_ROWS=$2 python bench.py --no-pmc --no-cpu-baseline --steps 4 > gpurun_out/r3c_bench_$1_$2.json 2> gpurun_out/r3c_bench_$1_$2.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r3c_bench_$1_$2.json"))
print("lanes $1 rows $2", "%.3g"%d["value"], "%.3f"%d["ms_per_step"], "pcie %.4g"%d.get("value_pcie_inclusive"), "nosrc %.4g"%d["pcie_inclusive"]["value_without_src"], d["pcie_inclusive"]["matches_device_entry"], "single %.3f py %.3f"%(d["single_frame"]["c_abi_pinned"]["ms"], d["single_frame"]["python_augment_pageable"]["ms"]))
PY
done
