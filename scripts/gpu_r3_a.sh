#!/bin/bash
# round 3, first GPU pass: pipeline parity test, bench with the one-context PCIe figure at three chunk sizes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pipelined or full_size_batch or L5_augment_device or ragged" 2>&1 | tail -15 > gpurun_out/r3a_pytest.log
for R in 1048576 2097152 4194304; do
  SNOWGPU_PIPE_ROWS=$R python bench.py --no-pmc --no-cpu-baseline > gpurun_out/r3a_bench_$R.json 2> gpurun_out/r3a_bench_$R.err
done
tail -5 gpurun_out/r3a_pytest.log
for R in 1048576 2097152 4194304; do python - <<PY
import json
d=json.load(open("gpurun_out/r3a_bench_$R.json"))
print($R, d["value"], d["ms_per_step"], d.get("value_pcie_inclusive"), d["pcie_inclusive"]["value_without_src"], d["pcie_inclusive"]["matches_device_entry"], d.get("single_frame"))
PY
done
