#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py 2>/dev/null > gpurun_out/final_bench.json; python -c "
import json
d=json.load(open('gpurun_out/final_bench.json'))
print(d['value'], d['ms_per_step'], d['value_pcie_inclusive'], d['pcie_inclusive']['value_without_src'], d['pcie_inclusive']['in_process_with_pytorch'], d['pcie_inclusive']['matches_device_entry'], d['single_frame']['c_abi_pinned']['ms'], d['cpu_baseline']['gpu_output_matches'])"
grep -h "L8 native" gpurun_out/fullsize_parity.jsonl | tail -4
