#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD
export SNOWGPU_LINK_BLOCKS=0 SNOWGPU_PIPE_LANES=1 SNOWGPU_PIPE_ROWS=1048576
cd /tmp && export TMPDIR=/tmp
for M in none torch; do
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/r3h_$M -o t -- python $R/scripts/probe/pipe_engine_probe.py $M 2>&1 | grep "points/s"
python - <<PY
import csv, collections
d="$R/gpurun_out/r3h_$M/"
rows=list(csv.DictReader(open(d+'t_memory_copy_trace.csv')))
print("$M", collections.Counter(r['Direction'] for r in rows))
k=list(csv.DictReader(open(d+'t_kernel_trace.csv')))
print(collections.Counter(r['Kernel_Name'][:30] for r in k if 'rocclr' in r['Kernel_Name']))
PY
done
