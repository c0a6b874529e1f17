#!/bin/bash
cd /tmp && export TMPDIR=/tmp
export SNOWGPU_PIPE_LANES=2 SNOWGPU_PIPE_ROWS=1048576
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3g_trace -o t -- python $GRAFT_REPO_ROOT/bench.py --no-pmc --no-cpu-baseline --steps 1 --warmup 1 --frames 64 > $GRAFT_REPO_ROOT/gpurun_out/r3g_trace.log 2>&1
