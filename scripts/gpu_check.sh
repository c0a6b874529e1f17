#!/bin/bash
# GPU side (through gpurun): parity suite, a plain bench run, and one kernel-trace of a few steps with its timeline.
# usage: scripts/gpu_check.sh [tests|notests] [bench args...]
export R=$GRAFT_REPO_ROOT; cd $R
MODE=${1:-tests}; shift
if [ "$MODE" = tests ]; then timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_fullsize.py 2>&1 | tail -5; fi
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-pcie "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('BENCH', round(d['value']/1e9,3), 'Gpts/s', round(d['ms_per_step'],3), 'ms/step region', round(d['roofline']['avg_launch_ms'],3), d['config'].get('beams_per_capacity_tier'))"
cd /tmp && export TMPDIR=/tmp; O=$R/gpurun_out/trace; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o b --output-format csv -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --no-pcie "$@" > $O/log.txt 2>&1
python $R/scripts/trace_timeline.py $O
