#!/bin/bash
cd "$(dirname "$0")/.."
run() { name=$1; shift; echo -n "$name: "; env "$@" python bench.py --no-pmc --no-cpu-baseline --no-pcie --steps 10 $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%.4g pts/s %.3f ms/step region %.3f ms tiers %s'%(d['value'],d['ms_per_step'],d['roofline']['avg_launch_ms'],d['config']['beams_per_capacity_tier']))"; }
run base A=1
run tiercap64 SNOWGPU_TIER_CAP=64
EXTRA="--workload C2far" run far_base A=1
EXTRA="--workload C2far" run far_tiercap64 SNOWGPU_TIER_CAP=64
EXTRA="--frames 16" run f16_base A=1
EXTRA="--frames 16" run f16_tiercap64 SNOWGPU_TIER_CAP=64
EXTRA="--frames 16" run f16_serial SNOWGPU_SERIAL=1
EXTRA="--frames 1" run f1_base A=1
EXTRA="--frames 1" run f1_tiercap64 SNOWGPU_TIER_CAP=64
EXTRA="--frames 1" run f1_serial SNOWGPU_SERIAL=1
