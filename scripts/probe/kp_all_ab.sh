#!/bin/bash
# A/B of the received-power phase's schedule on one box: three persistent kernels (rounds 2 - 5) against the one work queue (k_power_all),
# its grid (waves per CU) and how its waves draw items.  usage: scripts/probe/kp_all_ab.sh [workloads...]  -> gpurun_out/kp_all_ab.txt
out=gpurun_out/kp_all_ab.txt
mkdir -p gpurun_out
: > $out
for wl in "${@:-C2}"; do
  for cfg in "SNOWGPU_KP_ALL=0" "SNOWGPU_KP_ALL=1 SNOWGPU_KP_ALL_WAVES=6" "SNOWGPU_KP_ALL=1 SNOWGPU_KP_ALL_WAVES=8" "SNOWGPU_KP_ALL=1 SNOWGPU_KP_ALL_WAVES=5" \
             "SNOWGPU_KP_ALL=1 SNOWGPU_KP_ALL_WAVES=6 SNOWGPU_KP_ALL_TICKET=1" "SNOWGPU_KP_ALL=1 SNOWGPU_KP_ALL_WAVES=8 SNOWGPU_KP_ALL_TICKET=1" "SNOWGPU_KP_ALL=0"; do
    line=$(env $cfg python bench.py --workload $wl --steps 20 --warmup 3 --no-pmc --no-pcie --no-cpu-baseline 2>/dev/null | tail -1)
    ms=$(python -c "import json,sys; d=json.loads(sys.argv[1]); print('%.3f ms/step  %.3f G  region %.3f ms' % (d['ms_per_step'], d['value']/1e9, d['roofline']['avg_launch_ms']))" "$line" 2>/dev/null)
    echo "$wl  $cfg  ->  $ms" | tee -a $out
  done
done
