#!/bin/bash
# A/B of the received-power phase's schedule on one box: three persistent kernels (rounds 2 - 5) against the one work queue (k_power_all),
# its grid (waves per CU), and one against two batches in flight (bench.py --lanes).  usage: scripts/probe/kp_all_ab.sh [workloads...]  -> gpurun_out/kp_all_ab.txt
out=gpurun_out/kp_all_ab.txt
mkdir -p gpurun_out
: > $out
for wl in "${@:-C2}"; do
  for cfg in "SNOWGPU_KP_ALL=0 LANES=1" "SNOWGPU_KP_ALL=1 SNOWGPU_KP_ALL_WAVES=8 LANES=1" "SNOWGPU_KP_ALL=1 SNOWGPU_KP_ALL_WAVES=7 LANES=1" "SNOWGPU_KP_ALL=1 SNOWGPU_KP_ALL_WAVES=6 LANES=1" \
             "SNOWGPU_KP_ALL=0 LANES=2" "SNOWGPU_KP_ALL=1 SNOWGPU_KP_ALL_WAVES=8 LANES=2" "SNOWGPU_KP_ALL=1 SNOWGPU_KP_ALL_WAVES=6 LANES=2" "SNOWGPU_KP_ALL=1 SNOWGPU_KP_ALL_WAVES=8 LANES=3" "SNOWGPU_KP_ALL=0 LANES=1"; do
    lanes=$(echo "$cfg" | grep -o "LANES=[0-9]*" | cut -d= -f2)
    line=$(env $cfg python bench.py --workload $wl --steps 20 --warmup 4 --lanes $lanes --no-pmc --no-pcie --no-cpu-baseline 2>gpurun_out/kp_all_ab.err | tail -1)
    ms=$(python -c "import json,sys; d=json.loads(sys.argv[1]); print('%.3f ms/step  %.3f G  region %.3f ms' % (d['ms_per_step'], d['value']/1e9, d['roofline']['avg_launch_ms']))" "$line" 2>/dev/null)
    [ -z "$ms" ] && ms="FAILED: $(tail -2 gpurun_out/kp_all_ab.err | tr '\n' ' ')"
    echo "$wl  $cfg  ->  $ms" | tee -a $out
  done
done
