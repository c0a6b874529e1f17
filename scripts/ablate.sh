#!/bin/bash
# Timing experiment: build variants of libsnowgpu.so with phases of the per-beam kernel cut out (SG_ABLATE=1..4, results
# are wrong on purpose) next to the real one, so that `SNOWGPU_LIB=... python bench.py` can price each phase in real time.
#   1: no received-power phase   2: + no occlusion dict / amplitudes   3: + no candidate scan   4: rows copied only
set -e
cd "$(dirname "$0")/.."
C=lidar_snow_sim_amd/csrc; O=$C/_obj; D=lidar_snow_sim_amd/_ablate; mkdir -p $D
python -m lidar_snow_sim_amd.build > /dev/null
for n in 1 2 3 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DSG_ABLATE=$n -Iinclude -x hip -c $C/snowgpu_kernels.hip -o $D/k$n.o &
done
wait
for n in 1 2 3 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libsnowgpu_ab$n.so $D/k$n.o $O/snowgpu_prepass.o $O/snowgpu_sampler.o $O/snowgpu_api.o
done
ls -la $D/*.so
