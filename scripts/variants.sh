#!/bin/bash
# Timing experiment: libsnowgpu.so variants built with extra -D flags, e.g.  scripts/variants.sh nopack=-DSG_NO_PACK=1 nopre=-DSG_NO_PREFILTER=1
# -> lidar_snow_sim_amd/_ablate/libsnowgpu_<name>.so; run with SNOWGPU_LIB=<path> python bench.py
set -e
cd "$(dirname "$0")/.."
C=lidar_snow_sim_amd/csrc; O=$C/_obj; D=lidar_snow_sim_amd/_ablate; mkdir -p $D
python -m lidar_snow_sim_amd.build > /dev/null
for v in "$@"; do
  n=${v%%=*}; f=${v#*=}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $f -Iinclude -x hip -c $C/snowgpu_kernels.hip -o $D/k_$n.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libsnowgpu_$n.so $D/k_$n.o $O/snowgpu_prepass.o $O/snowgpu_sampler.o $O/snowgpu_api.o ) &
done
wait
ls $D/*.so
