#!/bin/bash
# Build lidar_snow_sim_amd/_variants/libsnowgpu_<name>.so with EVERY translation unit compiled with the extra -D flags (constants shared by
# the kernels, the table filing and the API: scripts/build_variant.sh recompiles the kernels only).   usage: scripts/build_variant_full.sh <name> [-DX=1 ...]
set -e
cd "$(dirname "$0")/.."
N=$1; shift
C=lidar_snow_sim_amd/csrc; V=lidar_snow_sim_amd/_variants; O=$C/_obj_$N; mkdir -p $V $O
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
for s in snowgpu_kernels.hip snowgpu_rows.hip snowgpu_prepass.hip snowgpu_plane.hip snowgpu_sampler.hip snowgpu_tables.hip snowgpu_api.cpp; do
  hipcc $F "$@" -x hip -c $C/$s -o $O/${s%.*}.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libsnowgpu_$N.so $O/*.o -ldl
rm -rf $O
echo $V/libsnowgpu_$N.so
