#!/bin/bash
# Build lidar_snow_sim_amd/_variants/libsnowgpu_<name>.so: the kernels compiled with extra -D flags (same-box A/B runs copy a
# variant over libsnowgpu.so on the GPU side).   usage: scripts/build_variant.sh <name> [-DX=1 ...]
set -e
cd "$(dirname "$0")/.."
N=$1; shift
C=lidar_snow_sim_amd/csrc; V=lidar_snow_sim_amd/_variants; mkdir -p $V $C/_obj
python -m lidar_snow_sim_amd.build > /dev/null
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
hipcc $F "$@" -x hip -c $C/snowgpu_kernels.hip -o $C/_obj/kernels_$N.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libsnowgpu_$N.so $C/_obj/kernels_$N.o $C/_obj/snowgpu_prepass.o $C/_obj/snowgpu_sampler.o $C/_obj/snowgpu_tables.o $C/_obj/snowgpu_api.o
echo $V/libsnowgpu_$N.so
