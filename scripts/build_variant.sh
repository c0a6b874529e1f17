#!/bin/bash
# Build lidar_snow_sim_amd/_variants/libsnowgpu_<name>.so: snowgpu_kernels.hip compiled with extra -D flags, the other translation
# units as they are (same-box A/B runs copy a variant over libsnowgpu.so on the GPU side).   usage: scripts/build_variant.sh <name> [-DX=1 ...]
set -e
cd "$(dirname "$0")/.."
N=$1; shift
C=lidar_snow_sim_amd/csrc; V=lidar_snow_sim_amd/_variants; mkdir -p $V $C/_obj
python -m lidar_snow_sim_amd.build > /dev/null
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
hipcc $F "$@" -x hip -c $C/snowgpu_kernels.hip -o $C/_obj/kernels_$N.o
OTHERS=$(ls $C/_obj/snowgpu_*.o | grep -v snowgpu_kernels.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libsnowgpu_$N.so $C/_obj/kernels_$N.o $OTHERS -ldl
echo $V/libsnowgpu_$N.so
