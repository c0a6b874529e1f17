"""Build libsnowgpu.so (hand-written HIP for gfx950) in-tree with hipcc -- and libsnowcpu.so, the same per-beam device code compiled for the host
(the CPU twin of include/snowgpu_cpu.h: a measurement baseline and parity check, never loaded by the package).

    python -m lidar_snow_sim_amd.build [--force]

The shared library lands next to the package (lidar_snow_sim_amd/libsnowgpu.so); it is git-ignored
but travels to the GPU box with the working tree.
"""
from __future__ import annotations

import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libsnowgpu.so"
SOURCES = ["snowgpu_kernels.hip", "snowgpu_rows.hip", "snowgpu_prepass.hip", "snowgpu_plane.hip", "snowgpu_sampler.hip", "snowgpu_tables.hip", "snowgpu_api.cpp"]
# -ffp-contract=off: every decision of the reference is made on separately rounded float64/float32
# operations (NumPy never fuses a multiply into an add); a contracted FMA would change them.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


CPU_LIB = PKG / "libsnowcpu.so"          # the CPU twin (include/snowgpu_cpu.h): the kernels' device code compiled for the host -- a baseline, never loaded by the package


def build_cpu_twin(force: bool = False, verbose: bool = True) -> Path:
    src = CSRC / "snowcpu.cpp"
    deps = [src] + [p for p in CSRC.glob("*.h")] + [PKG.parent / "include" / "snowgpu_cpu.h"]
    if not force and CPU_LIB.exists() and all(p.stat().st_mtime <= CPU_LIB.stat().st_mtime for p in deps):
        return CPU_LIB
    cmd = [hipcc(), "--cuda-host-only", "-x", "hip", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-pthread", "-w",
           "-I", str(CSRC), "-I", str(PKG.parent / "include"), str(src), "-o", str(CPU_LIB)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return CPU_LIB


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found: libsnowgpu.so cannot be built")
    return exe


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = [p for p in CSRC.glob("*") if p.is_file() and p.name != "snowcpu.cpp"] + [PKG.parent / "include" / "snowgpu.h"]
    return any(p.stat().st_mtime > t for p in deps)


def build(force: bool = False, verbose: bool = True) -> Path:
    build_cpu_twin(force, verbose)
    if not force and not needs_build():
        return LIB
    objs = []
    (CSRC / "_obj").mkdir(exist_ok=True)
    headers = [p for p in CSRC.glob("*.h")] + [PKG.parent / "include" / "snowgpu.h"]
    newest_header = max(p.stat().st_mtime for p in headers)
    procs = []
    for src in SOURCES:
        obj = CSRC / "_obj" / (src.rsplit(".", 1)[0] + ".o")
        objs.append(str(obj))
        if not force and obj.exists() and obj.stat().st_mtime > max((CSRC / src).stat().st_mtime, newest_header):
            continue                                               # this translation unit is up to date
        cmd = [hipcc(), *FLAGS, "-x", "hip", "-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))                 # the translation units compile side by side
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *objs, "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
