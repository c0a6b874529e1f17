#!/usr/bin/env python3
"""C5 dry run on ONE GPU: N synthetic STF `.bin` frames (64 x 2048, float32) through the frame-stream driver
(lidar_snow_sim_amd.stream: reader threads -> GPU workers -> writer threads), 2.5 mm/h @ 1.6 m/s gunn tables.

    python scripts/gpu_stream_c5.py [--frames 10000] [--distinct 256] [--batch 64] [--workers 2]

The N frame files are hard links to `distinct` different sweeps (bounded disk use; every frame is still read, uploaded,
augmented, downloaded and written); outputs are unlinked right after they have been written unless --keep.  Prints one JSON
line (files/s, points/s, stage busy times) and stores it under gpurun_out/."""
import argparse
import json
import os
import random
import shutil
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=10000)
    ap.add_argument("--distinct", type=int, default=256)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--workers", type=int, default=2)
    ap.add_argument("--readers", type=int, default=6)
    ap.add_argument("--writers", type=int, default=6)
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--dir", default=None, help="scratch directory (default: a temporary one)")
    args = ap.parse_args()
    import torch  # noqa: F401  -- one HIP runtime per process: before libsnowgpu.so
    import bench
    from lidar_snow_sim_amd import stream
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    from lidar_snow_sim_amd.tools.snowfall import sampling as smp
    base = Path(args.dir or tempfile.mkdtemp(prefix="snowgpu_c5_"))
    lidar = base / "lidar_hdl64_strongest"
    lidar.mkdir(parents=True, exist_ok=True)
    t0 = time.perf_counter()
    n_dist = min(args.distinct, args.frames)
    for i in range(n_dist):
        synthetic_sweep(64, 2048, seed=1000 + i, intensity="lambert").tofile(lidar / f"src_{i:05d}.bin")
    ids = []
    for i in range(args.frames):
        name = f"2018-02-03_{i:05d}"
        dst = lidar / f"{name}.bin"
        if not dst.exists():
            os.link(lidar / f"src_{i % n_dist:05d}.bin", dst)
        ids.append(name)
    tables = bench.make_tables(64, 2.5, 1.6)
    occ, rate = smp.compute_occupancy(2.5, 1.6), smp.snowfall_rate_to_rainfall_rate(2.5, 1.6)
    prefix = f"gunn_{rate}_{occ}"
    setup_s = time.perf_counter() - t0
    # warm-up: table upload, allocations, page-locked pools
    random.seed(0)
    stream.run(lidar, ids[:2 * args.batch], modes=("gunn",), combos=[(rate, occ)], batch=args.batch, particles_by_prefix={prefix: tables},
               planes=([0.0, 0.0, -1.0], -1.7), workers=args.workers, readers=args.readers, writers=args.writers, keep_outputs=False)
    rep = {}
    random.seed(1)
    n = stream.run(lidar, ids, modes=("gunn",), combos=[(rate, occ)], batch=args.batch, particles_by_prefix={prefix: tables},
                   planes=([0.0, 0.0, -1.0], -1.7), workers=args.workers, readers=args.readers, writers=args.writers,
                   keep_outputs=args.keep, report=rep)
    out = {"what": "C5 dry run, one GPU: synthetic 64 x 2048 STF frames through lidar_snow_sim_amd.stream (read .bin -> H2D -> augment -> D2H -> write .bin)",
           "frames": n, "distinct_sweeps": n_dist, "files_per_s": n / rep["wall_s"], "points_per_s": rep["points_in"] / rep["wall_s"],
           "bytes_per_s_in_plus_out": (rep["points_in"] + rep["points_out"]) * 20 / rep["wall_s"], "setup_s": setup_s, **rep,
           "host_cores": os.cpu_count()}
    print(json.dumps(out), flush=True)
    d = ROOT / "gpurun_out"
    d.mkdir(exist_ok=True)
    (d / "stream_c5.json").write_text(json.dumps(out, indent=1) + "\n")
    if not args.dir:
        shutil.rmtree(base, ignore_errors=True)


if __name__ == "__main__":
    main()
