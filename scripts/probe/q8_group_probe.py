#!/usr/bin/env python3
"""q8='numpy' by rows per group of the threshold-callback pipeline, with the time the calling thread spends inside the callback, against
q8='first' at the default chunk size.  python scripts/probe/q8_group_probe.py   (no PyTorch in the process)"""
import random
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import bench
from lidar_snow_sim_amd import engine
from lidar_snow_sim_amd.tools.snowfall import simulation as S
from lidar_snow_sim_amd.tools.wet_ground import augmentation as A

F = 256
tables = bench.make_tables(64, 2.5, 1.6)
eng = engine.get_engine(0)
frames, orders = [], []
for f in range(F):
    frames.append(bench.make_frame(64, 2048, 1000 + f, 1.0))
    random.seed(1000 + f); o = list(range(64)); random.shuffle(o); orders.append(o)
n_per = frames[0].shape[0]
pin = eng.ctx.pinned_empty((F * n_per, 5), np.float32)
pin[...] = np.concatenate(frames)
fb = S.FlatBatch(pin, np.arange(F + 1, dtype=np.int64) * n_per)
planes = [([0.0, 0.0, -1.0], -1.7)] * F
inside = [0.0, 0]
orig = A.noise_polys_from_device_stats
def timed(*a, **k):
    t0 = time.perf_counter()
    r = orig(*a, **k)
    inside[0] += time.perf_counter() - t0; inside[1] += 1
    return r
A.noise_polys_from_device_stats = timed
if hasattr(S, "noise_polys_from_device_stats"):
    S.noise_polys_from_device_stats = timed

def run(reps=4, **kw):
    S.augment_batch(fb, "unused", bench.BEAM_DIV, particles=tables, orders=orders, planes=planes, **kw)
    inside[0] = 0.0; inside[1] = 0
    t0 = time.perf_counter()
    for _ in range(reps):
        S.augment_batch(fb, "unused", bench.BEAM_DIV, particles=tables, orders=orders, planes=planes, **kw)
    return (time.perf_counter() - t0) / reps, inside[0] / reps, inside[1] // reps

import os
for mode in ("rows", "packed"):
    eng.ctx.set_result_transfer(mode, 8)
    eng.ctx.set_pipeline(3 << 19)
    a, _, _ = run()
    print(f"{mode} transfer: q8='first', default chunks: {F * n_per / a / 1e9:.3f} G ({a * 1e3:.1f} ms); usable cpus {A._usable_cpus()}")
    for sweeps in (24, 32, 40, 48, 56):
        eng.ctx.set_pipeline(sweeps << 17)
        res = [run(q8="numpy") for _ in range(3)]
        print(f"  q8='numpy', {sweeps} sweeps per group: " + ", ".join(f"{F * n_per / b / 1e9:.3f} G, share {a / b:.2f} ({cb * 1e3:.1f} ms in {ncb} callbacks)" for b, cb, ncb in res))
eng.ctx.set_result_transfer("rows", 0)
eng.ctx.set_pipeline(3 << 19)
