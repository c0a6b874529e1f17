#!/usr/bin/env python3
"""q8='numpy' through the threshold callback (snowgpu_set_threshold_callback) against q8='first', 256 C2 sweeps in page-locked memory, for a few
chunk sizes of the host pipeline.   python scripts/probe/q8_cb_probe.py   (no PyTorch in the process: the DMA engine moves the downloads)"""
import json
import random
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))


def main():
    import bench
    from lidar_snow_sim_amd import engine
    from lidar_snow_sim_amd.tools.snowfall.simulation import FlatBatch, augment_batch
    F = 256
    tables = bench.make_tables(64, 2.5, 1.6)
    eng = engine.get_engine(0)
    frames, orders = [], []
    for f in range(F):
        frames.append(bench.make_frame(64, 2048, 1000 + f, 1.0))
        random.seed(1000 + f)
        o = list(range(64))
        random.shuffle(o)
        orders.append(o)
    n_per = frames[0].shape[0]
    pin = eng.ctx.pinned_empty((F * n_per, 5), np.float32)
    pin[...] = np.concatenate(frames)
    fb = FlatBatch(pin, np.arange(F + 1, dtype=np.int64) * n_per)
    planes = [([0.0, 0.0, -1.0], -1.7)] * F

    def run(**kw):
        augment_batch(fb, "unused", bench.BEAM_DIV, particles=tables, orders=orders, planes=planes, **kw)
        t0 = time.perf_counter()
        for _ in range(3):
            augment_batch(fb, "unused", bench.BEAM_DIV, particles=tables, orders=orders, planes=planes, **kw)
        return (time.perf_counter() - t0) / 3

    out = {}
    for rows in (3 << 19, 3 << 20, 3 << 21):
        eng.ctx.set_pipeline(rows)
        a, b = run(), run(q8="numpy")
        out[str(rows >> 17) + " sweeps per chunk"] = {"q8_first_G": F * n_per / a / 1e9, "q8_numpy_G": F * n_per / b / 1e9, "share": a / b}
        print(rows >> 17, out[str(rows >> 17) + " sweeps per chunk"], flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
