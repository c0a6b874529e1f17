#!/usr/bin/env python3
"""Packed result transfer: throughput by host-thread count, and the per-chunk event trace (SNOWGPU_PIPE_TRACE=1) of one call.  python scripts/probe/packed_trace.py [threads ...]"""
import json, os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import bench, random
from lidar_snow_sim_amd import engine
F = 256
eng = engine.get_engine(0)
tables = bench.make_tables(64, 2.5, 1.6)
frames, ids = [], []
for f in range(F):
    frames.append(bench.make_frame(64, 2048, 1000 + f, 1.0))
    random.seed(1000 + f); o = list(range(64)); random.shuffle(o)
    ids.append(eng.table_ids_from_arrays(tables, o))
n_per = frames[0].shape[0]; n_total = n_per * F
pin_in = eng.ctx.pinned_empty((n_total, 5), np.float32); np.concatenate(frames, out=pin_in)
pin_out = eng.ctx.pinned_empty((n_total, 5), np.float32); pin_src = eng.ctx.pinned_empty(n_total, np.int32)
off = np.arange(F + 1, dtype=np.int64) * n_per; h_ids = np.asarray(ids, np.int32); planes = np.asarray([[0.0, 0.0, -1.0, -1.7]] * F)
def call(): return eng.ctx.augment_batch(pin_in, off, h_ids, bench.BEAM_DIV, plane=planes, out_rows=pin_out, out_src=pin_src)
def timed(n=4):
    call(); t0 = time.perf_counter()
    for _ in range(n): call()
    return (time.perf_counter() - t0) / n
res = {"rows": [round(n_total / timed() / 1e9, 3) for _ in range(3)]}
for thr in [int(a) for a in sys.argv[1:]] or [14, 8, 6]:
    eng.ctx.set_result_transfer("packed", thr)
    res[f"packed_{thr}"] = [round(n_total / timed() / 1e9, 3) for _ in range(4)]
    res[f"packed_{thr}_times"] = eng.ctx.transfer_times()
eng.ctx.set_result_transfer("rows")
res["rows_again"] = [round(n_total / timed() / 1e9, 3) for _ in range(2)]
print(json.dumps(res))
