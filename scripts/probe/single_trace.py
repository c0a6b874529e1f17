#!/usr/bin/env python3
"""One sweep end to end through the C ABI, 60 times (rocprofv3 --kernel-trace --memory-copy-trace around this), then print the median wall time.
   python scripts/probe/single_trace.py [--serial]"""
import json, os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import bench, random
from lidar_snow_sim_amd import engine
eng = engine.get_engine(0)
tables = bench.make_tables(64, 2.5, 1.6)
pc = bench.make_frame(64, 2048, 1000, 1.0)
random.seed(1000); o = list(range(64)); random.shuffle(o)
ids = np.asarray([eng.table_ids_from_arrays(tables, o)], np.int32)
n = pc.shape[0]
pin_in = eng.ctx.pinned_empty((n, 5), np.float32); pin_in[...] = pc
pin_out = eng.ctx.pinned_empty((n, 5), np.float32); pin_src = eng.ctx.pinned_empty(n, np.int32)
off = np.array([0, n], np.int64); planes = np.asarray([[0.0, 0.0, -1.0, -1.7]])
def one(): eng.ctx.augment_batch(pin_in, off, ids, bench.BEAM_DIV, plane=planes, out_rows=pin_out, out_src=pin_src)
for _ in range(10): one()
ts = []
for _ in range(60):
    c0 = time.perf_counter(); one(); ts.append(time.perf_counter() - c0)
print(json.dumps({"single_ms_median": float(np.median(ts) * 1e3), "min": float(np.min(ts) * 1e3)}))
