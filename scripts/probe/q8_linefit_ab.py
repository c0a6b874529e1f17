import sys, time, random, json
from pathlib import Path
import numpy as np
sys.path.insert(0, ".")
import bench
from lidar_snow_sim_amd import engine
from lidar_snow_sim_amd.tools.snowfall.simulation import FlatBatch, augment_batch
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
fb = FlatBatch(pin, np.arange(F + 1, dtype=np.int64) * n_per)
planes = [([0.0, 0.0, -1.0], -1.7)] * F
def run(**kw):
    augment_batch(fb, "unused", bench.BEAM_DIV, particles=tables, orders=orders, planes=planes, **kw)
    t0 = time.perf_counter()
    for _ in range(4):
        augment_batch(fb, "unused", bench.BEAM_DIV, particles=tables, orders=orders, planes=planes, **kw)
    return (time.perf_counter() - t0) / 4
new = A._linregress_line
def old(x, y):
    xmean, ymean = np.mean(x), np.mean(y)
    ssxm, ssxym, _, _ = np.cov(x, y, bias=1).flat
    m0 = ssxym / ssxm
    return m0, ymean - m0 * xmean
for rep in range(2):
    a = run()
    A._linregress_line = old; b_old = run(q8="numpy")
    A._linregress_line = new; b_new = run(q8="numpy")
    print(f"first {F*n_per/a/1e9:.3f} G; numpy old {F*n_per/b_old/1e9:.3f} G share {a/b_old:.3f}; numpy new {F*n_per/b_new/1e9:.3f} G share {a/b_new:.3f}", flush=True)
