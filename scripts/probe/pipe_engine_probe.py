"""Which engine moves the downloads of a pipelined host batch?  Run under rocprofv3 --kernel-trace --memory-copy-trace.
   argv[1] = 'torch' imports (and initialises) PyTorch first."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
if len(sys.argv) > 1 and sys.argv[1] == 'torch':
    import torch
    torch.zeros(4, device='cuda')
import numpy as np
from lidar_snow_sim_amd import engine
from lidar_snow_sim_amd.synthetic import synthetic_sweep
from lidar_snow_sim_amd.tools.snowfall import sampling as smp
eng = engine.get_engine(0)
occ, rate = smp.compute_occupancy(2.5, 1.6), smp.snowfall_rate_to_rainfall_rate(2.5, 1.6)
tabs = [smp.dart_throwing(occ, rate, 40.0, np.random.default_rng(42 + i), 'gunn') for i in range(4)]
tables = [tabs[i % 4] for i in range(64)]
F = 32
frames = [synthetic_sweep(64, 2048, seed=1000 + f, intensity='lambert') for f in range(F)]
n = frames[0].shape[0]
pin_in = eng.ctx.pinned_empty((F * n, 5), np.float32); pin_in[...] = np.concatenate(frames)
pin_out = eng.ctx.pinned_empty((F * n, 5), np.float32); pin_src = eng.ctx.pinned_empty(F * n, np.int32)
off = np.arange(F + 1, dtype=np.int64) * n
ids = np.asarray([eng.table_ids_from_arrays(tables, list(range(64))) for _ in range(F)], np.int32)
planes = np.asarray([[0.0, 0.0, -1.0, -1.7]] * F)
for _ in range(3):
    t = time.perf_counter()
    eng.ctx.augment_batch(pin_in, off, ids, float(np.degrees(3e-3)), plane=planes, out_rows=pin_out, out_src=pin_src)
    dt = time.perf_counter() - t
print(f"{F * n / dt / 1e9:.3f} G points/s")
