"""Experiment: per-phase cycle totals of the per-beam kernel on the bench workload (one batch)."""
import ctypes, random, sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from lidar_snow_sim_amd import _native, engine
from lidar_snow_sim_amd.synthetic import synthetic_sweep
sys.argv = sys.argv[:1]
import bench
eng = engine.get_engine(0)
tables = bench.make_tables(64)
F = 8
frames, tids, planes = [], [], []
for f in range(F):
    pc = synthetic_sweep(64, 2048, seed=1000 + f, intensity="lambert")
    random.seed(1000 + f); order = list(range(64)); random.shuffle(order)
    frames.append(pc); tids.append(eng.table_ids_from_arrays(tables, order)); planes.append([0, 0, -1.0, -1.7])
rows = np.concatenate(frames); off = np.arange(F + 1) * frames[0].shape[0]
L = _native.lib()
L.snowgpu_debug_phase_cycles.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
eng.ctx.augment_batch(rows, off, tids, bench.BEAM_DIV, plane=planes)
L.snowgpu_debug_phase_cycles(eng.ctx.handle, 1, None)
t = time.time(); eng.ctx.augment_batch(rows, off, tids, bench.BEAM_DIV, plane=planes); dt = time.time() - t
out = (ctypes.c_ulonglong * 64)()
L.snowgpu_debug_phase_cycles(eng.ctx.handle, 0, out)
v = list(out)
print('host call', dt, '(lane-0 view; 100 MHz ticks x24 = cycles @2.4 GHz)')
for ti, name in enumerate(('tier 4', 'tier 8', 'tier 16', 'tier 63')):
    b = v[8 * ti: 8 * ti + 8]
    w = max(b[5], 1)
    print(f'{name}: waves {b[5]}  load {b[0]/w*24:9.0f}  P1 {b[1]/w*24:9.0f}  P2 {b[2]/w*24:9.0f}  P3a {b[3]/w*24:9.0f}  P3b {b[4]/w*24:9.0f} cycles/wave;  lane0 mean L {b[6]/w:.2f} candidates {b[7]/w:.1f}')
for ti, name in enumerate(('tier 4', 'tier 8', 'tier 16', 'tier 63')):
    b = v[32 + 8 * ti: 32 + 8 * ti + 8]
    w = max(b[0], 1); ln = max(b[1], 1)
    print(f'{name}: power-phase waves {b[0]} lanes/wave {b[1]/w:.1f}  per wave max: scatterers {b[2]/w:.1f} groups evaluated {b[3]/w:.1f};  per lane mean: {b[4]/ln:.1f} / {b[5]/ln:.1f}')
