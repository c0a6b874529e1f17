"""PCIe-inclusive rate of the host-pointer entry (numpy in, numpy out), one context and two pipelined contexts."""
import sys, time, threading, random
import numpy as np
sys.path.insert(0, '/root/repo')
sys.argv = sys.argv[:1]
import bench
from lidar_snow_sim_amd.synthetic import synthetic_sweep
from lidar_snow_sim_amd.tools.snowfall.simulation import augment_batch
tables = bench.make_tables(64)
F = 32
frames = [synthetic_sweep(64, 2048, seed=1000 + f, intensity="lambert") for f in range(F)]
planes = [([0.0, 0.0, -1.0], -1.7)] * F
orders = [list(range(64))] * F
def run(slot, reps):
    for _ in range(reps):
        augment_batch(frames, "x", bench.BEAM_DIV, particles=tables, planes=planes, orders=orders, slot=slot)
run(0, 1); run(1, 1)
t = time.time(); run(0, 4); dt = time.time() - t
print(f"one context : {4 * F * 131072 / dt / 1e6:8.1f} M points/s (host arrays in/out, {dt / 4 * 1e3:.1f} ms per {F}-frame batch)")
t = time.time()
th = [threading.Thread(target=run, args=(s, 4)) for s in (0, 1)]
[x.start() for x in th]; [x.join() for x in th]
dt = time.time() - t
print(f"two contexts: {8 * F * 131072 / dt / 1e6:8.1f} M points/s (two host threads, copies of one overlap kernels of the other)")
