"""PCIe-inclusive rate of the host-pointer entry snowgpu_augment_batch (NumPy in, NumPy out):
pageable arrays allocated per call (what a naive caller does), page-locked buffers reused across calls
(Context.pinned_empty), and two host threads driving two contexts so that the copies of one overlap the kernels
of the other."""
import sys, time, threading
import numpy as np
sys.path.insert(0, '/root/repo')
sys.argv = sys.argv[:1]
import bench
from lidar_snow_sim_amd import engine
from lidar_snow_sim_amd.synthetic import synthetic_sweep
tables = bench.make_tables(64)
F = 32
N = 131072
frames = [synthetic_sweep(64, 2048, seed=1000 + f, intensity="lambert") for f in range(F)]
rows = np.concatenate(frames)
off = np.arange(F + 1, dtype=np.int64) * N
planes = [[0.0, 0.0, -1.0, -1.7]] * F
engs = [engine.get_engine(0, s) for s in (0, 1)]
tids = [[e.table_ids_from_arrays(tables, list(range(64))) for _ in range(F)] for e in engs]


def pageable(slot, reps):
    e = engs[slot]
    for _ in range(reps):
        e.ctx.augment_batch(rows, off, tids[slot], bench.BEAM_DIV, plane=planes)


pin = []
for e in engs:
    r = e.ctx.pinned_empty((F * N, 5), np.float32)
    r[:] = rows
    pin.append((r, e.ctx.pinned_empty((F * N, 5), np.float32), e.ctx.pinned_empty(F * N, np.int32)))


def pinned(slot, reps):
    e = engs[slot]
    r, o, s = pin[slot]
    for _ in range(reps):
        e.ctx.augment_batch(r, off, tids[slot], bench.BEAM_DIV, plane=planes, out_rows=o, out_src=s)


def rate(fn, threads, reps=6):
    for s in range(threads):
        fn(s, 1)
    t = time.time()
    th = [threading.Thread(target=fn, args=(s, reps)) for s in range(threads)]
    [x.start() for x in th]
    [x.join() for x in th]
    dt = time.time() - t
    return threads * reps * F * N / dt / 1e6, dt / reps * 1e3


for name, fn in (("pageable, fresh output arrays", pageable), ("page-locked, reused buffers  ", pinned)):
    for threads in (1, 2):
        r, ms = rate(fn, threads)
        print(f"{name}  {threads} context(s): {r:8.1f} M points/s  ({ms:.1f} ms per {F}-frame batch per context)")
from lidar_snow_sim_amd.tools.snowfall.simulation import augment_batch as py_batch


def dropin(slot, reps):
    for _ in range(reps):
        py_batch(frames, "x", bench.BEAM_DIV, particles=tables, planes=[([0.0, 0.0, -1.0], -1.7)] * F, orders=[list(range(64))] * F, slot=slot)


for threads in (1, 2):
    r, ms = rate(dropin, threads)
    print(f"augment_batch(list of frames) -> list of arrays  {threads} thread(s): {r:8.1f} M points/s  ({ms:.1f} ms per {F}-frame batch)")
a = engs[0].ctx.augment_batch(rows, off, tids[0], bench.BEAM_DIV, plane=planes)
r, o, s = pin[0]
b = engs[0].ctx.augment_batch(r, off, tids[0], bench.BEAM_DIV, plane=planes, out_rows=o, out_src=s)
ok = all(np.array_equal(a[0][f * N:f * N + int(a[2][f])], b[0][f * N:f * N + int(b[2][f])]) for f in range(F)) and np.array_equal(a[2], b[2])
print("pinned path returns the same rows:", ok)
