"""Probe: host time to ENQUEUE one snowgpu_augment_batch_device call vs the time the device needs for it."""
import sys, time
import numpy as np, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.argv = sys.argv[:1]
import bench
from lidar_snow_sim_amd import engine
from lidar_snow_sim_amd.synthetic import synthetic_sweep
eng = engine.get_engine(0)
tables = bench.make_tables(64)
dev = torch.device("cuda:0")
for F in (1, 8, 16, 32, 64):
    frames = [synthetic_sweep(64, 2048, seed=1000 + f, intensity="lambert") for f in range(F)]
    n = frames[0].shape[0]
    rows = torch.from_numpy(np.concatenate(frames)).to(dev)
    off = torch.arange(F + 1, dtype=torch.int64, device=dev) * n
    tids = torch.tensor([eng.table_ids_from_arrays(tables, list(range(64))) for _ in range(F)], dtype=torch.int32, device=dev)
    plane = torch.tensor([[0.0, 0.0, -1.0, -1.7]] * F, dtype=torch.float64, device=dev)
    out = torch.empty_like(rows); src = torch.empty(F * n, dtype=torch.int32, device=dev)
    cnt = torch.zeros(F, dtype=torch.int64, device=dev); st = torch.zeros(F, 3, dtype=torch.int64, device=dev)
    status = torch.zeros(8, dtype=torch.int32, device=dev)

    def call():
        eng.ctx.augment_batch_device(F, F * n, n, off.data_ptr(), rows.data_ptr(), 0, tids.data_ptr(), bench.BEAM_DIV, 0, plane.data_ptr(),
                                     0.7, 0, out.data_ptr(), src.data_ptr(), cnt.data_ptr(), st.data_ptr(), 0, status.data_ptr(), 0)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    # enqueue cost with an idle device (nothing to wait for) ...
    ts = []
    for _ in range(10):
        t = time.perf_counter(); call(); ts.append(time.perf_counter() - t); torch.cuda.synchronize()
    # ... and back to back
    t = time.perf_counter()
    for _ in range(20):
        call()
    enq = (time.perf_counter() - t) / 20
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t) / 20
    print(f"F={F}: enqueue idle {np.median(ts) * 1e3:.3f} ms, enqueue back-to-back {enq * 1e3:.3f} ms, device-paced {tot * 1e3:.3f} ms/call", flush=True)
