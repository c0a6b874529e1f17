"""Experiment: single-frame latency of snowgpu_augment_batch_device, plain launches vs a captured HIP graph."""
import sys, time, random
import numpy as np, torch
sys.path.insert(0, '/root/repo')
sys.argv = sys.argv[:1]
import bench
from lidar_snow_sim_amd import engine
from lidar_snow_sim_amd.synthetic import synthetic_sweep
eng = engine.get_engine(0)
tables = bench.make_tables(64)
dev = torch.device("cuda:0")
for F in (1, 4):
    frames = [synthetic_sweep(64, 2048, seed=1000 + f, intensity="lambert") for f in range(F)]
    n = frames[0].shape[0]
    rows = torch.from_numpy(np.concatenate(frames)).to(dev)
    off = torch.arange(F + 1, dtype=torch.int64, device=dev) * n
    tids = torch.tensor([eng.table_ids_from_arrays(tables, list(range(64))) for _ in range(F)], dtype=torch.int32, device=dev)
    plane = torch.tensor([[0.0, 0.0, -1.0, -1.7]] * F, dtype=torch.float64, device=dev)
    out = torch.empty_like(rows); src = torch.empty(F * n, dtype=torch.int32, device=dev)
    cnt = torch.zeros(F, dtype=torch.int64, device=dev); st = torch.zeros(F, 3, dtype=torch.int64, device=dev)
    status = torch.zeros(8, dtype=torch.int32, device=dev)
    s = torch.cuda.Stream()

    def call(stream):
        eng.ctx.augment_batch_device(F, F * n, n, off.data_ptr(), rows.data_ptr(), 0, tids.data_ptr(), bench.BEAM_DIV, 0, plane.data_ptr(),
                                     0.7, 0, out.data_ptr(), src.data_ptr(), cnt.data_ptr(), st.data_ptr(), 0, status.data_ptr(), stream.cuda_stream)

    with torch.cuda.stream(s):
        for _ in range(3):
            call(s)
        s.synchronize()
        t = time.perf_counter()
        for _ in range(50):
            call(s)
        s.synchronize()
        plain = (time.perf_counter() - t) / 50
        ref = (out.clone(), cnt.clone(), st.clone())
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                call(s)
            out.zero_(); cnt.zero_()
            g.replay(); s.synchronize()
            same = bool(torch.equal(cnt, ref[1]) and torch.equal(st, ref[2]) and torch.equal(out[:int(cnt[0])], ref[0][:int(cnt[0])]))
            t = time.perf_counter()
            for _ in range(50):
                g.replay()
            s.synchronize()
            graph = (time.perf_counter() - t) / 50
            print(f"F={F}: plain {plain * 1e3:.3f} ms/call, graph replay {graph * 1e3:.3f} ms/call, same result: {same}, status {status.tolist()}")
        except Exception as e:
            print(f"F={F}: plain {plain * 1e3:.3f} ms/call, graph capture failed: {type(e).__name__}: {str(e)[:200]}")
