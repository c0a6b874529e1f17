#!/usr/bin/env python3
"""Does a device-resident batch gain from running as two half batches on two contexts / streams (memory-bound head and tail of one
half beside the VALU-bound middle of the other)?   python scripts/probe/two_lane_probe.py [--frames 256] [--steps 20]"""
import argparse
import json
import random
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--workload", default="C2")
    args = ap.parse_args()
    import torch
    import bench
    from lidar_snow_sim_amd import engine
    layers, azimuths, snowfall, velocity, rscale = bench.WORKLOADS[args.workload]
    dev = torch.device("cuda:0")
    tables = bench.make_tables(layers, snowfall, velocity, distinct=min(layers, 64))
    F = args.frames
    frames, orders = [], []
    for f in range(F):
        seed = 1000 + f
        frames.append(bench.make_frame(layers, azimuths, seed, rscale))
        random.seed(seed)
        o = list(range(layers))
        random.shuffle(o)
        orders.append(o)
    n_per = frames[0].shape[0]

    class Half:
        def __init__(self, slot, lo, hi, stream):
            self.eng = engine.get_engine(0, slot)
            if layers != 64:
                self.eng.set_lasers(engine.load_lasers() * (layers // 64))
            self.F = hi - lo
            self.n_total = self.F * n_per
            self.rows = torch.from_numpy(np.concatenate(frames[lo:hi])).to(dev)
            self.off = torch.arange(0, self.F + 1, dtype=torch.int64, device=dev) * n_per
            self.tids = torch.tensor([self.eng.table_ids_from_arrays(tables, orders[f]) for f in range(lo, hi)], dtype=torch.int32, device=dev)
            self.plane = torch.tensor([[0.0, 0.0, -1.0, -1.7]] * self.F, dtype=torch.float64, device=dev)
            self.out_rows = torch.empty((self.n_total, 5), dtype=torch.float32, device=dev)
            self.out_src = torch.empty(self.n_total, dtype=torch.int32, device=dev)
            self.counts = torch.zeros(self.F, dtype=torch.int64, device=dev)
            self.stats = torch.zeros(self.F, 3, dtype=torch.int64, device=dev)
            self.status = torch.zeros(8, dtype=torch.int32, device=dev)
            self.stream = stream

        def step(self):
            self.eng.ctx.augment_batch_device(self.F, self.n_total, n_per, self.off.data_ptr(), self.rows.data_ptr(), 0, self.tids.data_ptr(),
                                              bench.BEAM_DIV, 0, self.plane.data_ptr(), 0.7, 0, self.out_rows.data_ptr(), self.out_src.data_ptr(),
                                              self.counts.data_ptr(), self.stats.data_ptr(), 0, self.status.data_ptr(), self.stream.cuda_stream)

    def timed(parts, steps):
        for _ in range(3):
            for p in parts:
                p.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            for p in parts:
                p.step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    s0, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    res = {}
    whole = Half(0, 0, F, s0)
    res["one_batch_ms"] = timed([whole], args.steps)
    a, b = Half(1, 0, F // 2, s1), Half(2, F // 2, F, s2)
    res["half_alone_ms"] = timed([a], args.steps)
    res["two_halves_two_streams_ms"] = timed([a, b], args.steps)
    # the halves staggered: b starts when a is half way (what a steady pipeline looks like)
    q = [Half(3 + i, i * (F // 4), (i + 1) * (F // 4), [s1, s2][i % 2]) for i in range(4)]
    res["four_quarters_two_streams_ms"] = timed(q, args.steps)
    whole2 = Half(7, 0, F, s1)
    res["two_batches_two_streams_ms_per_batch"] = timed([whole, whole2], args.steps) / 2
    res["digest"] = [int(whole.counts.sum()), int(a.counts.sum() + b.counts.sum())]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
