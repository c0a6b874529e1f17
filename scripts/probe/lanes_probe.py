#!/usr/bin/env python3
"""Two (or four) compute lanes for a device-resident batch, each lane ONE stream (SNOWGPU_SERIAL=1: no side streams, so no two streams
share a hardware queue): does the memory-bound head / tail of one part hide behind the per-beam kernels of another?
    python scripts/probe/lanes_probe.py [--frames 256] [--steps 20] [--workload C2]"""
import argparse
import json
import os
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
        frames.append(bench.make_frame(layers, azimuths, seed, rscale, args.workload in bench.FIRING_ORDER))
        random.seed(seed)
        o = list(range(layers))
        random.shuffle(o)
        orders.append(o)
    n_per = frames[0].shape[0]
    slot = [0]

    class Part:
        def __init__(self, lo, hi, serial, pool=None):
            if serial:
                os.environ["SNOWGPU_SERIAL"] = "1"
            else:
                os.environ.pop("SNOWGPU_SERIAL", None)
            slot[0] += 1
            self.eng = engine.Engine(0)
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

        def step(self):
            self.eng.ctx.augment_batch_device(self.F, self.n_total, n_per, self.off.data_ptr(), self.rows.data_ptr(), 0, self.tids.data_ptr(),
                                              bench.BEAM_DIV, 0, self.plane.data_ptr(), 0.7, 0, self.out_rows.data_ptr(), self.out_src.data_ptr(),
                                              self.counts.data_ptr(), self.stats.data_ptr(), 0, self.status.data_ptr(), 0)

        def close(self):
            self.eng.ctx.close()

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

    res = {}

    def run(name, cuts, serial, div=1.0, pools=None):
        parts = [Part(lo, hi, serial, pools[i] if pools else None) for i, (lo, hi) in enumerate(cuts)]
        res[name] = round(timed(parts, args.steps) / div, 3)
        n = int(sum(int(p.counts.sum()) for p in parts))
        for p in parts:
            p.close()
        return n

    d0 = run("one_batch_4streams_ms", [(0, F)], False)
    run("one_batch_serial_ms", [(0, F)], True)
    run("two_halves_serial_ms", [(0, F // 2), (F // 2, F)], True)
    run("two_batches_serial_ms_per_batch", [(0, F), (0, F)], True, 2.0)
    run("two_halves_4streams_each_ms", [(0, F // 2), (F // 2, F)], False)
    run("two_batches_4streams_each_ms_per_batch", [(0, F), (0, F)], False, 2.0)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
