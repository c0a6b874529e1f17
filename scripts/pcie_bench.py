#!/usr/bin/env python3
"""PCIe-inclusive throughput of the C-ABI host entry (snowgpu_augment_batch) in a process WITHOUT PyTorch.

    python scripts/pcie_bench.py [--frames 256] [--reps 4] [--workload C2]

The frames sit in page-locked memory; one call per step; upload, all kernels and download inside the clock.  bench.py runs
this as a child process for its `value_pcie_inclusive` leg: inside a process that has initialised PyTorch, the HIP runtime
moves device-to-host copies with a blit kernel instead of the DMA engine (traced), which stalls the kernels beside it."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def pin_to_device_node(device):
    """sched_setaffinity to the CPUs of the NUMA node HIP device `device` hangs on (snowgpu_device_numa_node: sysfs by PCI bus id), as a
    launcher would place a rank.  Returns a description of the placement, or None if the node cannot be told."""
    import os
    from lidar_snow_sim_amd import _native
    try:
        node = int(_native.lib().snowgpu_device_numa_node(int(device)))
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return f"node {node}: {len(cpus)} cpus"
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--seed-base", type=int, default=1000)
    ap.add_argument("--fast", action="store_true", help="the two headline figures only (A/B runs)")
    ap.add_argument("--numa", default="device", choices=("device", "none"),
                    help="device (default): run on the CPUs of the NUMA node the GPU hangs on, as a launcher would place a rank (numactl): on a "
                         "two-socket host page-locked buffers touched from the far socket cost the packed transfer a quarter of its rate")
    args = ap.parse_args()
    affinity = None
    if args.numa == "device":
        affinity = pin_to_device_node(args.device)
    import bench                      # helpers only: bench.py imports torch lazily, inside main()
    assert "torch" not in sys.modules
    import random
    from lidar_snow_sim_amd import engine
    layers, azimuths, snowfall, velocity, rscale = bench.WORKLOADS[args.workload]
    eng = engine.get_engine(args.device)
    if layers != 64:
        eng.set_lasers(engine.load_lasers() * (layers // 64))
    tables = bench.make_tables(layers, snowfall, velocity, distinct=min(layers, 64))
    F = args.frames
    frames, ids = [], []
    for f in range(F):
        seed = args.seed_base + f
        frames.append(bench.make_frame(layers, azimuths, seed, rscale, args.workload in bench.FIRING_ORDER))
        random.seed(seed)
        order = list(range(layers))
        random.shuffle(order)
        ids.append(eng.table_ids_from_arrays(tables, order))
    n_per = frames[0].shape[0]
    n_total = n_per * F
    pin_in = eng.ctx.pinned_empty((n_total, 5), np.float32)
    np.concatenate(frames, out=pin_in)
    pin_out = eng.ctx.pinned_empty((n_total, 5), np.float32)
    pin_src = eng.ctx.pinned_empty(n_total, np.int32)
    off = np.arange(F + 1, dtype=np.int64) * n_per
    h_ids = np.asarray(ids, np.int32)
    planes = np.asarray([[0.0, 0.0, -1.0, -1.7]] * F)

    def call(want_src, plane="injected"):
        return eng.ctx.augment_batch(pin_in, off, h_ids, bench.BEAM_DIV, plane=planes if plane == "injected" else None, out_rows=pin_out,
                                     out_src=pin_src, want_src=want_src)

    def timed(want_src, plane="injected"):
        call(want_src, plane)
        best = 1e9
        t0 = time.perf_counter()
        for _ in range(args.reps):
            c0 = time.perf_counter()
            call(want_src, plane)
            best = min(best, time.perf_counter() - c0)
        return (time.perf_counter() - t0) / args.reps, best

    s_src, b_src = timed(True)
    _, _, counts, stats, _ = call(True)
    digest = [int(counts.sum()), int(stats[:, 0].sum()), int(stats[:, 1].sum()), int(stats[:, 2].sum()),
              float(pin_out[:int(counts[0]), 3].sum()), int(pin_src[:int(counts[0])].astype(np.int64).sum())]
    s_nosrc, b_nosrc = timed(False)
    # packed result transfer (snowgpu_set_result_transfer): source row | label + intensity per kept row over the link, the library's host
    # threads assemble the caller's rows from those and from the input rows -- same bytes in pin_out / pin_src
    packed = {}
    for thr in ([0] if args.fast else [0, 8, 4]):
        eng.ctx.set_result_transfer("packed", thr)
        try:
            s_pk, b_pk = timed(True)
            _, _, c2, st2, _ = call(True)
            tt = eng.ctx.transfer_times()
            dg = [int(c2.sum()), int(st2[:, 0].sum()), int(st2[:, 1].sum()), int(st2[:, 2].sum()),
                  float(pin_out[:int(c2[0]), 3].sum()), int(pin_src[:int(c2[0])].astype(np.int64).sum())]
            s_pkn, _ = timed(False)
            packed[str(tt["host_threads"])] = {"points_per_s": n_total / s_pk, "points_per_s_best": n_total / b_pk, "points_per_s_without_src": n_total / s_pkn,
                                               "last_call_ms": tt, "same_digest_as_rows_mode": dg == digest}
        finally:
            eng.ctx.set_result_transfer("rows")
    # compact input (snowgpu_augment_batch_compact): (x, y, z, intensity) float32 + one channel byte per row up the link -- 17 B per point
    # instead of 20 --, packed result transfer down; the caller's out_rows / out_src receive the same bytes
    compact = None
    try:
        pin_x4 = eng.ctx.pinned_empty((n_total, 4), np.float32)
        pin_x4[...] = pin_in[:, :4]
        pin_ch = eng.ctx.pinned_empty(n_total, np.uint8)
        pin_ch[...] = pin_in[:, 4].astype(np.uint8)

        def call_c(want_src):
            return eng.ctx.augment_batch_compact(pin_x4, pin_ch, off, h_ids, bench.BEAM_DIV, plane=planes, out_rows=pin_out, out_src=pin_src, want_src=want_src)

        compact = {}
        for mode in ("rows", "packed"):
            eng.ctx.set_result_transfer(mode)
            try:
                call_c(True)
                t0 = time.perf_counter()
                for _ in range(args.reps):
                    call_c(True)
                s_c = (time.perf_counter() - t0) / args.reps
                _, _, c3, st3, _ = call_c(True)
                dg = [int(c3.sum()), int(st3[:, 0].sum()), int(st3[:, 1].sum()), int(st3[:, 2].sum()),
                      float(pin_out[:int(c3[0]), 3].sum()), int(pin_src[:int(c3[0])].astype(np.int64).sum())]
                compact[mode] = {"points_per_s": n_total / s_c, "same_digest_as_rows_mode": dg == digest}
            finally:
                eng.ctx.set_result_transfer("rows")
        compact["bytes_per_point_up"] = 17
    except Exception as ex:
        compact = {"error": repr(ex)}
    if args.fast:
        print(json.dumps({"points_per_s": n_total / s_src, "points_per_s_without_src": n_total / s_nosrc, "packed": packed, "compact_input": compact, "process_affinity": affinity}))
        return
    # plane = NULL at the C ABI: calculate_plane (simulation.py:449) on the device inside the batch -- the reference's default call
    s_ref, _ = timed(True, "device")                                     # method 'reference': the plane the reference returns today
    eng.ctx.set_plane_method("lsq")
    s_lsq, _ = timed(True, "device")                                     # method 'lsq': one more pass over the rows + one block per frame
    eng.ctx.set_plane_method("reference")
    # ... and through the Python entry the reference's callers use (augment_batch: per-frame order / table-id bookkeeping on the host)
    from lidar_snow_sim_amd.tools.snowfall.simulation import FlatBatch, augment_batch
    orders = []
    for f in range(F):
        random.seed(args.seed_base + f)
        o = list(range(layers))
        random.shuffle(o)
        orders.append(o)
    fb = FlatBatch(pin_in, off)

    def py_batch(warm=1, **kw):
        for _ in range(warm):
            augment_batch(fb, "unused", bench.BEAM_DIV, particles=tables, orders=orders, **kw)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            augment_batch(fb, "unused", bench.BEAM_DIV, particles=tables, orders=orders, **kw)
        return (time.perf_counter() - t0) / args.reps

    py_inj = py_batch(planes=[([0.0, 0.0, -1.0], -1.7)] * F)
    py_ref = py_batch()
    py_lsq = py_batch(plane_method="lsq")
    # q8='numpy': the histogram's row minima from THIS process' NumPy (quirk Q8); the device makes histogram and sums (prepass_stats),
    # the rows cross the link twice
    # (six untimed calls first: the selection's thread pool, its per-thread scratch and the allocator's thresholds for NumPy's 1 MB index
    # arrays settle over the first few calls -- scripts/probe/q8_group_probe.py shows the first timed run of a process 20 % below the later ones)
    py_q8 = py_batch(warm=6, planes=[([0.0, 0.0, -1.0], -1.7)] * F, q8="numpy")
    # one sweep end to end through the C ABI (page-locked) and through the Python augment() (pageable input)
    one_off = np.array([0, n_per], np.int64)

    def one_abi():
        eng.ctx.augment_batch(pin_in[:n_per], one_off, h_ids[:1], bench.BEAM_DIV, plane=planes[:1], out_rows=pin_out[:n_per], out_src=pin_src[:n_per])

    def med(fn, n=40):
        for _ in range(5):
            fn()
        ts = []
        for _ in range(n):
            c0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - c0)
        return float(np.median(ts) * 1e3), float(np.min(ts) * 1e3)

    abi_ms, abi_min = med(one_abi)
    from lidar_snow_sim_amd.tools.snowfall.simulation import augment as py_augment
    pageable = np.array(frames[0])
    order0 = list(range(layers))
    random.seed(args.seed_base)
    random.shuffle(order0)

    def one_py():
        py_augment(pageable, "unused", bench.BEAM_DIV, only_camera_fov=False, plane=([0.0, 0.0, -1.0], -1.7), order=order0, particles=tables)

    py_ms, py_min = med(one_py)

    def one_py_default():                                                # the literal reference call: no plane, no order
        py_augment(pageable, "unused", bench.BEAM_DIV, only_camera_fov=False, particles=tables)

    def one_py_lsq():
        py_augment(pageable, "unused", bench.BEAM_DIV, only_camera_fov=False, particles=tables, plane_method="lsq")

    pyd_ms, pyd_min = med(one_py_default)
    pyl_ms, pyl_min = med(one_py_lsq)
    print(json.dumps({"points_per_s": n_total / s_src, "points_per_s_best": n_total / b_src, "points_per_s_without_src": n_total / s_nosrc,
                      "points_per_s_without_src_best": n_total / b_nosrc, "frames": F, "reps": args.reps, "points_per_frame": n_per,
                      "digest": digest, "packed": packed, "compact_input": compact, "process_affinity": affinity, "single_frame_c_abi_ms": abi_ms, "single_frame_c_abi_min_ms": abi_min,
                      "single_frame_python_ms": py_ms, "single_frame_python_min_ms": py_min, "torch_loaded": "torch" in sys.modules,
                      "default_plane": {"c_abi_points_per_s_reference": n_total / s_ref, "c_abi_points_per_s_lsq": n_total / s_lsq,
                                        "python_points_per_s_injected": n_total / py_inj, "python_points_per_s_reference": n_total / py_ref,
                                        "python_points_per_s_lsq": n_total / py_lsq},
                      "q8_numpy": {"python_points_per_s": n_total / py_q8, "share_of_q8_first": py_inj / py_q8,
                                   "note": "augment_batch(..., q8='numpy') vs the same call with q8='first' (python_points_per_s_injected): "
                                           "device half of the prepass + np.argpartition of 50 x 2555 float64 rows per frame on a thread pool"},
                      "single_frame_python_default_ms": pyd_ms, "single_frame_python_default_min_ms": pyd_min,
                      "single_frame_python_lsq_ms": pyl_ms, "single_frame_python_lsq_min_ms": pyl_min}))


if __name__ == "__main__":
    main()
