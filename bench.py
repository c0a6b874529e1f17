#!/usr/bin/env python3
"""bench.py -- snowfall-augmentation throughput on MI355X, BASELINE.json's metric on its config C2.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path -- channel sort, noise-threshold prepass, per-beam occlusion /
received-power simulation, noise-floor filter, compaction, statistics (augment(), simulation.py:427-544,
only_camera_fov=False) -- over one batch of F synthetic 64 x 2048 sweeps (2.5 mm/h @ 1.6 m/s, gunn
tables, R0 = 80 m) that already sit in HBM when the timed region starts.  Ranks own independent
batches (frames shard with no data-path collective): weak scaling, value = points of all ranks / max time.

Prints ONE JSON line on rank 0 with `roofline` (per-beam kernel, HIP events on its launch stream) and
`cpu_baseline` (the CPU oracle on four frames of the same workload, channels spread over the host cores; one core on frame 0 beside it).
"""
import argparse
import ctypes
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md
DEFAULT_FRAMES = 256       # sweeps per step and GPU (BASELINE.json's C3 streams 256-frame batches of the C2 sweep)
BEAM_DIV = float(np.degrees(3e-3))
SNOWFALL, VELOCITY = 2.5, 1.6


WORKLOADS = {   # name: (layers, azimuths, snowfall mm/h, terminal velocity m/s, range scale)  -- SURVEY 8 d
    "C2": (64, 2048, 2.5, 1.6, 1.0),       # BASELINE.json configs[1]: the headline workload
    "C2far": (64, 2048, 2.5, 1.6, 1.8),    # same sweep with every range stretched (clipped at 119 m): long scatterer lists
    "C1": (64, 2048, 0.5, 2.0, 1.0),       # configs[0]'s table density (40 k flakes per line)
    "C4": (128, 4096, 10.0, 1.6, 1.0),     # configs[3]: 128 x 4096 dense sweep, heavy snowfall, tiled laser table
}


def make_tables(n_lines=64, snowfall=SNOWFALL, velocity=VELOCITY, distinct=None):
    from lidar_snow_sim_amd.tools.snowfall import sampling as smp
    occ = smp.compute_occupancy(snowfall, velocity)
    rate = smp.snowfall_rate_to_rainfall_rate(snowfall, velocity)
    distinct = n_lines if distinct is None else distinct
    tabs = [smp.dart_throwing(occ, rate, 80.0, np.random.default_rng(42 + line), "gunn") for line in range(1, distinct + 1)]
    return [tabs[i % distinct] for i in range(n_lines)]


def make_frame(layers, azimuths, seed, scale):
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    pc = synthetic_sweep(layers, azimuths, seed=seed, intensity="lambert")
    if scale != 1.0:
        r = np.linalg.norm(pc[:, :3].astype(np.float64), axis=1)
        f = np.minimum(r * scale, 119.0) / r
        pc[:, :3] = (pc[:, :3] * f[:, None]).astype(np.float32)
    return pc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=DEFAULT_FRAMES, help="frames per batch (per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-prepass", action="store_true", help="feed precomputed threshold polynomials (debug)")
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS), help="C2 (default) is BASELINE.json's metric config")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from lidar_snow_sim_amd import _native, engine
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    from lidar_snow_sim_amd.tools.wet_ground.augmentation import noise_threshold_poly

    layers, azimuths, snowfall, velocity, rscale = WORKLOADS[args.workload]
    eng = engine.get_engine(local_rank)
    if layers != 64:                                       # SURVEY 8 d: 128-entry laser table = the 64-entry one tiled
        eng.set_lasers(engine.load_lasers() * (layers // 64))
    tables = make_tables(layers, snowfall, velocity, distinct=min(layers, 64))
    ktot = sum(t.shape[0] for t in tables)
    F = args.frames
    import random
    frames, table_ids, planes, polys = [], [], [], []
    plane = ([0.0, 0.0, -1.0], -1.7)
    for f in range(F):
        seed = 1000 + rank * F + f
        pc = make_frame(layers, azimuths, seed, rscale)
        random.seed(seed)
        order = list(range(layers))
        random.shuffle(order)
        frames.append(pc)
        table_ids.append(eng.table_ids_from_arrays(tables, order))
        planes.append([*plane[0], plane[1]])
        if args.host_prepass:
            polys.append(noise_threshold_poly(pc, plane[0], plane[1], 0.7))
    n_per = frames[0].shape[0]
    n_total = n_per * F
    rows = torch.from_numpy(np.concatenate(frames)).to(dev)
    off = torch.arange(0, F + 1, dtype=torch.int64, device=dev) * n_per
    tids = torch.tensor(table_ids, dtype=torch.int32, device=dev)
    d_plane = torch.tensor(planes, dtype=torch.float64, device=dev)
    d_poly = torch.tensor(np.asarray(polys), dtype=torch.float64, device=dev) if args.host_prepass else None
    out_rows = torch.empty_like(rows)
    out_src = torch.empty(n_total, dtype=torch.int32, device=dev)
    out_counts = torch.zeros(F, dtype=torch.int64, device=dev)
    out_stats = torch.zeros(F, 3, dtype=torch.int64, device=dev)
    status = torch.zeros(8, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        eng.ctx.augment_batch_device(F, n_total, n_per, off.data_ptr(), rows.data_ptr(), 0, tids.data_ptr(), BEAM_DIV,
                                     d_poly.data_ptr() if d_poly is not None else 0,
                                     0 if d_poly is not None else d_plane.data_ptr(), 0.7, 0,
                                     out_rows.data_ptr(), out_src.data_ptr(), out_counts.data_ptr(),
                                     out_stats.data_ptr(), 0, status.data_ptr(), stream)

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    st = status.cpu().numpy()
    if st[0] != 0:
        raise RuntimeError(f"device status {st} after warmup")
    eng.ctx.profile_begin(args.steps)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    beam_ms, n_launch = eng.ctx.profile_end()
    if distributed:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    st = status.cpu().numpy()
    if st[0] != 0:
        raise RuntimeError(f"device status {st} after the timed region")

    if rank == 0:
        pts_per_step = n_total * world
        value = pts_per_step * args.steps / elapsed
        # algorithmic bytes of one per-beam launch (SURVEY 8 d): 20 B read + 20 B written per point, and each
        # channel's K x 3 float64 table read once per frame
        alg_bytes = 40.0 * n_total + 24.0 * ktot * F
        avg_ms = beam_ms / max(n_launch, 1)
        achieved = alg_bytes / (avg_ms * 1e-3) if avg_ms > 0 else 0.0
        traffic = None
        pmc = ROOT / "profiles" / "hbm_traffic.json"
        if pmc.exists():
            try:
                rec = json.loads(pmc.read_text())
                if rec.get("frames") == F:
                    traffic = rec.get("bytes_per_launch")
            except Exception:
                traffic = None
        result = {
            "metric": "augmented points/s/GPU (64x2048 sweep, 2.5 mm/h); HBM roofline %",
            "value": value, "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: synthetic {layers}-layer x {azimuths}-azimuth sweeps, snowfall_rate={snowfall} mm/h, "
                                   f"terminal_velocity={velocity} m/s, gunn tables R0=80 m ({ktot // layers} flakes/line), "
                                   f"beam_divergence=3 mrad, noise_floor=0.7, float32 rows resident in HBM"
                                   + ("" if rscale == 1.0 else f", ranges x{rscale} (clipped at 119 m)"),
                       "frames_per_step_per_gpu": F, "points_per_frame": n_per,
                       "prepass": "host (outside the timed region)" if args.host_prepass else "device (timed)",
                       "sharding": f"frame-parallel x{world}, no collective",
                       "beams_per_capacity_tier": [int(n_total)] + [int(v) for v in st[2:5]]},
            "per_gpu_value": value / world,
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": traffic,
                         "kernel": "per-beam kernels of one step: k_beams<float,LMAX,BLOCK,LIST> (capacity tiers 4/8/16/63), k_power and the k_list_* builders between them, one HIP event pair around the region", "avg_launch_ms": avg_ms, "launches": n_launch,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "40 B/point + 24 B per flake per channel per frame (tables counted, 251.3 B/point)"},
        }
        if not args.no_cpu_baseline and world == 1:      # rank 0, N = 1 only
            from oracle import snow_oracle as so
            # (i) one host core on frame 0, (ii) all host cores (channels on a thread pool, as the reference's
            # ThreadPool(cpu_count()) path, simulation.py:498) on frames 0..3: ~12 s of CPU work in all
            cores = max(1, min(os.cpu_count() or 1, layers))
            n_cpu = min(4, F)
            same, cpu_s, one_s = True, 0.0, 0.0
            las = so.load_lasers() * (layers // 64)
            for fi in range(n_cpu):
                random.seed(1000 + fi)
                order = list(range(layers))
                random.shuffle(order)
                poly = noise_threshold_poly(frames[fi], plane[0], plane[1], 0.7)
                if fi == 0:
                    c0 = time.perf_counter()
                    so.augment(frames[fi], tables, BEAM_DIV, order, plane=plane, thr_poly=poly, lasers=las)
                    one_s = time.perf_counter() - c0
                c0 = time.perf_counter()
                s_ref, a_ref, src_ref = so.augment(frames[fi], tables, BEAM_DIV, order, plane=plane, thr_poly=poly,
                                                   lasers=las, threads=cores)
                cpu_s += time.perf_counter() - c0
                n0 = int(out_counts[fi].item())
                lo = fi * n_per
                got = out_rows[lo:lo + n0].cpu().numpy()
                got_src = out_src[lo:lo + n0].cpu().numpy()
                same = same and (n0 == a_ref.shape[0] and np.array_equal(got_src, src_ref)
                                 and np.array_equal(got[:, 3:], a_ref[:, 3:])
                                 and np.allclose(got[:, :3], a_ref[:, :3], rtol=1e-6, atol=0))
            result["cpu_baseline"] = {"value": n_cpu * n_per / cpu_s, "unit": "points/s", "cores": cores, "kind": "port",
                                      "sample": f"frames 0..{n_cpu - 1} of the batch ({n_cpu * n_per} points), "
                                                f"oracle/snow_oracle.c (scalar C restatement, per-beam scan of the whole "
                                                f"table) with the NumPy frame driver, channels on {cores} threads, "
                                                f"{cpu_s:.1f} s wall; one core on frame 0: {n_per / one_s:.0f} points/s",
                                      "single_core_value": n_per / one_s,
                                      "gpu_output_matches": bool(same)}
        print(json.dumps(result), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
