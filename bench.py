#!/usr/bin/env python3
"""bench.py -- snowfall-augmentation throughput on MI355X, BASELINE.json's metric on its config C2.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F] [--workload C2|C2far|C1|C4|C3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path -- channel sort, noise-threshold prepass, per-beam occlusion /
received-power simulation, noise-floor filter, compaction, statistics (augment(), simulation.py:427-544,
only_camera_fov=False; workload C3 chains ground_water_augmentation() behind it, pointcloud_viewer.py:2807-2821) --
over one batch of F synthetic sweeps that already sit in HBM when the timed region starts.  Ranks own independent
batches (frames shard with no data-path collective): weak scaling, value = points of all ranks / max time.

Prints ONE JSON line on rank 0:
  value                  points/s with rows resident in HBM (the device entry of the C ABI)
  value_pcie_inclusive   the same frames through the HOST entry: H2D of the rows and D2H of the results inside the clock
                         (page-locked buffers, two contexts), SURVEY 8(d)'s definition of the metric; never `value`
  roofline               per-beam region: achieved = algorithmic bytes / HIP-event time; `traffic` and `valu` from rocprofv3
                         --pmc passes of this same command, run as child processes (N = 1 only; --no-pmc skips them)
  cpu_baseline           the CPU oracle on four frames of the same workload (all host cores, and one core)
"""
import argparse
import csv
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md
PCIE_PEAK = 63.0e9         # B/s per direction, PCIe Gen5 x16 (same guide)
DEFAULT_FRAMES = 256       # sweeps per step and GPU (BASELINE.json's C3 streams 256-frame batches of the C2 sweep)
BEAM_DIV = float(np.degrees(3e-3))
SNOWFALL, VELOCITY = 2.5, 1.6
WET = dict(water_height=0.0008, pavement_depth=0.001, noise_floor=0.7, power_factor=15.0, flat_earth=False, delta=0.5, replace=False)


WORKLOADS = {   # name: (layers, azimuths, snowfall mm/h, terminal velocity m/s, range scale)  -- SURVEY 8 d
    "C2": (64, 2048, 2.5, 1.6, 1.0),       # BASELINE.json configs[1]: the headline workload
    "C2far": (64, 2048, 2.5, 1.6, 1.8),    # same sweep with every range stretched (clipped at 119 m): long scatterer lists
    "C1": (64, 2048, 0.5, 2.0, 1.0),       # configs[0]'s table density (40 k flakes per line)
    "C4": (128, 4096, 10.0, 1.6, 1.0),     # configs[3]: 128 x 4096 dense sweep, heavy snowfall, tiled laser table
    "C3": (64, 2048, 2.5, 1.6, 1.0),       # configs[2]: the C2 sweeps through snowfall + wet ground, fused on the device
}
REGION_KERNELS = ("k_beams", "k_power", "k_tier")    # the per-beam region of roofline.avg_launch_ms


def make_tables(n_lines=64, snowfall=SNOWFALL, velocity=VELOCITY, distinct=None):
    """dart_throwing(occupancy, rain rate, 80 m, default_rng(42 + line), 'gunn') per line (SURVEY 8 d); cached on disk for
    the child processes of the counter passes (same seeds, same tables)."""
    from lidar_snow_sim_amd.tools.snowfall import sampling as smp
    occ = smp.compute_occupancy(snowfall, velocity)
    rate = smp.snowfall_rate_to_rainfall_rate(snowfall, velocity)
    distinct = n_lines if distinct is None else distinct
    cache = Path(tempfile.gettempdir()) / f"snowgpu_bench_tables_{snowfall}_{velocity}_{distinct}.npz"
    tabs = None
    if cache.exists():
        try:
            z = np.load(cache)
            tabs = [z[f"t{i}"] for i in range(distinct)]
        except Exception:
            tabs = None
    if tabs is None:
        tabs = [smp.dart_throwing(occ, rate, 80.0, np.random.default_rng(42 + line), "gunn") for line in range(1, distinct + 1)]
        try:
            tmp = cache.with_suffix(f".{os.getpid()}.npz")
            np.savez(tmp, **{f"t{i}": t for i, t in enumerate(tabs)})
            os.replace(tmp, cache)
        except OSError:
            pass
    return [tabs[i % distinct] for i in range(n_lines)]


def make_frame(layers, azimuths, seed, scale):
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    pc = synthetic_sweep(layers, azimuths, seed=seed, intensity="lambert")
    if scale != 1.0:
        r = np.linalg.norm(pc[:, :3].astype(np.float64), axis=1)
        f = np.minimum(r * scale, 119.0) / r
        pc[:, :3] = (pc[:, :3] * f[:, None]).astype(np.float32)
    return pc


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def short_kernel(name):
    return re.sub(r"\(.*", "", name).replace("void ", "")


def pmc_pass(counters, argv, steps_total):
    """One rocprofv3 --pmc pass of this script (child process, --inner).  Returns {kernel: {counter: mean per step}}."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not Path(exe).exists():
        return None
    out = tempfile.mkdtemp(prefix="snowgpu_pmc_")
    try:
        cmd = [exe, "--pmc", *counters, "--kernel-trace", "-d", out, "-o", "b", "--output-format", "csv", "--",
               sys.executable, str(ROOT / "bench.py"), "--inner", *argv]
        env = dict(os.environ, TMPDIR=tempfile.gettempdir())
        r = subprocess.run(cmd, cwd=tempfile.gettempdir(), env=env, capture_output=True, text=True, timeout=600)
        files = list(Path(out).rglob("*counter_collection.csv"))
        if r.returncode != 0 or not files:
            return None
        acc = {}
        with open(files[0], newline="") as fh:
            for row in csv.DictReader(fh):
                k = short_kernel(row["Kernel_Name"])
                acc.setdefault(k, {}).setdefault(row["Counter_Name"], 0.0)
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        return {k: {c: v / steps_total for c, v in d.items()} for k, d in acc.items()}
    except Exception:
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=DEFAULT_FRAMES, help="frames per batch (per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes (roofline.traffic / valu from profiles/)")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-entry (PCIe-inclusive) measurement")
    ap.add_argument("--inner", action="store_true", help=argparse.SUPPRESS)       # child of a counter pass: timed loop only
    ap.add_argument("--host-prepass", action="store_true", help="feed precomputed threshold polynomials (debug)")
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS), help="C2 (default) is BASELINE.json's metric config")
    args = ap.parse_args()
    if args.inner:
        args.no_cpu_baseline = args.no_pmc = args.no_pcie = True

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

    from lidar_snow_sim_amd import engine
    from lidar_snow_sim_amd.tools.wet_ground.augmentation import noise_threshold_poly

    layers, azimuths, snowfall, velocity, rscale = WORKLOADS[args.workload]
    fused_wet = args.workload == "C3"
    eng = engine.get_engine(local_rank)
    if layers != 64:                                       # SURVEY 8 d: 128-entry laser table = the 64-entry one tiled
        eng.set_lasers(engine.load_lasers() * (layers // 64))
    tables = make_tables(layers, snowfall, velocity, distinct=min(layers, 64))
    ktot = sum(t.shape[0] for t in tables)
    F = args.frames
    import random
    frames, orders, table_ids, planes, polys = [], [], [], [], []
    plane = ([0.0, 0.0, -1.0], -1.7)
    for f in range(F):
        seed = 1000 + rank * F + f
        pc = make_frame(layers, azimuths, seed, rscale)
        random.seed(seed)
        order = list(range(layers))
        random.shuffle(order)
        frames.append(pc)
        orders.append(order)
        table_ids.append(eng.table_ids_from_arrays(tables, order))
        planes.append([*plane[0], plane[1]])
        if args.host_prepass:
            polys.append(noise_threshold_poly(pc, plane[0], plane[1], 0.7))
    n_per = frames[0].shape[0]
    n_total = n_per * F
    host_rows = np.concatenate(frames)
    rows = torch.from_numpy(host_rows).to(dev)
    off = torch.arange(0, F + 1, dtype=torch.int64, device=dev) * n_per
    tids = torch.tensor(table_ids, dtype=torch.int32, device=dev)
    d_plane = torch.tensor(planes, dtype=torch.float64, device=dev)
    d_poly = torch.tensor(np.asarray(polys), dtype=torch.float64, device=dev) if args.host_prepass else None
    out_rows = torch.empty((n_total, 5), dtype=torch.float64 if fused_wet else torch.float32, device=dev)
    out_src = torch.empty(n_total, dtype=torch.int32, device=dev)
    out_counts = torch.zeros(F, dtype=torch.int64, device=dev)
    out_stats = torch.zeros(F, 3, dtype=torch.int64, device=dev)
    out_flags = torch.zeros(F, dtype=torch.int32, device=dev)
    status = torch.zeros(8, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        poly_ptr = d_poly.data_ptr() if d_poly is not None else 0
        plane_ptr = 0 if d_poly is not None else d_plane.data_ptr()
        if fused_wet:
            eng.ctx.augment_wet_batch_device(F, n_total, n_per, off.data_ptr(), rows.data_ptr(), 0, tids.data_ptr(), BEAM_DIV, poly_ptr,
                                             plane_ptr, 0.7, 0, d_plane.data_ptr(), WET["water_height"], WET["pavement_depth"],
                                             WET["noise_floor"], WET["power_factor"], WET["flat_earth"], WET["delta"], WET["replace"],
                                             out_rows.data_ptr(), out_src.data_ptr(), out_counts.data_ptr(), out_stats.data_ptr(),
                                             out_flags.data_ptr(), status.data_ptr(), stream)
        else:
            eng.ctx.augment_batch_device(F, n_total, n_per, off.data_ptr(), rows.data_ptr(), 0, tids.data_ptr(), BEAM_DIV, poly_ptr,
                                         plane_ptr, 0.7, 0, out_rows.data_ptr(), out_src.data_ptr(), out_counts.data_ptr(),
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
    if args.inner:
        return

    # ---- the same frames through the HOST entry: H2D + D2H inside the clock (SURVEY 8 d; precompute.py:78 / :106 are the
    # reference's boundary).  Two host threads drive two contexts with page-locked buffers, 32-frame sub-batches: the
    # copies of one overlap the kernels of the other.
    pcie = None
    if not args.no_pcie and not fused_wet:
        sub = min(32, F)
        n_sub = F // sub
        engs = [eng, engine.get_engine(local_rank, 1)]
        if layers != 64:
            engs[1].set_lasers(engine.load_lasers() * (layers // 64))
        # the frames sit in page-locked memory, as they do when the application reads its .bin files into such a buffer
        # (precompute.py:78 np.fromfile -> readinto); results land in per-context page-locked buffers
        pin_in = eng.ctx.pinned_empty((n_total, 5), np.float32)
        pin_in[...] = host_rows
        bufs = [(e.ctx.pinned_empty((sub * n_per, 5), np.float32), e.ctx.pinned_empty(sub * n_per, np.int32)) for e in engs]
        sub_off = np.arange(sub + 1, dtype=np.int64) * n_per
        ids2 = [[e.table_ids_from_arrays(tables, o) for o in orders] for e in engs]

        def host_worker(w, reps):
            e, (rout, rsrc) = engs[w], bufs[w]
            for _ in range(reps):
                for b in range(w, n_sub, 2):
                    e.ctx.augment_batch(pin_in[b * sub * n_per:(b + 1) * sub * n_per], sub_off, ids2[w][b * sub:(b + 1) * sub], BEAM_DIV,
                                        plane=planes[:sub], out_rows=rout, out_src=rsrc)

        for w in (0, 1):
            host_worker(w, 1)
        reps = max(1, min(args.steps, 4))
        barrier()
        c0 = time.perf_counter()
        th = [threading.Thread(target=host_worker, args=(w, reps)) for w in (0, 1)]
        [x.start() for x in th]
        [x.join() for x in th]
        torch.cuda.synchronize()
        pcie_s = time.perf_counter() - c0
        if distributed:
            tt = torch.tensor([pcie_s], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            pcie_s = float(tt.item())
        pcie = {"value": reps * n_sub * sub * n_per * world / pcie_s, "steps": reps, "frames_per_call": sub, "contexts": 2,
                "bytes_per_point": {"h2d": 20, "d2h": 24},
                "link_bound_points_per_s": PCIE_PEAK / 24.0 * world,
                "note": "snowgpu_augment_batch (host pointers) on frames held in page-locked memory: H2D, all kernels, D2H of rows + source "
                        "indices, one synchronisation per call; ceiling = 63 GB/s per direction / 24 B per point (D2H side)"}

    if rank == 0:
        pts_per_step = n_total * world
        value = pts_per_step * args.steps / elapsed
        # ALGORITHMIC bytes of one per-beam launch (SURVEY 8 d): 20 B read + 20 B written per point; the flake tables
        # (24 B per flake per channel per frame) stay in L2 / Infinity Cache across a batch (profiles/: the region fetches
        # far less than one pass over them per frame), so the table term is dropped -- the figure with it is given beside.
        alg_bytes = 40.0 * n_total
        alg_bytes_tables = alg_bytes + 24.0 * ktot * F
        avg_ms = beam_ms / max(n_launch, 1)
        achieved = alg_bytes / (avg_ms * 1e-3) if avg_ms > 0 else 0.0
        traffic, traffic_src, valu = None, None, None
        inner_argv = ["--steps", "2", "--warmup", "1", "--frames", str(F), "--workload", args.workload]
        if not args.no_pmc and world == 1:
            fetch = pmc_pass(["FETCH_SIZE"], inner_argv, 3)
            write = pmc_pass(["WRITE_SIZE"], inner_argv, 3) if fetch else None
            sq = pmc_pass(["SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES"],
                          inner_argv, 3) if write else None
            if fetch and write:
                in_region = lambda k: k.startswith(REGION_KERNELS)      # noqa: E731
                fb = sum(v.get("FETCH_SIZE", 0.0) for k, v in fetch.items() if in_region(k)) * 1024
                wb = sum(v.get("WRITE_SIZE", 0.0) for k, v in write.items() if in_region(k)) * 1024
                traffic = fb + wb
                traffic_src = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate child passes of this command (--steps 2 --warmup 1), "
                                         "kernels k_beams* / k_power* / k_tier*, raw counters x 1024 B (4- and 8-byte accesses: the gfx950 "
                                         "x2 correction for wide coalesced reads does not apply)",
                               "fetch_bytes": fb, "write_bytes": wb,
                               "whole_step_bytes": (sum(v.get("FETCH_SIZE", 0.0) for v in fetch.values()) + sum(v.get("WRITE_SIZE", 0.0) for v in write.values())) * 1024}
            if sq:
                dom = max((k for k in sq if k.startswith("k_beams")), key=lambda k: sq[k].get("SQ_INSTS_VALU", 0.0), default=None)
                if dom:
                    d = sq[dom]
                    valu = {"kernel": dom,
                            "lane_utilisation": d["SQ_THREAD_CYCLES_VALU"] / (64.0 * d["SQ_ACTIVE_INST_VALU"]) if d.get("SQ_ACTIVE_INST_VALU") else None,
                            "valu_instructions_per_wave": d["SQ_INSTS_VALU"] / d["SQ_WAVES"] if d.get("SQ_WAVES") else None,
                            "valu_issue_share_of_wave_cycles": d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"] if d.get("SQ_WAVE_CYCLES") else None,
                            "note": "SQ counters of the dominant kernel (rocprofv3 --pmc child pass): the path is bound by VALU issue and "
                                    "latency, not by HBM -- these are the figures to read beside frac"}
        if traffic is None:
            pmc = ROOT / "profiles" / "hbm_traffic.json"
            if pmc.exists():
                try:
                    rec = json.loads(pmc.read_text())
                    if rec.get("frames") == F and rec.get("workload", "C2") == args.workload:
                        traffic = rec.get("bytes_per_launch")
                        traffic_src = {"source": "profiles/hbm_traffic.json (committed rocprofv3 passes of this command)"}
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
                                   + ("" if rscale == 1.0 else f", ranges x{rscale} (clipped at 119 m)")
                                   + (", snowfall + wet ground fused (snowgpu_augment_wet_batch_device)" if fused_wet else ""),
                       "frames_per_step_per_gpu": F, "points_per_frame": n_per,
                       "prepass": "host (outside the timed region)" if args.host_prepass else "device (timed)",
                       "sharding": f"frame-parallel x{world}, no collective",
                       "beams_per_capacity_tier": [int(n_total)] + [int(v) for v in st[2:6]]},
            "per_gpu_value": value / world,
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": traffic, "traffic_detail": traffic_src,
                         "frac_tables_counted": alg_bytes_tables / (avg_ms * 1e-3) / HBM_PEAK if avg_ms > 0 else 0.0,
                         "valu": valu,
                         "kernel": "per-beam region of one step: k_beams<float,LMAX,BLOCK,LIST,DICT> (the pass over all rows and the later capacity "
                                   "tiers), k_power_plan + k_power (received power), k_tier_* (tier lists), k_beams_huge; one HIP event pair "
                                   "around the region on its launch stream",
                         "avg_launch_ms": avg_ms, "launches": n_launch,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "40 B/point (20 read + 20 written); tables cache-resident across the batch, so SURVEY 8(d)'s table term "
                                 "(24 B per flake per channel per frame) is dropped -- frac_tables_counted keeps it (251.3 B/point on C2)"},
        }
        if pcie is not None:
            result["value_pcie_inclusive"] = pcie["value"]
            result["pcie_inclusive"] = pcie
        if not args.no_cpu_baseline and world == 1:      # rank 0, N = 1 only
            from oracle import snow_oracle as so
            # (i) one host core on frame 0, (ii) all host cores (channels on a thread pool, as the reference's
            # ThreadPool(cpu_count()) path, simulation.py:498) on frames 0..3: ~12 s of CPU work in all
            cores = max(1, min(os.cpu_count() or 1, layers))
            n_cpu = min(4, F)
            same, cpu_s, one_s = True, 0.0, 0.0
            las = so.load_lasers() * (layers // 64)
            for fi in range(n_cpu):
                poly = noise_threshold_poly(frames[fi], plane[0], plane[1], 0.7)
                if fi == 0:
                    c0 = time.perf_counter()
                    so.augment(frames[fi], tables, BEAM_DIV, orders[fi], plane=plane, thr_poly=poly, lasers=las)
                    one_s = time.perf_counter() - c0
                c0 = time.perf_counter()
                s_ref, a_ref, src_ref = so.augment(frames[fi], tables, BEAM_DIV, orders[fi], plane=plane, thr_poly=poly,
                                                   lasers=las, threads=cores)
                if fused_wet:
                    a_ref, wsrc = so.ground_water_augmentation(a_ref, water_height=WET["water_height"], pavement_depth=WET["pavement_depth"],
                                                               noise_floor=WET["noise_floor"], power_factor=WET["power_factor"],
                                                               flat_earth=WET["flat_earth"], delta=WET["delta"], replace=WET["replace"],
                                                               plane=plane, return_src=True)
                    src_ref = src_ref[wsrc]
                cpu_s += time.perf_counter() - c0
                n0 = int(out_counts[fi].item())
                lo = fi * n_per
                got = out_rows[lo:lo + n0].cpu().numpy()
                got_src = out_src[lo:lo + n0].cpu().numpy()
                ok = n0 == a_ref.shape[0] and np.array_equal(got_src, src_ref) and np.array_equal(got[:, 4], a_ref[:, 4]) \
                    and np.allclose(got[:, :3], a_ref[:, :3], rtol=1e-6, atol=0)
                ok = ok and (np.allclose(got[:, 3], a_ref[:, 3], rtol=1e-6, atol=0) if fused_wet else np.array_equal(got[:, 3], a_ref[:, 3]))
                same = same and bool(ok)
            result["cpu_baseline"] = {"value": n_cpu * n_per / cpu_s, "unit": "points/s", "cores": cores, "kind": "port",
                                      "cpu_model": cpu_model(), "host_logical_cpus": os.cpu_count(),
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
