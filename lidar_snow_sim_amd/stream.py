"""Frame-stream driver: the reference's batch pre-computation loop as a reader / augment / writer pipeline.

Counterpart of tools/snowfall/precompute.py::__main__ (:47-106): for every frame id of a split and every
(snowfall rate, terminal velocity) pair, read the float32 N x 5 STF `.bin` (:78), crop to the camera field of view
(:96-99), augment with the reference's defaults (:103-104: beam_divergence = degrees(3e-3), shuffle=True,
only_camera_fov=True) and write float32 rows to
    <lidar>/../snowfall_simulation/<mode>/<lidar_folder>_rainrate_<int(rain_rate)>/<id>.bin   (:85-89, :106)
skipping outputs that already exist (:91-92).

The reference walks mode -> frame -> combo one augment() at a time.  Here the same work items are grouped into batches
per (mode, combo) -- one flake-table set per batch -- and flow through three stages joined by bounded queues:

    reader threads  read the batch's .bin files straight into ONE page-locked batch buffer (no per-frame array, no staging
                    copy on the GPU worker's thread), `depth` batches ahead
    GPU workers     one host thread + engine context (stream, scratch, page-locked staging) each: upload, crop on the device,
                    augment, crop again (augment's only_camera_fov), download; the copies of one context overlap the
                    kernels of the other
    writer threads  float32 `.bin` files

The channel permutations are drawn from the global `random` in the reference's nesting order -- per mode, per frame, per
combo, one shuffle per output that does not exist yet -- BEFORE the items are regrouped AND before they are sharded: every
rank of a multi-GPU run draws for all frames and keeps the items of its own (frame i belongs to rank i mod W,
lidar_snow_sim_amd.dist), so a seeded run gives every (mode, frame, combo) the permutation the reference's sequential loop
gives it, on one rank or on eight.  "Exists" is decided against a listing taken before the run writes anything.

    python -m lidar_snow_sim_amd.stream --lidar <dir> --split <file> --particles <npy dir> [--batch 32]
"""
from __future__ import annotations

import argparse
import os
import queue
import threading
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Iterable, List, Sequence

import numpy as np

from . import dist as sdist
from .tools.snowfall.sampling import compute_occupancy, snowfall_rate_to_rainfall_rate
from .tools.snowfall.simulation import FlatBatch, augment_batch

SNOWFALL_RATES = [0.5, 1.0, 2.0, 2.5, 1.5]        # mm/h   (precompute.py:20)
TERMINAL_VELOCITIES = [2.0, 1.6, 2.0, 1.6, 0.6]   # m/s    (precompute.py:21)


def rate_combos(rates: Sequence[float] = SNOWFALL_RATES, velocities: Sequence[float] = TERMINAL_VELOCITIES):
    """(rainfall_rate, occupancy_ratio) per pair (precompute.py:56-60)."""
    assert len(rates) == len(velocities), 'you need to provide an equal amount of snowfall_rates and terminal velocities'
    return [(snowfall_rate_to_rainfall_rate(r, v), compute_occupancy(r, v)) for r, v in zip(rates, velocities)]


def read_split(split_file) -> List[str]:
    """Frame ids in the reference's processing order (precompute.py:62-68)."""
    ids = sorted('_'.join(x.strip().split(',')) for x in open(split_file).readlines() if x.strip())
    k, m = divmod(len(ids), 2)
    first, second = ids[:k + min(1, m)], ids[k + min(1, m):]
    return second + first[::-1]


def output_path(lidar_folder: Path, mode: str, rainfall_rate: float, sample_id: str) -> Path:
    return (lidar_folder.parent / 'snowfall_simulation' / mode /
            f'{lidar_folder.name}_rainrate_{int(rainfall_rate)}' / f'{sample_id}.bin')


def existing_outputs(lidar_folder: Path, modes, combos):
    """The outputs that exist NOW, as a set of paths: one directory listing per (mode, combo) folder.  Taken once, before the
    pipeline starts -- the reference's skip test (precompute.py:91-92) is then a set lookup that neither the writers of this run
    nor the other ranks of a sharded run can change while the permutations are being drawn."""
    have = set()
    for mode in modes:
        for rainfall_rate, _occ in combos:
            folder = output_path(lidar_folder, mode, rainfall_rate, 'x').parent
            try:
                have.update(str(folder / name) for name in os.listdir(folder))
            except FileNotFoundError:
                pass
    return have


def plan_iter(lidar_folder: Path, ids: Sequence[str], modes, combos, batch: int, n_lasers: int = 64, rank: int = 0, world: int = 1,
              existing=None):
    """The work items in the reference's order (precompute.py:70-92: mode -> frame -> combo, existing outputs skipped), one
    `random.shuffle` per item in that order (simulation.py:483-486), regrouped into batches of one (mode, combo).  A generator:
    a batch is handed out as soon as its group holds `batch` items, so the pipeline runs while later permutations are drawn.

    Sharding (rank r of `world` owns frames r, r + world, ...) happens AFTER the draw: every rank walks ALL ids and draws every
    item's permutation from the global `random`, keeping its own items only -- so a seeded run gives (mode, frame, combo) the
    permutation the reference's sequential loop gives it whatever the number of ranks (64-int shuffles: 10 000 frames x 10 items
    are ~1 s per rank).  `existing`: the set of output paths present when the run started (existing_outputs); the skip test never
    looks at the live file system, whose state changes while writers and other ranks run.  An item planned earlier in this
    generator counts as existing too (two combos with the same int(rainfall_rate), a repeated id)."""
    import random
    if existing is None:
        existing = existing_outputs(lidar_folder, modes, combos)
    planned = set()
    groups = {}

    def job(mode, ci, items):
        rainfall_rate, occupancy = combos[ci]
        prefix = f'{mode}_{rainfall_rate}_{occupancy}'                                   # precompute.py:101
        return (mode, rainfall_rate, prefix, [c[0] for c in items], [c[1] for c in items])

    for mode in modes:
        for si, s in enumerate(ids):
            for ci, (rainfall_rate, _occ) in enumerate(combos):
                out = str(output_path(lidar_folder, mode, rainfall_rate, s))
                if out in existing or out in planned:                                    # :91-92
                    continue
                planned.add(out)
                order = list(range(n_lasers))
                random.shuffle(order)
                if si % world != rank:                                                   # another rank's frame: drawn, not kept
                    continue
                items = groups.setdefault((mode, ci), [])
                items.append((s, order))
                if len(items) >= batch:
                    yield job(mode, ci, items)
                    groups[(mode, ci)] = []
    for (mode, ci), items in groups.items():
        if items:
            yield job(mode, ci, items)


def plan(lidar_folder: Path, ids: Sequence[str], modes, combos, batch: int, n_lasers: int = 64, rank: int = 0, world: int = 1,
         existing=None):
    return list(plan_iter(lidar_folder, ids, modes, combos, batch, n_lasers, rank, world, existing))


def run(lidar_folder, sample_ids: Iterable[str], particle_root=None, modes=('gunn', 'sekhon'), combos=None,
        batch: int = 32, calib=None, device: int = 0, rank: int = 0, world: int = 1, particles_by_prefix=None,
        planes=None, workers: int = 2, readers: int = 4, writers: int = 4, depth: int = 4, keep_outputs: bool = True,
        report: dict = None, plane_method: str = 'reference', plane_seed: int = 0, existing=None, sample_missing: bool = False,
        backend=None) -> int:
    """Process this rank's share of `sample_ids`; returns the number of files written.

    workers   GPU worker threads, each with its own engine context on `device`
    readers / writers   file I/O threads; depth: batches the readers may run ahead, results the writers may lag behind
    planes    None (calculate_plane per frame on the device, by `plane_method`: 'reference' = the plane the reference returns today,
              'lsq', 'ransac' seeded with `plane_seed`) or one (w, h) used for every frame
    sample_missing  particle tables whose <prefix>_<line>.npy does not exist are sampled on the device (augment_batch(particles=
              'missing')): an empty particle directory is enough to run
    existing  set of output paths to treat as present (default: a listing of the output folders taken before anything is
              written).  Ranks of a sharded run must agree on it: start them against a quiescent output tree (they each list it
              before writing; a launcher-started run lists behind a barrier, see main()), or pass the same set to all
    keep_outputs=False  unlink every output right after it has been written (throughput dry runs on a small disk)
    report    optional dict that receives wall time, files, points in / out and the per-stage busy times
    backend   None: the product (libsnowgpu.so through simulation.augment_batch, page-locked batch buffers from the engine).  Tests and
              `bench.py --dry` inject an object with alloc_rows(n_rows) -> float32 (n, 5) array and augment_batch(frames, prefix,
              beam_divergence, **kw) to drive the reader / writer / sharding machinery on a machine without a GPU"""
    lidar_folder = Path(lidar_folder)
    combos = rate_combos() if combos is None else combos
    ids = list(sample_ids)
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    if existing is None:
        existing = existing_outputs(lidar_folder, modes, combos)   # before the first write of this run (and see `existing` above)
        if world > 1:
            # A sharded run's ranks must take the listing before ANY of them writes: a rank that started late would otherwise skip items
            # another rank has just written -- and their random.shuffle draws with them, after which its seeded permutations drift from
            # the one-rank sequence.  With a process group up, every rank lists and then waits for the others here; without one there is
            # nothing to wait on, and the caller has to pass the same `existing` set to every rank.
            try:
                import torch.distributed as _td
                grouped = _td.is_available() and _td.is_initialized()
            except ImportError:
                grouped = False
            if not grouped:
                raise ValueError("stream.run(rank, world > 1) without `existing`: list the outputs once (stream.existing_outputs) and pass "
                                 "the same set to every rank, or initialise torch.distributed so that the ranks can list behind a barrier")
            _td.barrier()
    # drawn by the feeder thread while the pipeline runs; every rank draws for ALL ids and keeps its own (see plan_iter)
    jobs = plan_iter(lidar_folder, ids, modes, combos, batch, rank=rank, world=world, existing=existing)
    n_jobs = [0]
    t_start = time.perf_counter()
    tally = {'files': 0, 'points_in': 0, 'points_out': 0, 'read_s': 0.0, 'gpu_s': 0.0, 'write_s': 0.0}
    tally_lock = threading.Lock()
    errors = []

    # page-locked batch buffers, recycled: a reader fills one, the GPU worker uploads from it and hands it back
    from . import engine as _engine
    pin_pool, pin_lock = [], threading.Lock()

    def take_buffer(n_rows):
        with pin_lock:
            for i, b in enumerate(pin_pool):
                if b.shape[0] >= n_rows:
                    return pin_pool.pop(i)
        if backend is not None:
            return backend.alloc_rows(max(n_rows, 1) * 9 // 8 + 1024)
        return _engine.get_engine(device, 0).ctx.pinned_empty((max(n_rows, 1) * 9 // 8 + 1024, 5), np.float32)

    def give_buffer(b):
        with pin_lock:
            if len(pin_pool) < 2 * (max(workers, 1) + max(depth, 1)):
                pin_pool.append(b)

    def read_job(job):
        t0 = time.perf_counter()
        paths = [lidar_folder / f'{s}.bin' for s in job[3]]
        sizes = []
        for p in paths:
            nbytes = p.stat().st_size
            if nbytes % 20:                                                             # np.fromfile(...).reshape((-1, 5)) raises there (precompute.py:78)
                raise ValueError(f'{p}: {nbytes} bytes is not a whole number of float32 N x 5 rows')
            sizes.append(nbytes // 20)
        off = np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)
        buf = take_buffer(int(off[-1]))
        for p, a, b in zip(paths, off[:-1], off[1:]):
            with open(p, 'rb', buffering=0) as fh:
                view = memoryview(buf[int(a):int(b)]).cast('B')
                got = fh.readinto(view)
                while got < len(view):                                                  # short reads
                    more = fh.readinto(view[got:])
                    if not more:
                        raise OSError(f'{p}: truncated')
                    got += more
        with tally_lock:
            tally['read_s'] += time.perf_counter() - t0
        return FlatBatch(buf, off)

    made_dirs = set()

    def write_job(job, results):
        t0 = time.perf_counter()
        mode, rainfall_rate = job[0], job[1]
        n_out = 0
        for s, (stats, aug) in zip(job[3], results):
            out = output_path(lidar_folder, mode, rainfall_rate, s)
            if out.parent not in made_dirs:                                              # one mkdir per output folder, not per file
                out.parent.mkdir(parents=True, exist_ok=True)
                made_dirs.add(out.parent)
            aug.astype(np.float32, copy=False).tofile(out)                               # :106
            n_out += aug.shape[0]
            if not keep_outputs:
                out.unlink()
        with tally_lock:
            tally['files'] += len(results)
            tally['points_out'] += n_out
            tally['write_s'] += time.perf_counter() - t0

    q_read = queue.Queue(maxsize=max(depth, 1))        # (job, future of its frames): the readers' lead
    q_write = queue.Queue(maxsize=max(depth, 1))       # write futures: the writers' lag
    with ThreadPoolExecutor(max_workers=max(readers, 1)) as read_pool, ThreadPoolExecutor(max_workers=max(writers, 1)) as write_pool:

        def feeder():
            try:
                for job in jobs:
                    n_jobs[0] += 1
                    q_read.put((job, read_pool.submit(read_job, job)))
            except BaseException as e:                                                   # noqa: BLE001 -- reported after the drain
                errors.append(e)
            finally:
                for _ in range(max(workers, 1)):
                    q_read.put(None)

        def gpu_worker(slot):
            while True:
                item = q_read.get()
                if item is None:
                    return
                job, fut = item
                try:
                    frames = fut.result()
                    t0 = time.perf_counter()
                    try:
                        results = (augment_batch if backend is None else backend.augment_batch)(
                                                frames, job[2], float(np.degrees(3e-3)), shuffle=False, root_path=particle_root,
                                                particles=(particles_by_prefix[job[2]] if particles_by_prefix is not None
                                                           else ('missing' if sample_missing else None)),
                                                planes=None if planes is None else [planes] * len(frames),
                                                orders=job[4], device=device, slot=slot, calib=calib, pre_crop=calib is not None,
                                                plane_method=plane_method, plane_seed=plane_seed)
                    finally:
                        give_buffer(frames.rows)
                    with tally_lock:
                        tally['gpu_s'] += time.perf_counter() - t0
                        tally['points_in'] += int(frames.offsets[-1])
                    q_write.put(write_pool.submit(write_job, job, results))
                except BaseException as e:                                               # noqa: BLE001 -- reported after the drain
                    errors.append(e)

        def drainer():
            while True:
                fut = q_write.get()
                if fut is None:
                    return
                try:
                    fut.result()
                except BaseException as e:                                               # noqa: BLE001
                    errors.append(e)

        threads = [threading.Thread(target=feeder, daemon=True), threading.Thread(target=drainer, daemon=True)]
        gpu_threads = [threading.Thread(target=gpu_worker, args=(w,), daemon=True) for w in range(max(workers, 1))]
        for t in threads + gpu_threads:
            t.start()
        for t in gpu_threads:
            t.join()
        q_write.put(None)
        for t in threads:
            t.join()
    if errors:
        raise errors[0]
    if report is not None:
        report.update(tally, wall_s=time.perf_counter() - t_start, batches=n_jobs[0], workers=workers, readers=readers,
                      writers=writers, batch=batch)
    return tally['files']


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('--lidar', required=True, help='folder of <id>.bin float32 N x 5 sweeps')
    ap.add_argument('--split', required=True, help="split file with 'date,frame' lines")
    ap.add_argument('--particles', default=None, help='root_path holding training/snowflakes/npy/<prefix>_<line>.npy')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--calib', default=None, help='KITTI-style calibration file for the camera-FOV crop')
    ap.add_argument('--workers', type=int, default=2, help='GPU worker threads (engine contexts) per GPU')
    ap.add_argument('--readers', type=int, default=4)
    ap.add_argument('--writers', type=int, default=4)
    ap.add_argument('--seed', type=int, default=None, help='random.seed() before the permutations are drawn (same on every rank)')
    ap.add_argument('--plane-method', default='reference', choices=('reference', 'lsq', 'ransac'))
    ap.add_argument('--plane-seed', type=int, default=0, help="seed of --plane-method ransac (Philox draws keyed by it: same cloud + same seed = same plane)")
    ap.add_argument('--sample-missing', action='store_true', help='sample particle tables that have no .npy file on the device')
    args = ap.parse_args(argv)
    rank, local_rank, world = sdist.env_rank_world()
    calib = None
    if args.calib:
        from .calibration import Calibration
        calib = Calibration(args.calib)
    rep = {}
    existing = None
    if world > 1:
        # every rank lists the outputs BEFORE any rank writes: the skip decisions (and with them the permutations a seeded run
        # draws) are then the same on all ranks.  gloo is enough: this barrier is the only collective of the driver.
        existing = existing_outputs(Path(args.lidar), ('gunn', 'sekhon'), rate_combos())
        d = sdist.init('gloo', rank, world)
        d.barrier()
    if args.seed is not None:
        import random
        random.seed(args.seed)                                  # the same seed on every rank: the reference's sequential draw order
    n = run(args.lidar, read_split(args.split), particle_root=args.particles, batch=args.batch, calib=calib,
            device=local_rank, rank=rank, world=world, workers=args.workers, readers=args.readers, writers=args.writers, report=rep,
            existing=existing, plane_method=args.plane_method, plane_seed=args.plane_seed, sample_missing=args.sample_missing)
    print(f'rank {rank}/{world}: wrote {n} files in {rep.get("wall_s", 0.0):.1f} s')


if __name__ == '__main__':
    main()
