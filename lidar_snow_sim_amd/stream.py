"""Frame-stream driver: the reference's batch pre-computation loop on top of augment_batch.

Counterpart of tools/snowfall/precompute.py::__main__ (:47-106): for every frame id of a split and every
(snowfall rate, terminal velocity) pair, read the float32 N x 5 STF `.bin` (:78), optionally crop to the camera
field of view (:96-99), augment with the reference's defaults (:103-104: beam_divergence = degrees(3e-3),
shuffle=True) and write float32 rows to
    <lidar>/../snowfall_simulation/<mode>/<lidar_folder>_rainrate_<int(rain_rate)>/<id>.bin   (:85-89, :106)
skipping outputs that already exist (:91-92).  Frames are independent, so a multi-GPU run shards the frame
list round-robin over ranks (lidar_snow_sim_amd.dist) and batches frames per launch.

    python -m lidar_snow_sim_amd.stream --lidar <dir> --split <file> --particles <npy dir> [--batch 32]
"""
from __future__ import annotations

import argparse
from pathlib import Path
from typing import Iterable, List, Sequence

import numpy as np

from . import dist as sdist
from .tools.snowfall.sampling import compute_occupancy, snowfall_rate_to_rainfall_rate
from .tools.snowfall.simulation import augment_batch

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


def run(lidar_folder, sample_ids: Iterable[str], particle_root=None, modes=('gunn', 'sekhon'), combos=None,
        batch: int = 32, calib=None, device: int = 0, rank: int = 0, world: int = 1, particles_by_prefix=None,
        planes=None, workers: int = 1) -> int:
    """Process this rank's share of `sample_ids`; returns the number of files written.

    workers > 1: that many host threads, each with its own engine context (stream + scratch) on `device`, take
    batches in turn -- file reads, H2D/D2H copies and file writes of one batch overlap the kernels of another.
    The channel permutations are drawn from the global `random` in the main thread, frame by frame in processing
    order (as the reference's sequential loop would), so the output does not depend on `workers`."""
    import random
    from concurrent.futures import ThreadPoolExecutor
    lidar_folder = Path(lidar_folder)
    combos = rate_combos() if combos is None else combos
    ids = list(sample_ids)
    mine = [ids[i] for i in sdist.shard_indices(len(ids), rank, world)]
    n_lasers = 64

    def do_chunk(job):
        slot, mode, rainfall_rate, prefix, chunk, orders = job
        frames = []
        for s in chunk:
            pts = np.fromfile(str(lidar_folder / f'{s}.bin'), dtype=np.float32).reshape(-1, 5)       # :78
            if calib is not None:                                                                    # :96-99
                from .calibration import get_fov_flag
                pts = pts[get_fov_flag(calib.lidar_to_rect(pts[:, 0:3]), (1024, 1920), calib)]
            frames.append(pts)
        results = augment_batch(frames, prefix, float(np.degrees(3e-3)), shuffle=False, root_path=particle_root,
                                particles=None if particles_by_prefix is None else particles_by_prefix[prefix],
                                planes=planes, orders=orders, device=device, slot=slot)
        for s, (stats, aug) in zip(chunk, results):
            if calib is not None:                                        # augment()'s only_camera_fov default
                from .calibration import get_fov_flag
                aug = aug[get_fov_flag(calib.lidar_to_rect(aug[:, 0:3]), (1024, 1920), calib)]
            out = output_path(lidar_folder, mode, rainfall_rate, s)
            out.parent.mkdir(parents=True, exist_ok=True)
            aug.astype(np.float32).tofile(out)                           # :106
        return len(chunk)

    jobs = []
    for mode in modes:
        for rainfall_rate, occupancy in combos:
            prefix = f'{mode}_{rainfall_rate}_{occupancy}'                      # precompute.py:101
            todo = [s for s in mine if not output_path(lidar_folder, mode, rainfall_rate, s).is_file()]   # :91-92
            for b0 in range(0, len(todo), batch):
                chunk = todo[b0:b0 + batch]
                orders = []
                for _ in chunk:                                          # simulation.py:483-486, one draw per frame
                    order = list(range(n_lasers))
                    random.shuffle(order)
                    orders.append(order)
                jobs.append((len(jobs) % max(workers, 1), mode, rainfall_rate, prefix, chunk, orders))
    if workers <= 1:
        return sum(do_chunk(j) for j in jobs)
    # one thread per slot so that a context is never used by two threads at once
    per_slot = [[j for j in jobs if j[0] == w] for w in range(workers)]
    with ThreadPoolExecutor(max_workers=workers) as pool:
        return sum(pool.map(lambda js: sum(do_chunk(j) for j in js), per_slot))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('--lidar', required=True, help='folder of <id>.bin float32 N x 5 sweeps')
    ap.add_argument('--split', required=True, help="split file with 'date,frame' lines")
    ap.add_argument('--particles', default=None, help='root_path holding training/snowflakes/npy/<prefix>_<line>.npy')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--calib', default=None, help='KITTI-style calibration file for the camera-FOV crop')
    ap.add_argument('--workers', type=int, default=2, help='host threads (engine contexts) per GPU')
    args = ap.parse_args(argv)
    rank, local_rank, world = sdist.env_rank_world()
    calib = None
    if args.calib:
        from .calibration import Calibration
        calib = Calibration(args.calib)
    n = run(args.lidar, read_split(args.split), particle_root=args.particles, batch=args.batch, calib=calib,
            device=local_rank, rank=rank, world=world, workers=args.workers)
    print(f'rank {rank}/{world}: wrote {n} files')


if __name__ == '__main__':
    main()
