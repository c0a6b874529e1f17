"""Offline particle-table generator: the `npy/<mode>_<rate>_<occupancy>_<line>.npy` tree the augmentation reads.

Counterpart of tools/snowfall/sampling.py::__main__ (:360-413) with its resumable writer (save_array / sampling_exists /
do_in_parallel, :327-357):

    2 distributions ('gunn', 'sekhon', :393) x 50 (rainfall rate, occupancy) pairs -- the product of
    np.linspace(0.5, 2.5, 5) mm/h snowfall rates and np.linspace(0.2, 2, 10) m/s terminal velocities (:384-390), sorted by
    occupancy, largest first (:395-397) -- x 64 lines (:399) = 6400 tables of R_0 = 80 m (:379),
    each saved as f'{dist}_{rate}_{ratio}_{line}.npy' (:344) unless that file exists already (:346-347).

    python -m lidar_snow_sim_amd.sample_tables --out <npy dir> [--device] [--rng serial|per-table] [--seed 42]
                                                [--modes gunn sekhon] [--snowfall-rates ...] [--velocities ...] [--lines 1 64]

Two samplers:

  host (default)   tools/snowfall/sampling.py::dart_throwing through this package's mirror, which consumes a NumPy Generator in the
                   reference's draw order: a table is bit for bit what the reference computes from the same generator state (L7
                   fixtures).  Which state a table sees is where the reference itself is loose: its one module-level
                   default_rng(42) (:381) is copied into every worker process of process_map (:413), so its tables depend on the
                   schedule.  `--rng serial` is that code with ONE worker: a single default_rng(seed) consumed table after table in
                   the reference's order (a skipped table consumes nothing, as there).  `--rng per-table` gives every table its
                   own default_rng([seed, mode index, pair index, line]): any subset, any order, any number of processes
                   (`--jobs`) -- and a resumed run -- write the same bytes.
  --device         snowgpu_sample_table (csrc/snowgpu_sampler.hip): the same process from a Philox stream keyed by
                   engine.table_seed(prefix, line) -- statistically the reference's tables, ~10^3 tables/s, and exactly
                   the tables augment(particles='device') makes on the fly, so a run from the files and a run without them agree.

The frame-stream driver consumes the tree: `python -m lidar_snow_sim_amd.stream --particles <root>` looks under
<root>/training/snowflakes/npy (simulation.py:324-325); `--root-path <root>` here writes there.
"""
from __future__ import annotations

import argparse
import itertools
import os
import sys
import time
from pathlib import Path

import numpy as np

from .tools.snowfall.sampling import compute_occupancy, dart_throwing, gunn_marshall, sekhon_srivastava, snowfall_rate_to_rainfall_rate

R_0 = 80.0                                                     # sampling.py:379
MODES = ('gunn', 'sekhon')                                     # sampling.py:393
SNOWFALL_RATES = np.linspace(0.5, 2.5, 5)                      # mm/h, sampling.py:384
TERMINAL_VELOCITIES = np.linspace(0.2, 2, 10)                  # m/s,  sampling.py:385


def rate_pairs(snowfall_rates=SNOWFALL_RATES, velocities=TERMINAL_VELOCITIES):
    """The (rainfall rate, occupancy ratio) rows of sampling.py:387-397: product of the two grids, sorted by occupancy, largest first."""
    p = list(itertools.product(snowfall_rates, velocities))                                    # :387
    r_r_s = [snowfall_rate_to_rainfall_rate(r_s, v_s) for r_s, v_s in p]                       # :389
    ratios = [compute_occupancy(r_s, v_s) for r_s, v_s in p]                                   # :390
    runs = np.column_stack((r_r_s, ratios))                                                    # :395
    runs = runs[runs[:, 1].argsort()]                                                          # :396
    return runs[::-1]                                                                          # :397


def table_name(dist, rate, ratio, line) -> str:
    """f'{dist}_{rate}_{ratio}_{line}' (sampling.py:344) -- rate and ratio print as Python floats do, which is also how
    precompute.py:101 / pointcloud_viewer.py:2802 build the prefix the augmentation looks up."""
    return f'{dist}_{rate}_{ratio}_{line}'


def plan(modes=MODES, runs=None, lines=range(1, 65)):
    """[(mode index, pair index, dist, rate, ratio, line)] in the order of itertools.product(m, runs, n) (sampling.py:399-406)."""
    runs = rate_pairs() if runs is None else np.asarray(runs, np.float64).reshape(-1, 2)
    return [(mi, ri, dist, runs[ri, 0], runs[ri, 1], int(line))
            for (mi, dist), ri, line in itertools.product(list(enumerate(modes)), range(len(runs)), lines)]


def sampling_exists(out_dir, name) -> bool:                     # sampling.py:334-338
    return (Path(out_dir) / f'{name}.npy').is_file()


def save_array(out_dir, name, samples) -> None:                 # sampling.py:327-331 (written under a temporary name first: a killed run
    tmp = Path(out_dir) / f'.{name}.{os.getpid()}.tmp.npy'      #  leaves no half-written table that the next one would skip)
    np.save(tmp, samples)
    os.replace(tmp, Path(out_dir) / f'{name}.npy')


def _table_rng(seed, mi, ri, line):
    return np.random.default_rng([int(seed), int(mi), int(ri), int(line)])


def _host_one(job):
    out_dir, seed, r0, (mi, ri, dist, rate, ratio, line) = job
    name = table_name(dist, rate, ratio, line)
    if sampling_exists(out_dir, name):                          # sampling.py:346-347
        return name, -1
    particles = dart_throwing(occupancy_ratio=ratio, precipitation_rate=rate, R_0=r0, distribution=dist, rng=_table_rng(seed, mi, ri, line))
    save_array(out_dir, name, particles)
    return name, int(particles.shape[0])


def generate(out_dir, modes=MODES, runs=None, lines=range(1, 65), *, device=None, rng='serial', seed=42, r0=R_0, jobs=1, verbose=True):
    """Write every missing table of the plan; returns {'written': n, 'skipped': n, 'flakes': n, 'seconds': s, 'names': [...]}."""
    out_dir = Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    items = plan(modes, runs, lines)
    t0 = time.perf_counter()
    written = skipped = flakes = 0
    names = []

    def note(name, k):
        nonlocal written, skipped, flakes
        names.append(name)
        if k < 0:
            skipped += 1
            if verbose:
                print(f'{name} skipped', flush=True)                 # sampling.py:347
        else:
            written += 1
            flakes += k
            if verbose:
                print(f'{name}  ({k} flakes)', flush=True)           # sampling.py:349

    if device is not None:
        from . import engine
        eng = engine.get_engine(int(device))
        for mi, ri, dist, rate, ratio, line in items:
            name = table_name(dist, rate, ratio, line)
            if sampling_exists(out_dir, name):
                note(name, -1)
                continue
            prefix = f'{dist}_{rate}_{ratio}'
            rate_parameter = gunn_marshall(rate) if dist == 'gunn' else sekhon_srivastava(rate)
            rows = eng.ctx.sample_table(-1, ratio, (1 / rate_parameter) * 10, r0, engine.table_seed(prefix, line))
            save_array(out_dir, name, rows)
            note(name, int(rows.shape[0]))
    elif rng == 'serial':
        gen = np.random.default_rng(seed)                        # sampling.py:381 -- one generator, one worker
        for mi, ri, dist, rate, ratio, line in items:
            name = table_name(dist, rate, ratio, line)
            if sampling_exists(out_dir, name):
                note(name, -1)
                continue
            particles = dart_throwing(occupancy_ratio=ratio, precipitation_rate=rate, R_0=r0, distribution=dist, rng=gen)   # :351-352
            save_array(out_dir, name, particles)
            note(name, int(particles.shape[0]))
    elif rng == 'per-table':
        work = [(str(out_dir), seed, r0, it) for it in items]
        if jobs > 1:
            import multiprocessing
            with multiprocessing.get_context('spawn').Pool(jobs) as pool:
                for name, k in pool.imap(_host_one, work, chunksize=1):
                    note(name, k)
        else:
            for w in work:
                note(*_host_one(w))
    else:
        raise ValueError("rng must be 'serial' or 'per-table'")
    return {'written': written, 'skipped': skipped, 'flakes': flakes, 'seconds': time.perf_counter() - t0, 'names': names}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    where = ap.add_mutually_exclusive_group(required=True)
    where.add_argument('--out', help='directory for the .npy files (the reference writes <repo>/npy, sampling.py:20)')
    where.add_argument('--root-path', help="write <root>/training/snowflakes/npy (augment's root_path, simulation.py:324-325)")
    ap.add_argument('--device', type=int, nargs='?', const=0, default=None, help='sample on this GPU (Philox; statistical parity) instead of on the host')
    ap.add_argument('--rng', default='serial', choices=('serial', 'per-table'))
    ap.add_argument('--seed', type=int, default=42)
    ap.add_argument('--modes', nargs='+', default=list(MODES), choices=list(MODES))
    ap.add_argument('--snowfall-rates', type=float, nargs='+', default=None, help='mm/h (default: np.linspace(0.5, 2.5, 5))')
    ap.add_argument('--velocities', type=float, nargs='+', default=None, help='m/s (default: np.linspace(0.2, 2, 10)); product with the rates')
    ap.add_argument('--pairs', type=float, nargs='+', default=None,
                    help='explicit (snowfall rate, terminal velocity) pairs instead of the product, e.g. precompute.py:20-21: 0.5 2.0 1.0 1.6 ...')
    ap.add_argument('--lines', type=int, nargs=2, default=[1, 64], metavar=('FIRST', 'LAST'))
    ap.add_argument('--r0', type=float, default=R_0)
    ap.add_argument('--jobs', type=int, default=1, help='host processes (--rng per-table only)')
    ap.add_argument('--quiet', action='store_true')
    args = ap.parse_args(argv)
    out = Path(args.out) if args.out else Path(args.root_path) / 'training' / 'snowflakes' / 'npy'
    if args.pairs:
        if len(args.pairs) % 2:
            ap.error('--pairs takes an even number of values')
        pr = np.asarray(args.pairs).reshape(-1, 2)
        runs = np.column_stack(([snowfall_rate_to_rainfall_rate(r, v) for r, v in pr], [compute_occupancy(r, v) for r, v in pr]))
        runs = runs[runs[:, 1].argsort()][::-1]
    else:
        runs = rate_pairs(SNOWFALL_RATES if args.snowfall_rates is None else args.snowfall_rates,
                          TERMINAL_VELOCITIES if args.velocities is None else args.velocities)
    if args.jobs > 1 and (args.rng != 'per-table' or args.device is not None):
        ap.error('--jobs needs --rng per-table on the host (a serial generator has one consumer)')
    rep = generate(out, args.modes, runs, range(args.lines[0], args.lines[1] + 1), device=args.device, rng=args.rng, seed=args.seed,
                   r0=args.r0, jobs=args.jobs, verbose=not args.quiet)
    print(f"{rep['written']} tables written ({rep['flakes']} flakes), {rep['skipped']} skipped, {rep['seconds']:.1f} s"
          + (f", {rep['written'] / rep['seconds']:.1f} tables/s" if rep['written'] and rep['seconds'] > 0 else '') + f" -> {out}")
    return 0


if __name__ == '__main__':
    sys.exit(main())
