"""Snowfall physics helpers and the dart-throwing particle sampler.

Mirror of tools/snowfall/sampling.py (compute_occupancy :23-32, rainfall_rate_to_snowfall_rate :35-52,
snowfall_rate_to_rainfall_rate :55-69, sekhon_srivastava :72-78, gunn_marshall :81-87,
dart_throwing :90-194).  Host logic: the sampler runs offline, once per (mode, rate, occupancy, line).

`dart_throwing` consumes the caller's NumPy Generator in exactly the reference's order
(uniform, uniform, exponential until <= 20 mm, uniform), so a seeded call reproduces the reference's
table bit for bit; the O(K) NumPy overlap test per dart (sampling.py:170) is replaced by a uniform
grid of cells that returns the same boolean.
"""
import math

import numpy as np

PI = np.pi


def compute_occupancy(snowfall_rate: float, terminal_velocity: float, snow_density: float = 0.1) -> float:
    water_density = 1.0
    return (water_density * snowfall_rate) / ((3.6 * 10 ** 6) * (snow_density * terminal_velocity))


def rainfall_rate_to_snowfall_rate(rainfall_rate: float, terminal_velocity: float,
                                   snowflake_density: float = 0.1, snowflake_diameter: float = 0.003) -> float:
    return 487 * snowflake_density * snowflake_diameter * terminal_velocity * (rainfall_rate ** (2 / 3))


def snowfall_rate_to_rainfall_rate(snowfall_rate: float, terminal_velocity: float,
                                   snowflake_density: float = 0.1, snowflake_diameter: float = 0.003) -> float:
    return np.sqrt((snowfall_rate / (487 * snowflake_density * snowflake_diameter * terminal_velocity)) ** 3)


def sekhon_srivastava(precipitation_rate: float) -> float:
    return 22.9 * precipitation_rate ** -0.45


def gunn_marshall(precipitation_rate: float) -> float:
    return 25.5 * precipitation_rate ** -0.48


def dart_throwing(occupancy_ratio: float, precipitation_rate: float, R_0: float, rng: np.random.Generator,
                  distribution: str = 'sekhon_srivastava', show_progessbar: bool = False) -> np.ndarray:
    """N x 3 array (x, y, disk radius) of non-overlapping snowflake disks in a disc of radius R_0."""
    if distribution == 'sekhon':
        rate_parameter = sekhon_srivastava(precipitation_rate)
    elif distribution == 'gunn':
        rate_parameter = gunn_marshall(precipitation_rate)
    else:
        raise NotImplementedError('Distribution model unknown.')          # sampling.py:113 (also the default!)
    scale = 1 / rate_parameter                                             # cm
    target = occupancy_ratio * PI * R_0 ** 2                               # sampling.py:124
    area = 0.0
    out = []
    # Uniform grid over [-R_0, R_0]^2.  A disk of radius <= 10 mm overlaps only disks whose centres are
    # within 20 mm, so with cells >= 25 mm the 3 x 3 neighbourhood is a superset of the candidates.
    cell = max(0.025, 2.0 * R_0 / 4096.0)
    inv = 1.0 / cell
    grid = {}
    uniform, exponential = rng.uniform, rng.exponential
    sqrt, cos, sin = np.sqrt, np.cos, np.sin
    r0sq = R_0 ** 2
    while area < target:                                                   # sampling.py:142
        length = sqrt(uniform(0, r0sq))                                    # :145
        angle = uniform(0, 2) * PI                                         # :146
        x = length * cos(angle)
        y = length * sin(angle)
        diameter = np.inf
        while diameter > 20:                                               # :153-154
            diameter = exponential(scale * 10)
        diameter = diameter / 1000                                         # :157
        height = uniform(-diameter / 2, diameter / 2)                      # :160
        radius = sqrt((diameter / 2) ** 2 - height ** 2)                   # :163
        if x ** 2 + y ** 2 <= radius ** 2:                                 # :166
            continue
        cx, cy = int(math.floor(x * inv)), int(math.floor(y * inv))
        overlap = False
        for gx in (cx - 1, cx, cx + 1):
            for gy in (cy - 1, cy, cy + 1):
                for (sx, sy, sr) in grid.get((gx, gy), ()):
                    if (sx - x) ** 2 + (sy - y) ** 2 <= (sr + radius) ** 2:   # :170
                        overlap = True
                        break
                if overlap:
                    break
            if overlap:
                break
        if overlap:                                                        # :173-174
            continue
        area += PI * radius ** 2                                           # :181-182
        out.append((x, y, radius))
        grid.setdefault((cx, cy), []).append((x, y, radius))
    return np.array(out, dtype=np.float64).reshape(-1, 3)


def dart_throwing_device(occupancy_ratio: float, precipitation_rate: float, R_0: float, seed: int,
                         distribution: str = 'gunn', device: int = 0, file_table: bool = False, want_rows: bool = True):
    """dart_throwing on the GPU (libsnowgpu `snowgpu_sample_table`): the same sampling process driven by a
    counter-based Philox stream instead of a sequential NumPy Generator, so tables are statistically -- not bit for
    bit -- those of `dart_throwing`.

    Returns the K x 3 rows.  With file_table=True the table is also filed on the device (derived, binned and sorted by
    kernels: it never visits the host) under a fresh engine table id, and (rows, table_id) is returned -- rows is None
    with want_rows=False.  Such ids go into the `table_ids` rows of the C ABI / `Context.augment_batch`."""
    if distribution == 'sekhon':
        rate_parameter = sekhon_srivastava(precipitation_rate)
    elif distribution == 'gunn':
        rate_parameter = gunn_marshall(precipitation_rate)
    else:
        raise NotImplementedError('Distribution model unknown.')
    from ... import engine
    eng = engine.get_engine(device)
    scale_mm = (1 / rate_parameter) * 10
    if not file_table:
        return eng.ctx.sample_table(-1, occupancy_ratio, scale_mm, R_0, seed)
    tid = eng.user_table_id()                       # never an id of the engine's own file / array caches
    res = eng.ctx.sample_table(tid, occupancy_ratio, scale_mm, R_0, seed, want_rows=want_rows)
    return (res if want_rows else None), tid
