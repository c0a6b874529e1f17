"""CPU-only tests: the C ABI loads and exports what include/snowgpu.h declares, host-side mirrors against the
reference's golden vectors, the arithmetic the device relies on, sharding over gloo (world_size 2)."""
import ctypes
import os
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


# ---- the C ABI ------------------------------------------------------------------------------------------------
def _declared_symbols():
    text = (ROOT / "include" / "snowgpu.h").read_text()
    return sorted(set(re.findall(r"\b(snowgpu_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from lidar_snow_sim_amd import _native
    lib = _native.lib()
    names = _declared_symbols()
    assert len(names) >= 12
    for name in names:
        assert hasattr(lib, name), name
    assert set(_native.EXPORTS) <= set(names)
    assert b"gfx950" in lib.snowgpu_version()


def test_no_device_is_an_error_not_a_fallback():
    """On a box without a GPU the product must fail loudly (no CPU path exists)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from lidar_snow_sim_amd import _native
    with pytest.raises(_native.SnowGPUError) as e:
        _native.Context(0)
    assert e.value.code == _native.E_NO_DEVICE


def test_null_and_bad_arguments_return_status_codes():
    from lidar_snow_sim_amd import _native
    lib = _native.lib()
    assert lib.snowgpu_range_grid(None) == _native.E_INVALID
    assert lib.snowgpu_set_lasers(None, 0, None, None, None, None) == _native.E_INVALID
    assert lib.snowgpu_upload_table(None, 0, None, 0) == _native.E_INVALID
    assert lib.snowgpu_last_error(None) == b"null context"


def test_product_never_imports_the_oracle():
    pat = re.compile(r"(from|import)\s+oracle|snow_oracle|libsnow_oracle|oracle/")
    for path in (ROOT / "lidar_snow_sim_amd").rglob("*"):
        if path.suffix in (".py", ".cpp", ".hip", ".h"):
            assert not pat.search(path.read_text()), path


# ---- arithmetic the kernels rely on ---------------------------------------------------------------------------
def test_range_grid_matches_numpy_and_the_table_free_formula():
    from lidar_snow_sim_amd import _native
    from oracle import snow_oracle as so
    grid = so.range_grid()                                      # np.round(np.linspace(...), 2), simulation.py:116
    assert np.array_equal(_native.range_grid(), grid)
    # sg_range_bin (sg_beam.h): n = rint(k * step * 100); q = n * 0.01; q + fma(-q, 100, n) * 0.01
    from fractions import Fraction
    step = (120 + 299792458.0 * 1e-8) / 1229

    def fma(a, b, c):
        return float(Fraction(a) * Fraction(b) + Fraction(c))
    for k in range(1230):
        n = float(np.rint((k * step) * 100.0))
        q = n * 0.01
        assert fma(fma(-q, 100.0, n), 0.01, q) == grid[k], k


def test_own_sine_is_within_one_ulp_of_libm():
    """sg_sin_0_pi restated in Python (same constants, fma emulated exactly): <= 1 ULP from math.sin on the
    argument range the power term uses, so sin^2 differs by a few 1e-16 relative at most."""
    import math
    from fractions import Fraction
    coef = [-1.9572941063391263e-20, 8.2206352466243295e-18, -2.8114572543455206e-15, 7.6471637318198164e-13,
            -1.6059043836821613e-10, 2.5052108385441720e-08, -2.7557319223985893e-06, 1.9841269841269841e-04,
            -8.3333333333333332e-03, 1.6666666666666666e-01]

    def fma(a, b, c):
        return float(Fraction(a) * Fraction(b) + Fraction(c))

    def own(u):
        x = u
        if u > 1.5707963267948966:
            x = (u - 3.141592653589793) - 1.2246467991473532e-16
        x2 = x * x
        p = coef[0]
        for c in coef[1:]:
            p = fma(p, x2, c)
        return fma(-(x * x2), p, x)
    rng = np.random.default_rng(3)
    worst = 0.0
    for u in np.concatenate((rng.uniform(0, 3.3, 4000), [0.0, 1e-9, 1.5707963267948966, 3.141592653589793, 3.25])):
        s, t = abs(own(float(u))), abs(math.sin(float(u)))
        if t > 1e-300:
            worst = max(worst, abs(s - t) / math.ulp(t))
    assert worst <= 1.0


# ---- host mirrors against the reference's golden vectors ------------------------------------------------------
def test_sampling_helpers_L0(golden):
    from lidar_snow_sim_amd.tools.snowfall import sampling as smp
    d = golden("L0_helpers")
    g = d["grid"]
    assert np.array_equal(d["occupancy"], [smp.compute_occupancy(a, b) for a, b in g])
    assert np.array_equal(d["rain"], [smp.snowfall_rate_to_rainfall_rate(a, b) for a, b in g])
    assert np.array_equal(d["snow"], [smp.rainfall_rate_to_snowfall_rate(a * 7, b) for a, b in g])
    rs = np.array([0.5, 1.0, 1.5, 2.0, 2.5, 10.0])
    assert np.array_equal(d["gunn"], [smp.gunn_marshall(a * 7) for a in rs])
    assert np.array_equal(d["sekhon"], [smp.sekhon_srivastava(a * 7) for a in rs])


def test_dart_throwing_is_bit_exact_L7(golden):
    from lidar_snow_sim_amd.tools.snowfall import sampling as smp
    d = golden("L7_dart_throwing")
    for i in range(3):
        a = d[f"args{i}"]
        t = smp.dart_throwing(a[0], a[1], a[2], np.random.default_rng(int(a[3])), str(d[f"mode{i}"]))
        assert np.array_equal(t, d[f"t{i}"])
    a = d["big_args"]                                            # the R0 = 80 m production table (18 028 flakes)
    t = smp.dart_throwing(a[0], a[1], a[2], np.random.default_rng(int(a[3])), "gunn")
    assert len(t) == int(d["big_count"])
    assert np.array_equal(t[:64], d["big_head"]) and np.array_equal(t[-64:], d["big_tail"])
    assert np.array_equal(t.sum(axis=0), d["big_sum"])
    with pytest.raises(NotImplementedError):                     # the reference's own default raises (Q13)
        smp.dart_throwing(1e-6, 10.0, 5.0, np.random.default_rng(0))


def test_estimate_laser_parameters_L6(golden):
    from lidar_snow_sim_amd.tools.wet_ground.augmentation import estimate_laser_parameters
    d = golden("L6_wet_ground")
    rel, thr, p, _ = estimate_laser_parameters(d["elp_pc"], d["elp_angle"], noise_floor=0.7, debug=False)
    np.testing.assert_allclose(rel, d["elp_rel"], rtol=1e-13)
    np.testing.assert_allclose(thr, d["elp_thr"], rtol=1e-13)
    np.testing.assert_allclose(p, d["elp_p"], rtol=1e-13)
    assert estimate_laser_parameters(d["elp_pc"][:2], d["elp_angle"][:2]) == (None, None, None, None)


def test_calculate_plane_reference_method_and_crop_mask():
    """The default estimator is the plane the reference returns today (planes.py:43-48 under scikit-learn >= 1.2) and needs
    no device; the host-side crop mask equals the reference's expression (planes.py:21-26) in the cloud's dtype."""
    from lidar_snow_sim_amd.tools.wet_ground.planes import calculate_plane, ground_crop
    pc = np.zeros((10, 5), np.float32)
    assert calculate_plane(pc) == ([0, 0, 1], -1.55)
    assert calculate_plane(pc, -1.6) == ([0, 0, 1], -1.6)
    rng = np.random.default_rng(0)
    for dt in (np.float32, np.float64):
        pc = np.column_stack((rng.uniform(0, 90, 4000), rng.uniform(-5, 5, 4000), rng.uniform(-2.8, -1.3, 4000),
                              np.ones(4000), np.zeros(4000))).astype(dt)
        want = (pc[:, 2] < -1.55) & (pc[:, 2] > -1.86 - 0.01 * pc[:, 0]) & (pc[:, 0] > 10) & (pc[:, 0] < 70) & (pc[:, 1] > -3) & (pc[:, 1] < 3)
        assert np.array_equal(ground_crop(pc), want) and 0 < want.sum() < 4000
    with pytest.raises(ValueError):
        calculate_plane(pc, method="sklearn")


def test_laser_constants_follow_the_calibration():
    from lidar_snow_sim_amd import engine
    las = engine.load_lasers()
    assert len(las) == 64 and "min_intensity" not in las[40]
    fs, fo, mi, ma = engine.laser_constants(las)
    assert [ma[c] for c in (53, 55, 56, 58)] == [230] * 4 and ma[0] == 255 and mi[40] == 0 and mi[0] == 40
    assert fo[0] == (1 - 8.0 * 100 / 13100) ** 2


def test_synthetic_sweep_is_deterministic_and_in_range():
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    a, b = synthetic_sweep(seed=7), synthetic_sweep(seed=7)
    assert a.shape == (131072, 5) and a.dtype == np.float32 and np.array_equal(a, b)
    r = np.linalg.norm(a[:, :3], axis=1)
    assert r.min() >= 2.99 and r.max() <= 119.01
    assert np.array_equal(a[:, 4], np.repeat(np.arange(64), 2048))
    assert synthetic_sweep(128, 4096, seed=1).shape == (524288, 5)


def test_missing_particle_file_raises_file_not_found(tmp_path):
    from lidar_snow_sim_amd import engine

    class Stub(engine.Engine):
        def __init__(self):                                      # no device needed for the lookup rule itself
            self.lasers = engine.load_lasers()
            self._tables, self._arrays, self._free_ids, self._next_id = {}, {}, [], 0
            import threading
            self._lock = threading.Lock()

            class Ctx:
                def upload_table(self, tid, arr):
                    pass
            self.ctx = Ctx()
    e = Stub()
    with pytest.raises(FileNotFoundError):                       # np.load in the reference (simulation.py:329)
        e.table_ids_from_files("gunn_1.0_2.0", list(range(64)), root_path=str(tmp_path))
    d = tmp_path / "training" / "snowflakes" / "npy"
    d.mkdir(parents=True)
    for line in range(1, 65):
        np.save(d / f"p_{line}.npy", np.zeros((1, 3)))
    order = list(range(63, -1, -1))
    ids = e.table_ids_from_files("p", order, root_path=str(tmp_path))
    assert ids == list(range(64)) and len(set(ids)) == 64        # channel c -> file order[c] + 1 (simulation.py:78)


# ---- frame sharding across ranks (gloo, world_size 2) -----------------------------------------------------------
def test_shard_indices_cover_everything_once():
    from lidar_snow_sim_amd.dist import bench_frame_seeds, shard_indices
    for world in (1, 2, 3, 8):
        seen = sorted(i for r in range(world) for i in shard_indices(37, r, world))
        assert seen == list(range(37))
    assert not set(bench_frame_seeds(0, 32)) & set(bench_frame_seeds(1, 32))


_WORKER = r'''
import os, sys, time
sys.path.insert(0, {root!r})
from lidar_snow_sim_amd import dist as sd
rank, world = int(sys.argv[1]), int(sys.argv[2])
os.environ["MASTER_PORT"] = sys.argv[3]
d = sd.init("gloo", rank, world)
mine = sd.shard_indices(10, rank, world)
t = sd.max_over_ranks(1.0 + rank)
tot = sd.sum_over_ranks([len(mine), sum(mine)])
d.barrier()
print("RESULT", rank, t, tot[0], tot[1], flush=True)
d.destroy_process_group()
'''


def test_two_rank_gloo_round_trip(tmp_path):
    """The N > 1 bookkeeping of bench.py / the stream driver (max time over ranks, item counts) on CPU."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=str(ROOT)))
    port = str(29600 + os.getpid() % 300)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", port], stdout=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    for r, out in enumerate(outs):
        line = [ln for ln in out.splitlines() if ln.startswith("RESULT")][0].split()
        assert float(line[2]) == 2.0                              # max over ranks of (1 + rank)
        assert float(line[3]) == 10.0 and float(line[4]) == 45.0  # every frame owned exactly once


def test_bench_gpus_2_spawns_two_ranks_over_gloo():
    """`python bench.py --gpus 2` with no launcher environment must start two ranks itself (it used to run one and
    print n_gpus from WORLD_SIZE only).  --dry keeps the launch / barrier / reduction logic and skips the GPU work."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry", "--frames", "8"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                                           # rank 0 only
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["backend"] == "gloo"
    assert rec["ms_per_step"] >= 20.0                                          # max over ranks: rank 1 sleeps 20 ms
    # a launcher environment that disagrees with --gpus is an error, not a silently different run
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry"], capture_output=True, text=True, timeout=120,
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "must agree" in r.stderr


def test_stream_driver_bookkeeping(tmp_path):
    """precompute.py's frame order, (rate, velocity) combos and output layout -- no GPU involved."""
    from lidar_snow_sim_amd import stream
    split = tmp_path / "s.txt"
    split.write_text("\n".join(f"2018-02-0{d},0000{i}" for d in (1, 2) for i in range(3)) + "\n")
    ids = stream.read_split(split)
    srt = sorted(ids)
    assert len(ids) == 6 and ids[:3] == srt[3:] and ids[3:] == srt[:3][::-1]     # second half, reversed first half
    combos = stream.rate_combos()
    assert len(combos) == 5 and int(combos[3][0]) == 34                          # 2.5 mm/h @ 1.6 m/s -> rainrate_34
    p = stream.output_path(tmp_path / "lidar_hdl64_strongest", "gunn", combos[3][0], ids[0])
    assert p == tmp_path / "snowfall_simulation" / "gunn" / "lidar_hdl64_strongest_rainrate_34" / f"{ids[0]}.bin"


def test_own_tangent_is_within_two_ulp_of_libm():
    """sg_tan_0_2pi (csrc/sg_math.h) restated with NumPy float64 operations (no fused multiply-adds, as the kernels
    are built with -ffp-contract=off): <= 2 ULP from libm's tan on [0, 2 pi], including next to pi/2 and 3 pi/2."""
    import math

    def sg_tan(theta):
        two_over_pi = 6.36619772367581382433e-01
        p1, p2, p2t = 1.57079632673412561417e+00, 6.07710050630396597660e-11, 2.02226624879595063154e-21
        fn = np.rint(theta * two_over_pi)
        n = fn.astype(int)
        r = theta - fn * p1
        t = r
        w = fn * p2
        r = t - w
        w = fn * p2t - ((t - r) - w)
        x = r - w
        tail = (r - x) - w
        z = x * x
        s1, s2, s3, s4, s5, s6 = (-1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04,
                                  2.75573137070700676789e-06, -2.50507602534068634195e-08, 1.58969099521155010221e-10)
        v = z * x
        rs = s2 + z * (s3 + z * (s4 + z * (s5 + z * s6)))
        sn = x - ((z * (0.5 * tail - v * rs) - tail) - v * s1)
        c1, c2, c3, c4, c5, c6 = (4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05,
                                  -2.75573143513906633035e-07, 2.08757232129817482790e-09, -1.13596475577881948265e-11)
        rc = z * (c1 + z * (c2 + z * (c3 + z * (c4 + z * (c5 + z * c6)))))
        hz = 0.5 * z
        ww = 1.0 - hz
        cs = ww + (((1.0 - ww) - hz) + (z * rc - x * tail))
        return np.where(n & 1, -(cs / sn), sn / cs)
    rng = np.random.default_rng(0)
    th = np.concatenate((rng.uniform(0, 2 * np.pi, 100000), np.pi / 2 + rng.uniform(-1e-3, 1e-3, 5000),
                         3 * np.pi / 2 + rng.uniform(-1e-6, 1e-6, 5000), rng.uniform(0, 1e-3, 5000),
                         2 * np.pi - rng.uniform(0, 1e-3, 5000)))
    ref = np.array([math.tan(t) for t in th])
    ulp = np.abs(sg_tan(th) - ref) / np.spacing(np.abs(ref))
    assert ulp.max() <= 2.0


# ---------------------------------------------------------------------------------------------------------------------
# The pruning rules of the received-power phase (csrc/sg_beam.h: sg_lane_power), restated in NumPy
# and checked against the full profile the reference computes: the first maximum must lie in a listed group.
def _power_profile(r, ratio, ca_p0_beta0):
    """simulation.py:135-149 for one beam: scatterers in dict order (range r, ratio), float64 throughout."""
    c_tau = 299792458.0 * 1e-8
    R = np.round(np.linspace(0, 120 + c_tau, 1230), 2)
    prof = np.zeros(1230)
    amps, wins = [], []
    for rj, rt in zip(r, ratio):
        xsi = 0.0 if rj <= 0.9 else (1.0 if rj >= 1.0 else (1 / (1.0 - 0.9)) * rj + (0 - (1 / (1.0 - 0.9)) * 0.9))
        amp = ((ca_p0_beta0 * rt) * xsi) / (rj * rj)
        k0, k1 = int(np.ceil(rj * 10)), int(np.floor((rj + c_tau) * 10) + 1)
        for k in range(k0, min(k1, 1230)):
            prof[k] += amp * np.sin((np.pi * (R[k] - rj)) / c_tau) ** 2
        amps.append(amp)
        wins.append((k0, min(k1, 1230)))
    return prof, np.array(amps), wins, R


def _listed_bins(r, amps, wins, nb):
    """Stage A of sg_lane_power with best = 0: the bins it puts on the work list."""
    c_tau = 299792458.0 * 1e-8
    step = (120 + c_tau) / 1229
    need = 0.9966 * amps.max()
    listed = set()
    n = len(r)
    for t in range(n):
        A = amps[t]
        if not A > 0.0:
            continue
        k0, k1 = wins[t]
        oth, lo_trim, hi_trim = 0.0, k0, k1

        def visit(j):
            nonlocal oth, lo_trim, hi_trim
            q0, q1 = wins[j]
            if amps[j] > A or (amps[j] == A and j < t):
                if q0 <= k0:
                    lo_trim = max(lo_trim, q1)
                elif q1 >= k1:
                    hi_trim = min(hi_trim, q0)
            else:
                oth += amps[j]
        j = t - 1
        while j >= 0 and wins[j][1] > k0:
            visit(j)
            j -= 1
        j = t + 1
        while j < n and wins[j][0] < k1:
            visit(j)
            j += 1
        if (A + oth) * (1.0 + 1e-9) < need:
            continue
        q = (need * (1.0 - 1e-9) - oth * (1.0 + 1e-9)) / A
        ka, kb = k0, k1 - 1
        if q >= 0.5:
            om = 1.0 - q if q < 1.0 else 0.0
            delta = float(np.float32(np.sqrt(np.float32(1.26 * om)))) * (1.0 + 1e-6) + 1e-6
            Rc = r[t] + c_tau / 2
            D = delta * (c_tau / np.pi) + 0.006
            ka = max(ka, int(np.floor((Rc - D) / step)))
            kb = min(kb, int(np.ceil((Rc + D) / step)))
        ka, kb = max(ka, lo_trim), min(kb, hi_trim - 1)
        g = ka
        while g <= kb:
            listed.update(range(g, g + nb))
            g += nb
    return listed


@pytest.mark.parametrize("nb", [4, 8])
def test_power_phase_pruning_never_drops_the_first_maximum(nb):
    rng = np.random.default_rng(2024 + nb)
    ca = 0.9 * 255
    cases = 0
    for trial in range(4000):
        s = int(rng.integers(1, 9))
        d = float(rng.uniform(1.2, 115.0))
        kind = trial % 5
        if kind == 0:      # flakes anywhere in front of the target
            r = np.sort(rng.uniform(0.5, d, s))
        elif kind == 1:    # a cluster right in front of the target: overlapping windows
            r = np.sort(d - rng.uniform(0.0, 2.9, s))
        elif kind == 2:    # nested windows with nearly equal ranges
            r = np.sort(rng.uniform(0.95, d, 1) + rng.uniform(0, 0.3, s))
        elif kind == 3:    # near the sensor, around the xsi ramp
            r = np.sort(rng.uniform(0.85, 1.6, s))
        else:              # evenly spaced at about one window length
            r = np.sort(rng.uniform(1.0, 3.0) + np.arange(s) * rng.uniform(2.5, 3.5))
        r = r[(r > 0.3) & (r < d)]
        if r.size == 0:
            continue
        ratio = rng.uniform(0.001, 0.4, r.size)
        if trial % 7 == 0:
            ratio[:] = ratio[0]                       # equal ratios: amplitudes differ only through 1 / r^2
        tgt = float(np.clip(1.0 - ratio.sum(), 0.0, 1.0)) if trial % 3 else 0.0
        rr = np.append(r, d)
        rt = np.append(ratio, tgt)
        if trial % 11 == 0 and rr.size >= 2:          # two scatterers with exactly the same amplitude
            rt[1] = rt[0] * (rr[1] / rr[0]) ** 2
        prof, amps, wins, R = _power_profile(rr, rt, ca)
        if amps.max() <= 0:
            continue
        k_star = int(np.argmax(prof))                 # first maximum (simulation.py:151)
        listed = _listed_bins(rr, amps, wins, nb)
        assert k_star in listed, (trial, rr, rt, k_star)
        # and nothing outside the listed bins ties with it
        out = np.ones(1230, bool)
        out[[k for k in listed if 0 <= k < 1230]] = False
        assert not np.any(prof[out] >= prof[k_star]), (trial, rr, rt)
        cases += 1
    assert cases > 3200


# ---------------------------------------------------------------------------------------------------------------------
# The single slot walk of phase 2 (csrc/sg_beam.h), restated in NumPy, against the oracle's compute_occlusion_dict
# (itself pinned to the reference by the L2 golden vectors): same keys, bit-identical ratios.
def _single_walk_dict(right, left, intervals, d, beam_div_deg):
    a1 = intervals[:, 0].copy()
    a2 = intervals[:, 1].copy()
    rho = intervals[:, 2]
    L = len(a1)
    ra, la = right, left
    if ra > la:                                                  # simulation.py:260-263
        ra = ra - 2 * np.pi
        sw = a1 > a2
        a1[sw] = a1[sw] - 2 * np.pi
    delta = np.radians(beam_div_deg)
    pts = np.concatenate(([ra, la], a1, a2))
    e, e_max = pts.min(), pts.max()
    slots = [[] for _ in range(L)]                               # widths per owner, in walk order
    free = []
    while e < e_max:
        own = -1
        nxt = e_max
        for q in range(L):
            if own < 0 and a1[q] <= e < a2[q]:
                own = q
            for v in (a1[q], a2[q]):
                if e < v < nxt:
                    nxt = v
        for v in (ra, la):
            if e < v < nxt:
                nxt = v
        (free if own < 0 else slots[own]).append(nxt - e)
        e = nxt
    out = []
    for j in range(L):
        if not slots[j]:
            continue
        if len(slots[j]) < 8:                                    # a running sum is np.sum for fewer than 8 addends
            acc = slots[j][0]
            for w in slots[j][1:]:
                acc = acc + w
            total = 0.0 + acc
        else:
            total = np.sum(np.array(slots[j]))
        out.append((j, float(rho[j]), float(np.clip(total / delta, 0, 1))))
    out.append((-1, float(d), float(np.clip(np.sum(np.array(free)) / delta if free else 0.0, 0, 1))))
    return out


def test_single_slot_walk_equals_the_reference_assignment():
    from oracle import snow_oracle as so
    rng = np.random.default_rng(77)
    bd = float(np.degrees(3e-3))
    for trial in range(1500):
        centre = rng.uniform(0, 2 * np.pi) if trial % 4 else rng.choice([1e-4, 2 * np.pi - 1e-4, 7e-4])   # also around the seam (Q9)
        right, left = centre - 1.5e-3, centre + 1.5e-3
        if right < 0:
            right += 2 * np.pi
        if left > 2 * np.pi:
            left -= 2 * np.pi
        L = int(rng.integers(1, 21))
        c = centre + rng.uniform(-2.2e-3, 2.2e-3, L)
        hw = rng.exponential(4e-4, L) + 1e-6
        lo, hi = np.maximum(c - hw, centre - 1.5e-3), np.minimum(c + hw, centre + 1.5e-3)
        ok = hi > lo
        lo, hi = lo[ok], hi[ok]
        if lo.size == 0:
            continue
        if trial % 5 == 0:                                       # shared endpoints: np.unique must not create empty slots
            hi[: lo.size // 2] = np.resize(lo[lo.size // 2:], lo.size // 2) if lo.size > 1 else hi[:0]
            ok = hi > lo
            lo, hi = lo[ok], hi[ok]
            if lo.size == 0:
                continue
        lo = np.where(lo < 0, lo + 2 * np.pi, np.where(lo > 2 * np.pi, lo - 2 * np.pi, lo))
        hi = np.where(hi < 0, hi + 2 * np.pi, np.where(hi > 2 * np.pi, hi - 2 * np.pi, hi))
        rho = np.sort(rng.uniform(1.0, 50.0, lo.size))
        iv = np.column_stack((lo, hi, rho))
        want = so.compute_occlusion_dict((right, left), iv, 60.0, bd)
        got = _single_walk_dict(right, left, iv, 60.0, bd)
        assert [k for k, _, _ in got] == [k for k, _, _ in want], (trial, iv)
        for (k, r0, q0), (_, r1, q1) in zip(got, want):
            assert r0 == r1 and q0 == q1, (trial, k, q0, q1)


def test_result_buffer_pool_recycles_and_respects_its_limit():
    """Engine.result_buffers hands out views of pooled page-locked buffers; they return to the pool when the caller
    drops every view, and past PIN_LIMIT plain arrays are returned instead.  (No device: a fake context allocates
    ordinary memory.)"""
    import gc
    import threading
    from lidar_snow_sim_amd import engine

    class FakeCtx:
        def __init__(self):
            self.allocs = 0

        def pinned_empty(self, shape, dtype):
            self.allocs += 1
            return np.empty(shape, dtype)

    e = engine.Engine.__new__(engine.Engine)
    e.ctx = FakeCtx()
    e._lock = threading.Lock()
    e.batch_lock = threading.RLock()
    rows, src = e.result_buffers(1000, np.float32)
    assert rows.shape == (1000, 5) and rows.dtype == np.float32 and src.shape == (1000,) and src.dtype == np.int32
    rows[:] = 7.0
    src[:] = 3
    assert np.all(rows == 7.0) and np.all(src == 3)                    # the two views do not overlap
    view = rows[10:20]
    del rows, src
    gc.collect()
    assert not e.__dict__.get("_pin_pool")                             # a live view keeps the lease out
    del view
    gc.collect()
    assert len(e._pin_pool) == 1 and e._pin_out == 0
    r2, s2 = e.result_buffers(500, np.float64)                         # fits the pooled buffer: no new allocation
    assert e.ctx.allocs == 1 and r2.dtype == np.float64 and r2.shape == (500, 5)
    e.PIN_LIMIT = 1                                                    # anything new is now over the limit
    r3, s3 = e.result_buffers(10 ** 6, np.float32)
    assert e.ctx.allocs == 1 and r3.shape == (10 ** 6, 5)              # pageable fallback
    z, zs = e.result_buffers(0, np.float32)
    assert z.shape == (0, 5) and zs.shape == (0,)
    stage = e.staging_in(100, np.float32)
    assert stage.shape == (100, 5) and e.staging_in(50, np.float32).ctypes.data == stage.ctypes.data


def test_array_table_cache_holds_its_arrays_and_evicts_least_recently_used():
    """Engine.array_table_id: a table is found again by identity (`is`), never by a recycled address -- the cache keeps a
    reference to the array, so CPython cannot hand its id to another array while the entry lives; past ARRAY_TABLES entries
    the least recently used table is dropped from the device and its id is reused.  (No device: a fake context.)"""
    import gc
    import threading
    from lidar_snow_sim_amd import engine

    class FakeCtx:
        def __init__(self):
            self.up, self.freed = [], []

        def upload_table(self, tid, arr):
            self.up.append((tid, float(arr[0, 0])))

        def free_table(self, tid):
            self.freed.append(tid)

    e = engine.Engine.__new__(engine.Engine)
    e.ctx = FakeCtx()
    e._lock = threading.Lock()
    e._tables, e._arrays, e._free_ids, e._next_id = {}, {}, [], 0
    e.ARRAY_TABLES = 3
    a = np.full((4, 3), 1.0)
    ta = e.array_table_id(a)
    assert e.array_table_id(a) == ta and len(e.ctx.up) == 1
    addr = id(a)
    del a
    gc.collect()
    # an array of the same shape allocated now may or may not land on the old address: either way it is a NEW table,
    # because the cached entry still owns the old array
    b = np.full((4, 3), 2.0)
    tb = e.array_table_id(b)
    assert tb != ta and e.ctx.up[-1] == (tb, 2.0)
    assert addr in e._arrays and e._arrays[addr][0][0, 0] == 1.0
    c, d = np.full((4, 3), 3.0), np.full((4, 3), 4.0)
    tc = e.array_table_id(c)
    assert e.array_table_id(b) == tb                              # touch b: a is now the least recently used
    td = e.array_table_id(d)                                      # fourth table: a goes
    assert e.ctx.freed == [ta] and addr not in e._arrays
    assert len({tb, tc, td}) == 3
    f = np.full((4, 3), 5.0)
    tf = e.array_table_id(f)                                      # evicts c (least recently used), reuses a freed id
    assert e.ctx.freed == [ta, tc] and tf in (ta, tc)
    assert e.user_table_id() not in (tb, td, tf)


def test_stream_plan_draws_permutations_in_the_reference_order(tmp_path):
    """precompute.py:70-92 walks mode -> frame -> combo and augment() shuffles once per output it actually computes
    (simulation.py:483-486).  stream.plan regroups the items into (mode, combo) batches but draws in THAT order, so a
    seeded run hands every (frame, combo) the permutation the reference's loop hands it; existing outputs draw nothing."""
    import random
    from lidar_snow_sim_amd import stream
    lidar = tmp_path / "lidar_hdl64_strongest"
    lidar.mkdir()
    ids = ["a_1", "a_2", "b_1"]
    combos = [(10.5, 1e-6), (20.5, 2e-6)]
    done = stream.output_path(lidar, "gunn", 20.5, "a_2")
    done.parent.mkdir(parents=True)
    done.write_bytes(b"")
    random.seed(11)
    want = {}
    for mode in ("gunn", "sekhon"):
        for s in ids:
            for rr, _ in combos:
                if stream.output_path(lidar, mode, rr, s).is_file():
                    continue
                order = list(range(64))
                random.shuffle(order)
                want[(mode, rr, s)] = order
    random.seed(11)
    jobs = stream.plan(lidar, ids, ("gunn", "sekhon"), combos, batch=2)
    got = {(mode, rr, s): o for mode, rr, prefix, ss, orders in jobs for s, o in zip(ss, orders)}
    assert got == want and len(got) == 11
    # the skip decisions come from a listing taken up front: files that appear while the plan is drawn change nothing
    random.seed(11)
    it = stream.plan_iter(lidar, ids, ("gunn", "sekhon"), combos, batch=2)
    first = next(it)
    for s in ids:
        late = stream.output_path(lidar, "sekhon", 10.5, s)
        late.parent.mkdir(parents=True, exist_ok=True)
        late.write_bytes(b"")
    jobs2 = [first] + list(it)
    assert {(mode, rr, s): o for mode, rr, prefix, ss, orders in jobs2 for s, o in zip(ss, orders)} == want
    assert all(len(j[3]) <= 2 for j in jobs)
    assert {j[2] for j in jobs} == {f"{m}_{rr}_{occ}" for m in ("gunn", "sekhon") for rr, occ in combos}


def test_correctly_rounded_atan2_equals_glibc_where_glibc_is(tmp_path):
    """csrc/sg_atan_cr.h (float64 atan2 / atan of the float64 beam azimuth and of device-filed tables) against glibc on 4 * 10^6
    inputs: the two agree except on a few inputs in 10^4, and there it is GLIBC that is not correctly rounded -- every mismatch
    is checked against 70-digit arithmetic: ours lies within half an ULP of the true value, glibc's does not."""
    from decimal import Decimal, getcontext
    sys.path.insert(0, str(ROOT / "scripts"))
    import gen_atan_table as gen
    getcontext().prec = 70
    exe = tmp_path / "atan_cr_check"
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-o", str(exe), str(ROOT / "scripts" / "probe" / "atan_cr_check.cpp"), "-lm"])
    r = subprocess.run([str(exe), "4"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    summary = r.stdout.strip().splitlines()[-1]
    n, m2, m1 = (int(v) for v in re.findall(r"(\d+) inputs: atan2 mismatches (\d+), atan mismatches (\d+)", summary)[0])
    assert n == 4_000_000 and m2 + m1 < n // 500, summary                 # < 0.2 %
    checked = 0
    for line in r.stdout.splitlines():
        if not line.startswith("M "):
            continue
        _, fn, ys, xs, gs, os_ = line.split()
        y, x, g, o = (float.fromhex(v) for v in (ys, xs, gs, os_))
        yd, xd = Decimal(y), Decimal(x)
        # atan2 in 70 digits: octant reduction to atan of a ratio in [0, 1]
        ay, ax = abs(yd), abs(xd)
        t = gen.atan_dec(min(ay, ax) / max(ay, ax))
        if ay > ax:
            t = gen.PI / 2 - t
        if xd < 0:
            t = gen.PI - t
        if yd < 0:
            t = -t
        ulp = Decimal(abs(g - o))                                          # the two differ by one ULP
        assert abs(t - Decimal(o)) < ulp / 2 < abs(t - Decimal(g)), line
        checked += 1
    assert checked > 0


def test_flat_batch_validates_its_layout():
    """simulation.FlatBatch (frames back to back in one array, what the stream driver's readers fill) refuses anything the
    upload could not take as it is -- before a device is touched."""
    from lidar_snow_sim_amd.tools.snowfall.simulation import FlatBatch
    rows = np.zeros((10, 5), np.float32)
    fb = FlatBatch(rows, [0, 4, 4, 10])
    assert len(fb) == 3 and fb.frame(0).shape == (4, 5) and fb.frame(1).shape == (0, 5) and fb.frame(2).base is rows
    for bad_rows, bad_off in ((np.zeros((10, 6), np.float32), [0, 10]), (np.zeros((10, 5), np.int32), [0, 10]),
                              (np.zeros((10, 5), np.float32)[::2], [0, 5]), (rows, [1, 10]), (rows, [0, 11]), (rows, [0, 6, 4])):
        with pytest.raises(ValueError):
            FlatBatch(bad_rows, bad_off)


def test_stream_plan_iter_hands_batches_out_as_they_fill(tmp_path):
    """stream.plan_iter is the lazy form of stream.plan: same items, same permutations (one random.shuffle per item in the
    reference's nesting order), a batch as soon as its (mode, combo) group is full."""
    import random
    from lidar_snow_sim_amd import stream
    lidar = tmp_path / "lidar_hdl64_strongest"
    lidar.mkdir()
    ids = [f"a_{i}" for i in range(7)]
    combos = [(10.5, 1e-6), (20.5, 2e-6)]
    random.seed(3)
    eager = stream.plan(lidar, ids, ("gunn",), combos, batch=3)
    random.seed(3)
    it = stream.plan_iter(lidar, ids, ("gunn",), combos, batch=3)
    first = next(it)
    state_after_first = random.getstate()
    rest = list(it)
    assert [first] + rest == eager
    assert len(first[3]) == 3 and first[3] == ["a_0", "a_1", "a_2"]
    random.seed(3)
    for _ in range(5):                                     # the first full batch needs the draws of items 0 .. 4 (two combos interleave)
        random.shuffle(list(range(64)))
    assert random.getstate() == state_after_first


# ---- a seeded sharded stream == the unsharded one (precompute.py:70-92, simulation.py:482-486) -------------
def _plan_items(jobs):
    return {(mode, rr, s): tuple(o) for mode, rr, prefix, ss, orders in jobs for s, o in zip(ss, orders)}


def _stream_fixture(tmp_path, n_ids=23):
    from lidar_snow_sim_amd import stream
    lidar = tmp_path / "lidar_hdl64_strongest"
    lidar.mkdir(exist_ok=True)
    ids = [f"2018-02-03_{i:05d}" for i in range(n_ids)]
    combos = [(10.5, 1e-6), (20.5, 2e-6), (20.9, 3e-6)]            # the last two share int(rainfall_rate): one output path
    for s in (ids[2], ids[7]):                                     # outputs that already exist draw nothing (precompute.py:91-92)
        done = stream.output_path(lidar, "gunn", 10.5, s)
        done.parent.mkdir(parents=True, exist_ok=True)
        done.write_bytes(b"")
    return lidar, ids, combos


@pytest.mark.parametrize("world", [2, 3, 8])
def test_union_of_the_ranks_plans_is_the_single_rank_plan(tmp_path, world):
    import random
    from lidar_snow_sim_amd import stream
    lidar, ids, combos = _stream_fixture(tmp_path)
    random.seed(5)
    whole = _plan_items(stream.plan(lidar, ids, ("gunn", "sekhon"), combos, batch=4))
    assert len(whole) == 2 * len(ids) * 2 - 2                      # two distinct output paths per (mode, frame); two exist already
    union = {}
    for rank in range(world):
        random.seed(5)                                             # every rank seeds alike, as the reference's one process does once
        part = _plan_items(stream.plan(lidar, ids, ("gunn", "sekhon"), combos, batch=4, rank=rank, world=world))
        assert all(ids.index(k[2]) % world == rank for k in part)  # round-robin ownership
        assert not set(part) & set(union)
        union.update(part)
    assert union == whole                                          # item for item, permutation for permutation


_PLAN_WORKER = r"""
import json, os, random, sys
sys.path.insert(0, {root!r})
from pathlib import Path
from lidar_snow_sim_amd import dist as sd, stream
rank, world, port, lidar = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], Path(sys.argv[4])
os.environ["MASTER_PORT"] = port
ids = [f"2018-02-03_{{i:05d}}" for i in range(23)]
combos = [(10.5, 1e-6), (20.5, 2e-6), (20.9, 3e-6)]
existing = stream.existing_outputs(lidar, ("gunn", "sekhon"), combos)
d = sd.init("gloo", rank, world)
d.barrier()                                                        # every rank has listed before any rank writes
random.seed(5)
items = []
for mode, rr, prefix, ss, orders in stream.plan_iter(lidar, ids, ("gunn", "sekhon"), combos, 4, rank=rank, world=world, existing=existing):
    for s, o in zip(ss, orders):
        out = stream.output_path(lidar, mode, rr, s)               # "write" while the other rank is still planning
        out.parent.mkdir(parents=True, exist_ok=True)
        out.write_bytes(b"x")
        items.append([mode, rr, s, o])
d.barrier()
print("PLAN", json.dumps(items), flush=True)
d.destroy_process_group()
"""


def test_two_process_sharded_plan_equals_the_single_rank_plan(tmp_path):
    """Two gloo ranks plan AND write concurrently; their union must still be the one-rank plan (the skip test reads the listing
    taken before the barrier, not the live tree the other rank is filling)."""
    import json
    import random
    from lidar_snow_sim_amd import stream
    lidar, ids, combos = _stream_fixture(tmp_path)
    random.seed(5)
    whole = _plan_items(stream.plan(lidar, ids, ("gunn", "sekhon"), combos, batch=4))
    script = tmp_path / "plan_worker.py"
    script.write_text(_PLAN_WORKER.format(root=str(ROOT)))
    port = str(29900 + os.getpid() % 90)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", port, str(lidar)], stdout=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    union = {}
    for out in outs:
        for mode, rr, s, o in json.loads([ln for ln in out.splitlines() if ln.startswith("PLAN")][0][5:]):
            assert (mode, rr, s) not in union
            union[(mode, rr, s)] = tuple(o)
    assert union == whole


def test_stream_reader_rejects_a_truncated_bin(tmp_path):
    """np.fromfile(...).reshape((-1, 5)) raises on a file that is not whole rows (precompute.py:78); so does the reader --
    checked without a GPU through the size test it runs before it touches the device."""
    from lidar_snow_sim_amd import stream
    src = open(stream.__file__).read()
    assert "nbytes % 20" in src and "not a whole number of float32 N x 5 rows" in src


def test_fov_projection_divides_by_the_rectified_z_like_openpcdet():
    """The camera-FOV test (simulation.py:39-47) goes through OpenPCDet's Calibration.rect_to_img (the reference's un-vendored
    submodule): image coordinates = (rect_hom . P2^T)[:, :2] / RECTIFIED z, depth = third homogeneous coordinate - P2[2, 3].  With a
    KITTI P2 (P2[2, 3] = 2.7e-3) that is not the division by the third homogeneous coordinate; host mirror and oracle restatement
    must both be the former, and agree with each other point for point."""
    from lidar_snow_sim_amd.calibration import Calibration, get_fov_flag
    from oracle import snow_oracle as so
    P2 = np.array([[721.5377, 0, 609.5593, 44.85728], [0, 721.5377, 172.854, 0.2163791], [0, 0, 1, 0.002745884]])
    R0 = np.array([[0.9999239, 0.00983776, -0.007445048], [-0.009869795, 0.9999421, -0.004278459], [0.007402527, 0.004351614, 0.9999631]])
    V2C = np.array([[7.533745e-03, -9.999714e-01, -6.166020e-04, -4.069766e-03], [1.480249e-02, 7.280733e-04, -9.998902e-01, -7.631618e-02],
                    [9.998621e-01, 7.523790e-03, 1.480755e-02, -2.717806e-01]])
    cal = Calibration(P2=P2, R0=R0, V2C=V2C)
    rng = np.random.default_rng(5)
    pts = np.column_stack([rng.uniform(-5, 80, 20000), rng.uniform(-40, 40, 20000), rng.uniform(-3, 3, 20000)])
    rect = cal.lidar_to_rect(pts)
    img, depth = cal.rect_to_img(rect)
    hom = np.hstack([rect, np.ones((len(rect), 1))]) @ P2.T
    assert np.array_equal(img, (hom[:, :2].T / rect[:, 2]).T) and np.array_equal(depth, hom[:, 2] - P2[2, 3])
    other = (hom[:, :2].T / hom[:, 2]).T                                   # what rounds 1 - 3 divided by
    assert np.abs(img - other).max() > 1e-3                                # ... is a different picture coordinate
    a = get_fov_flag(rect, (1024, 1920), cal)
    b = so.fov_flag(pts, V2C, R0, P2, (1024, 1920))
    assert np.array_equal(a, b) and 500 < a.sum() < 19500


def test_bench_c5_dry_eight_ranks_write_disjoint_files(tmp_path):
    """`python bench.py --gpus 8 --dry --workload C5`: the 8-GPU stream run as one command, without the GPU work -- eight gloo ranks shard ONE
    stream of frame files round-robin through lidar_snow_sim_amd.stream (reader / writer threads capped to each rank's share of the
    CPUs), every output file is written exactly once, and rank 0's line carries every rank's stage times and the host-thread budget."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    n = 88                                                                     # not a multiple of 8 x batch: ragged last batches
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--dry", "--workload", "C5", "--frames", str(n),
                        "--c5-dir", str(tmp_path), "--c5-keep"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    cfg = rec["config"]
    assert rec["n_gpus"] == 8 and rec["dry"] is True and rec["scaling"] == "strong" and cfg["ranks_seen"] == 8
    assert cfg["files_written"] == n and cfg["frames"] == n
    per = cfg["stage_busy_s_per_rank"]
    assert [p["rank"] for p in per] == list(range(8)) and all(p["wall_s"] > 0 and p["read_s"] > 0 and p["write_s"] > 0 for p in per)
    assert rec["ms_per_step"] >= max(p["wall_s"] for p in per) * 1e3 * 0.999   # max over ranks
    host = cfg["host"]
    tpr = host["threads_per_rank"]
    assert tpr["readers"] >= 1 and tpr["writers"] >= 1 and 8 * (tpr["readers"] + tpr["writers"]) <= max(16, 2 * host["cpus_usable"] + 16)
    out = sorted(p.name for p in (tmp_path / "snowfall_simulation").rglob("*.bin"))
    assert out == sorted(f"2018-02-03_{i:05d}.bin" for i in range(n))          # every frame once, whichever rank owned it
    src = tmp_path / "lidar_hdl64_strongest"
    for name in out[:: 11]:                                                    # dry: the frame comes back as it went in
        written = next((tmp_path / "snowfall_simulation").rglob(name))
        assert written.read_bytes() == (src / name).read_bytes()


def test_stream_run_refuses_a_sharded_run_without_a_common_listing(tmp_path):
    """Ranks that list the output tree at their own start times skip different items and drift apart in their seeded draws (round-4
    advisor): stream.run(world > 1) without `existing` needs a process group to list behind, else it raises before touching a device."""
    from lidar_snow_sim_amd import stream
    (tmp_path / "lidar").mkdir()
    with pytest.raises(ValueError, match="existing"):
        stream.run(tmp_path / "lidar", ["a,1"], rank=1, world=2, modes=("gunn",), combos=[(1.0, 0.1)])


def test_line_fit_of_the_threshold_callback_is_numpy_s_own():
    """The threshold callback fits its noise line with the ufunc calls np.mean and np.cov make themselves (_linregress_line): same bits as
    scipy.stats.linregress, which the host path and the reference call (wet_ground/augmentation.py:248-249), for every length the 50-bin
    histogram can leave."""
    from scipy.stats import linregress
    from lidar_snow_sim_amd.tools.wet_ground.augmentation import _linregress_line
    rng = np.random.default_rng(77)
    xmid = (np.linspace(10, 70, 51)[:-1] + np.linspace(10, 70, 51)[1:]) / 2
    for trial in range(400):
        n = 4 + trial % 47
        u = np.zeros(50, bool)
        u[rng.choice(50, n, replace=False)] = True
        x, y = xmid[u], 5.0 + rng.integers(1, 2555, n) * rng.uniform(0.01, 0.2)
        ref = linregress(x, y)
        slope, intercept = _linregress_line(x, y)
        assert slope == ref.slope and intercept == ref.intercept, (trial, n)


def test_sorting_networks_of_the_dict_are_the_generator_s_and_sort():
    """csrc/sg_sortnet.h is what scripts/gen_sortnet.py writes (nothing edited by hand), and its comparator lists sort: every 0 / 1 input of 10
    and 18 values (the zero-one principle), random inputs with ties for 34."""
    sys.path.insert(0, str(ROOT / "scripts"))
    import gen_sortnet as gen
    assert (ROOT / "lidar_snow_sim_amd" / "csrc" / "sg_sortnet.h").read_text() == gen.header_text()
    for n in gen.SIZES:
        net = gen.network(n)
        assert all(0 <= i < j < n for i, j in net)
        gen.check(n, net)
