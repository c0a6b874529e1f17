"""-m gpu: the device ground-plane estimate (csrc/snowgpu_plane.hip; the counterpart of tools/wet_ground/planes.py:12-50).

The reference's own estimator is unpinned (unseeded RANSAC that raises with scikit-learn >= 1.2: SURVEY 8 c), so these tests
hold the engine's estimators to what they claim: 'reference' = the plane the reference returns today, 'lsq' = NumPy's lstsq on
the rows of the reference's crop, 'ransac' = deterministic per seed, robust to outliers, equal to 'lsq' on its consensus set.
"""
import numpy as np
import pytest
import torch  # noqa: F401  -- before libsnowgpu.so is loaded (see test_gpu_parity.py)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from lidar_snow_sim_amd import engine
    return engine.get_engine(0)


def _road(n, seed, dtype, tilt=(0.012, -0.02), height=-1.72, noise=0.01, outliers=0.0):
    """A strip of road z = tilt . (x, y) + height (+ noise) inside the reference's crop window, plus clutter around it."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(5.0, 80.0, n)
    y = rng.uniform(-6.0, 6.0, n)
    z = tilt[0] * x + tilt[1] * y + height + rng.normal(0.0, noise, n)
    bad = rng.random(n) < outliers
    z[bad] += rng.uniform(-0.12, 0.12, bad.sum())
    pc = np.column_stack((x, y, z, rng.integers(1, 200, n), rng.integers(0, 64, n))).astype(dtype)
    return pc


def _lstsq_plane(pc):
    from lidar_snow_sim_amd.tools.wet_ground.planes import ground_crop
    sub = pc[ground_crop(pc)].astype(np.float64)
    A = np.column_stack((sub[:, 0], sub[:, 1], np.ones(len(sub))))
    c, *_ = np.linalg.lstsq(A, sub[:, 2], rcond=None)
    w = np.array([c[0], c[1], -1.0])
    return w / np.linalg.norm(w), c[2], len(sub)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_lsq_plane_equals_numpy_lstsq_on_the_reference_crop(eng, dtype):
    from lidar_snow_sim_amd.tools.wet_ground.planes import calculate_plane
    for seed in range(4):
        pc = _road(20000 + 7777 * seed, seed, dtype)
        w, h, info = calculate_plane(pc, method="lsq", return_info=True)
        w0, h0, m = _lstsq_plane(pc)
        assert info["model"] == "lsq" and info["strip_rows"] == m and info["fit_rows"] == m     # the device crop is the reference's, row for row
        np.testing.assert_allclose(w, w0, rtol=0, atol=1e-12)
        np.testing.assert_allclose(h, h0, rtol=1e-12, atol=1e-12)


def test_reference_method_is_the_flat_earth_plane_and_small_crops_fall_back(eng):
    from lidar_snow_sim_amd.tools.wet_ground.planes import calculate_plane
    pc = _road(5000, 1, np.float32)
    assert calculate_plane(pc) == ([0, 0, 1], -1.55)                       # what the reference returns today (planes.py:43-48)
    assert calculate_plane(pc, standart_height=-1.6) == ([0, 0, 1], -1.6)
    few = pc[:2000].copy()
    few[:, 0] = 5.0                                                        # nothing in the strip (x <= 10)
    few[:5, :3] = [[20, 0, -1.7], [30, 1, -1.7], [40, -1, -1.75], [50, 2, -1.8], [60, 0, -1.9]]
    for method in ("lsq", "ransac"):
        w, h, info = calculate_plane(few, method=method, return_info=True)  # 5 rows <= 5 columns (planes.py:29-32)
        assert (w, h) == ([0, 0, 1], -1.55) and info["model"] == "flat_earth" and info["strip_rows"] == 5
    six = np.column_stack((few, np.zeros(len(few)))).astype(np.float32)    # a sixth column raises the bar to 6 rows
    six[5, :3] = [25, 0.5, -1.71]
    assert calculate_plane(six, method="lsq", return_info=True)[2]["model"] == "flat_earth"
    few[5, :3] = [25, 0.5, -1.71]
    assert calculate_plane(few, method="lsq", return_info=True)[2]["model"] == "lsq"          # 6 rows > 5 columns


def test_ransac_plane_is_seeded_and_ignores_what_least_squares_does_not(eng):
    from lidar_snow_sim_amd.tools.wet_ground.planes import calculate_plane
    tilt, height = (0.001, -0.003), -1.7                                  # a road that stays inside the strip out to 70 m
    pc = _road(30000, 5, np.float32, tilt=tilt, height=height, noise=0.004)
    rng = np.random.default_rng(6)
    pot = rng.random(len(pc)) < 0.05
    pc[pot, 2] -= 0.25                                                     # potholes: 5 % of the rows a quarter of a metre low
    w1, h1, i1 = calculate_plane(pc, method="ransac", seed=11, return_info=True)
    w2, h2, i2 = calculate_plane(pc, method="ransac", seed=11, return_info=True)
    assert np.array_equal(w1, w2) and h1 == h2 and i1 == i2                # same cloud, same seed: same plane, bit for bit
    assert i1["model"] == "ransac" and i1["valid_trials"] > 900 and 3 <= i1["fit_rows"] < i1["strip_rows"]
    true_w = np.array([tilt[0], tilt[1], -1.0])
    true_w /= np.linalg.norm(true_w)
    wl, hl = calculate_plane(pc, method="lsq")
    assert np.abs(w1 - true_w).max() < 1e-3 and abs(h1 - height) < 3e-3    # the consensus set leaves the potholes out ...
    assert abs(hl - height) > 8e-3                                         # ... least squares over the whole strip cannot
    w3, h3 = calculate_plane(pc, method="ransac", seed=12)
    assert np.abs(w3 - w1).max() < 1e-3 and abs(h3 - h1) < 3e-3            # another seed: other samples, the same road
    w4, h4, i4 = calculate_plane(pc, method="ransac", seed=11, trials=64, return_info=True)
    assert i4["valid_trials"] <= 64 and np.abs(w4 - true_w).max() < 2e-3


def test_batch_with_empty_and_ragged_frames_and_planes_inside_augment(eng):
    """plane=None at the C ABI: calculate_plane runs on the device inside the batch (simulation.py:449)."""
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    from lidar_snow_sim_amd.tools.snowfall.simulation import augment_batch
    from lidar_snow_sim_amd.tools.snowfall import sampling as smp
    occ, rate = smp.compute_occupancy(2.5, 1.6), smp.snowfall_rate_to_rainfall_rate(2.5, 1.6)
    tabs = [smp.dart_throwing(occ, rate, 30.0, np.random.default_rng(42 + i), "gunn") for i in range(4)]
    tl = [tabs[i % 4] for i in range(64)]
    frames = [synthetic_sweep(64, 2048, seed=1000 + f, intensity="lambert").reshape(64, 2048, 5)[:, ::16].reshape(-1, 5) for f in range(3)]
    frames.insert(1, frames[0][::3].copy())
    bd = float(np.degrees(3e-3))
    orders = [list(np.random.default_rng(f).permutation(64)) for f in range(len(frames))]
    eng.ctx.set_plane_method("lsq")
    try:
        off = np.concatenate(([0], np.cumsum([len(f) for f in frames])))
        planes, info = eng.ctx.estimate_planes(np.concatenate(frames), off)
    finally:
        eng.ctx.set_plane_method("reference")
    assert info[0, 1] == 1 and abs(planes[0, 3] + 1.7) < 0.02 and np.allclose(planes[0, :3], [0, 0, -1], atol=2e-3)   # ground at z = -1.7 (+ wall hits)
    # the device's own estimate inside the batch == that plane injected
    got = augment_batch(frames, "unused", bd, particles=tl, orders=orders, plane_method="lsq", return_src=True)
    ref = augment_batch(frames, "unused", bd, particles=tl, orders=orders, return_src=True,
                        planes=[(planes[i, :3], planes[i, 3]) if info[i, 1] else ([0, 0, 1], -1.55) for i in range(len(frames))])
    for (s1, a1, i1), (s2, a2, i2) in zip(got, ref):
        assert tuple(s1) == tuple(s2) and np.array_equal(i1, i2) and np.array_equal(a1, a2)
    # default method: the reference's flat-earth plane, as an injected ([0, 0, 1], -1.55) gives
    frames2 = [f.copy() for f in frames]
    for f in frames2:
        f[:, 2] += 3.25                       # ground rows now sit at z = +1.55, where ([0, 0, 1], -1.55) looks for them
    got = augment_batch(frames2, "unused", bd, particles=tl, orders=orders, return_src=True)
    ref = augment_batch(frames2, "unused", bd, particles=tl, orders=orders, return_src=True, planes=[([0, 0, 1], -1.55)] * len(frames2))
    for (s1, a1, i1), (s2, a2, i2) in zip(got, ref):
        assert tuple(s1) == tuple(s2) and np.array_equal(i1, i2) and np.array_equal(a1, a2)


def test_wet_ground_estimates_its_plane_on_the_device(eng):
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    from lidar_snow_sim_amd.tools.wet_ground.augmentation import ground_water_augmentation
    pc = synthetic_sweep(64, 2048, seed=1003, intensity="lambert")
    kw = dict(water_height=0.0008, pavement_depth=0.001, noise_floor=0.7, power_factor=15, flat_earth=False, delta=0.5, replace=False, debug=False)
    from lidar_snow_sim_amd.tools.wet_ground.planes import calculate_plane
    a = ground_water_augmentation(pc, plane_method="lsq", **kw)
    w, h = calculate_plane(pc, method="lsq")
    assert np.allclose(w, [0, 0, -1], atol=2e-3) and abs(h + 1.7) < 0.02    # the synthetic ground is the plane z = -1.7 (+ wall hits in the strip)
    b = ground_water_augmentation(pc, plane=(w, h), **kw)                   # the same plane injected: the same rows, bit for bit
    assert a.shape == b.shape and np.array_equal(a, b) and 1000 < a.shape[0] < pc.shape[0]
    c = ground_water_augmentation(pc, **kw)                                 # default: the plane the reference returns today ...
    d = ground_water_augmentation(pc, plane=([0, 0, 1], -1.55), **kw)       # ... (rows of the far wall around z = +1.55 pass for "ground")
    assert np.array_equal(c, d)


@pytest.mark.parametrize("method", ["lsq", "ransac"])
def test_fewer_than_three_strip_rows_give_the_flat_earth_plane_whatever_min_rows_is(eng, method):
    """A direct C-ABI caller may set min_rows below 2 (the Python mirror always passes the column count): a strip of 0, 1 or 2 rows
    still has no plane, and the RANSAC draws -- which index rows m - 1 and m - 2 -- must not run (round-4 advisor)."""
    pc = _road(3000, 5, np.float32)
    pc[:, 0] = 5.0                                                         # nothing in the strip (x <= 10) ...
    frames = []
    for k in (0, 1, 2, 3):
        f = pc.copy()
        for i in range(k):
            f[i, :3] = (20 + 7 * i, 0.5 * i, -1.7 - 0.01 * i)                # ... except k rows
        frames.append(f)
    rows = np.concatenate(frames)
    off = np.arange(5) * pc.shape[0]
    for min_rows in (0, 1):
        eng.ctx.set_plane_method(method, seed=3, trials=64, min_rows=min_rows)
        try:
            planes, info = eng.ctx.estimate_planes(rows, off)
        finally:
            eng.ctx.set_plane_method("reference")
        for k in (0, 1, 2):
            assert tuple(planes[k]) == (0.0, 0.0, 1.0, -1.55), (method, min_rows, k, planes[k])
        assert np.isfinite(planes[3]).all()
