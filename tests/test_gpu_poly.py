"""-m gpu: estimation_method='poly' of the wet-ground model (tools/wet_ground/augmentation.py:171-192, :223-229, :243-246) on the device.

Parity is unpinned by construction -- the reference draws its RANSAC samples from NumPy's unseeded global generator, and with NumPy >= 1.23
the branch raises before it gets there (tests/golden/make_golden_poly.py) -- so the tests hold the device to
  * the reference's own deterministic parts on the L6 clouds (fixture L9, made by importing the reference): the laser-power quadratic
    np.polyfit(dist, I / cos, 2), the inputs of ransac_polyfit, and -- on these clouds no RANSAC trial of any of the reference's 32 runs
    undercut the fit over all points -- the noise quadratic and the output rows;
  * a NumPy restatement of ransac_polyfit fed with the SAME Philox draws, through the tap snowgpu_debug_ransac_polyfit, on data where
    trials do win;
  * determinism per seed.
"""
import warnings

import numpy as np
import pytest
import torch  # noqa: F401  -- before libsnowgpu.so is loaded (one HIP runtime per process)

pytestmark = pytest.mark.gpu

PLANE = (np.array([0.0, 0.0, -1.0]), -1.7)
M32 = 0xFFFFFFFF


def philox4x32_10(seed, idx, group, tag):
    """Philox4x32-10 block (idx, group, tag) under key `seed` -- csrc/sg_philox.h::philox_u32x4 in Python integers."""
    c = [idx & M32, (idx >> 32) & M32, group & M32, tag & M32]
    k = [seed & M32, (seed >> 32) & M32]
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & M32, (p0 >> 32) ^ c[3] ^ k[1], p0 & M32]
        k = [(k[0] + 0x9E3779B9) & M32, (k[1] + 0xBB67AE85) & M32]
    return c


def ransac_polyfit_with_device_draws(x, y, seed, frame, n=15, k=100, t=0.1, d=15, f=0.8):
    """augmentation.py:171-192 with order 2, statement for statement, except that trial kk draws its n indices from the Philox
    block the device uses instead of np.random.randint, and that a trial whose draws hold fewer than three distinct abscissae is
    skipped (np.polyfit would answer with a rank warning and the minimum-norm fit; the device skips)."""
    m = len(x)
    bestfit = np.polyfit(x, y, 2)
    besterr = np.sum(np.abs(np.polyval(bestfit, x) - y))
    win = -1
    for kk in range(k):
        words = []
        for d4 in range((n + 3) // 4):
            words += philox4x32_10(seed, frame, kk * 4 + d4, 0x504F4C59)
        maybe = np.array([(w * m) >> 32 for w in words[:n]])
        if len(set(x[maybe])) < 3:
            continue
        maybemodel = np.polyfit(x[maybe], y[maybe], 2)
        also = np.abs(np.polyval(maybemodel, x) - y) < t
        if sum(also) > d and sum(also) > len(x) * f:
            better = np.polyfit(x[also], y[also], 2)
            err = np.sum(np.abs(np.polyval(better, x[also]) - y[also]))
            if err < besterr:
                bestfit, besterr, win = better, err, kk
    return bestfit, win


@pytest.fixture(scope="module")
def eng():
    from lidar_snow_sim_amd import engine
    return engine.get_engine(0)


@pytest.fixture(scope="module")
def l9(golden):
    return golden("L9_wet_poly")


def _curve_close(a, b, rtol):
    d = np.linspace(3.0, 119.0, 117)
    np.testing.assert_allclose(np.polyval(a, d), np.polyval(b, d), rtol=rtol, atol=0)


@pytest.mark.parametrize("kind", ["clean", "outliers", "few", "steep"])
def test_device_ransac_polyfit_equals_numpy_with_the_same_draws(eng, kind):
    rng = np.random.default_rng({"clean": 1, "outliers": 2, "few": 3, "steep": 4}[kind])
    xm = np.linspace(10, 70, 51)
    xm = (xm[:-1] + xm[1:]) / 2
    for rep in range(6):
        if kind == "few":
            x = np.sort(rng.choice(xm, size=int(rng.integers(3, 9)), replace=False))
        else:
            x = np.sort(rng.choice(xm, size=int(rng.integers(38, 51)), replace=False))
        true = np.array([0.004, -0.2, 9.0]) if kind != "steep" else np.array([0.02, -1.9, 60.0])
        y = np.polyval(true, x) + rng.normal(0, 0.02, x.shape[0])
        if kind in ("outliers", "steep"):
            bad = rng.choice(x.shape[0], size=max(1, x.shape[0] // 8), replace=False)
            y[bad] += rng.uniform(2.0, 9.0, bad.shape[0]) * rng.choice([-1.0, 1.0], bad.shape[0])
        for seed in (0, 7, 2 ** 40 + 3):
            for frame in (0, 5):
                coef, trial = eng.ctx.debug_ransac_polyfit(x, y, seed=seed, frame=frame)
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    ref, win = ransac_polyfit_with_device_draws(x, y, seed, frame)
                assert trial == win, (kind, rep, seed, frame, trial, win)
                _curve_close(coef, ref, 1e-9)
                if kind == "outliers":
                    assert trial >= 0                    # a trial without an outlier among its 15 draws exists, and its refit wins


@pytest.mark.parametrize("case", range(8))
def test_L9_poly_on_the_reference_clouds(eng, l9, golden, case):
    """The L6 clouds through ground_water_augmentation(estimation_method='poly'): the laser-power quadratic equals the reference's
    np.polyfit, the noise quadratic its ransac_polyfit (whose 32 runs all ended with the fit over all points on these clouds, as the
    device's trials do), the rows its output; and the same seed gives the same bytes."""
    from lidar_snow_sim_amd.tools.wet_ground.augmentation import ground_water_augmentation
    d6 = golden("L6_wet_ground")
    pc = d6[f"c{case}_pc"]
    kw = dict(water_height=0.0008, pavement_depth=0.001, noise_floor=0.7, power_factor=15, estimation_method="poly",
              flat_earth=bool(d6[f"c{case}_flat"]), debug=False, delta=0.5, replace=bool(d6[f"c{case}_replace"]), plane=PLANE, return_src=True)
    assert len({tuple(np.round(q, 10)) for q in l9[f"c{case}_pmin"]}) == 1 and len(set(l9[f"c{case}_rows"].tolist())) == 1
    outs = []
    for seed in (0, 1, 0):
        out, src = ground_water_augmentation(pc, poly_seed=seed, **kw)
        fit = eng.ctx.wet_last_fit(1)[0]
        outs.append((out, src, fit))
        assert int(fit[6]) == int(l9[f"c{case}_n_ground"])
        _curve_close(fit[0:3], l9[f"c{case}_p"], 1e-9 if pc.dtype == np.float64 else 1e-6)
        # the device's (x, min_vals) are the reference's: its RANSAC on the fixture's points with the device's draws gives the device's curve
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref, win = ransac_polyfit_with_device_draws(l9[f"c{case}_x"], l9[f"c{case}_y"], seed, 0)
        assert int(fit[7]) == win
        _curve_close(fit[3:6], ref, 1e-8)
        if win < 0:
            _curve_close(fit[3:6], l9[f"c{case}_pmin"][0], 1e-8)
            ref_out = l9[f"c{case}_out0"]
            assert out.dtype == np.float64 and out.shape == ref_out.shape
            assert np.array_equal(out[:, [0, 1, 2, 4]], ref_out[:, [0, 1, 2, 4]])
            np.testing.assert_allclose(out[:, 3], ref_out[:, 3], rtol=1e-9 if pc.dtype == np.float64 else 1e-6, atol=1e-9)
        assert np.array_equal(pc[src, :3].astype(np.float64), out[:, :3])
    assert outs[0][0].tobytes() == outs[2][0].tobytes() and np.array_equal(outs[0][1], outs[2][1]) and np.array_equal(outs[0][2], outs[2][2])
    # 'linear' is untouched by the switch: the call that follows a 'poly' call gives the L6 rows
    out_lin = ground_water_augmentation(pc, **dict(kw, estimation_method="linear", return_src=False))
    assert out_lin.shape == d6[f"c{case}_out"].shape
    assert eng.ctx.wet_last_fit(1)[0][0] == 0.0 and eng.ctx.wet_last_fit(1)[0][7] == -1


def test_poly_argument_handling(eng, golden):
    from lidar_snow_sim_amd.tools.wet_ground.augmentation import ground_water_augmentation, estimate_laser_parameters
    pc = golden("L6_wet_ground")["c0_pc"]
    with pytest.raises(ValueError):
        ground_water_augmentation(pc, estimation_method="cubic", plane=PLANE, debug=False)
    with pytest.raises(ValueError):
        ground_water_augmentation(pc, estimation_method="poly", q8="numpy", plane=PLANE, debug=False)
    with pytest.raises(NotImplementedError):
        estimate_laser_parameters(pc[:100], np.full(100, 0.3), debug=False, estimation_method="poly")
    out = ground_water_augmentation(pc[:900], estimation_method="poly", plane=PLANE, debug=False)
    assert out is pc[:900] or out.shape == pc[:900].shape                # fewer than 1000 ground rows: the input comes back (:51-52)
    # A cloud whose ground rows all lie in one or two range rows of the histogram: np.polyfit of degree 2 over fewer than 3 points answers
    # with the minimum-norm solution and a RankWarning, and ransac_polyfit returns that fit (no trial can gather d = 15 inliers,
    # augmentation.py:179-191) -- so does the device; only an EMPTY x raises (np.polyfit's TypeError).
    ring = pc[np.abs(np.linalg.norm(pc[:, :3], axis=1) - 20.0) < 0.5]
    if ring.shape[0] >= 1000:
        import warnings
        out = ground_water_augmentation(ring, estimation_method="poly", plane=PLANE, debug=False)
        fit = eng.ctx.wet_last_fit(1)[0]
        w = np.asarray(PLANE[0], np.float64)
        hog = ring[:, :3].astype(np.float64) @ w
        g = ring[np.abs(hog + PLANE[1]) < 0.5].astype(np.float64)
        dist = np.linalg.norm(g[:, :3], axis=1)
        norm = g[:, 3] / ((g[:, :3] @ w) / (dist * np.linalg.norm(w)))
        hist, xe, ye = np.histogram2d(dist, norm, bins=(50, 2555), range=((10, 70), (5, np.abs(np.max(norm)))))
        hist[hist == 0] = len(g)
        mv = ye[np.argmin(hist, axis=1)]
        sel = np.where(mv > 5)[0]
        assert 1 <= len(sel) <= 2 and out.shape[1] == 5
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = np.polyfit((xe[sel] + xe[sel + 1]) / 2, mv[sel], 2)
        np.testing.assert_allclose(fit[3:6], want, rtol=1e-8, atol=1e-12)


def test_fused_snow_and_wet_batch_with_poly_equals_the_chained_calls(eng, golden, tables):
    """pointcloud_viewer.py:2807-2821 with estimation_method='poly' (:2820): augment() then ground_water_augmentation(..., 'poly').  The
    fused entry (snowgpu_augment_wet_batch, estimation set on the context) must give the rows of the two chained calls -- frame f of the
    fused batch draws the RANSAC samples of (seed; f), so the chained call for frame f is checked through the curves it fitted."""
    from lidar_snow_sim_amd.tools.snowfall.simulation import augment
    from lidar_snow_sim_amd.tools.wet_ground.augmentation import ground_water_augmentation
    d = golden("L6_wet_ground")
    frames = [d["c0_pc"], d["c2_pc"], d["c1_pc"]]
    tl = [tables["t"][i % 4] for i in range(64)]
    bd = float(np.degrees(3e-3))
    order = list(range(64))
    off = np.concatenate(([0], np.cumsum([f.shape[0] for f in frames])))
    tids = [eng.table_ids_from_arrays(tl, order)] * len(frames)
    pl = [[0.0, 0.0, -1.0, -1.7]] * len(frames)
    eng.ctx.set_wet_estimation("poly", 5)
    try:
        out, src, counts, stats, flags = eng.ctx.augment_wet_batch(
            np.concatenate(frames), off, tids, bd, wet_plane=pl, plane=pl, water_height=0.0008, pavement_depth=0.001,
            wet_noise_floor=0.7, power_factor=15, flat_earth=False, delta=0.5, replace=False)
        fits = eng.ctx.wet_last_fit(len(frames))
    finally:
        eng.ctx.set_wet_estimation("linear")
    assert (fits[:, 0] != 0).all()                                     # quadratics, not lines
    for i, f in enumerate(frames):
        st, aug = augment(f, "unused", bd, only_camera_fov=False, plane=PLANE, order=order, particles=tl)
        ref = ground_water_augmentation(aug, water_height=0.0008, pavement_depth=0.001, noise_floor=0.7, power_factor=15, estimation_method="poly",
                                        flat_earth=False, debug=False, delta=0.5, replace=False, plane=PLANE, poly_seed=5)
        fit1 = eng.ctx.wet_last_fit(1)[0]
        np.testing.assert_allclose(fits[i][0:3], fit1[0:3], rtol=1e-12, atol=0)       # np.polyfit of the same rows
        if fits[i][7] == -1 and fit1[7] == -1:                         # no RANSAC trial won in either (the draws of frame i and of frame 0 differ)
            n = int(counts[i])
            got = out[off[i]:off[i] + n]
            assert got.shape == ref.shape and np.array_equal(got[:, [0, 1, 2, 4]], ref[:, [0, 1, 2, 4]])
            np.testing.assert_allclose(got[:, 3], ref[:, 3], rtol=1e-12, atol=0)
