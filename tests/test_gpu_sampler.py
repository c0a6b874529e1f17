"""-m gpu: the on-device snowflake sampler against the reference's sampling process (statistical parity)."""
import numpy as np
import pytest

import torch  # noqa: F401  -- before libsnowgpu.so is loaded (one HIP runtime per process)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def smp():
    from lidar_snow_sim_amd.tools.snowfall import sampling
    return sampling


def _params(smp, rs, vt):
    return smp.compute_occupancy(rs, vt), smp.snowfall_rate_to_rainfall_rate(rs, vt)


def test_device_table_is_a_valid_dart_throw(smp):
    """Every invariant of sampling.py:142-183 on a full-size table (2.5 mm/h @ 1.6 m/s, R0 = 80 m)."""
    from scipy.spatial import cKDTree
    occ, rate = _params(smp, 2.5, 1.6)
    t = smp.dart_throwing_device(occ, rate, 80.0, seed=7, distribution="gunn")
    x, y, r = t[:, 0], t[:, 1], t[:, 2]
    assert t.shape[1] == 3 and np.isfinite(t).all()
    assert (x * x + y * y <= 80.0 ** 2).all()                    # inside the domain (:145)
    assert (x * x + y * y > r * r).all()                         # no disk over the origin (:166)
    assert (r > 0).all() and (r <= 0.010).all()                  # diameters are cut at 20 mm (:153)
    pairs = cKDTree(t[:, :2]).query_pairs(0.0201, output_type="ndarray")
    if len(pairs):                                               # no overlaps (:170-174)
        dist = np.hypot(x[pairs[:, 0]] - x[pairs[:, 1]], y[pairs[:, 0]] - y[pairs[:, 1]])
        assert (dist > r[pairs[:, 0]] + r[pairs[:, 1]]).all()
    target = occ * np.pi * 80.0 ** 2
    area = np.cumsum(np.pi * r * r)                              # stop rule (:142): the last dart is the one that fills it
    assert area[-1] >= target and area[-2] < target


def test_device_table_matches_the_reference_distribution(smp):
    """Counts, radii and radial density against tables thrown with the reference's own sequential sampler."""
    from scipy.stats import ks_2samp, kstest
    occ, rate = _params(smp, 2.5, 1.6)
    ref = [smp.dart_throwing(occ, rate, 80.0, np.random.default_rng(100 + i), "gunn") for i in range(3)]
    dev = [smp.dart_throwing_device(occ, rate, 80.0, seed=100 + i, distribution="gunn") for i in range(3)]
    k_ref, k_dev = np.mean([len(t) for t in ref]), np.mean([len(t) for t in dev])
    assert abs(k_dev - k_ref) / k_ref < 0.03                     # ~18 000 flakes either way
    r_ref, r_dev = np.concatenate([t[:, 2] for t in ref]), np.concatenate([t[:, 2] for t in dev])
    assert abs(r_dev.mean() - r_ref.mean()) / r_ref.mean() < 0.02
    assert ks_2samp(r_ref, r_dev).pvalue > 1e-3                  # same radius law
    rho2 = np.concatenate([t[:, 0] ** 2 + t[:, 1] ** 2 for t in dev]) / 80.0 ** 2
    assert kstest(rho2, "uniform").pvalue > 1e-3                 # uniform in area
    phi = np.concatenate([np.arctan2(t[:, 1], t[:, 0]) for t in dev])
    assert kstest((phi + np.pi) / (2 * np.pi), "uniform").pvalue > 1e-3


def test_device_sampler_is_deterministic_per_seed_and_mode(smp):
    occ, rate = _params(smp, 1.0, 1.6)
    a = smp.dart_throwing_device(occ, rate, 30.0, seed=5, distribution="gunn")
    b = smp.dart_throwing_device(occ, rate, 30.0, seed=5, distribution="gunn")
    c = smp.dart_throwing_device(occ, rate, 30.0, seed=6, distribution="gunn")
    d = smp.dart_throwing_device(occ, rate, 30.0, seed=5, distribution="sekhon")
    assert np.array_equal(a, b) and not np.array_equal(a[:50], c[:50]) and len(d) != len(a)
    with pytest.raises(NotImplementedError):
        smp.dart_throwing_device(occ, rate, 30.0, seed=5, distribution="sekhon_srivastava")


def test_sampled_tables_feed_the_simulation(smp):
    """Tables made on the device and FILED on the device (no host round trip) feed augment -- and the CPU oracle, given the
    same rows, agrees row for row."""
    from lidar_snow_sim_amd import engine
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    from oracle import snow_oracle as so
    occ, rate = _params(smp, 2.5, 1.6)
    eng = engine.get_engine(0)
    made = [smp.dart_throwing_device(occ, rate, 40.0, seed=900 + i, file_table=True) for i in range(2)]
    tabs, ids = [m[0] for m in made], [m[1] for m in made]
    assert ids[0] != ids[1] and all(t.shape[0] > 1000 for t in tabs)
    tl = [tabs[i % 2] for i in range(64)]
    full = synthetic_sweep(64, 2048, seed=31, intensity="lambert").reshape(64, 2048, 5)
    pc = np.ascontiguousarray(full[:, ::32, :].reshape(-1, 5))
    order = list(range(64))
    bd = float(np.degrees(3e-3))
    tids = [ids[order[c] % 2] for c in range(64)]
    out, src, counts, stats, _ = eng.ctx.augment_batch(pc, [0, pc.shape[0]], [tids], bd, plane=[[0, 0, -1.0, -1.7]])
    s0, a0, src0 = so.augment(pc, tl, bd, order, plane=([0.0, 0.0, -1.0], -1.7))
    n = int(counts[0])
    assert tuple(int(v) for v in stats[0]) == tuple(int(v) for v in s0)
    assert np.array_equal(src[:n], src0) and np.array_equal(out[:n, 3:], a0[:, 3:])
    # rows that never left the device: only the count comes back
    none_rows, tid3 = smp.dart_throwing_device(occ, rate, 40.0, seed=900, file_table=True, want_rows=False)
    assert none_rows is None and tid3 not in ids
    q3, q0 = eng.ctx.debug_table(tid3, tabs[0].shape[0]), eng.ctx.debug_table(ids[0], tabs[0].shape[0])
    assert np.array_equal(q3, q0)                                   # same seed -> same table, filed the same way


def test_filed_per_flake_quantities_match_the_reference_geometry(golden):
    """The per-flake quantities hoisted out of get_occlusions' beam loop -- range, azimuth, tangent angles (geometry.py:138-190,
    :32-80) -- read back through the debug tap against the reference's own L1 outputs: bit for bit when the table is filed on
    the host (snowgpu_upload_table, glibc libm); when it is filed by the device kernels, whose atan2 / atan are rounded correctly
    (csrc/sg_atan_cr.h) where glibc's are off by up to 0.506 ULP, all but a few values in a thousand are identical and the rest
    differ by one ULP."""
    import torch
    from lidar_snow_sim_amd import engine
    d = golden("L1_geometry")
    disks = np.ascontiguousarray(d["disks"])
    k = disks.shape[0]
    eng = engine.get_engine(0)
    t_host, t_dev = eng.user_table_id(), eng.user_table_id()
    eng.ctx.upload_table(t_host, disks)
    dd = torch.from_numpy(disks).to("cuda:0")
    eng.ctx.file_table_device(t_dev, dd.data_ptr(), k)
    qh, qd = eng.ctx.debug_table(t_host, k), eng.ctx.debug_table(t_dev, k)
    want = np.column_stack((d["rho"], d["phi"], d["tangent_angles"]))
    assert np.array_equal(qh, want)                                  # host filing: the reference's numbers exactly
    np.testing.assert_allclose(qd, want, rtol=2.3e-16, atol=0)       # device filing: within 1 ulp ...
    assert (qd != want).mean() < 0.005                               # ... and that only where glibc is not correctly rounded
    # both filings put every flake into the same bins in the same order: a sweep over either gives the same rows
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    full = synthetic_sweep(64, 2048, seed=77, intensity="lambert").reshape(64, 2048, 5)
    pc = np.ascontiguousarray(full[:, ::64, :].reshape(-1, 5))
    res = [eng.ctx.augment_batch(pc, [0, pc.shape[0]], [[t] * 64], float(np.degrees(3e-2)), thr_poly=[[0.0, 0.0, -1.0]]) for t in (t_host, t_dev)]
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][0][:, 3:], res[1][0][:, 3:])
    assert (res[0][0][:, 4] > 0).sum() > 5
    with pytest.raises(Exception):                                   # a disk over the origin is refused by the device path too
        bad = torch.tensor([[0.001, 0.001, 0.01]], dtype=torch.float64, device="cuda:0")
        eng.ctx.file_table_device(eng.user_table_id(), bad.data_ptr(), 1)


def test_stream_driver_with_an_empty_particle_directory_samples_on_the_device(smp, tmp_path):
    """no <prefix>_<line>.npy anywhere -- the stream driver (precompute.py:74-106) samples the 64 tables of every
    prefix on the device (seed = f(prefix, line)), caches the ids, and every output equals the CPU oracle run on the read-back
    tables; augment(particles='device') reaches the same tables by name."""
    import random
    from lidar_snow_sim_amd import engine, stream
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    from lidar_snow_sim_amd.tools.snowfall.simulation import augment
    from oracle import snow_oracle as so
    eng = engine.get_engine(0)
    eng.keep_sampled_rows = True
    try:
        lidar = tmp_path / "lidar_hdl64_strongest"
        lidar.mkdir()
        (tmp_path / "training" / "snowflakes" / "npy").mkdir(parents=True)       # the reference's directory, empty
        ids = ["2018-02-03_00001", "2018-02-03_00002"]
        full = synthetic_sweep(64, 2048, seed=17, intensity="lambert").reshape(64, 2048, 5)
        frames = {}
        for i, s in enumerate(ids):
            frames[s] = np.ascontiguousarray(full[:, i::64, :].reshape(-1, 5))
            frames[s].tofile(lidar / f"{s}.bin")
        combos = stream.rate_combos()[3:4]                                        # 2.5 mm/h @ 1.6 m/s
        prefix = f"gunn_{combos[0][0]}_{combos[0][1]}"
        with pytest.raises(FileNotFoundError):                                    # the reference's behaviour without the switch (simulation.py:329)
            stream.run(lidar, ids, particle_root=str(tmp_path), modes=("gunn",), combos=combos, batch=2)
        random.seed(8)
        # one GPU worker: the batch then runs on the engine this test reads the sampled rows back from (slot 0)
        n = stream.run(lidar, ids, particle_root=str(tmp_path), modes=("gunn",), combos=combos, batch=2, sample_missing=True, workers=1)
        assert n == 2
        tabs = [eng.sampled_rows[(prefix, line)] for line in range(1, 65)]        # the tables as the device made them
        assert all(t.shape[0] > 10000 for t in tabs) and eng.sampled_flakes[(prefix, 1)] == tabs[0].shape[0]
        assert len({t.shape[0] for t in tabs}) > 8                                # 64 different tables, not one table 64 times
        random.seed(8)
        bd = float(np.degrees(3e-3))
        for s in ids:
            order = list(range(64))
            random.shuffle(order)
            got = np.fromfile(stream.output_path(lidar, "gunn", combos[0][0], s), dtype=np.float32).reshape(-1, 5)
            _, exp, _ = so.augment(frames[s], tabs, bd, order, plane=None)
            assert got.shape == exp.shape and np.array_equal(got[:, 3:], exp[:, 3:])
        n_tables = eng.ctx._L.snowgpu_table_count(eng.ctx.handle)
        # the same names through augment(): cached ids, no new table
        random.seed(8)
        order = list(range(64))
        random.shuffle(order)
        stats, aug = augment(frames[ids[0]], prefix, bd, only_camera_fov=False, particles="device", order=order)
        s0, a0, _ = so.augment(frames[ids[0]], tabs, bd, order, plane=None)
        assert tuple(int(v) for v in stats) == tuple(int(v) for v in s0) and np.array_equal(aug[:, 3:], a0[:, 3:])
        assert eng.ctx._L.snowgpu_table_count(eng.ctx.handle) == n_tables
        with pytest.raises(ValueError):
            augment(frames[ids[0]], "mytables", bd, only_camera_fov=False, particles="device")     # a prefix that names no (mode, rate, occupancy)
    finally:
        eng.keep_sampled_rows = False


def test_offline_generator_in_device_mode_writes_the_tables_augment_samples_on_the_fly(smp, tmp_path):
    """python -m lidar_snow_sim_amd.sample_tables --device (tools/snowfall/sampling.py:360-413 with the device sampler): the files of a
    (mode, pair) hold exactly the rows augment(particles='device') samples for the same prefix and line (seed = f(prefix, line)), carry
    the reference's names, and a second run skips them."""
    from lidar_snow_sim_amd import engine
    from lidar_snow_sim_amd import sample_tables as st
    occ, rate = _params(smp, 2.5, 1.6)
    rep = st.generate(tmp_path, ["gunn"], [[rate, occ]], range(1, 4), device=0, verbose=False)
    assert rep["written"] == 3 and rep["flakes"] > 3 * 15000
    prefix = f"gunn_{rate}_{occ}"
    assert sorted(p.name for p in tmp_path.iterdir()) == [f"{prefix}_{k}.npy" for k in (1, 2, 3)]
    e = engine.Engine(0)
    try:
        e.keep_sampled_rows = True
        for line in (1, 2, 3):
            e.sampled_table_id(prefix, line)
            assert np.load(tmp_path / f"{prefix}_{line}.npy").tobytes() == e.sampled_rows[(prefix, line)].tobytes()
    finally:
        e.ctx.close()
    again = st.generate(tmp_path, ["gunn"], [[rate, occ]], range(1, 5), device=0, verbose=False)
    assert again["written"] == 1 and again["skipped"] == 3
