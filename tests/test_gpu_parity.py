"""-m gpu: the HIP path (through the C ABI) against the CPU oracle and the reference's golden vectors."""
import numpy as np
import pytest
import torch  # noqa: F401  -- before libsnowgpu.so is loaded: PyTorch bundles its own HIP runtime, and the process must end up with one

from conftest import canonical

pytestmark = pytest.mark.gpu

PLANE = (np.array([0.0, 0.0, -1.0]), -1.7)


@pytest.fixture(scope="module")
def eng():
    from lidar_snow_sim_amd import engine
    return engine.get_engine(0)


@pytest.fixture(scope="module")
def so():
    from oracle import snow_oracle
    return snow_oracle


def _tables64(tables):
    return [tables["t"][i % 4] for i in range(64)]


def test_library_is_the_hip_build(eng):
    from lidar_snow_sim_amd import _native
    assert b"gfx950" in _native.lib().snowgpu_version()
    assert eng.ctx.handle


@pytest.mark.parametrize("tag", ["float32", "float64"])
def test_occlusion_dicts_match_oracle(eng, so, golden, tables, tag):
    """get_occlusions / compute_occlusion_dict on the L4 inputs: scatterer lists bit for bit."""
    d = golden("L4_process_single_channel")
    pc = d[f"pc_{tag}"]
    tl = _tables64(tables)
    tids = eng.table_ids_from_arrays(tl, list(range(64)))
    cnt, rj, ratio, src = eng.ctx.debug_occlusions(pc, tids, float(d["bd"]))
    assert np.array_equal(pc[src, 4], np.sort(pc[:, 4], kind="stable"))
    las = so.load_lasers()
    bad = 0
    for ch in range(64):
        rows = np.where(pc[:, 4] == ch)[0]
        _, _, (c0, k0, r0, q0) = so.process_single_channel(pc[rows], tl[ch], float(d["bd"]), las, ch, dump=True)
        pos = np.where(pc[src, 4] == ch)[0]
        assert np.array_equal(src[pos], rows)
        start = np.concatenate(([0], np.cumsum(c0)))
        for i, p in enumerate(pos):
            n = int(c0[i])
            ok = cnt[p] == n and np.array_equal(rj[p, :n], r0[start[i]:start[i] + n])
            if tag == "float32":
                ok = ok and np.array_equal(ratio[p, :n], q0[start[i]:start[i] + n])
            else:
                # float64 rows: the beam azimuth is a float64 atan2 -- OCML's and glibc's differ in the last
                # bit for some inputs, which moves ratios by ~1e-16 of a beam (no float32 rounding to hide in)
                ok = ok and np.allclose(ratio[p, :n], q0[start[i]:start[i] + n], rtol=0, atol=1e-12)
            bad += not ok
    assert bad == 0


@pytest.mark.parametrize("case", range(8))
def test_L5_augment_matches_reference(eng, golden, tables, case):
    from lidar_snow_sim_amd.tools.snowfall.simulation import augment
    d = golden("L5_augment")
    pc = d[f"c{case}_pc"]
    plane = (d[f"c{case}_plane_w"], float(d[f"c{case}_plane_h"]))
    stats, aug, src = augment(pc, "unused", float(d["bd"]), only_camera_fov=False, plane=plane,
                              order=list(d[f"c{case}_order"]), particles=_tables64(tables), return_src=True,
                              device_prepass=False)
    assert tuple(int(s) for s in stats) == tuple(int(v) for v in d[f"c{case}_stats"])
    a1, s1 = canonical(aug, src)
    a2, s2 = canonical(d[f"c{case}_aug"], d[f"c{case}_src"])
    assert np.array_equal(s1, s2)                              # bit-exact kept-point indices
    assert a1.dtype == a2.dtype
    assert np.array_equal(a1[:, 4], a2[:, 4])                  # bit-exact labels
    assert np.array_equal(a1[:, 3], a2[:, 3])                  # intensities are integers: exact
    np.testing.assert_allclose(a1[:, :3], a2[:, :3], rtol=1e-6 if pc.dtype == np.float32 else 1e-12, atol=0)
    # output order: channel-sorted, stable
    assert np.array_equal(src, s1[np.argsort(pc[s1, 4], kind="stable")])


def test_Q5_unsimulated_channels(eng, golden, tables):
    from lidar_snow_sim_amd.tools.snowfall.simulation import augment
    d = golden("L5_augment")
    stats, aug, src = augment(d["q5_pc"], "unused", float(d["bd"]), shuffle=False, only_camera_fov=False, plane=PLANE,
                              particles=_tables64(tables), return_src=True, device_prepass=False)
    assert tuple(int(s) for s in stats) == tuple(int(v) for v in d["q5_stats"])
    a1, _ = canonical(aug, src)
    a2, _ = canonical(d["q5_aug"], d["q5_src"])
    assert np.array_equal(a1, a2)


@pytest.mark.parametrize("tag", ["far", "far64", "wide", "wide64", "near", "near64", "dense"])
def test_L4_special_channels(eng, so, golden, tables, tag):
    """Grid end, xsi ramp under NEP 50, lists beyond the 16/32-entry fast paths."""
    d = golden("L4_process_single_channel")
    if tag == "dense":
        pc, ch, bd, tab, ref = d["dense_pc"], 5, float(d["bd"]), tables["dense"], d["dense_out"]
        pc = pc[pc[:, 4] == ch]
    else:
        pc, ch, bd, ref = d[f"{tag}_pc"], int(d[f"{tag}_ch"]), float(d[f"{tag}_bd"]), d[f"{tag}_out"]
        tab = {"dense": tables["dense"], "nearflakes": d["nearflakes_xyr"]}.get(str(d[f"{tag}_table"]), tables["t"][0])
    tids = [eng.array_table_id(tab)] * 64
    off = np.array([0, pc.shape[0]])
    out, src, counts, stats, _ = eng.ctx.augment_batch(pc, off, [tids], bd, thr_poly=[[0.0, 0.0, -1.0]])
    assert counts[0] == pc.shape[0]                            # threshold -1: nothing is filtered
    got = np.empty_like(out)
    got[src] = out
    exp = ref.copy()
    exp[:, 3] = np.round(exp[:, 3])
    assert np.array_equal(got[:, 4], exp[:, 4])
    assert np.array_equal(got[:, 3], exp[:, 3])
    np.testing.assert_allclose(got[:, :3], exp[:, :3], rtol=1e-6 if pc.dtype == np.float32 else 1e-12, atol=0)
    diff = float(d[f"{tag}_diff"])
    n_att = int((exp[:, 4] == 1).sum())
    assert int(stats[0, 0]) == n_att
    assert int(stats[0, 2]) == (int(diff / n_att) if n_att else 0)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_synthetic_frame_against_oracle(eng, so, tables, dtype):
    """A 64 x 128 sub-sweep at production beam divergence: every row against the CPU oracle."""
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    from lidar_snow_sim_amd.tools.snowfall.simulation import augment
    full = synthetic_sweep(64, 2048, seed=1003, intensity="lambert", dtype=np.float64).reshape(64, 2048, 5)
    pc = full[:, ::16, :].reshape(-1, 5).astype(dtype)
    rng = np.random.default_rng(5)
    pc = pc[rng.permutation(pc.shape[0])]                      # unsorted input: exercises the device sort
    order = list(rng.permutation(64))
    tl = _tables64(tables)
    bd = float(np.degrees(3e-3))
    s0, a0, src0 = so.augment(pc, tl, bd, order, plane=PLANE)
    s1, a1, src1 = augment(pc, "unused", bd, only_camera_fov=False, plane=PLANE, order=order, particles=tl,
                           return_src=True, device_prepass=False)
    assert tuple(int(v) for v in s1) == tuple(int(v) for v in s0)
    assert np.array_equal(src1, src0)                          # same rows kept, same (stable) order
    assert np.array_equal(a1[:, 3:], a0[:, 3:])
    np.testing.assert_allclose(a1[:, :3], a0[:, :3], rtol=1e-6 if dtype == np.float32 else 1e-12, atol=0)


def test_empty_and_ragged_batch(eng, so, tables):
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    from lidar_snow_sim_amd.tools.snowfall.simulation import augment_batch
    full = synthetic_sweep(64, 2048, seed=1004, intensity="lambert").reshape(64, 2048, 5)
    f0 = full[:, ::64, :].reshape(-1, 5)
    f1 = np.zeros((0, 5), np.float32)
    f2 = full[:7, 5::100, :].reshape(-1, 5)
    tl = _tables64(tables)
    bd = float(np.degrees(3e-3))
    polys = [[0.0, 0.01, 2.0]] * 3
    orders = [list(range(64))] * 3
    res = augment_batch([f0, f1, f2], "unused", bd, particles=tl, orders=orders, thr_polys=polys, return_src=True)
    for f, (st, aug, src) in zip((f0, f1, f2), res):
        if f.shape[0] == 0:
            assert aug.shape == (0, 5) and tuple(int(v) for v in st) == (0, 0, 0)
            continue
        s0, a0, src0 = so.augment(f, tl, bd, orders[0], thr_poly=np.array(polys[0]))
        assert tuple(int(v) for v in st) == tuple(int(v) for v in s0)
        assert np.array_equal(src, src0) and np.array_equal(aug[:, 3:], a0[:, 3:])


def test_range_beyond_grid_raises_index_error(eng, tables):
    from lidar_snow_sim_amd.tools.snowfall.simulation import augment
    pc = np.array([[125.0, 1.0, 0.0, 30.0, 3.0], [10.0, 1.0, -1.0, 30.0, 3.0]], np.float32)
    with pytest.raises(IndexError):
        augment(pc, "unused", float(np.degrees(3e-2)), only_camera_fov=False, particles=_tables64(tables),
                thr_poly=[0.0, 0.0, 0.0], shuffle=False)


# ---- device prepass (simulation.py:449-467 on the GPU) ---------------------------------------------------------
@pytest.mark.parametrize("case", range(8))
def test_L5_augment_device_prepass(eng, golden, tables, case):
    """Same L5 fixtures with the noise-threshold prepass on the device: the polynomial agrees with the
    NumPy/SciPy one to 1e-7 relative (different summation order and least-squares solver) and the kept
    rows / labels / intensities are the reference's."""
    from lidar_snow_sim_amd.tools.snowfall.simulation import augment
    from lidar_snow_sim_amd.tools.wet_ground.augmentation import noise_threshold_poly
    d = golden("L5_augment")
    pc = d[f"c{case}_pc"]
    plane = (d[f"c{case}_plane_w"], float(d[f"c{case}_plane_h"]))
    order = list(d[f"c{case}_order"])
    tl = _tables64(tables)
    tids = eng.table_ids_from_arrays(tl, order)
    _, _, _, _, thr = eng.ctx.augment_batch(pc, [0, pc.shape[0]], [tids], float(d["bd"]),
                                            plane=[[*plane[0], plane[1]]], want_thr=True)
    srt = pc[np.argsort(pc[:, 4], kind="stable")]
    host = noise_threshold_poly(srt, plane[0], plane[1], 0.7)
    dist = np.linspace(3, 80, 50)
    # float64 rows: 1e-12.  float32 rows: np.polyfit builds its Vandermonde matrix in float32 and divides it by
    # float32 column norms before the float64 solve, which perturbs ITS answer by ~1e-6 in threshold units (and
    # differently for every row order); the device solves the same normal equations in float64.
    tol = dict(rtol=1e-5, atol=1e-4) if pc.dtype == np.float32 else dict(rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(np.polyval(thr[0], dist), np.polyval(host, dist), **tol)
    stats, aug, src = augment(pc, "unused", float(d["bd"]), only_camera_fov=False, plane=plane, order=order,
                              particles=tl, return_src=True)
    assert tuple(int(s) for s in stats) == tuple(int(v) for v in d[f"c{case}_stats"])
    a1, s1 = canonical(aug, src)
    a2, s2 = canonical(d[f"c{case}_aug"], d[f"c{case}_src"])
    assert np.array_equal(s1, s2) and np.array_equal(a1[:, 3:], a2[:, 3:])


def test_too_few_ground_points_raise_type_error(eng, tables):
    from lidar_snow_sim_amd.tools.snowfall.simulation import augment
    pc = np.array([[10.0, 1.0, 5.0, 30.0, 3.0], [12.0, 1.0, 6.0, 30.0, 3.0]], np.float32)   # nothing near the plane
    with pytest.raises(TypeError):
        augment(pc, "unused", float(np.degrees(3e-3)), only_camera_fov=False, particles=_tables64(tables),
                plane=PLANE, shuffle=False)


# ---- wet ground (wet_ground/augmentation.py:25-161 on the GPU) --------------------------------------------------
@pytest.mark.parametrize("case", range(8))
def test_L6_wet_ground(eng, golden, case):
    from lidar_snow_sim_amd.tools.wet_ground.augmentation import ground_water_augmentation
    d = golden("L6_wet_ground")
    pc = d[f"c{case}_pc"]
    out, src = ground_water_augmentation(pc, water_height=0.0008, pavement_depth=0.001, noise_floor=0.7, power_factor=15,
                                         estimation_method="linear", flat_earth=bool(d[f"c{case}_flat"]), debug=False,
                                         delta=0.5, replace=bool(d[f"c{case}_replace"]), plane=PLANE, return_src=True)
    ref = d[f"c{case}_out"]
    assert out.dtype == np.float64 and out.shape == ref.shape            # same rows kept, float64 output (Q10)
    assert np.array_equal(out[:, [0, 1, 2, 4]], ref[:, [0, 1, 2, 4]])   # [non-ground ; kept ground] order, labels
    if pc.dtype == np.float64:
        np.testing.assert_allclose(out[:, 3], ref[:, 3], rtol=1e-9, atol=0)
    else:
        # float32 rows: scipy's linregress takes np.mean of the float32 range column (a float32 pairwise sum) for
        # the intercept; the device reproduces that sum operation for operation (k_pre_mean32), which matters because
        # the laser-power line nearly cancels at short range on these frames.
        np.testing.assert_allclose(out[:, 3], ref[:, 3], rtol=1e-7, atol=0)
    assert np.array_equal(pc[src, :3].astype(np.float64), out[:, :3])


def test_wet_ground_returns_input_below_1000_ground_rows(eng, golden):
    from lidar_snow_sim_amd.tools.wet_ground.augmentation import ground_water_augmentation
    d = golden("L6_wet_ground")
    pc = d["c0_pc"][:900]
    out = ground_water_augmentation(pc, plane=PLANE, debug=False)
    assert out is pc                                                     # augmentation.py:51-52


def test_chained_snow_then_wet_like_the_viewer(eng, so, golden, tables):
    """pointcloud_viewer.py:2807-2821: augment(...) then ground_water_augmentation(..., replace=False)."""
    from lidar_snow_sim_amd.tools.snowfall.simulation import augment
    from lidar_snow_sim_amd.tools.wet_ground.augmentation import ground_water_augmentation
    d = golden("L6_wet_ground")
    pc = d["c0_pc"]
    tl = _tables64(tables)
    bd = float(np.degrees(3e-3))
    order = list(range(64))
    stats, aug = augment(pc, "unused", bd, only_camera_fov=False, plane=PLANE, order=order, particles=tl)
    out = ground_water_augmentation(aug, water_height=0.0008, pavement_depth=0.001, noise_floor=0.7, power_factor=15,
                                    flat_earth=False, debug=False, delta=0.5, replace=False, plane=PLANE)
    s0, a0, _ = so.augment(pc, tl, bd, order, plane=PLANE)
    o0 = so.ground_water_augmentation(a0, water_height=0.0008, pavement_depth=0.001, flat_earth=False, replace=False, plane=PLANE)
    assert out.shape == o0.shape and np.array_equal(out[:, 4], o0[:, 4])
    np.testing.assert_allclose(out[:, :3], o0[:, :3], rtol=1e-6, atol=0)
    np.testing.assert_allclose(out[:, 3], o0[:, 3], rtol=1e-6, atol=0)


def test_fast_sine_equals_libm_mode_on_a_full_sweep(eng, tables):
    """64 x 2048 sweep (BASELINE config C2 shape): the engine's own sin^2 against the device libm + true
    division (snowgpu_set_exact_math) -- same kept rows, labels and intensities."""
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    from lidar_snow_sim_amd.tools.snowfall.simulation import augment
    pc = synthetic_sweep(64, 2048, seed=1007, intensity="lambert")
    tl = _tables64(tables)
    bd = float(np.degrees(3e-3))
    kw = dict(only_camera_fov=False, plane=PLANE, order=list(range(64)), particles=tl, return_src=True)
    s_fast, a_fast, src_fast = augment(pc, "unused", bd, **kw)
    eng.ctx.set_exact_math(True)
    try:
        s_ex, a_ex, src_ex = augment(pc, "unused", bd, **kw)
    finally:
        eng.ctx.set_exact_math(False)
    assert tuple(int(v) for v in s_fast) == tuple(int(v) for v in s_ex)
    assert np.array_equal(src_fast, src_ex) and np.array_equal(a_fast, a_ex)
    assert (a_fast[:, 4] == 1).sum() > 1000 and (a_fast[:, 4] == 2).sum() > 100


def test_stream_driver_writes_reference_layout(eng, so, tables, tmp_path):
    """tools/snowfall/precompute.py:74-106 through lidar_snow_sim_amd.stream: .bin in, .bin out, skip-if-exists."""
    import random
    from lidar_snow_sim_amd import stream
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    lidar = tmp_path / "lidar_hdl64_strongest"
    lidar.mkdir()
    ids = ["2018-02-03_00001", "2018-02-03_00002", "2018-02-04_00001"]
    full = synthetic_sweep(64, 2048, seed=11, intensity="lambert").reshape(64, 2048, 5)
    frames = {}
    for i, s in enumerate(ids):
        frames[s] = np.ascontiguousarray(full[:, i::64, :].reshape(-1, 5))
        frames[s].tofile(lidar / f"{s}.bin")
    combos = stream.rate_combos()[3:4]
    prefix = f"gunn_{combos[0][0]}_{combos[0][1]}"
    tl = _tables64(tables)
    random.seed(5)
    n = stream.run(lidar, ids, modes=("gunn",), combos=combos, batch=2, particles_by_prefix={prefix: tl},
                   planes=None)
    assert n == 3
    random.seed(5)
    bd = float(np.degrees(3e-3))
    for s in ids:                                                # same global-`random` draws, frame by frame
        order = list(range(64))
        random.shuffle(order)
        got = np.fromfile(stream.output_path(lidar, "gunn", combos[0][0], s), dtype=np.float32).reshape(-1, 5)
        _, exp, _ = so.augment(frames[s], tl, bd, order, plane=None)
        assert got.shape == exp.shape and np.array_equal(got[:, 3:], exp[:, 3:])
    assert stream.run(lidar, ids, modes=("gunn",), combos=combos, batch=2, particles_by_prefix={prefix: tl}) == 0
    # two host threads / engine contexts: same files
    first = {s: np.fromfile(stream.output_path(lidar, "gunn", combos[0][0], s), dtype=np.float32) for s in ids}
    for s in ids:
        stream.output_path(lidar, "gunn", combos[0][0], s).unlink()
    random.seed(5)
    assert stream.run(lidar, ids, modes=("gunn",), combos=combos, batch=1, particles_by_prefix={prefix: tl}, workers=2) == 3
    for s in ids:
        assert np.array_equal(np.fromfile(stream.output_path(lidar, "gunn", combos[0][0], s), dtype=np.float32), first[s])


def test_sharded_stream_writes_the_same_bytes_as_the_single_rank_run(eng, tables, tmp_path):
    """stream.run with rank / world = 0/2 and 1/2 (one GPU, one after the other, each seeded like the single
    rank) writes exactly the files of the world = 1 run -- every (mode, frame, combo) gets the permutation the reference's
    sequential loop draws for it (precompute.py:70-92, simulation.py:482-486)."""
    import random
    import shutil
    from lidar_snow_sim_amd import stream
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    lidar = tmp_path / "lidar_hdl64_strongest"
    lidar.mkdir()
    ids = [f"2018-02-03_{i:05d}" for i in range(7)]
    full = synthetic_sweep(64, 2048, seed=13, intensity="lambert").reshape(64, 2048, 5)
    for i, s in enumerate(ids):
        np.ascontiguousarray(full[:, i::64, :].reshape(-1, 5)).tofile(lidar / f"{s}.bin")
    combos = stream.rate_combos()[2:4]
    tl = _tables64(tables)
    kw = dict(modes=("gunn", "sekhon"), combos=combos, batch=3, particles_by_prefix={f"{m}_{rr}_{occ}": tl for m in ("gunn", "sekhon") for rr, occ in combos})
    out_root = lidar.parent / "snowfall_simulation"

    def snapshot():
        files = {str(p.relative_to(out_root)): p.read_bytes() for p in sorted(out_root.rglob("*.bin"))}
        shutil.rmtree(out_root)
        return files

    random.seed(21)
    assert stream.run(lidar, ids, **kw) == len(ids) * 4
    whole = snapshot()
    existing = set()                                             # both ranks start against the same (empty) output tree
    random.seed(21)
    n0 = stream.run(lidar, ids, rank=0, world=2, existing=existing, **kw)
    random.seed(21)
    n1 = stream.run(lidar, ids, rank=1, world=2, existing=existing, **kw)
    assert n0 + n1 == len(ids) * 4 and n0 == 4 * 4 and n1 == 3 * 4
    sharded = snapshot()
    assert sharded.keys() == whole.keys() and all(sharded[k] == whole[k] for k in whole)


# ---- BASELINE.json configs as parity cases ------------------------------------------------------------------------
def test_config_C1_dense_table_half_mm_per_hour(so, tables):
    """C1: 0.5 mm/h @ 2.0 m/s (40 112 flakes per line at R0 = 80 m): the capacity tiers start at 8 entries."""
    from lidar_snow_sim_amd import engine
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    from lidar_snow_sim_amd.tools.snowfall import sampling as smp
    from lidar_snow_sim_amd.tools.snowfall.simulation import augment
    occ, rate = smp.compute_occupancy(0.5, 2.0), smp.snowfall_rate_to_rainfall_rate(0.5, 2.0)
    tabs = [smp.dart_throwing(occ, rate, 80.0, np.random.default_rng(42 + i), "gunn") for i in range(2)]
    assert tabs[0].shape[0] > 39000
    tl = [tabs[i % 2] for i in range(64)]
    full = synthetic_sweep(64, 2048, seed=1000, intensity="lambert").reshape(64, 2048, 5)
    pc = full[:, ::20, :].reshape(-1, 5).copy()                # 103 rows per channel: several 64-row blocks per segment
    pc[:, :3] *= 1.7                                            # push targets out to ~90 m: long lists
    order = list(range(64))
    bd = float(np.degrees(3e-3))
    poly = [0.0, 0.0, 3.0]
    eng2 = engine.Engine(0)                                     # own context: max table size drives the tier choice
    try:
        tids = eng2.table_ids_from_arrays(tl, order)
        out, src, counts, stats, _ = eng2.ctx.augment_batch(pc, [0, pc.shape[0]], [tids], bd, thr_poly=[poly])
    finally:
        eng2.ctx.close()
    s0, a0, src0 = so.augment(pc, tl, bd, order, thr_poly=np.array(poly))
    n = int(counts[0])
    assert tuple(int(v) for v in stats[0]) == tuple(int(v) for v in s0)
    assert np.array_equal(src[:n], src0) and np.array_equal(out[:n, 3:], a0[:, 3:])
    assert (a0[:, 4] == 2).sum() > 20


def test_config_C4_128_layers(so, tables):
    """C4: 128-layer sweep, 10 mm/h tables, a 128-entry laser table made by tiling the 64-entry one (SURVEY 8 d).
    The reference stops at 64 channels (simulation.py:474-483); the oracle is process_single_channel per channel."""
    from lidar_snow_sim_amd import engine
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    from lidar_snow_sim_amd.tools.snowfall import sampling as smp
    occ, rate = smp.compute_occupancy(10.0, 1.6), smp.snowfall_rate_to_rainfall_rate(10.0, 1.6)
    tabs = [smp.dart_throwing(occ, rate, 80.0, np.random.default_rng(42 + i), "gunn") for i in range(2)]
    lasers = engine.load_lasers() * 2
    full = synthetic_sweep(128, 4096, seed=3, intensity="lambert").reshape(128, 4096, 5)
    pc = np.ascontiguousarray(full[:, ::128, :].reshape(-1, 5))
    bd = float(np.degrees(3e-3))
    eng2 = engine.Engine(0, lasers=lasers)
    try:
        tids = [eng2.table_id(("c4", i % 2), lambda i=i: tabs[i % 2]) for i in range(128)]
        out, src, counts, stats, _ = eng2.ctx.augment_batch(pc, [0, pc.shape[0]], [tids], bd, thr_poly=[[0.0, 0.0, -1.0]])
    finally:
        eng2.ctx.close()
    assert counts[0] == pc.shape[0]
    got = np.empty_like(out)
    got[src] = out
    las = so.load_lasers() * 2
    diff_sum, n_att = 0.0, 0
    for ch in range(128):
        rows = np.where(pc[:, 4] == ch)[0]
        # channels 64.. reuse laser ch % 64, incl. its max-intensity class (53, 55, 56, 58 -> 230)
        d, exp = so.process_single_channel(pc[rows], tabs[ch % 2], bd, las, ch % 64)
        assert np.array_equal(got[rows, 4], exp[:, 4]) and np.array_equal(got[rows, 3], np.round(exp[:, 3])), ch
        np.testing.assert_allclose(got[rows, :3], exp[:, :3], rtol=1e-6, atol=0)
        diff_sum += d
        n_att += int((exp[:, 4] == 1).sum())
    assert int(stats[0, 0]) == n_att and int(stats[0, 2]) == (int(diff_sum / n_att) if n_att else 0)


def test_camera_fov_crop_on_device_parity_unpinned(eng, so, tables):
    """only_camera_fov=True with an explicit KITTI-style calibration: the crop runs inside the compaction kernels and is
    checked against the ORACLE's projection applied to the oracle's augment() output (simulation.py:532-540).  Parity with
    the reference itself is unpinned (un-vendored calibration_kitti, missing calib_hdl64.txt: SURVEY 8 c)."""
    from lidar_snow_sim_amd.calibration import Calibration
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    from lidar_snow_sim_amd.tools.snowfall.simulation import augment
    cal = Calibration(P2=np.array([[700.0, 0, 960, 0], [0, 700.0, 512, 0], [0, 0, 1, 0]]), R0=np.eye(3),
                      V2C=np.array([[0, -1.0, 0, 0], [0, 0, -1.0, 0], [1.0, 0, 0, 0]]))
    full = synthetic_sweep(64, 2048, seed=21, intensity="lambert").reshape(64, 2048, 5)
    bd = float(np.degrees(3e-3))
    tl = _tables64(tables)
    for dtype in (np.float32, np.float64):
        pc = np.ascontiguousarray(full[:, ::8, :].reshape(-1, 5)).astype(dtype)
        kw = dict(plane=PLANE, order=list(range(64)), particles=tl, return_src=True)
        s_fov, a_fov, src_fov = augment(pc, "unused", bd, only_camera_fov=True, calib=cal, **kw)
        s0, a0, src0 = so.augment(pc, tl, bd, list(range(64)), plane=PLANE)
        flag = so.fov_flag(a0[:, :3], cal.V2C, cal.R0, cal.P2, (1024, 1920))
        assert 0 < flag.sum() < len(flag)
        assert np.array_equal(src_fov, src0[flag]) and np.array_equal(a_fov[:, 3:], a0[flag][:, 3:])
        np.testing.assert_allclose(a_fov[:, :3], a0[flag][:, :3], rtol=1e-6 if dtype == np.float32 else 1e-12, atol=0)
        # num_removed counts the cropped rows, num_attenuated / avg are taken before the crop (simulation.py:525-538)
        assert int(s_fov[0]) == int(s0[0]) and int(s_fov[2]) == int(s0[2])
        assert int(s_fov[1]) == int(s0[1]) + int((~flag).sum())
        # the crop is a per-call state: the next call without it sees every row again
        s_all, a_all, _ = augment(pc, "unused", bd, only_camera_fov=False, **kw)
        assert a_all.shape[0] == a0.shape[0] and tuple(int(v) for v in s_all) == tuple(int(v) for v in s0)
    with pytest.raises(AssertionError):                          # missing calibration file (simulation.py:35)
        augment(pc, "unused", bd, only_camera_fov=True, **kw)


def test_config_C3_snow_and_wet_fused_batch(eng, so, golden, tables):
    """C3: a batch of sweeps through snowfall + wet ground in one call, the intermediate cloud staying on the device
    (pointcloud_viewer.py:2807-2821 chains the two on the host)."""
    d = golden("L6_wet_ground")
    frames = [d["c0_pc"], d["c1_pc"][:2500], d["c2_pc"]]
    tl = _tables64(tables)
    bd = float(np.degrees(3e-3))
    order = list(range(64))
    off = np.concatenate(([0], np.cumsum([f.shape[0] for f in frames])))
    tids = [eng.table_ids_from_arrays(tl, order)] * 3
    pl = [[0.0, 0.0, -1.0, -1.7]] * 3
    out, src, counts, stats, flags = eng.ctx.augment_wet_batch(
        np.concatenate(frames), off, tids, bd, wet_plane=pl, plane=pl, water_height=0.0008, pavement_depth=0.001,
        wet_noise_floor=0.7, power_factor=15, flat_earth=False, delta=0.5, replace=False)
    for i, f in enumerate(frames):
        s0, a0, src0 = so.augment(f, tl, bd, order, plane=PLANE)
        o0, wsrc0 = so.ground_water_augmentation(a0, water_height=0.0008, pavement_depth=0.001, flat_earth=False,
                                                 replace=False, plane=PLANE, return_src=True)
        n = int(counts[i])
        got = out[off[i]:off[i] + n]
        assert tuple(int(v) for v in stats[i]) == tuple(int(v) for v in s0)
        assert got.shape == o0.shape and np.array_equal(got[:, 4], o0[:, 4])
        assert np.array_equal(src[off[i]:off[i] + n], src0[wsrc0])
        np.testing.assert_allclose(got[:, :3], o0[:, :3], rtol=1e-6, atol=0)
        np.testing.assert_allclose(got[:, 3], o0[:, 3], rtol=1e-6, atol=0)


def test_noise_line_fallback_uses_numpy_float32_mean(eng, tables):
    """Few occupied histogram rows -> pmin = p (augmentation.py:250-251): the intercept then carries np.mean of the
    float32 range column, which the device reproduces with NumPy's pairwise float32 sum."""
    from lidar_snow_sim_amd.tools.wet_ground.augmentation import noise_threshold_poly
    rng = np.random.default_rng(9)
    n = 3000
    az = rng.uniform(-np.pi, np.pi, n)
    d = rng.uniform(3.0, 9.5, n)                                # every ground row nearer than 10 m: empty histogram
    pc = np.column_stack((d * np.cos(az), d * np.sin(az), np.full(n, -1.7), rng.integers(5, 120, n),
                          rng.integers(0, 64, n))).astype(np.float32)
    tids = eng.table_ids_from_arrays(_tables64(tables), list(range(64)))
    _, _, _, _, thr = eng.ctx.augment_batch(pc, [0, n], [tids], float(np.degrees(3e-3)), plane=[[0, 0, -1.0, -1.7]],
                                            want_thr=True)
    srt = pc[np.argsort(pc[:, 4], kind="stable")]
    host = noise_threshold_poly(srt, PLANE[0], PLANE[1], 0.7)
    dist = np.linspace(3, 10, 20)
    np.testing.assert_allclose(np.polyval(thr[0], dist), np.polyval(host, dist), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("seed", range(6))
def test_randomised_frames_tables_and_divergences(so, seed):
    """Random tables (incl. flakes close to the sensor), random beam divergences (1-30 mrad), random channel mixes,
    ranges from 0.8 m to 119 m, unsorted rows, float32 and float64: every row against the CPU oracle."""
    from lidar_snow_sim_amd import engine
    rng = np.random.default_rng(1000 + seed)
    n_tab = 3
    tabs = []
    for _ in range(n_tab):
        k = int(rng.integers(200, 6000))
        rho = np.sqrt(rng.uniform(0.3 ** 2, 70.0 ** 2, k))
        rho[: k // 20] = rng.uniform(0.3, 2.0, k // 20)             # a crowd of near flakes: wide angular intervals
        phi = rng.uniform(0, 2 * np.pi, k)
        r = np.minimum(rng.exponential(1.5e-3, k) + 1e-4, 0.01)
        tabs.append(np.column_stack((rho * np.cos(phi), rho * np.sin(phi), r)))
    tl = [tabs[i % n_tab] for i in range(64)]
    n = 1500
    az = rng.uniform(-np.pi, np.pi, n)
    az[:40] = rng.uniform(-4e-3, 4e-3, 40)                          # around the 0 / 2 pi seam (Q9)
    az[40:60] = np.pi / 2 + rng.uniform(-2e-3, 2e-3, 20)            # around a vertical beam limit
    el = rng.uniform(-0.4, 0.05, n)
    d = np.exp(rng.uniform(np.log(0.8), np.log(119.0), n))
    pc = np.column_stack((d * np.cos(el) * np.cos(az), d * np.cos(el) * np.sin(az), d * np.sin(el),
                          rng.integers(0, 256, n), rng.integers(0, 64, n)))
    dtype = np.float32 if seed % 2 == 0 else np.float64
    pc = pc.astype(dtype)
    bd = float(np.degrees(rng.choice([1e-3, 3e-3, 8e-3, 3e-2])))
    order = list(rng.permutation(64))
    poly = [0.002, -0.1, 12.0]
    eng2 = engine.Engine(0)
    try:
        tids = eng2.table_ids_from_arrays(tl, order)
        out, src, counts, stats, _ = eng2.ctx.augment_batch(pc, [0, n], [tids], bd, thr_poly=[poly])
    finally:
        eng2.ctx.close()
    s0, a0, src0 = so.augment(pc, tl, bd, order, thr_poly=np.array(poly))
    m = int(counts[0])
    assert tuple(int(v) for v in stats[0]) == tuple(int(v) for v in s0)
    assert np.array_equal(src[:m], src0)
    assert np.array_equal(out[:m, 3:], a0[:, 3:])
    np.testing.assert_allclose(out[:m, :3], a0[:, :3], rtol=1e-6 if dtype == np.float32 else 1e-12, atol=0)


def test_error_paths_report_instead_of_computing_garbage(eng, tables):
    from lidar_snow_sim_amd import _native
    with pytest.raises(_native.SnowGPUError) as e:                   # a disk over the origin: sqrt(< 0) in geometry.py:161
        eng.ctx.upload_table(900, np.array([[0.001, 0.001, 0.01]]))
    assert e.value.code == _native.E_TABLE
    pc = np.array([[10.0, 1.0, -1.0, 30.0, 3.0]] * 8, np.float32)
    with pytest.raises(_native.SnowGPUError) as e:                   # table id never uploaded
        eng.ctx.augment_batch(pc, [0, 8], [[12345] * 64], float(np.degrees(3e-3)), thr_poly=[[0, 0, 0.0]])
    assert e.value.code == _native.E_INVALID
    with pytest.raises(_native.SnowGPUError):                        # beam divergence outside (0, 45) degrees
        eng.ctx.augment_batch(pc, [0, 8], [eng.table_ids_from_arrays(_tables64(tables), list(range(64)))], 60.0,
                              thr_poly=[[0, 0, 0.0]])
    with pytest.raises(TypeError):                                   # integer rows
        eng.ctx.augment_batch(pc.astype(np.int32), [0, 8], [[0] * 64], 0.17, thr_poly=[[0, 0, 0.0]])


def test_table_ordered_launch_equals_sorted_row_order(tables, monkeypatch):
    """The first capacity tier walks (table, frame, channel) segments; SNOWGPU_LINEAR_ORDER=1 keeps sorted-row order.
    Same bytes either way -- ragged frames, an empty frame, channels without lasers (Q5) and rows of several
    channels that share one table included."""
    from lidar_snow_sim_amd import engine
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    rng = np.random.default_rng(77)
    full = synthetic_sweep(64, 2048, seed=1011, intensity="lambert").reshape(64, 2048, 5)
    f0 = full[:, ::16, :].reshape(-1, 5).copy()
    f1 = np.zeros((0, 5), np.float32)
    f2 = full[3:40, 7::50, :].reshape(-1, 5).copy()
    f2[::11, 4] = 70.0                                   # no such laser: copied through (Q5)
    f3 = full[:, 1::32, :].reshape(-1, 5).copy()
    rng.shuffle(f3, axis=0)
    frames = [f0, f1, f2, f3]
    rows = np.concatenate(frames)
    off = np.concatenate(([0], np.cumsum([f.shape[0] for f in frames])))
    tl = _tables64(tables)
    bd = float(np.degrees(3e-3))
    polys = [[0.0, 0.01, 2.0]] * 4
    results = []
    for linear in ("0", "1"):
        monkeypatch.setenv("SNOWGPU_LINEAR_ORDER", linear)
        e = engine.Engine(0)
        try:
            orders = [list(np.random.default_rng(5 + i).permutation(64)) for i in range(len(frames))]
            tids = [e.table_ids_from_arrays(tl, o) for o in orders]
            results.append(e.ctx.augment_batch(rows, off, tids, bd, thr_poly=polys))
        finally:
            e.ctx.close()
    (o0, s0, c0, st0, _), (o1, s1, c1, st1, _) = results
    assert np.array_equal(c0, c1) and np.array_equal(st0, st1)
    for f in range(len(frames)):
        a, m = int(off[f]), int(c0[f])
        assert np.array_equal(s0[a:a + m], s1[a:a + m])
        assert o0[a:a + m].tobytes() == o1[a:a + m].tobytes()
    assert int(c0[1]) == 0 and int(c0[0]) > 0 and int(st0[:, 0].sum()) > 0


def test_results_stay_valid_while_the_pool_recycles_buffers(eng, tables):
    """Results live in page-locked buffers that return to a pool when the caller drops them: a later call must not
    overwrite rows somebody still holds, and dropped buffers must be reused."""
    import gc
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    from lidar_snow_sim_amd.tools.snowfall.simulation import augment_batch
    tl = _tables64(tables)
    bd = float(np.degrees(3e-3))
    fr = [synthetic_sweep(64, 128, seed=1020 + i, intensity="lambert") for i in range(3)]
    kw = dict(particles=tl, orders=[list(range(64))] * 3, thr_polys=[[0.0, 0.01, 2.0]] * 3)
    first = augment_batch(fr, "unused", bd, **kw)
    snap = [a.copy() for _, a in first]
    second = augment_batch(fr[::-1], "unused", bd, **kw)
    for (_, a), s in zip(first, snap):
        assert np.array_equal(a, s)
    for (_, a), (_, b) in zip(first, second[::-1]):
        assert np.array_equal(a, b)
    addr = first[0][1].__array_interface__["data"][0]
    del first, second, a, b
    gc.collect()
    again = augment_batch(fr, "unused", bd, **kw)
    pool_addrs = {again[0][1].__array_interface__["data"][0]}
    assert addr in pool_addrs or len(eng.__dict__.get("_pin_pool", [])) >= 1
    for (_, a), s in zip(again, snap):
        assert np.array_equal(a, s)


def test_odd_channel_values_fall_back_to_a_host_permutation(eng, so, tables):
    """Channel values the device counting sort does not take (non-integers, > 255, negative) make the library report
    SNOWGPU_E_CHANNELS; the Python mirror then sorts on the host and runs the batch again.  Such rows are copied
    through like any channel without a laser (Q5)."""
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    from lidar_snow_sim_amd.tools.snowfall.simulation import augment_batch
    pc = synthetic_sweep(64, 96, seed=1031, intensity="lambert")
    pc[5::37, 4] = 3.5
    pc[7::41, 4] = 300.0
    pc[11::43, 4] = -2.0
    tl = _tables64(tables)
    bd = float(np.degrees(3e-3))
    poly = [0.0, 0.01, 2.0]
    (st, aug, src), = augment_batch([pc], "unused", bd, particles=tl, orders=[list(range(64))], thr_polys=[poly], return_src=True)
    s0, a0, src0 = so.augment(pc, tl, bd, list(range(64)), thr_poly=np.array(poly))
    assert tuple(int(v) for v in st) == tuple(int(v) for v in s0)
    assert np.array_equal(src, src0) and np.array_equal(aug[:, 3:], a0[:, 3:])


def test_full_size_batch_properties(eng, tables):
    """BASELINE-sized sweeps (64 x 2048) in one batch, checked through properties that do not need the oracle:
    batching invariance, run-to-run determinism, bookkeeping identities, untouched rows bit-identical to their source,
    scattered points on the ray of their source and nearer than it."""
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    F, N = 8, 64 * 2048
    frames = [synthetic_sweep(64, 2048, seed=1000 + f, intensity="lambert") for f in range(F)]
    rows = np.concatenate(frames)
    tl = _tables64(tables)
    rng = np.random.default_rng(3)
    tids = [eng.table_ids_from_arrays(tl, list(rng.permutation(64))) for _ in range(F)]
    bd = float(np.degrees(3e-3))
    planes = [[0.0, 0.0, -1.0, -1.7]] * F
    off = np.arange(F + 1, dtype=np.int64) * N

    def run(lo, hi):
        o, s, c, st, _ = eng.ctx.augment_batch(rows[lo * N:hi * N], off[:hi - lo + 1], tids[lo:hi], bd, plane=planes[lo:hi])
        return [(o[i * N:i * N + int(c[i])].copy(), s[i * N:i * N + int(c[i])].copy(), st[i].copy()) for i in range(hi - lo)]

    whole = run(0, F)
    again = run(0, F)
    halves = run(0, F // 2) + run(F // 2, F)
    n_moved = 0
    for f in range(F):
        o, s, st = whole[f]
        for other in (again[f], halves[f]):
            assert o.tobytes() == other[0].tobytes() and np.array_equal(s, other[1]) and np.array_equal(st, other[2])
        assert int(st[1]) == N - o.shape[0]                                  # num_removed (simulation.py:522)
        assert int(st[0]) == int((o[:, 4] == 1).sum())                       # num_attenuated (:525)
        assert set(np.unique(o[:, 4])) <= {0.0, 1.0, 2.0}
        src = frames[f][s]
        assert np.all(np.diff(src[:, 4]) >= 0)                               # channel-sorted output order (:447)
        same = o[:, 4] == 0
        assert np.array_equal(o[same, :4], src[same, :4])
        mv = o[:, 4] == 2
        n_moved += int(mv.sum())
        po, pi = o[mv, :3].astype(np.float64), src[mv, :3].astype(np.float64)
        d_out, d_in = np.linalg.norm(po, axis=1), np.linalg.norm(pi, axis=1)
        assert np.all(d_out < d_in)
        cosang = np.einsum("ij,ij->i", po, pi) / (d_out * d_in)
        # on the ray -- or on its mirror image: a beam whose only power comes from flakes nearer than 0.9 m (xsi = 0)
        # has an all-zero power profile, argmax 0, d_max = -c tau / 2 and a negative scale (simulation.py:151-180)
        assert np.all(np.abs(cosang) > 1 - 1e-6)
        assert np.all((cosang > 0) | (np.abs(d_out - 299792458.0 * 1e-8 / 2) < 1e-5))
    assert n_moved > 100


@pytest.mark.parametrize("extra,first_tier", [(60000, 16), (110000, 63)])
def test_dense_tables_start_at_a_higher_capacity_tier(so, tables, extra, first_tier):
    """The first capacity tier follows the table size (choose_tiers): 16 entries above ~50 k flakes per line, the
    63-entry tier above ~84 k.  Those direct-mode passes, their 64-thread blocks, segment order and k_power<16/63> are
    exercised by padding a table with flakes beyond every target (they never intersect a beam, but they count)."""
    from lidar_snow_sim_amd import engine
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    rng = np.random.default_rng(extra)
    base = tables["t"][0]
    rho = rng.uniform(150.0, 200.0, extra)
    phi = rng.uniform(0, 2 * np.pi, extra)
    far = np.column_stack((rho * np.cos(phi), rho * np.sin(phi), np.full(extra, 1e-3)))
    big = np.concatenate((base, far))
    tl = [big if i % 2 == 0 else tables["t"][1] for i in range(64)]
    full = synthetic_sweep(64, 2048, seed=1040, intensity="lambert").reshape(64, 2048, 5)
    pc = full[:, ::24, :].reshape(-1, 5).copy()                 # 86 rows per channel: two 64-row blocks per segment
    order = list(range(64))
    bd = float(np.degrees(3e-3))
    poly = [0.0, 0.01, 2.0]
    eng2 = engine.Engine(0)                                     # own context: max table size drives the tier choice
    try:
        tids = eng2.table_ids_from_arrays(tl, order)
        out, src, counts, stats, _ = eng2.ctx.augment_batch(pc, [0, pc.shape[0]], [tids], bd, thr_poly=[poly])
    finally:
        eng2.ctx.close()
    s0, a0, src0 = so.augment(pc, tl, bd, order, thr_poly=np.array(poly), threads=8)
    n = int(counts[0])
    assert tuple(int(v) for v in stats[0]) == tuple(int(v) for v in s0)
    assert np.array_equal(src[:n], src0) and np.array_equal(out[:n, 3:], a0[:, 3:])
    np.testing.assert_allclose(out[:n, :3], a0[:, :3], rtol=1e-6, atol=0)
    assert (a0[:, 4] == 1).sum() > 50


def test_device_entry_can_be_captured_into_a_hip_graph(eng, tables):
    """snowgpu_augment_batch_device neither allocates (after the first call of a given size), nor synchronises, nor copies
    from host memory: with its side streams forking from and joining back into the caller's stream, the whole launch
    sequence is capturable and a replay gives the same rows."""
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    dev = torch.device("cuda:0")
    F, n = 2, 64 * 256
    frames = [synthetic_sweep(64, 256, seed=1050 + f, intensity="lambert") for f in range(F)]
    rows = torch.from_numpy(np.concatenate(frames)).to(dev)
    off = torch.arange(F + 1, dtype=torch.int64, device=dev) * n
    tids = torch.tensor([eng.table_ids_from_arrays(_tables64(tables), list(range(64))) for _ in range(F)], dtype=torch.int32, device=dev)
    plane = torch.tensor([[0.0, 0.0, -1.0, -1.7]] * F, dtype=torch.float64, device=dev)
    out = torch.empty_like(rows)
    src = torch.empty(F * n, dtype=torch.int32, device=dev)
    cnt = torch.zeros(F, dtype=torch.int64, device=dev)
    st = torch.zeros(F, 3, dtype=torch.int64, device=dev)
    status = torch.zeros(8, dtype=torch.int32, device=dev)
    s = torch.cuda.Stream()

    def call():
        eng.ctx.augment_batch_device(F, F * n, n, off.data_ptr(), rows.data_ptr(), 0, tids.data_ptr(), float(np.degrees(3e-3)), 0,
                                     plane.data_ptr(), 0.7, 0, out.data_ptr(), src.data_ptr(), cnt.data_ptr(), st.data_ptr(), 0,
                                     status.data_ptr(), s.cuda_stream)

    other = [synthetic_sweep(64, 256, seed=1070 + f, intensity="lambert") for f in range(F)]
    inputs = [rows.clone(), torch.from_numpy(np.concatenate(other)).to(dev), rows.clone()]

    def result():
        return out.clone(), src.clone(), cnt.clone(), st.clone()

    with torch.cuda.stream(s):
        want = []
        for inp in inputs:                       # what plain calls give for each input (two calls each: the second allocates nothing)
            rows.copy_(inp)
            call()
            call()
            s.synchronize()
            want.append(result())
        assert not torch.equal(want[0][3], want[1][3])          # the inputs really differ
        rows.copy_(inputs[0])
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            call()
        # Replayed three times on changing input: everything a call counts up from zero (the prepass histogram, the queue counters) has to
        # be cleared INSIDE the graph -- a fill issued on a side stream before it joined the capture ran once, at capture time, and the
        # second replay would fit its noise threshold to a histogram that still held the first one's counts.
        for k, inp in enumerate(inputs):
            rows.copy_(inp)
            out.zero_(); src.zero_(); cnt.zero_(); st.zero_()
            g.replay()
            s.synchronize()
            assert int(status[0]) == 0 and torch.equal(cnt, want[k][2]), k
            assert torch.equal(st, want[k][3]), (k, st, want[k][3])
            for f in range(F):
                m = int(cnt[f])
                assert m > 0 and torch.equal(out[f * n:f * n + m], want[k][0][f * n:f * n + m]), (k, f)
                assert torch.equal(src[f * n:f * n + m], want[k][1][f * n:f * n + m]), (k, f)


def test_compact_input_gives_the_bytes_of_the_row_entry(tables):
    """snowgpu_augment_batch_compact: frames as (x, y, z, intensity) float32 rows + one channel byte per row (17 B per point up the link instead
    of 20; k_expand_rows makes the rows on the device) -- one sweep (single chunk) and 40 sweeps (four chunks of the pipeline on two lanes), rows
    and packed result transfer, with rows of a channel that has no laser (their output keeps the channel value, quirk Q5): out_rows, out_src,
    counts and statistics are byte for byte those of snowgpu_augment_batch on the five-column rows."""
    from lidar_snow_sim_amd import engine
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    tl = _tables64(tables)
    eng = engine.Engine(0)
    try:
        for n_frames in (1, 40):
            frames = [synthetic_sweep(64, 2048, seed=1400 + f, intensity="lambert") for f in range(n_frames)]
            for f in frames:
                f[5::997, 4] = 200.0                                # a channel without a laser
            rows = np.concatenate(frames)
            off = np.arange(n_frames + 1, dtype=np.int64) * frames[0].shape[0]
            tids = [eng.table_ids_from_arrays(tl, list(np.random.default_rng(f).permutation(64))) for f in range(n_frames)]
            planes = [[0.0, 0.0, -1.0, -1.7]] * n_frames
            want = eng.ctx.augment_batch(rows, off, tids, float(np.degrees(3e-3)), plane=planes)
            xyzi, ch = np.ascontiguousarray(rows[:, :4]), rows[:, 4].astype(np.uint8)
            for mode in ("rows", "packed"):
                eng.ctx.set_result_transfer(mode)
                try:
                    got = eng.ctx.augment_batch_compact(xyzi, ch, off, tids, float(np.degrees(3e-3)), plane=planes)
                finally:
                    eng.ctx.set_result_transfer("rows")
                assert np.array_equal(got[2], want[2]) and np.array_equal(got[3], want[3]), (n_frames, mode)
                for f in range(n_frames):
                    a, m = int(off[f]), int(want[2][f])
                    assert got[0][a:a + m].tobytes() == want[0][a:a + m].tobytes() and np.array_equal(got[1][a:a + m], want[1][a:a + m]), (n_frames, mode, f)
            assert (want[0][:int(want[2][0]), 4] == 200.0).sum() > 50
    finally:
        eng.ctx.close()


def _stretched_subsweep(step=8, scale=1.8, seed=1060):
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    full = synthetic_sweep(64, 2048, seed=seed, intensity="lambert").reshape(64, 2048, 5)
    pc = np.ascontiguousarray(full[:, ::step, :].reshape(-1, 5))
    r = np.linalg.norm(pc[:, :3].astype(np.float64), axis=1)
    pc[:, :3] = (pc[:, :3] * (np.minimum(r * scale, 119.0) / r)[:, None]).astype(np.float32)
    return pc


def test_tier_hand_over_buffer_overflow_runs_in_place(so, tables, monkeypatch):
    """The later capacity tiers hand their occlusion dicts to k_power through a buffer sized for a fraction of the batch;
    the entries beyond it run the received-power phase in place.  SNOWGPU_TIER_CAP=48 forces that path: same bytes as with
    the default buffers, and both equal the oracle."""
    from lidar_snow_sim_amd import engine
    pc = _stretched_subsweep()
    tl = _tables64(tables)
    bd = float(np.degrees(3e-3))
    order = list(range(64))
    poly = [0.0, 0.01, 2.0]
    results = []
    for cap in ("0", "48"):
        monkeypatch.setenv("SNOWGPU_TIER_CAP", cap)
        e = engine.Engine(0)
        try:
            tids = e.table_ids_from_arrays(tl, order)
            results.append(e.ctx.augment_batch(pc, [0, pc.shape[0]], [tids], bd, thr_poly=[poly]))
            st = e.ctx.last_status()
        finally:
            e.ctx.close()
    assert st[2] > 48, st                                        # the second tier really had more beams than the buffer holds
    (o0, s0, c0, st0, _), (o1, s1, c1, st1, _) = results
    n = int(c0[0])
    assert np.array_equal(c0, c1) and np.array_equal(st0, st1) and np.array_equal(s0[:n], s1[:n])
    assert o0[:n].tobytes() == o1[:n].tobytes()
    r_stats, r_aug, r_src = so.augment(pc, tl, bd, order, thr_poly=np.array(poly))
    assert tuple(int(v) for v in st0[0]) == tuple(int(v) for v in r_stats)
    assert np.array_equal(s0[:n], r_src) and np.array_equal(o0[:n, 3:], r_aug[:, 3:])


@pytest.mark.parametrize("first", [4, 8, 16, 63])
def test_every_first_tier_gives_the_same_rows(so, tables, monkeypatch, first):
    """SNOWGPU_FIRST_TIER picks the capacity the pass over all rows starts with (normally chosen from the table size):
    k_beams / k_power of every capacity in direct mode, the remaining tiers in list mode -- all against the oracle."""
    from lidar_snow_sim_amd import engine
    pc = _stretched_subsweep(step=16, seed=1061)
    tl = _tables64(tables)
    bd = float(np.degrees(3e-3))
    order = list(np.random.default_rng(first).permutation(64))
    poly = [0.0, 0.01, 2.0]
    monkeypatch.setenv("SNOWGPU_FIRST_TIER", str(first))
    e = engine.Engine(0)
    try:
        out, src, counts, stats, _ = e.ctx.augment_batch(pc, [0, pc.shape[0]], [e.table_ids_from_arrays(tl, order)], bd, thr_poly=[poly])
    finally:
        e.ctx.close()
    r_stats, r_aug, r_src = so.augment(pc, tl, bd, order, thr_poly=np.array(poly))
    n = int(counts[0])
    assert tuple(int(v) for v in stats[0]) == tuple(int(v) for v in r_stats)
    assert np.array_equal(src[:n], r_src) and np.array_equal(out[:n, 3:], r_aug[:, 3:])
    np.testing.assert_allclose(out[:n, :3], r_aug[:, :3], rtol=1e-6, atol=0)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_more_than_63_flakes_in_one_beam_take_the_global_list_tier(so, tables, dtype):
    """The reference's per-beam lists are unbounded (simulation.py:413-419).  Beams that meet more flakes than the largest
    LDS list holds run with lists in global memory: a column of 150 / 400 small flakes strung along one azimuth."""
    from lidar_snow_sim_amd import engine
    rng = np.random.default_rng(63)
    base = tables["t"][0]
    cols = []
    for az, m in ((0.7, 150), (2.1, 400)):
        rho = np.linspace(4.0, 55.0, m) + rng.uniform(-0.02, 0.02, m)
        phi = az + rng.uniform(-8e-4, 8e-4, m)
        cols.append(np.column_stack((rho * np.cos(phi), rho * np.sin(phi), rng.uniform(5e-4, 3e-3, m))))
    big = np.concatenate([base] + cols)
    tl = [big] * 64
    n = 600
    az = rng.uniform(-np.pi, np.pi, n)
    az[:40] = 0.7 + rng.uniform(-2e-3, 2e-3, 40)
    az[40:80] = 2.1 + rng.uniform(-2e-3, 2e-3, 40)
    d = rng.uniform(20.0, 110.0, n)
    el = rng.uniform(-0.3, 0.03, n)
    pc = np.column_stack((d * np.cos(el) * np.cos(az), d * np.cos(el) * np.sin(az), d * np.sin(el), rng.integers(0, 256, n),
                          rng.integers(0, 64, n))).astype(dtype)
    bd = float(np.degrees(3e-3))
    order = list(range(64))
    poly = [0.0, 0.0, 1.0]
    e = engine.Engine(0)
    try:
        out, src, counts, stats, _ = e.ctx.augment_batch(pc, [0, n], [e.table_ids_from_arrays(tl, order)], bd, thr_poly=[poly])
        st = e.ctx.last_status()
    finally:
        e.ctx.close()
    assert st[5] >= 10, st                                       # beams handed to the global-list tier
    r_stats, r_aug, r_src = so.augment(pc, tl, bd, order, thr_poly=np.array(poly))
    m = int(counts[0])
    assert tuple(int(v) for v in stats[0]) == tuple(int(v) for v in r_stats)
    assert np.array_equal(src[:m], r_src) and np.array_equal(out[:m, 3:], r_aug[:, 3:])
    np.testing.assert_allclose(out[:m, :3], r_aug[:, :3], rtol=1e-6 if dtype == np.float32 else 1e-12, atol=0)


def test_fused_snow_and_wet_device_entry_is_capturable(eng, so, tables):
    """snowgpu_augment_wet_batch_device: the snowfall rows feed the wet-ground kernels on the same stream with no host
    copy, no synchronisation and no allocation (after the first call of a size) -- so the chain can be captured into a HIP
    graph, and the replay equals the oracle chain (pointcloud_viewer.py:2807-2821)."""
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    dev = torch.device("cuda:0")
    F, n = 2, 64 * 512
    frames = [synthetic_sweep(64, 512, seed=1070 + f, intensity="lambert") for f in range(F)]
    tl = _tables64(tables)
    order = list(range(64))
    bd = float(np.degrees(3e-3))
    rows = torch.from_numpy(np.concatenate(frames)).to(dev)
    off = torch.arange(F + 1, dtype=torch.int64, device=dev) * n
    tids = torch.tensor([eng.table_ids_from_arrays(tl, order) for _ in range(F)], dtype=torch.int32, device=dev)
    plane = torch.tensor([[0.0, 0.0, -1.0, -1.7]] * F, dtype=torch.float64, device=dev)
    out = torch.zeros(F * n, 5, dtype=torch.float64, device=dev)
    src = torch.zeros(F * n, dtype=torch.int32, device=dev)
    cnt = torch.zeros(F, dtype=torch.int64, device=dev)
    st = torch.zeros(F, 3, dtype=torch.int64, device=dev)
    flags = torch.zeros(F, dtype=torch.int32, device=dev)
    status = torch.zeros(8, dtype=torch.int32, device=dev)
    s = torch.cuda.Stream()

    def call():
        eng.ctx.augment_wet_batch_device(F, F * n, n, off.data_ptr(), rows.data_ptr(), 0, tids.data_ptr(), bd, 0, plane.data_ptr(), 0.7,
                                         0, plane.data_ptr(), 0.0008, 0.001, 0.7, 15.0, False, 0.5, False, out.data_ptr(),
                                         src.data_ptr(), cnt.data_ptr(), st.data_ptr(), flags.data_ptr(), status.data_ptr(),
                                         s.cuda_stream)

    with torch.cuda.stream(s):
        call()
        call()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            call()
        for _ in range(3):                       # (replayed more than once: whatever a call clears has to be cleared inside the graph)
            out.zero_(); src.zero_(); cnt.zero_(); st.zero_()
            g.replay()
            s.synchronize()
    assert int(status[0]) == 0
    for f in range(F):
        s0, a0, src0 = so.augment(frames[f], tl, bd, order, plane=PLANE)
        o0, wsrc0 = so.ground_water_augmentation(a0, water_height=0.0008, pavement_depth=0.001, flat_earth=False, replace=False,
                                                 plane=PLANE, return_src=True)
        m = int(cnt[f])
        got = out[f * n:f * n + m].cpu().numpy()
        assert tuple(int(v) for v in st[f].cpu()) == tuple(int(v) for v in s0) and int(flags[f]) == 0
        assert got.shape == o0.shape and np.array_equal(got[:, 4], o0[:, 4])
        assert np.array_equal(src[f * n:f * n + m].cpu().numpy(), src0[wsrc0])
        np.testing.assert_allclose(got[:, :3], o0[:, :3], rtol=1e-6, atol=0)
        np.testing.assert_allclose(got[:, 3], o0[:, 3], rtol=1e-6, atol=0)


def test_pre_augment_crop_on_device_matches_crop_then_augment(eng, so, tables):
    """tools/snowfall/precompute.py:96-104: crop the frame to the camera's view, THEN augment (whose only_camera_fov default
    crops the result again).  pre_crop=True does the first crop on the device right after the upload; checked against the
    oracle chain crop -> augment -> crop, batch of two ragged frames (projection parity unpinned: SURVEY 8 c)."""
    from lidar_snow_sim_amd.calibration import Calibration
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    from lidar_snow_sim_amd.tools.snowfall.simulation import augment_batch
    cal = Calibration(P2=np.array([[700.0, 0, 960, 0], [0, 700.0, 512, 0], [0, 0, 1, 0]]), R0=np.eye(3),
                      V2C=np.array([[0, -1.0, 0, 0], [0, 0, -1.0, 0], [1.0, 0, 0, 0]]))
    full = synthetic_sweep(64, 2048, seed=23, intensity="lambert").reshape(64, 2048, 5)
    frames = [np.ascontiguousarray(full[:, ::4, :].reshape(-1, 5)), np.ascontiguousarray(full[:, 1::8, :].reshape(-1, 5))]
    bd = float(np.degrees(3e-3))
    tl = _tables64(tables)
    orders = [list(range(64)), list(range(63, -1, -1))]
    res = augment_batch(frames, "unused", bd, particles=tl, orders=orders, planes=[PLANE, PLANE], return_src=True, calib=cal,
                        pre_crop=True)
    for pc, order, (st, aug, src) in zip(frames, orders, res):
        flag1 = so.fov_flag(pc[:, :3], cal.V2C, cal.R0, cal.P2, (1024, 1920))
        kept1 = np.where(flag1)[0]
        assert 0 < kept1.size < pc.shape[0]
        s0, a0, src0 = so.augment(pc[flag1], tl, bd, order, plane=PLANE)
        flag2 = so.fov_flag(a0[:, :3], cal.V2C, cal.R0, cal.P2, (1024, 1920))
        assert np.array_equal(src, kept1[src0][flag2])                       # rows of the ORIGINAL frame
        assert np.array_equal(aug[:, 3:], a0[flag2][:, 3:])
        np.testing.assert_allclose(aug[:, :3], a0[flag2][:, :3], rtol=1e-6, atol=0)
        assert (int(st[0]), int(st[1]), int(st[2])) == (int(s0[0]), int(s0[1]) + int((~flag2).sum()), int(s0[2]))


@pytest.mark.parametrize("mode", ["-1"])
def test_per_lane_and_wave_scan_give_the_same_rows(so, tables, monkeypatch, mode):
    """The candidate scan has two forms: one beam per lane, and the wave-flattened one (every lane tests one (beam, record)
    pair, hits appended through LDS counters, each beam's entries ordered afterwards).  The pass over all rows uses the
    second and the tiers, by default, the first; SNOWGPU_PER_LANE_SCAN=-1 makes the tiers use the second too -- same bytes,
    and both equal the oracle.  Wide beams (30 mrad: wedges of ten bins) exercise the per-lane part behind the first two
    bins; SNOWGPU_FIRST_TIER=8 ... 63 (test_every_first_tier_gives_the_same_rows) runs every capacity in both forms."""
    from lidar_snow_sim_amd import engine
    tl = _tables64(tables)
    order = list(range(64))
    poly = [0.0, 0.01, 2.0]
    for bd in (float(np.degrees(3e-3)), float(np.degrees(3e-2))):
        pc = _stretched_subsweep(step=16 if bd < 0.5 else 64, seed=1080)
        outs = []
        for env in ("0", mode):
            monkeypatch.setenv("SNOWGPU_PER_LANE_SCAN", env)
            e = engine.Engine(0)
            try:
                outs.append(e.ctx.augment_batch(pc, [0, pc.shape[0]], [e.table_ids_from_arrays(tl, order)], bd, thr_poly=[poly]))
            finally:
                e.ctx.close()
        (o0, s0, c0, st0, _), (o1, s1, c1, st1, _) = outs
        n = int(c0[0])
        assert np.array_equal(c0, c1) and np.array_equal(st0, st1) and np.array_equal(s0[:n], s1[:n]) and o0[:n].tobytes() == o1[:n].tobytes()
        r_stats, r_aug, r_src = so.augment(pc, tl, bd, order, thr_poly=np.array(poly))
        assert tuple(int(v) for v in st0[0]) == tuple(int(v) for v in r_stats)
        assert np.array_equal(s0[:n], r_src) and np.array_equal(o0[:n, 3:], r_aug[:, 3:])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_pipelined_host_entry_equals_the_single_chunk_call(eng, tables, dtype):
    """snowgpu_augment_batch cuts a large host batch into chunks of whole frames and overlaps upload / kernels / download
    (snowgpu_set_pipeline).  Same bytes as the one-chunk call: ragged frames incl. an empty one, device prepass and caller
    polynomials, caller permutation, out_src = NULL, out_thr_poly, status words summed over the chunks."""
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    rng = np.random.default_rng(11)
    frames = []
    for f in range(11):
        full = synthetic_sweep(64, 2048, seed=1100 + f, intensity="lambert").reshape(64, 2048, 5)
        step = int(rng.integers(20, 60))
        frames.append(np.ascontiguousarray(full[:, f % 7::step, :].reshape(-1, 5)).astype(dtype))
    F = len(frames)
    tl = _tables64(tables)
    tids = [eng.table_ids_from_arrays(tl, list(rng.permutation(64))) for _ in range(F)]
    bd = float(np.degrees(3e-3))
    planes = [[0.0, 0.0, -1.0, -1.7]] * F
    polys = [[1e-4 * f, 0.01, 2.0] for f in range(F)]
    rows = off = None

    def run(chunk_rows, **kw):
        nonlocal rows, off
        fr = list(frames)
        if "thr_poly" in kw:
            fr[4] = np.zeros((0, 5), dtype)                  # an empty frame (with a plane the reference raises on it: Q7)
        rows = np.concatenate(fr)
        off = np.concatenate(([0], np.cumsum([f.shape[0] for f in fr]))).astype(np.int64)
        if "perm" in kw:
            kw = dict(kw, perm=np.concatenate([np.argsort(f[:, 4], kind="stable") for f in fr]).astype(np.int32))
        eng.ctx.set_pipeline(chunk_rows)
        try:
            o, s, c, st, thr = eng.ctx.augment_batch(rows, off, tids, bd, **kw)
            return o.copy(), None if s is None else s.copy(), c.copy(), st.copy(), thr, eng.ctx.last_status().copy()
        finally:
            eng.ctx.set_pipeline(3 << 19)

    for kw in (dict(plane=planes, want_thr=True), dict(thr_poly=polys), dict(thr_poly=polys, perm=True), dict(plane=planes, want_src=False)):
        one = run(0, **kw)
        for chunk_rows in (2500, 9000):                      # ~11 and ~4 chunks
            many = run(chunk_rows, **kw)
            assert np.array_equal(one[2], many[2]) and np.array_equal(one[3], many[3])
            for f in range(F):
                a, n = int(off[f]), int(one[2][f])
                assert one[0][a:a + n].tobytes() == many[0][a:a + n].tobytes()
                if one[1] is not None:
                    assert np.array_equal(one[1][a:a + n], many[1][a:a + n])
            assert (one[1] is None) == (many[1] is None)
            if kw.get("want_thr"):
                assert np.array_equal(one[4], many[4])
            assert np.array_equal(one[5][2:6], many[5][2:6])
    # an error in a middle chunk is reported like in the one-chunk call (range >= 120 m -> SNOWGPU_E_RANGE)
    from lidar_snow_sim_amd import _native
    run(0, thr_poly=polys)
    bad = rows.copy()
    bad[int(off[6]) + 3, :3] = (150.0, 0.0, 0.0)
    eng.ctx.set_pipeline(2500)
    try:
        with pytest.raises(_native.SnowGPUError) as ei:
            eng.ctx.augment_batch(bad, off, tids, bd, thr_poly=polys)
        assert ei.value.code == _native.E_RANGE
    finally:
        eng.ctx.set_pipeline(3 << 19)


# ---- the literal drop-in call: tables in .npy files, permutation from the seeded global `random`, no keyword extras ----------
def _write_table_files(directory, tables, prefix="gunn_x_y"):
    directory.mkdir(parents=True, exist_ok=True)
    for line in range(1, 65):
        np.save(directory / f"{prefix}_{line}.npy", tables["t"][(line - 1) % 4])


@pytest.mark.parametrize("case", range(8))
def test_L5_literal_drop_in_call(eng, golden, tables, tmp_path, case):
    """Exactly the call the L5 fixtures were captured with (tests/golden/make_golden.py): <root>/training/snowflakes/npy/
    <prefix>_<1..64>.npy on disk, random.seed(s), augment(pc6, prefix, bd, shuffle=..., only_camera_fov=False, root_path=root)
    with a sixth source-index column riding through (simulation.py:447-523).  Fallback-plane cases (even) go through the real
    calculate_plane; the injected-plane cases pass the fixture's plane, the one keyword extra they need."""
    import random
    from lidar_snow_sim_amd.tools.snowfall.simulation import augment
    d = golden("L5_augment")
    _write_table_files(tmp_path / "training" / "snowflakes" / "npy", tables)
    pc = d[f"c{case}_pc"]
    pc6 = np.column_stack((pc, np.arange(len(pc)))).astype(pc.dtype)
    shuffle, seed = ((True, 3), (False, 0))[(case // 2) % 2]
    injected = bool(d[f"c{case}_injected"])
    kw = dict(plane=(d[f"c{case}_plane_w"], float(d[f"c{case}_plane_h"]))) if injected else {}
    random.seed(seed)
    stats, aug = augment(pc6, "gunn_x_y", float(d["bd"]), shuffle=shuffle, show_progressbar=False, only_camera_fov=False,
                         noise_floor=0.7, root_path=str(tmp_path), **kw)
    assert tuple(int(s) for s in stats) == tuple(int(v) for v in d[f"c{case}_stats"])
    assert aug.shape[1] == 6 and aug.dtype == pc.dtype
    a1, s1 = canonical(aug[:, :5], aug[:, 5].astype(np.int64))
    a2, s2 = canonical(d[f"c{case}_aug"], d[f"c{case}_src"])
    assert np.array_equal(s1, s2)                              # the sixth column came through: bit-exact kept-point indices
    assert np.array_equal(a1[:, 3:], a2[:, 3:])
    np.testing.assert_allclose(a1[:, :3], a2[:, :3], rtol=1e-6 if pc.dtype == np.float32 else 1e-12, atol=0)


@pytest.mark.parametrize("case", range(4))
def test_L8_viewer_chain_literal(eng, golden, tables, tmp_path, case):
    """pointcloud_viewer.py:2807-2821 with its keyword arguments: augment(pc=..., only_camera_fov=..., particle_file_prefix=...,
    noise_floor=..., beam_divergence=..., shuffle=True, show_progressbar=True) reading <repo>/npy, then
    ground_water_augmentation(pc, ..., debug=False, delta=..., replace=False); against what the reference returned (L8)."""
    import random
    from lidar_snow_sim_amd import engine
    from lidar_snow_sim_amd.tools.snowfall.simulation import augment
    from lidar_snow_sim_amd.tools.wet_ground.augmentation import ground_water_augmentation
    from test_oracle_golden import match_rows_by_xyz
    d = golden("L8_viewer_chain")
    _write_table_files(tmp_path / "npy", tables)
    pc = d[f"c{case}_pc"]
    pc6 = np.column_stack((pc, np.arange(len(pc)))).astype(pc.dtype)
    kw = dict(plane=PLANE) if bool(d[f"c{case}_inject"]) else {}
    engine.set_particle_dir(tmp_path / "npy")                  # the reference's <repo>/npy (simulation.py:326-327)
    try:
        random.seed(int(d[f"c{case}_seed"]))
        stats, snow = augment(pc=pc6, only_camera_fov=False, particle_file_prefix="gunn_x_y", noise_floor=0.7,
                              beam_divergence=float(np.degrees(3e-3)), shuffle=True, show_progressbar=True, **kw)
    finally:
        engine.set_particle_dir(None)
    assert tuple(int(s) for s in stats) == tuple(int(v) for v in d[f"c{case}_stats"])
    a1, s1 = canonical(snow[:, :5], snow[:, 5].astype(np.int64))
    a2, s2 = canonical(d[f"c{case}_snow"], d[f"c{case}_snow_src"])
    assert np.array_equal(s1, s2) and np.array_equal(a1[:, 3:], a2[:, 3:])
    tol = 1e-6 if pc.dtype == np.float32 else 1e-12
    np.testing.assert_allclose(a1[:, :3], a2[:, :3], rtol=tol, atol=0)
    out = ground_water_augmentation(snow[:, :5], water_height=0.0008, pavement_depth=0.001, noise_floor=0.7, power_factor=15,
                                    flat_earth=False, estimation_method="linear", debug=False, delta=0.5, replace=False, **kw)
    ref = d[f"c{case}_out"]
    assert out.dtype == ref.dtype and out.shape == ref.shape
    ids = match_rows_by_xyz(out, snow[:, :5], snow[:, 5].astype(np.int64))
    ref_ids = match_rows_by_xyz(ref, d[f"c{case}_snow"], d[f"c{case}_snow_src"])
    assert np.array_equal(np.sort(ids), np.sort(ref_ids))      # same rows kept by both stages
    o1, o2 = out[np.argsort(ids, kind="stable")], ref[np.argsort(ref_ids, kind="stable")]
    assert np.array_equal(o1[:, 4], o2[:, 4])
    np.testing.assert_allclose(o1[:, :3], o2[:, :3], rtol=tol, atol=0)
    np.testing.assert_allclose(o1[:, 3], o2[:, 3], rtol=1e-7 if pc.dtype == np.float32 else 1e-9, atol=0)


def test_pre_crop_with_odd_channel_values_and_plane_from_the_cropped_cloud(eng, so, tables):
    """Two advisor findings on the pre-augment crop (precompute.py:96-104).  (1) Channel values the device sort refuses make the
    mirror sort on the host; with pre_crop that used to fail (the device pre-crop takes no caller permutation): the retry now
    crops on the host.  (2) With planes=None the plane has to be fitted on the CROPPED cloud, as the reference does: the rows
    handed to calculate_plane are the camera-view part of its own crop window."""
    from lidar_snow_sim_amd.calibration import Calibration, get_fov_flag
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    from lidar_snow_sim_amd.tools.snowfall import simulation as sim
    cal = Calibration(P2=np.array([[700.0, 0, 960, 0], [0, 700.0, 512, 0], [0, 0, 1, 0]]), R0=np.eye(3),
                      V2C=np.array([[0, -1.0, 0, 0], [0, 0, -1.0, 0], [1.0, 0, 0, 0]]))
    pc = np.ascontiguousarray(synthetic_sweep(64, 2048, seed=29, intensity="lambert").reshape(64, 2048, 5)[:, ::4, :].reshape(-1, 5))
    pc[5::37, 4] = 3.5
    pc[7::41, 4] = 300.0
    tl = _tables64(tables)
    bd = float(np.degrees(3e-3))
    order = list(range(64))
    (st, aug, src), = sim.augment_batch([pc], "unused", bd, particles=tl, orders=[order], planes=[PLANE], return_src=True, calib=cal,
                                        pre_crop=True)
    flag1 = so.fov_flag(pc[:, :3], cal.V2C, cal.R0, cal.P2, (1024, 1920))
    kept1 = np.where(flag1)[0]
    s0, a0, src0 = so.augment(pc[flag1], tl, bd, order, plane=PLANE)
    flag2 = so.fov_flag(a0[:, :3], cal.V2C, cal.R0, cal.P2, (1024, 1920))
    assert np.array_equal(src, kept1[src0][flag2]) and np.array_equal(aug[:, 3:], a0[flag2][:, 3:])
    assert (int(st[0]), int(st[1]), int(st[2])) == (int(s0[0]), int(s0[1]) + int((~flag2).sum()), int(s0[2]))
    # (2) planes=None: calculate_plane runs on the device AFTER the device crop, i.e. on the cropped cloud.  A narrow camera
    # (the strip of planes.py:21-27 is wider than its view) and a road that banks outside the view make the two planes differ.
    from lidar_snow_sim_amd.tools.wet_ground.planes import calculate_plane
    narrow = Calibration(P2=np.array([[7000.0, 0, 960, 0], [0, 700.0, 512, 0], [0, 0, 1, 0]]), R0=np.eye(3),
                         V2C=np.array([[0, -1.0, 0, 0], [0, 0, -1.0, 0], [1.0, 0, 0, 0]]))
    road = np.ascontiguousarray(synthetic_sweep(64, 2048, seed=31, intensity="lambert"))
    bank = np.abs(road[:, 1]) > 1.0
    road[bank, 2] -= (0.02 * (np.abs(road[bank, 1]) - 1.0)).astype(np.float32)
    seen = get_fov_flag(narrow.lidar_to_rect(road[:, 0:3]), (1024, 1920), narrow)
    w_all, h_all = calculate_plane(road, method="lsq")
    w_crop, h_crop = calculate_plane(road[seen], method="lsq")
    assert abs(h_all - h_crop) > 1e-3                                    # the crop matters for this cloud
    (st1, a1, s1), = sim.augment_batch([road], "unused", bd, particles=tl, orders=[order], return_src=True, calib=narrow, pre_crop=True,
                                       plane_method="lsq")
    (st2, a2, s2), = sim.augment_batch([road], "unused", bd, particles=tl, orders=[order], return_src=True, calib=narrow, pre_crop=True,
                                       planes=[(w_crop, h_crop)])
    assert tuple(st1) == tuple(st2) and np.array_equal(s1, s2) and np.array_equal(a1, a2)
    # (3) the host-fitted polynomial (q8='numpy' forces it) sees the cropped cloud too (advisor, round 3): same result as
    # cropping first and augmenting the cropped cloud without the pre-crop
    (st3, a3, s3), = sim.augment_batch([road], "unused", bd, particles=tl, orders=[order], return_src=True, calib=narrow, pre_crop=True,
                                       planes=[(w_crop, h_crop)], q8="numpy")
    kept = np.where(seen)[0]
    (st4, a4, s4), = sim.augment_batch([road[seen]], "unused", bd, particles=tl, orders=[order], return_src=True, calib=narrow,
                                       planes=[(w_crop, h_crop)], q8="numpy")
    assert tuple(st3) == tuple(st4) and np.array_equal(s3, kept[s4]) and np.array_equal(a3, a4)


@pytest.mark.parametrize("lanes", ["2", "3"])
def test_pipeline_compute_lanes_give_the_same_rows(tables, monkeypatch, lanes):
    """SNOWGPU_PIPE_LANES > 1: chunk c of a pipelined host batch computes on lane c mod L -- a sub-context with its own stream,
    events and scratch on the root's tables -- so that the launch chains of consecutive chunks overlap.  Same bytes as the
    one-chunk call, device prepass and caller polynomials, ragged frames."""
    from lidar_snow_sim_amd import engine
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    rng = np.random.default_rng(13)
    frames = []
    for f in range(9):
        full = synthetic_sweep(64, 2048, seed=1200 + f, intensity="lambert").reshape(64, 2048, 5)
        frames.append(np.ascontiguousarray(full[:, f % 5::int(rng.integers(24, 50)), :].reshape(-1, 5)))
    rows = np.concatenate(frames)
    off = np.concatenate(([0], np.cumsum([f.shape[0] for f in frames]))).astype(np.int64)
    tl = _tables64(tables)
    bd = float(np.degrees(3e-3))
    planes = [[0.0, 0.0, -1.0, -1.7]] * len(frames)
    polys = [[1e-4 * f, 0.01, 2.0] for f in range(len(frames))]
    monkeypatch.setenv("SNOWGPU_PIPE_LANES", lanes)
    e = engine.Engine(0)
    try:
        tids = [e.table_ids_from_arrays(tl, list(rng.permutation(64))) for _ in frames]
        for kw in (dict(plane=planes), dict(thr_poly=polys)):
            e.ctx.set_pipeline(0)
            one = [np.copy(x) if x is not None else None for x in e.ctx.augment_batch(rows, off, tids, bd, **kw)[:4]]
            e.ctx.set_pipeline(3000)                             # ~8 chunks over the lanes
            many = e.ctx.augment_batch(rows, off, tids, bd, **kw)
            assert np.array_equal(one[2], many[2]) and np.array_equal(one[3], many[3])
            for f in range(len(frames)):
                a, n = int(off[f]), int(one[2][f])
                assert one[0][a:a + n].tobytes() == many[0][a:a + n].tobytes() and np.array_equal(one[1][a:a + n], many[1][a:a + n])
    finally:
        e.ctx.close()


def test_device_fov_crop_against_the_float32_numpy_projection(eng, tables):
    """Advisor finding: OpenPCDet's lidar_to_rect / rect_to_img run in the cloud's dtype (float32 GEMMs for STF clouds), the
    device crop in float64 with a fixed operation order.  The two can only disagree for points whose image lies within float32
    rounding of the picture's edge or of depth 0: count them on 260 k points spread over the whole sweep, and check that every
    disagreement sits within 1e-3 px / 1e-4 m of a boundary (parity with the reference itself is unpinned anyway: SURVEY 8 c)."""
    from lidar_snow_sim_amd.calibration import Calibration, get_fov_flag
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    from lidar_snow_sim_amd.tools.snowfall.simulation import augment_batch
    cal = Calibration(P2=np.array([[721.5377, 0, 609.5593, 44.85728], [0, 721.5377, 172.854, 0.2163791], [0, 0, 1, 0.002745884]], np.float32),
                      R0=np.array([[0.9999239, 0.00983776, -0.007445048], [-0.009869795, 0.9999421, -0.004278459],
                                   [0.007402527, 0.004351614, 0.9999631]], np.float32),
                      V2C=np.array([[0.007533745, -0.9999714, -0.000616602, -0.004069766], [0.01480249, 0.0007280733, -0.9998902, -0.07631618],
                                    [0.9998621, 0.00752379, 0.01480755, -0.2717806]], np.float32))
    frames = [synthetic_sweep(64, 2048, seed=1800 + f, intensity="lambert") for f in range(2)]
    poly = [0.0, 0.0, -1.0]                                               # keep every row: only the crop decides
    kw = dict(particles=_tables64(tables), orders=[list(range(64))] * 2, thr_polys=[poly, poly], return_src=True)
    res = augment_batch(frames, "unused", float(np.degrees(3e-3)), calib=cal, **kw)
    everything = augment_batch(frames, "unused", float(np.degrees(3e-3)), **kw)          # no crop: the label of every row
    n_diff = 0
    for pc, (st, aug, src), (_, aug_all, src_all) in zip(frames, res, everything):
        assert aug_all.shape[0] == pc.shape[0]
        kept = np.zeros(pc.shape[0], bool)
        kept[src] = True
        unmoved = np.ones(pc.shape[0], bool)
        unmoved[src_all[aug_all[:, 4] == 2]] = False                     # scattered points are cropped at their NEW position
        ref = get_fov_flag(cal.lidar_to_rect(pc[:, 0:3]), (1024, 1920), cal)          # float32 all the way, as in the reference
        diff = np.where((kept != ref) & unmoved)[0]
        n_diff += diff.size
        if diff.size:
            x64 = pc[diff, :3].astype(np.float64)
            rect = np.hstack((x64, np.ones((diff.size, 1)))) @ (cal.V2C.astype(np.float64).T @ cal.R0.astype(np.float64).T)
            hom = np.hstack((rect, np.ones((diff.size, 1)))) @ cal.P2.astype(np.float64).T
            u, v, depth = hom[:, 0] / hom[:, 2], hom[:, 1] / hom[:, 2], hom[:, 2] - float(cal.P2[2, 3])
            edge = np.minimum.reduce([np.abs(u), np.abs(u - 1920), np.abs(v), np.abs(v - 1024)])
            assert np.all((edge < 1e-3) | (np.abs(depth) < 1e-4)), (edge.max(), np.abs(depth).min())
    assert n_diff <= 20          # a handful of boundary points in 262 144, if any


def test_firing_order_rows_give_the_rows_of_the_channel_major_sweep(eng, so, tables):
    """An STF .bin holds its rows in firing order -- azimuth-major, the 64 channels interleaved (precompute.py:78) --, for which the
    channel sort (simulation.py:447) is a real permutation: the device makes a sorted copy and every later kernel reads that.  The
    stable sort puts the rows of one channel in azimuth order whichever way they came, so the augmented rows equal those of the
    channel-major sweep byte for byte and the source indices map through the reordering; a batch may mix both kinds of frame (each
    frame decides for itself), ragged and shuffled ones included.  Float32 and float64; checked against the oracle too."""
    from lidar_snow_sim_amd.synthetic import synthetic_sweep, firing_order
    tl = _tables64(tables)
    bd = float(np.degrees(3e-3))
    for dtype in (np.float32, np.float64):
        full = synthetic_sweep(64, 2048, seed=1021, intensity="lambert").reshape(64, 2048, 5)
        cm = full[:, ::8, :].reshape(-1, 5).astype(dtype)                    # channel-major, 64 x 256
        fire = firing_order(cm, 64, 256)
        to_cm = np.arange(64 * 256).reshape(64, 256).T.reshape(-1)           # row of `fire` -> row of `cm`
        rng = np.random.default_rng(3)
        shuf_idx = rng.permutation(cm.shape[0])
        shuf = cm[shuf_idx]                                                  # no order at all (the stable sort keeps arrival order per channel)
        ragged = fire[: 64 * 100 + 17]                                       # a frame that ends inside an azimuth step
        frames = [cm, fire, shuf, ragged, cm[:5000], fire[:3000]]
        orders = [list(np.random.default_rng(40 + i).permutation(64)) for i in range(len(frames))]
        planes = [[*PLANE[0], PLANE[1]]] * len(frames)
        tids = [eng.table_ids_from_arrays(tl, o) for o in orders]
        # every frame alone ...
        singles = [eng.ctx.augment_batch(f, [0, f.shape[0]], [tids[i]], bd, plane=[planes[i]]) for i, f in enumerate(frames)]
        # ... and all of them in one batch (sorted and unsorted frames side by side)
        rows = np.concatenate(frames)
        off = np.concatenate(([0], np.cumsum([f.shape[0] for f in frames])))
        out, src, counts, stats, _ = eng.ctx.augment_batch(rows, off, tids, bd, plane=planes)
        for i, f in enumerate(frames):
            o1, s1, c1, st1, _ = singles[i]
            a, m = int(off[i]), int(counts[i])
            assert m == int(c1[0]) and np.array_equal(stats[i], st1[0])
            assert np.array_equal(src[a:a + m], s1[:m]) and out[a:a + m].tobytes() == o1[:m].tobytes()
            r_stats, r_aug, r_src = so.augment(f, tl, bd, orders[i], plane=PLANE)
            assert tuple(int(v) for v in stats[i]) == tuple(int(v) for v in r_stats)
            assert np.array_equal(s1[:m], r_src) and np.array_equal(o1[:m, 3:], r_aug[:, 3:])
            assert np.allclose(o1[:m, :3], r_aug[:, :3], rtol=1e-6 if dtype == np.float32 else 1e-12, atol=0)
        # same table order for the channel-major and the firing-order copy of one sweep: same rows, indices mapped
        o_cm, s_cm, c_cm, st_cm, _ = eng.ctx.augment_batch(cm, [0, cm.shape[0]], [tids[0]], bd, plane=[planes[0]])
        o_f, s_f, c_f, st_f, _ = eng.ctx.augment_batch(fire, [0, fire.shape[0]], [tids[0]], bd, plane=[planes[0]])
        n = int(c_cm[0])
        assert n == int(c_f[0]) and np.array_equal(st_cm, st_f) and o_cm[:n].tobytes() == o_f[:n].tobytes()
        assert np.array_equal(to_cm[s_f[:n]], s_cm[:n])
        assert int(st_cm[0][0]) > 0 and n < cm.shape[0]


@pytest.mark.parametrize("switch,values", [("SNOWGPU_SERIAL", ("1",)), ("SNOWGPU_FEW", ("0", "1", "3")), ("SNOWGPU_TIER_ROWS", ("0", "1"))])
def test_remaining_environment_switches_change_no_byte(so, tables, monkeypatch, switch, values):
    """The environment switches that survive in csrc/ (INTEGRATION.md lists them) pick a schedule or a kernel variant, never a result:
    SNOWGPU_SERIAL=1 (every kernel on the caller's stream: pure kernel times for the profiles), SNOWGPU_FEW=0..3 (beams with up to that
    many flakes go through the register kernel k_power_few; default 2), SNOWGPU_TIER_ROWS=0 / 1 (the later capacity tiers as row
    kernels: default only for batches up to four sweeps).  Same bytes as the default on stretched sweeps (long lists: every tier busy),
    float32 in firing order and float64 channel-major, and equal to the oracle."""
    from lidar_snow_sim_amd import engine
    from lidar_snow_sim_amd.synthetic import firing_order
    tl = _tables64(tables)
    bd = float(np.degrees(3e-3))
    order = list(np.random.default_rng(9).permutation(64))
    pc32 = firing_order(_stretched_subsweep(step=16, seed=1090), 64, 128)
    pc64 = _stretched_subsweep(step=16, seed=1091).astype(np.float64)
    results = []
    for v in (None,) + tuple(values):
        if v is None:
            monkeypatch.delenv(switch, raising=False)
        else:
            monkeypatch.setenv(switch, v)
        e = engine.Engine(0)
        try:
            tid = [e.table_ids_from_arrays(tl, order)]
            results.append([e.ctx.augment_batch(pc, [0, pc.shape[0]], tid, bd, plane=[[*PLANE[0], PLANE[1]]]) for pc in (pc32, pc64)])
        finally:
            e.ctx.close()
    for k, pc in enumerate((pc32, pc64)):
        o0, s0, c0, st0, _ = results[0][k]
        n = int(c0[0])
        for res in results[1:]:
            o1, s1, c1, st1, _ = res[k]
            assert np.array_equal(c0, c1) and np.array_equal(st0, st1) and np.array_equal(s0[:n], s1[:n]) and o0[:n].tobytes() == o1[:n].tobytes()
        r_stats, r_aug, r_src = so.augment(pc, tl, bd, order, plane=PLANE)
        assert tuple(int(v) for v in st0[0]) == tuple(int(v) for v in r_stats)
        assert np.array_equal(s0[:n], r_src) and np.array_equal(o0[:n, 3:], r_aug[:, 3:])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_packed_result_transfer_fills_the_callers_buffers_with_the_same_bytes(tables, dtype):
    """snowgpu_set_result_transfer(ctx, 1, threads): per kept row only its source row | label and its intensity (and the moved coordinates of
    scattered rows) cross the link, host threads of the library copy the rest from the caller's input rows.  The caller's out_rows / out_src /
    counts / statistics must hold the bytes of the default transfer: stretched sweeps (many scattered rows), firing order, ragged and empty
    frames, channels without a laser (their column 4 keeps the channel value, Q5), device prepass and caller polynomials, out_src = NULL,
    one and several host threads, an error in a middle chunk."""
    from lidar_snow_sim_amd import _native, engine
    from lidar_snow_sim_amd.synthetic import firing_order
    rng = np.random.default_rng(21)
    frames = []
    for f in range(9):
        pc = _stretched_subsweep(step=int(rng.integers(24, 48)), seed=1200 + f)
        n_az = pc.shape[0] // 64
        if f % 3 == 1:
            pc = firing_order(pc, 64, n_az)
        if f % 4 == 2:
            pc = pc[: pc.shape[0] - 37]
            pc[5::53, 4] = 70.0                              # no such laser: copied through with its channel value
        frames.append(pc.astype(dtype))
    frames[5] = np.zeros((0, 5), dtype)
    F = len(frames)
    tl = _tables64(tables)
    bd = float(np.degrees(3e-3))
    polys = [[1e-4 * f, 0.01, 2.0] for f in range(F)]
    rows = np.concatenate(frames)
    off = np.concatenate(([0], np.cumsum([f.shape[0] for f in frames]))).astype(np.int64)
    e = engine.Engine(0)
    try:
        tids = [e.table_ids_from_arrays(tl, list(rng.permutation(64))) for _ in range(F)]

        def run(mode, threads, **kw):
            e.ctx.set_result_transfer(mode, threads)
            e.ctx.set_pipeline(2500)
            try:
                o, s, c, st, _ = e.ctx.augment_batch(rows, off, tids, bd, **kw)
                if mode == "packed":
                    assert e.ctx.transfer_times()["host_threads"] == threads
                return o.copy(), None if s is None else s.copy(), c.copy(), st.copy()
            finally:
                e.ctx.set_pipeline(3 << 19)
                e.ctx.set_result_transfer("rows")

        n_scattered = 0
        for kw in (dict(thr_poly=polys), dict(thr_poly=polys, want_src=False)):
            ref = run("rows", 0, **kw)
            for threads in (1, 5):
                got = run("packed", threads, **kw)
                assert np.array_equal(ref[2], got[2]) and np.array_equal(ref[3], got[3])
                for f in range(F):
                    a, n = int(off[f]), int(ref[2][f])
                    assert ref[0][a:a + n].tobytes() == got[0][a:a + n].tobytes(), (kw.keys(), threads, f)
                    if ref[1] is not None:
                        assert np.array_equal(ref[1][a:a + n], got[1][a:a + n])
                    n_scattered += int((ref[0][a:a + n, 4] == 2).sum())
                assert (ref[1] is None) == (got[1] is None)
        assert n_scattered > 100 and (ref[0][:, 4] == 70).any()
        # the device prepass (plane given) on frames that all have ground rows
        keep = [f for f in range(F) if frames[f].shape[0] > 0]
        rows2 = np.concatenate([frames[f] for f in keep])
        off2 = np.concatenate(([0], np.cumsum([frames[f].shape[0] for f in keep]))).astype(np.int64)
        tids2 = [tids[f] for f in keep]
        outs = []
        for mode in ("rows", "packed"):
            e.ctx.set_result_transfer(mode, 3)
            e.ctx.set_pipeline(2500)
            try:
                o, s, c, st, _ = e.ctx.augment_batch(rows2, off2, tids2, bd, plane=[[0.0, 0.0, -1.0, -1.7]] * len(keep))
                outs.append((o.copy(), s.copy(), c.copy(), st.copy()))
            finally:
                e.ctx.set_pipeline(3 << 19)
                e.ctx.set_result_transfer("rows")
        assert np.array_equal(outs[0][2], outs[1][2]) and np.array_equal(outs[0][3], outs[1][3])
        for f in range(len(keep)):
            a, n = int(off2[f]), int(outs[0][2][f])
            assert outs[0][0][a:a + n].tobytes() == outs[1][0][a:a + n].tobytes() and np.array_equal(outs[0][1][a:a + n], outs[1][1][a:a + n])
        # an error in a middle chunk still comes back as the reference's exception type, with no thread left behind
        bad = rows.copy()
        bad[int(off[6]) + 3, :3] = (150.0, 0.0, 0.0)
        e.ctx.set_result_transfer("packed", 2)
        e.ctx.set_pipeline(2500)
        with pytest.raises(_native.SnowGPUError) as ei:
            e.ctx.augment_batch(bad, off, tids, bd, thr_poly=polys)
        assert ei.value.code == _native.E_RANGE
        good = run("packed", 2, thr_poly=polys)
        assert np.array_equal(good[2], ref[2])
    finally:
        e.ctx.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_flakes_a_hair_off_tangent_to_a_limit_ray_are_redone_with_the_reference_expression(so, tables, dtype):
    """geometry.py:131-135 decides `distance(flake centre, limit ray) < r` from tangent, root and quotient.  The pass over all rows tests
    |y cos - x sin| < r instead and leaves every test within 1e-12 (|x| + |y|) of equality undecided (csrc/sg_beam.h: sg_near_ray<DEFER>):
    such a beam is sent to the global-list tier, whose scan evaluates the reference's expression.  Here every beam has one flake whose
    centre lies OUTSIDE its wedge at r -+ 3e-13 .. 3e-12 m from a limit ray -- inside the band, far outside the last-bit noise of
    any tangent -- so the flake counts or not by that test alone: the beams must take the global-list tier (device status word 5),
    and occlusion dicts and output rows must be the oracle's."""
    from lidar_snow_sim_amd import engine
    rng = np.random.default_rng(131)
    n = 400
    bd = float(np.degrees(3e-3))
    az = rng.uniform(-np.pi, np.pi, n)
    d = rng.uniform(25.0, 100.0, n)
    el = rng.uniform(-0.2, 0.02, n)
    pc = np.column_stack((d * np.cos(el) * np.cos(az), d * np.cos(el) * np.sin(az), d * np.sin(el), rng.integers(0, 256, n), np.zeros(n))).astype(dtype)
    th = np.arctan2(pc[:, 1], pc[:, 0])                          # simulation.py:91-92 in the row dtype
    th = np.where(th < 0, th + dtype(2 * np.pi), th).astype(np.float64)
    half = np.radians(bd / 2)
    side = rng.integers(0, 2, n)                                 # 0: right limit ray, 1: left
    ray = np.where(side == 0, th - half, th + half)              # :96-101
    ray = np.where(ray < 0, ray + 2 * np.pi, ray)
    ray = np.where(ray > 2 * np.pi, ray - 2 * np.pi, ray)
    rho = rng.uniform(3.0, 15.0, n)
    fr = rng.uniform(1e-3, 4e-3, n)
    gap = np.exp(rng.uniform(np.log(3e-13), np.log(3e-12), n)) * rng.choice([-1.0, 1.0], n)
    L = np.longdouble
    c, s = np.cos(ray.astype(L)), np.sin(ray.astype(L))
    off = (fr.astype(L) + gap.astype(L)) * np.where(side == 0, -1, 1)          # beyond the right ray / beyond the left ray
    flakes = np.column_stack(((rho.astype(L) * c - off * s).astype(np.float64), (rho.astype(L) * s + off * c).astype(np.float64), fr))
    tl = [np.concatenate([tables["t"][0], flakes])] + _tables64(tables)[1:]
    order = list(range(64))
    poly = [0.0, 0.0, 1.0]
    las = so.load_lasers()
    _, _, (c0, k0, r0, q0) = so.process_single_channel(pc, tl[0], bd, las, 0, dump=True)
    _, _, (c1, _, _, _) = so.process_single_channel(pc, tables["t"][0], bd, las, 0, dump=True)
    assert 0.25 * n < int((c0 != c1).sum()) < 0.75 * n          # the test flake counts for about half of the beams: the sign of `gap` decides
    e = engine.Engine(0)
    try:
        tids = e.table_ids_from_arrays(tl, order)
        out, src, counts, stats, _ = e.ctx.augment_batch(pc, [0, n], [tids], bd, thr_poly=[poly])
        st = e.ctx.last_status()
        cnt, rj, ratio, dsrc = e.ctx.debug_occlusions(pc, tids, bd)
    finally:
        e.ctx.close()
    # float32 rows: NumPy's float32 arctan2 and the kernels' differ in the last bit for some rows -- those beams' flakes are not near-tangent
    assert st[5] >= (0.9 if dtype == np.float64 else 0.5) * n, st
    assert np.array_equal(dsrc, np.arange(n)) and np.array_equal(cnt, c0)
    start = np.concatenate(([0], np.cumsum(c0)))
    for i in range(n):
        assert np.array_equal(rj[i, :c0[i]], r0[start[i]:start[i + 1]])
        np.testing.assert_allclose(ratio[i, :c0[i]], q0[start[i]:start[i + 1]], rtol=0, atol=0 if dtype == np.float32 else 1e-12)
    r_stats, r_aug, r_src = so.augment(pc, tl, bd, order, thr_poly=np.array(poly))
    m = int(counts[0])
    assert tuple(int(v) for v in stats[0]) == tuple(int(v) for v in r_stats)
    assert np.array_equal(src[:m], r_src) and np.array_equal(out[:m, 3:], r_aug[:, 3:])
    np.testing.assert_allclose(out[:m, :3], r_aug[:, :3], rtol=1e-6 if dtype == np.float32 else 1e-12, atol=0)
