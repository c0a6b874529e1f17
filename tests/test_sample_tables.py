"""The offline particle-table generator (lidar_snow_sim_amd/sample_tables.py) against the reference's own tools/snowfall/sampling.py
`__main__` block (:360-413), captured in tests/golden/L10_table_plan.npz by tests/golden/make_golden_tables.py: the 6400 file names in
order, skip-if-exists, and one R_0 = 80 m table bit for bit; plus that the frame-stream's lookup finds what the generator wrote."""
import numpy as np
import pytest

from conftest import GOLDEN


@pytest.fixture(scope="module")
def l10():
    return np.load(GOLDEN / "L10_table_plan.npz")


def test_plan_names_every_table_the_reference_writes_in_its_order(l10):
    from lidar_snow_sim_amd import sample_tables as st
    runs = st.rate_pairs()
    assert runs.shape == (50, 2) and np.array_equal(runs, l10["runs"])                    # sampling.py:384-397, bit for bit
    assert float(l10["r0"]) == st.R_0
    items = st.plan()
    names = [st.table_name(d, rate, ratio, line) for _, _, d, rate, ratio, line in items]
    assert len(names) == int(l10["n_names"]) == 2 * 50 * 64
    assert names[:8] == list(l10["first_names"]) and names[-8:] == list(l10["last_names"])
    prefixes = list(dict.fromkeys(n.rsplit("_", 1)[0] for n in names))
    assert prefixes == list(l10["prefixes"])                                              # 80 distinct of the 100 (mode, pair): ratios repeat
    # the prefix augment() looks up (precompute.py:101, pointcloud_viewer.py:2802) is one of them
    from lidar_snow_sim_amd.tools.snowfall.sampling import compute_occupancy, snowfall_rate_to_rainfall_rate
    occ, rate = compute_occupancy(2.5, 1.6), snowfall_rate_to_rainfall_rate(2.5, 1.6)
    assert f"gunn_{rate}_{occ}" in prefixes


def test_serial_host_mode_writes_the_reference_table_and_skips_it_the_second_time(l10, tmp_path, capsys):
    """`--rng serial --seed 42` restricted to the table the fixture sampled (sparsest pair, gunn, line 1): the first table of a run sees
    a fresh default_rng(42), as in the reference's do_in_parallel (:341-357) -- same file name, same 40 112 rows bit for bit; a second
    run skips it (:346-347) and leaves the file alone."""
    from lidar_snow_sim_amd import sample_tables as st
    rc = st.main(["--out", str(tmp_path), "--modes", "gunn", "--pairs", "0.5", "2.0", "--lines", "1", "1", "--seed", "42"])
    assert rc == 0
    name = str(l10["table_name"])
    f = tmp_path / f"{name}.npy"
    assert [p.name for p in tmp_path.iterdir()] == [f.name]
    t = np.load(f)
    assert t.dtype == np.float64 and tuple(t.shape) == tuple(l10["table_shape"])
    assert np.array_equal(t[:32], l10["table_head"]) and np.array_equal(t[-32:], l10["table_tail"])
    assert np.array_equal(t.sum(axis=0), l10["table_sum"])
    before = f.stat().st_mtime_ns
    capsys.readouterr()
    rep = st.generate(tmp_path, ["gunn"], [l10["table_pair"]], range(1, 2))
    assert rep["written"] == 0 and rep["skipped"] == 1 and f.stat().st_mtime_ns == before
    assert f"{name} skipped" in capsys.readouterr().out


def test_per_table_mode_is_order_and_process_independent_and_resumable(tmp_path):
    """`--rng per-table`: every table from its own default_rng([seed, mode, pair, line]) -- a whole run, a run in two processes and a run
    resumed after half of the files were deleted all leave the same bytes (small R_0: seconds)."""
    from lidar_snow_sim_amd import sample_tables as st
    runs = st.rate_pairs()[[0, 49]]
    a, b = tmp_path / "a", tmp_path / "b"
    kw = dict(rng="per-table", seed=7, r0=6.0, verbose=False)
    ra = st.generate(a, ["gunn", "sekhon"], runs, range(1, 4), **kw)
    assert ra["written"] == 12 and ra["skipped"] == 0
    rb = st.generate(b, ["gunn", "sekhon"], runs, range(1, 4), jobs=2, **kw)
    assert rb["written"] == 12 and sorted(rb["names"]) == sorted(ra["names"])
    for i, n in enumerate(sorted(ra["names"])):
        if i % 2:
            (b / f"{n}.npy").unlink()
    rc = st.generate(b, ["sekhon", "gunn"][::-1], runs, range(1, 4), **kw)
    assert rc["written"] == 6 and rc["skipped"] == 6
    tabs = {}
    for n in ra["names"]:
        ta, tb = np.load(a / f"{n}.npy"), np.load(b / f"{n}.npy")
        assert ta.shape[0] > 5 and ta.shape[1] == 3 and ta.tobytes() == tb.tobytes()
        tabs[n] = ta
    assert len({t.tobytes() for t in tabs.values()}) == 12                                # twelve different tables


def test_the_augmentation_lookup_finds_the_generated_tree(tmp_path):
    """simulation.py:324-329: <root_path>/training/snowflakes/npy/<prefix>_<line>.npy -- `--root-path` writes there, and the engine's
    path rule (engine.file_table_id) resolves exactly those files."""
    from lidar_snow_sim_amd import sample_tables as st
    from lidar_snow_sim_amd.tools.snowfall.sampling import compute_occupancy, snowfall_rate_to_rainfall_rate
    rc = st.main(["--root-path", str(tmp_path), "--modes", "gunn", "--pairs", "2.5", "1.6", "--lines", "1", "2", "--rng", "per-table", "--r0", "5", "--quiet"])
    assert rc == 0
    occ, rate = compute_occupancy(2.5, 1.6), snowfall_rate_to_rainfall_rate(2.5, 1.6)
    d = tmp_path / "training" / "snowflakes" / "npy"
    assert sorted(p.name for p in d.iterdir()) == [f"gunn_{rate}_{occ}_1.npy", f"gunn_{rate}_{occ}_2.npy"]
