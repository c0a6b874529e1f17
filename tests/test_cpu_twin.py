"""libsnowcpu.so (include/snowgpu_cpu.h; SURVEY 8 b: "snowgpu_cpu_* twins running the C++ CPU restatement"): the kernels' own per-beam
device code (csrc/sg_beam.h, sg_table_host.h, sg_row.h) compiled for the host and driven by host threads.  No GPU needed: against the
reference's L5 golden fixtures (augment end to end, float32 and float64 rows, shuffled channel permutations), against the oracle on a
quarter sweep at the bench's table density, and that the package itself never loads it (it is a baseline, not a fallback)."""
import re

import numpy as np
import pytest

from conftest import ROOT, canonical


@pytest.fixture(scope="module")
def so():
    from oracle import snow_oracle
    snow_oracle.build()
    return snow_oracle


@pytest.fixture(scope="module")
def twin():
    from lidar_snow_sim_amd import build, _cpu_twin
    build.build_cpu_twin(verbose=False)
    assert b"snowcpu" in _cpu_twin.lib().snowgpu_cpu_version()
    return _cpu_twin


def test_cpu_twin_exports_what_its_header_declares(twin):
    text = (ROOT / "include" / "snowgpu_cpu.h").read_text()
    names = sorted(set(re.findall(r"\b(snowgpu_cpu_[a-z_]+)\s*\(", text)))
    assert names == ["snowgpu_cpu_augment_batch", "snowgpu_cpu_version"]
    for n in names:
        assert hasattr(twin.lib(), n), n
    assert twin.lib().snowgpu_cpu_augment_batch(0, None, None, 0, 0, None, None, None, 0, None, None, None, None, 0.1, None, 0.7, 1, None, None, None, None, None) == 1


def test_cpu_twin_reproduces_the_reference_L5_fixtures(twin, so, golden, tables):
    """All eight L5 cases (tools/snowfall/simulation.py::augment through the imported reference): kept rows, labels, intensities and
    statistics equal the fixture's; the whole result equals the oracle's byte for byte.  The threshold polynomial is the oracle's
    (the twin restates the per-beam path and the frame steps around it, not the prepass)."""
    d = golden("L5_augment")
    tl = [tables["t"][i % 4] for i in range(64)]
    for case in range(8):
        pc, order = d[f"c{case}_pc"], list(d[f"c{case}_order"])
        plane = (d[f"c{case}_plane_w"], float(d[f"c{case}_plane_h"]))
        s0, a0, src0, extra = so.augment(pc, tl, float(d["bd"]), order, plane=plane, return_full=True)
        (st, aug, src), = twin.augment_batch([pc], tl, [order], float(d["bd"]), [extra["thr_poly"]], threads=2)
        assert aug.dtype == pc.dtype and tuple(int(v) for v in st) == tuple(int(v) for v in d[f"c{case}_stats"])
        assert np.array_equal(src, src0) and aug.tobytes() == a0.tobytes()
        a1, s1 = canonical(aug, src)
        a2, s2 = canonical(d[f"c{case}_aug"], d[f"c{case}_src"])
        assert np.array_equal(s1, s2) and np.array_equal(a1[:, 3:], a2[:, 3:])
        np.testing.assert_allclose(a1[:, :3], a2[:, :3], rtol=1e-6 if pc.dtype == np.float32 else 1e-12, atol=0)


def test_cpu_twin_equals_the_oracle_on_a_ragged_batch_at_bench_density(twin, so):
    """Two frames of different sizes (a quarter and an eighth of a 64 x 2048 sweep, ranges stretched so that beams meet up to dozens of
    flakes) with R_0 = 80 m tables of the bench's density, any thread count: the oracle's bytes, statistics included."""
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    from lidar_snow_sim_amd.tools.snowfall import sampling as smp
    occ, rate = smp.compute_occupancy(2.5, 1.6), smp.snowfall_rate_to_rainfall_rate(2.5, 1.6)
    distinct = [smp.dart_throwing(occ, rate, 80.0, np.random.default_rng(42 + i), "gunn") for i in range(2)]
    tl = [distinct[i % 2] for i in range(64)]
    full = synthetic_sweep(64, 2048, seed=1000, intensity="lambert").reshape(64, 2048, 5)
    frames = [np.ascontiguousarray(full[:, ::4].reshape(-1, 5)), np.ascontiguousarray(full[:, 1::8].reshape(-1, 5))]
    r = np.linalg.norm(frames[1][:, :3].astype(np.float64), axis=1)
    frames[1][:, :3] = (frames[1][:, :3] * (np.minimum(r * 1.8, 119.0) / r)[:, None]).astype(np.float32)
    orders = [list(np.random.default_rng(3 + f).permutation(64)) for f in range(2)]
    bd = float(np.degrees(3e-3))
    plane = ([0.0, 0.0, -1.0], -1.7)
    refs = [so.augment(pc, tl, bd, o, plane=plane, return_full=True, threads=4) for pc, o in zip(frames, orders)]
    for threads in (1, 3):
        res = twin.augment_batch(frames, tl, orders, bd, [r[3]["thr_poly"] for r in refs], threads=threads)
        for (st, aug, src), (s0, a0, src0, _) in zip(res, refs):
            assert tuple(int(v) for v in st) == tuple(int(v) for v in s0)
            assert np.array_equal(src, src0) and aug.tobytes() == a0.tobytes()
    assert sum(int((r[1][:, 4] == 2).sum()) for r in refs) > 50


def test_cpu_twin_reports_the_reference_errors(twin, tables):
    tl = [tables["t"][i % 4] for i in range(64)]
    far = np.array([[125.0, 1.0, 0.0, 30.0, 3.0], [10.0, 1.0, -1.0, 30.0, 3.0]], np.float32)
    with pytest.raises(IndexError):                               # simulation.py:149: a simulated point at >= 120 m
        twin.augment_batch([far], tl, [list(range(64))], float(np.degrees(3e-2)), [[0.0, 0.0, 0.0]])


def test_the_package_never_loads_the_cpu_twin():
    """A baseline, not a fallback: only build.py (which compiles it) and _cpu_twin.py (its binding, imported by bench.py and tests) name it."""
    pat = re.compile(r"libsnowcpu|_cpu_twin|snowgpu_cpu_")
    for path in (ROOT / "lidar_snow_sim_amd").rglob("*.py"):
        if path.name in ("build.py", "_cpu_twin.py"):
            continue
        assert not pat.search(path.read_text()), path
    for path in (ROOT / "lidar_snow_sim_amd" / "csrc").glob("*"):
        if path.is_file() and path.name != "snowcpu.cpp" and path.suffix in (".cpp", ".hip", ".h"):
            assert "snowgpu_cpu_" not in path.read_text(), path
