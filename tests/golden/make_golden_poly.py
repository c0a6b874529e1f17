#!/usr/bin/env python3
"""Fixture for estimation_method='poly' of the wet-ground model (tests/golden/L9_wet_poly_portable.npz), made by IMPORTING the
reference (/root/reference, build container only) exactly as make_golden.py does: the L6 input clouds go through the reference's
ground_water_augmentation(..., estimation_method='poly') 32 times each, `np.random.seed(k)` before run k (the reference draws its
RANSAC samples from NumPy's global generator, tools/wet_ground/augmentation.py:183 -- unseeded in real use, so the 32 runs show its
run-to-run spread).  Stored per case: the deterministic parts (the power quadratic p = np.polyfit(dist, I / cos, 2); the inputs x,
min_vals of ransac_polyfit and the fit over all of them) and per run the noise quadratic, the number of output rows and run 0's
output.  (With this container's NumPy 2.2 the branch raises as it stands -- see `ransac` below -- so the fixture flattens one index
array, which is what NumPy < 1.23 did by itself.)  NumPy's SIMD dispatch is switched off (NPY_DISABLE_CPU_FEATURES: the "portable" flavour the parity gate pins, quirk Q8).

    python tests/golden/make_golden_poly.py
"""
import os
import subprocess
import sys
import types
from pathlib import Path

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference")
DISABLE = "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR AVX2 FMA3"

if __name__ == "__main__" and len(sys.argv) == 1:
    env = dict(os.environ, MPLBACKEND="Agg", NPY_DISABLE_CPU_FEATURES=DISABLE)
    subprocess.check_call([sys.executable, __file__, "portable"], env=env)
    sys.exit(0)

os.environ.setdefault("MPLBACKEND", "Agg")
import warnings  # noqa: E402

import numpy as np  # noqa: E402

sys.path.insert(0, str(REF))
for name in ("lib", "lib.OpenPCDet", "lib.OpenPCDet.pcdet", "lib.OpenPCDet.pcdet.utils",
             "lib.OpenPCDet.pcdet.utils.calibration_kitti"):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
sys.modules["lib.OpenPCDet.pcdet.utils"].calibration_kitti = sys.modules["lib.OpenPCDet.pcdet.utils.calibration_kitti"]
import tools.wet_ground.augmentation as wet  # noqa: E402

RUNS = 32


def main():
    l6 = np.load(HERE / "L6_wet_ground_portable.npz")
    n_cases = int(l6["n_cases"])
    out = {"n_cases": np.array(n_cases), "runs": np.array(RUNS), "numpy": np.array(np.__version__)}
    wet.calculate_plane = lambda _pc: (np.asarray([0.0, 0.0, -1.0]), -1.7)
    orig_elp, orig_ransac = wet.estimate_laser_parameters, wet.ransac_polyfit
    cap = {}

    def elp(planes, angle, *a, **kw):
        rel, thr, p, stat = orig_elp(planes, angle, *a, **kw)
        dist = np.linalg.norm(planes[:, :3], axis=1)
        nf = kw.get("noise_floor", 0.7)
        cap["p"] = np.asarray(p, np.float64)
        cap["pmin"] = np.polyfit(dist, thr / nf, 2)            # thr = nf * polyval(pmin, dist) exactly: the fit returns pmin
        cap["n_ground"] = planes.shape[0]
        return rel, thr, p, stat

    def ransac(x, y, **kw):
        # augmentation.py:240-241 builds x as xedges[[array]]: NumPy < 1.23 read the one-element list as a tuple (x is 1-D, as the
        # author meant); NumPy >= 1.23 makes it a fancy index and x comes out (1, m) -- np.polyfit then raises TypeError("expected 1D
        # vector for x"), i.e. TODAY the reference's 'poly' branch raises for every cloud.  The fixture holds
        # the intended computation: x flattened.
        x = np.ravel(x)
        cap["x"], cap["y"] = np.array(x, np.float64), np.array(y, np.float64)
        return orig_ransac(x, y, **kw)

    wet.estimate_laser_parameters, wet.ransac_polyfit = elp, ransac
    for c in range(n_cases):
        pc = l6[f"c{c}_pc"]
        flat, replace = bool(l6[f"c{c}_flat"]), bool(l6[f"c{c}_replace"])
        kw = dict(water_height=0.0008, pavement_depth=0.001, noise_floor=0.7, power_factor=15, estimation_method="poly",
                  flat_earth=flat, debug=False, delta=0.5, replace=replace)
        pmins, rows = [], []
        for k in range(RUNS):
            np.random.seed(k)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")                # np.polyfit's RankWarning on 15 draws with few distinct abscissae
                res = wet.ground_water_augmentation(pc.copy(), **kw)
            pmins.append(cap["pmin"])
            rows.append(res.shape[0])
            if k == 0:
                out[f"c{c}_out0"] = res
        out[f"c{c}_p"] = cap["p"]
        out[f"c{c}_x"], out[f"c{c}_y"] = cap["x"], cap["y"]
        out[f"c{c}_fit_all"] = np.polyfit(cap["x"], cap["y"], 2)
        out[f"c{c}_pmin"] = np.asarray(pmins)
        out[f"c{c}_rows"] = np.asarray(rows, np.int64)
        out[f"c{c}_n_ground"] = np.array(cap["n_ground"])
        distinct = len({tuple(np.round(q, 12)) for q in pmins})
        print(f"  L9 case {c}: {pc.dtype} flat={flat} replace={replace}: {cap['n_ground']} ground rows, {len(cap['x'])} histogram rows, "
              f"rows out {min(rows)}..{max(rows)}, {distinct} distinct noise curves in {RUNS} runs")
    np.savez_compressed(HERE / "L9_wet_poly_portable.npz", **out)
    print("wrote", HERE / "L9_wet_poly_portable.npz")


if __name__ == "__main__":
    main()
