#!/usr/bin/env python3
"""Fixture for the offline table generator (tests/golden/L10_table_plan.npz), made by RUNNING the reference's own
tools/snowfall/sampling.py `__main__` block (:360-413) in the build container with two names patched:

  * tqdm.contrib.concurrent.process_map -> a function that captures `paramlist` (the 2 x 50 x 64 parameter tuples, :399-406) instead of
    sampling 6400 tables;
  * SAVE_DIR -> a temporary directory,

and then calling the reference's `do_in_parallel` (:341-357) on ONE captured tuple -- the sparsest pair, 'gunn', line 1 -- twice: the first
call samples (from the module's default_rng(42), :381, fresh) and writes `<name>.npy`, the second finds the file and skips.

Stored: the 100 (dist, rate, ratio) prefixes in order, the first and last eight file names, the count, and of the sampled table its
name, shape, column sums, first and last 32 rows.  Inputs and outputs only; no line of the reference is kept.

    python tests/golden/make_golden_tables.py
"""
import os
import runpy
import sys
import tempfile
from pathlib import Path

os.environ.setdefault("MPLBACKEND", "Agg")
import numpy as np  # noqa: E402

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference")


def main():
    import tqdm.contrib.concurrent as tcc
    captured = {}

    def capture(fn, params, **kw):
        captured["fn"], captured["params"], captured["kw"] = fn, list(params), kw

    tcc.process_map = capture
    sys.path.insert(0, str(REF))
    g = runpy.run_path(str(REF / "tools" / "snowfall" / "sampling.py"), run_name="__main__")
    params = captured["params"]
    fn = captured["fn"]
    names = [f"{p[0]}_{p[1][0]}_{p[1][1]}_{p[2]}" for p in params]                      # (checked against the file written below)
    prefixes, seen = [], set()
    for p in params:
        key = f"{p[0]}_{p[1][0]}_{p[1][1]}"
        if key not in seen:
            seen.add(key)
            prefixes.append(key)
    # the sparsest pair (last of `runs`), gunn, line 1
    pick = next(p for p in params if p[0] == "gunn" and p[2] == 1 and np.array_equal(p[1], params[-1][1]))
    with tempfile.TemporaryDirectory() as tmp:
        fn.__globals__["SAVE_DIR"] = tmp
        fn(pick)
        files = sorted(os.listdir(tmp))
        assert files == [f"gunn_{pick[1][0]}_{pick[1][1]}_1.npy"], files
        t = np.load(Path(tmp) / files[0])
        mtime = (Path(tmp) / files[0]).stat().st_mtime_ns
        fn(pick)                                                                        # second call: "<name> skipped"
        assert (Path(tmp) / files[0]).stat().st_mtime_ns == mtime
    out = {"n_names": np.array(len(names)), "prefixes": np.array(prefixes), "first_names": np.array(names[:8]), "last_names": np.array(names[-8:]),
           "runs": np.array([p[1] for p in params[:50 * 64:64]], np.float64), "r0": np.array(g["r"]),
           "workers_kw": np.array(sorted(captured["kw"])),
           "table_name": np.array(files[0][:-4]), "table_shape": np.array(t.shape), "table_sum": t.sum(axis=0), "table_head": t[:32], "table_tail": t[-32:],
           "table_pair": np.asarray(pick[1], np.float64), "numpy": np.array(np.__version__)}
    np.savez_compressed(HERE / "L10_table_plan.npz", **out)
    print(f"L10: {len(names)} names, {len(prefixes)} prefixes; sampled {files[0]}: {t.shape[0]} flakes")


if __name__ == "__main__":
    main()
