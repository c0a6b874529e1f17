"""ctypes binding of libsnowcpu.so (include/snowgpu_cpu.h): the CPU twin of the augment path -- the kernels' own device code
(csrc/sg_beam.h, sg_table_host.h, sg_row.h) compiled for the host and driven by host threads.

A MEASUREMENT BASELINE AND A PARITY CHECK, NOT A FALLBACK: no module of this package imports this file (tests/test_host_logic.py holds
that); `augment()` / `augment_batch()` raise without a GPU.  bench.py times it next to the GPU (`cpu_twin` in the bench line) and the tests
compare its rows with the oracle's (CPU) and with the HIP path's (GPU).
"""
from __future__ import annotations

import ctypes
from pathlib import Path

import numpy as np

_LIB_PATH = Path(__file__).resolve().parent / "libsnowcpu.so"
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            raise RuntimeError(f"{_LIB_PATH} is missing: run `python -m lidar_snow_sim_amd.build`")
        L = ctypes.CDLL(str(_LIB_PATH))
        vp, dbl, ci = ctypes.c_void_p, ctypes.c_double, ctypes.c_int
        L.snowgpu_cpu_version.restype = ctypes.c_char_p
        L.snowgpu_cpu_augment_batch.restype = ci
        L.snowgpu_cpu_augment_batch.argtypes = [ci, vp, vp, ci, ci, vp, vp, vp, ci, vp, vp, vp, vp, dbl, vp, dbl, ci, vp, vp, vp, vp, vp]
        _lib = L
    return _lib


def augment_batch(frames, tables, orders, beam_divergence, thr_polys, lasers=None, noise_floor=0.7, threads=0):
    """frames: list of N_i x 5 float32 / float64 arrays; tables: sequence of K x 3 float64 tables (index = line - 1); orders: per frame the
    channel permutation (channel c reads tables[orders[f][c]], simulation.py:78); thr_polys: per frame (p0, p1, p2).
    Returns [(stats, aug_pc, src)] like augment_batch(..., return_src=True) of the HIP path."""
    from .engine import laser_constants, load_lasers
    L = lib()
    lasers = load_lasers() if lasers is None else lasers
    fs, fo, mi, ma = laser_constants(lasers)
    nl = len(fs)
    dt = frames[0].dtype
    if dt not in (np.float32, np.float64) or any(f.dtype != dt for f in frames):
        raise TypeError("frames must share one dtype, float32 or float64")
    nf = len(frames)
    off = np.zeros(nf + 1, np.int64)
    off[1:] = np.cumsum([f.shape[0] for f in frames])
    rows = np.ascontiguousarray(np.concatenate([f[:, :5] for f in frames]))
    uniq, index = [], {}
    ids = np.zeros((nf, nl), np.int32)
    for f in range(nf):
        for c in range(nl):
            t = tables[orders[f][c]]
            k = index.get(id(t))
            if k is None:
                k = index[id(t)] = len(uniq)
                uniq.append(np.ascontiguousarray(t, np.float64))
            ids[f, c] = k
    ptrs = (ctypes.c_void_p * len(uniq))(*[t.ctypes.data for t in uniq])
    ks = np.asarray([t.shape[0] for t in uniq], np.int64)
    fs, fo = np.ascontiguousarray(fs, np.float64), np.ascontiguousarray(fo, np.float64)
    mi, ma = np.ascontiguousarray(mi, np.int32), np.ascontiguousarray(ma, np.int32)
    thr = np.ascontiguousarray(thr_polys, np.float64).reshape(nf, 3)
    out = np.empty_like(rows)
    src = np.empty(rows.shape[0], np.int32)
    counts = np.zeros(nf, np.int64)
    stats = np.zeros((nf, 3), np.int64)
    status = np.zeros(2, np.int32)
    p = lambda a: ctypes.c_void_p(a.ctypes.data)   # noqa: E731
    rc = L.snowgpu_cpu_augment_batch(nf, p(off), p(rows), 0 if dt == np.float32 else 1, len(uniq), ctypes.cast(ptrs, ctypes.c_void_p), p(ks), p(ids), nl,
                                     p(fs), p(fo), p(mi), p(ma), float(beam_divergence), p(thr), float(noise_floor), int(threads), p(out), p(src),
                                     p(counts), p(stats), p(status))
    if rc:
        kind = {4: IndexError, 3: ValueError}.get(rc, RuntimeError)
        raise kind(f"libsnowcpu status {rc} at {int(status[1])}")
    res = []
    for f in range(nf):
        a, n = int(off[f]), int(counts[f])
        res.append(((np.int64(stats[f, 0]), np.int64(stats[f, 1]), int(stats[f, 2])), out[a:a + n], src[a:a + n]))
    return res
