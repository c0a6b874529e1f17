"""Wet-ground augmentation and the laser-parameter estimate it shares with the snowfall path.

Mirror of tools/wet_ground/augmentation.py: ``ground_water_augmentation`` (:25-161) and
``estimate_laser_parameters`` (:195-266).  The per-point model (incident angles, Fresnel chain,
intensity rewrite, adaptive noise-threshold drop, [non-ground ; kept ground] reordering) runs in
libsnowgpu.so; this file is argument handling.  `estimate_laser_parameters` is also kept as a host
function because it is part of the reference's importable surface (simulation.py:24).
"""
import numpy as np

from ... import engine as _engine
from ..._native import SnowGPUError as _SnowGPUError
from .planes import calculate_plane


def estimate_laser_parameters(pointcloud_planes, calculated_indicent_angle, power_factor=15, noise_floor=0.7,
                              debug=True, estimation_method='linear', *, q8='first', return_lines=False):
    """(relative_output_intensity, adaptive_noise_threshold, p, stat_values) -- augmentation.py:195-266.

    Host (NumPy/SciPy) version of the estimate, 'linear' mode.  The per-row minimum of the 50 x 2555
    histogram is taken as the FIRST minimum of the row, which is what ``np.argpartition(hist, 2)[:, 0]``
    (:236) returns through NumPy's portable selection code; NumPy's AVX2/AVX-512 builds return a
    different one of the three smallest bins, so the reference's own answer is
    machine-dependent there and this mirror pins the portable one by default (``q8='first'``).
    ``q8='numpy'`` evaluates the reference's expression with THIS process' NumPy instead -- whatever its
    selection code returns on this CPU, i.e. what the reference itself prints here.
    """
    from scipy.stats import linregress
    if estimation_method != 'linear':
        raise NotImplementedError("this host mirror fits estimation_method='linear' only; 'poly' (np.polyfit + a RANSAC over the "
                                  "unseeded global NumPy RNG, augmentation.py:171-192) runs on the device with seeded draws: "
                                  "ground_water_augmentation(..., estimation_method='poly', poly_seed=...)")
    normalized = pointcloud_planes[:, 3] / np.cos(calculated_indicent_angle)       # :207
    distance = np.linalg.norm(pointcloud_planes[:, :3], axis=1)                    # :208
    if len(normalized) < 3:                                                        # :213-214
        return None, None, None, None
    reg = linregress(distance, normalized)                                         # :216
    p = [reg[0], reg[1]]
    stat_values = reg[2:]
    relative_output_intensity = power_factor * (p[0] * distance + p[1])            # :221
    hist, xedges, yedges = np.histogram2d(distance, normalized, bins=(50, 2555),
                                          range=((10, 70), (5, np.abs(np.max(normalized)))))   # :232-233
    hist[hist == 0] = len(pointcloud_planes)                                       # :234-235
    if q8 == 'numpy':
        ymins = np.argpartition(hist, 2)[:, 0]                                     # :236, verbatim: follows the local NumPy build
    elif q8 == 'first':
        ymins = np.argmin(hist, axis=1)                                            # :236 through NumPy's portable selection code
    else:
        raise ValueError("q8 must be 'first' or 'numpy'")
    min_vals = yedges[ymins]                                                       # :237
    sel = np.where(min_vals > 5)[0]                                                # :238
    min_vals = min_vals[sel]
    x = (xedges[sel] + xedges[sel + 1]) / 2                                        # :240-241
    pmin = linregress(x, min_vals) if len(min_vals) > 3 else p                     # :248-251
    adaptive_noise_threshold = noise_floor * (pmin[0] * distance + pmin[1])        # :252-253
    if return_lines:
        return relative_output_intensity, adaptive_noise_threshold, p, stat_values, [pmin[0], pmin[1]]
    return relative_output_intensity, adaptive_noise_threshold, p, stat_values


_tls = None


def _scratch_f64(n):
    """A float64 scratch array of n elements, kept per thread."""
    global _tls
    import threading
    if _tls is None:
        _tls = threading.local()
    buf = getattr(_tls, "buf", None)
    if buf is None or buf.size < n:
        _tls.buf = buf = np.empty(n + n // 8, np.float64)
    return buf[:n]


_pool = None


def _thread_pool(workers):
    """One pool for the life of the process: its threads keep their scratch arrays."""
    global _pool
    from concurrent.futures import ThreadPoolExecutor
    if _pool is None or _pool._max_workers != workers:
        _pool = ThreadPoolExecutor(max_workers=workers)
    return _pool


def _usable_cpus():
    """CPUs this process may keep busy: its affinity mask, cut to the cgroup's CPU quota where one is set (a thread pool wider than the quota
    is throttled as a whole when the quota of the period is spent)."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def _linregress_line(x, y):
    """Slope and intercept of scipy.stats.linregress(x, y) for float64 vectors: xmean = np.mean(x), ymean = np.mean(y), ssxm, ssxym from
    np.cov(x, y, bias=1), slope = ssxym / ssxm, intercept = ymean - slope * xmean -- as the ufunc calls np.mean and np.cov make themselves
    (np.add.reduce / n; the two rows stacked, their means subtracted in place, np.dot(X, X.T.conj()), times np.true_divide(1, n)), without
    the two functions' argument handling: 10 instead of 37 us per frame, all of it under the GIL, and the same bits
    (tests/test_host_logic.py::test_line_fit_of_the_threshold_callback_is_numpy_s_own)."""
    n = x.shape[0]
    xmean, ymean = np.add.reduce(x) / n, np.add.reduce(y) / n                       # np.mean: umr_sum, true_divide by the count
    X = np.empty((2, n))
    X[0] = x
    X[1] = y                                                                        # np.cov: array(m, ndmin=2), concatenate((X, y))
    avg = np.add.reduce(X, axis=1)
    avg /= n                                                                        # np.average(X, axis=1) = X.mean(1)
    X -= avg[:, None]
    c = np.dot(X, X.T.conj())
    c *= np.true_divide(1, n)                                                       # bias=1: ddof = 0, fact = n
    slope = c[0, 1] / c[0, 0]
    return slope, ymean - slope * xmean


def noise_polys_from_device_stats(hist, rec, noise_floor=0.7, threads=None):
    """Threshold polynomials (n_frames x 3) for q8='numpy' from the device half of the prepass (Context.prepass_stats): the
    histogram's row minima are taken HERE, with this process' NumPy -- np.argpartition(hist, 2)[:, 0] on float64 rows of 2555
    bins whose empty bins hold the ground-row count, exactly the reference's expression (augmentation.py:232-236) -- and the
    rest of estimate_laser_parameters (:237-253) and the quadratic fit (simulation.py:462-467) follow from the device's sums.
    The row selections of a batch run on a thread pool (NumPy releases the GIL inside argpartition)."""
    import os
    nf = hist.shape[0]
    xedges = np.linspace(10, 70, 51)                                                # np.histogram2d's edges (:232-233)
    xmid = (xedges[:-1] + xedges[1:]) / 2                                           # :240-241

    n_ground, ymax, p0, p1 = rec[:, 0], rec[:, 4], rec[:, 5], rec[:, 6]
    step = (np.abs(ymax) - 5.0) / 2555.0
    m0, m1 = p0.copy(), p1.copy()                                                   # :250-251: too few rows -> the regression line p

    # :234-236: the device has already put the ground-row count into the empty bins; the histogram becomes the float64 array
    # np.histogram2d returns (one casting copy) and the selection is the reference's expression -- row by row, so any run of rows of the
    # (frames * 50) x 2555 array gives what fifty-row arrays give: the rows are dealt out in equal runs, whatever frame they belong to (24
    # frames on 16 threads: one and a half frames each instead of two for some and none for others).  Both steps release the GIL.
    rows_all = hist.reshape(-1, 2555)
    ymins = np.empty(nf * 50, np.intp)

    def select(lo, hi):
        h = _scratch_f64((hi - lo) * 2555).reshape(-1, 2555)                       # (reused per thread: no page faults after the first call)
        np.copyto(h, rows_all[lo:hi])
        ymins[lo:hi] = np.argpartition(h, 2)[:, 0]

    workers = threads or int(os.environ.get("SNOWGPU_Q8_THREADS", "0")) or min(16, _usable_cpus())
    if nf <= 2 or workers <= 1:
        select(0, nf * 50)
    else:
        per = -(-nf * 50 // workers)
        list(_thread_pool(workers).map(lambda lo: select(lo, min(lo + per, nf * 50)), range(0, nf * 50, per)))
    # the noise line per frame (augmentation.py:237-253), on the COMPRESSED arrays x[use], min_vals[use] with the expressions of
    # scipy.stats.linregress -- np.mean of each, np.cov(x, y, bias=1) -- so that the line is the one the host path
    # (noise_threshold_poly -> estimate_laser_parameters) fits, operation for operation (a masked sum over all 50 bins adds the
    # same numbers in another order: last-bit differences that a row sitting on the threshold can see)
    min_all = ymins.reshape(nf, 50) * step[:, None] + 5.0                           # yedges[ymins] with yedges = np.linspace(5, ymax, 2556) (:237)
    use_all = min_all > 5                                                           # :238
    enough = np.add.reduce(use_all, axis=1) > 3                                     # :248
    for f in range(nf):                                                             # (element by element the expressions of one frame)
        if enough[f]:
            u = use_all[f]
            m0[f], m1[f] = _linregress_line(xmid[u], min_all[f][u])                 # :249 scipy linregress
    q = rec[:, 7:18]
    a22, a21, a2, a11, a1, a2gc, a2c, a1gc, a1c, gc, c = (q[:, k] for k in range(11))
    # normal equations of np.polyfit(d, nf (m0 d + m1) c, 2), columns scaled by their norms as polyfit scales them
    rhs = noise_floor * np.stack([m0 * a2gc + m1 * a2c, m0 * a1gc + m1 * a1c, m0 * gc + m1 * c], axis=1)
    G = np.stack([np.stack([a22, a21, a2], axis=1), np.stack([a21, a11, a1], axis=1), np.stack([a2, a1, n_ground], axis=1)], axis=1)
    sc = np.sqrt(np.stack([a22, a11, n_ground], axis=1))
    sol = np.linalg.solve(G / (sc[:, :, None] * sc[:, None, :]), (rhs / sc)[:, :, None])[:, :, 0]
    return sol / sc


def noise_threshold_poly(pc_sorted, w, h, noise_floor=0.7, q8='first'):
    """Quadratic (p0, p1, p2) of the per-point noise threshold over range -- simulation.py:450-467 (host)."""
    w = np.asarray(w)
    height = np.matmul(pc_sorted[:, :3], w) + h
    ground = np.logical_and(height < 0.5, height > -0.5)
    pc_ground = pc_sorted[ground]
    angle = np.arccos(np.divide(np.matmul(pc_ground[:, :3], w),
                                np.linalg.norm(pc_ground[:, :3], axis=1) * np.linalg.norm(w)))
    _, thr, _, _ = estimate_laser_parameters(pc_ground, angle, noise_floor=noise_floor, debug=False, q8=q8)
    if thr is None:
        raise TypeError("unsupported operand type(s) for *=: 'NoneType' and 'float' "
                        "(fewer than 3 ground points, simulation.py:462)")
    thr = thr * np.cos(angle)
    return np.polyfit(np.linalg.norm(pc_ground[:, :3], axis=1), thr, 2)


def ground_water_augmentation(pointcloud, water_height=0.001, pavement_depth=0.0012, noise_floor=0.7, power_factor=15,
                              estimation_method='linear', flat_earth=False, debug=True, delta=0.5, replace=True,
                              *, plane=None, device=0, return_src=False, q8='first', plane_method='reference', plane_seed=0,
                              plane_trials=1000, poly_seed=0):
    """Drop-in for tools/wet_ground/augmentation.py::ground_water_augmentation (:25-161).

    `debug` is accepted and ignored (the reference's debug branch only draws matplotlib figures).
    Extra keyword-only arguments: plane=(w, h) to skip the plane estimate -- without it calculate_plane (:41) runs on the device by
    `plane_method` ('reference': the flat-earth plane the reference returns today; 'lsq'; 'ransac' seeded with `plane_seed`) --,
    device, return_src, and q8: 'first' (default) fits
    the noise line on the device through the FIRST sparsest histogram bin of every range row; 'numpy' fits the two lines on the
    host with THIS process' NumPy (np.argpartition verbatim, quirk Q8: what the reference itself computes on this machine) and
    hands them to the device, which does everything else.
    estimation_method='poly' (the viewer passes it through: pointcloud_viewer.py:2820, :2851): np.polyfit of degree 2 for the laser
    power (:223-229) and ransac_polyfit (:171-192) for the noise level (:243-246), both on the device.  The reference draws its RANSAC
    samples from NumPy's unseeded global generator, so it differs from run to run; here the draws are Philox numbers keyed by
    `poly_seed` -- same cloud + same seed = same result (parity unpinned; q8='numpy' is a 'linear'-only switch).
    Returns a float64 N' x 5 array (:150); the input object itself when fewer than 1000 ground rows
    exist (:51-52).
    """
    if estimation_method not in ('linear', 'poly'):
        raise ValueError("estimation_method must be 'linear' or 'poly' (augmentation.py:215, :223)")
    if estimation_method == 'poly' and q8 == 'numpy':
        raise ValueError("q8='numpy' fits the two LINES of estimation_method='linear' on the host; 'poly' runs on the device only")
    pc = np.asarray(pointcloud)
    rows = pc if pc.dtype in (np.float32, np.float64) else pc.astype(np.float64)
    eng = _engine.get_engine(device)
    lines = None
    if q8 == 'numpy' and plane is None:           # the two lines are fitted here, so the plane is needed here
        plane = calculate_plane(pc, method=plane_method, seed=plane_seed, trials=plane_trials, device=device)
    if q8 == 'numpy':
        w, h = plane
        # the reference's own steps up to the two fitted lines (augmentation.py:44-76), with the local NumPy
        wv = np.asarray(w)
        hog = np.matmul(rows[:, :3], wv)
        ground = np.logical_and(hog + h < delta, hog + h > -delta)
        planes = np.hstack((rows[ground, :5], hog.reshape((len(hog), 1))[ground]))
        if planes.shape[0] >= 1000:
            if not flat_earth:
                angle = np.arccos(np.divide(np.matmul(planes[:, :3], wv), np.linalg.norm(planes[:, :3], axis=1) * np.linalg.norm(w)))
            else:
                angle = np.arccos(-np.divide(np.matmul(planes[:, :3], np.asarray([0, 0, 1])),
                                             np.linalg.norm(planes[:, :3], axis=1) * np.linalg.norm([0, 0, 1])))
            _, _, p, _, pmin = estimate_laser_parameters(planes, angle, noise_floor=noise_floor, power_factor=power_factor, debug=False,
                                                         q8='numpy', return_lines=True)
            lines = [[p[0], p[1], pmin[0], pmin[1]]]
    elif q8 != 'first':
        raise ValueError("q8 must be 'first' or 'numpy'")
    with eng.batch_lock:
        if plane is None:                         # augmentation.py:41 calculate_plane(pointcloud): on the device
            eng.ctx.set_plane_method(plane_method, seed=plane_seed, trials=plane_trials, min_rows=pc.shape[1])
        if estimation_method == 'poly':
            eng.ctx.set_wet_estimation('poly', poly_seed)
        try:
            out, src, counts, flags = eng.ctx.wet_ground_batch(
                np.ascontiguousarray(rows[:, :5]), [0, rows.shape[0]],
                None if plane is None else [[plane[0][0], plane[0][1], plane[0][2], plane[1]]], water_height,
                pavement_depth, noise_floor, power_factor, flat_earth, delta, replace, lines=lines)
        except _SnowGPUError as e:
            if e.code == 7:                       # np.polyfit / linregress on too few points (augmentation.py:243)
                raise TypeError(str(e)) from None
            raise
        finally:
            if plane is None and plane_method != 'reference':
                eng.ctx.set_plane_method('reference')
            if estimation_method == 'poly':
                eng.ctx.set_wet_estimation('linear')
    if flags[0]:
        return (pointcloud, np.arange(rows.shape[0])) if return_src else pointcloud
    n = int(counts[0])
    return (out[:n], src[:n]) if return_src else out[:n]
