"""CPU oracle for the snowfall / wet-ground hot path of SysCV/LiDAR_snow_sim.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module; the product package ``lidar_snow_sim_amd`` never does.

Parity status: PINNED -- every function here is checked against golden vectors made by importing the
reference in the build container (``tests/golden/make_golden.py``; fixtures under ``tests/golden/``).

Layout
------
* per-beam physics (``process_single_channel`` / ``get_occlusions`` / ``compute_occlusion_dict`` /
  ``geometry.*`` / ``received_power``): plain C in ``snow_oracle.c`` (scalar, glibc libm), bound here
  with ctypes;
* frame driver, noise-threshold prepass, wet-ground model, sampling helpers and ``dart_throwing``:
  NumPy restatements below.

Citations are file:line in the reference checkout (``tools/snowfall/simulation.py`` = ``sim``,
``tools/wet_ground/augmentation.py`` = ``wet``, ``tools/snowfall/sampling.py`` = ``smp``).

Deliberate, documented differences from the reference ("canonical order"):
* the channel sort is *stable* (the reference's ``argsort`` is an unstable introsort, sim:447, so
  its within-channel order is implementation-defined); fixtures carry a source-index column and are
  compared in canonical order;
* the ground plane is an explicit input (``plane=(w, h)``); ``plane=None`` applies the flat-earth
  fallback the reference takes with every scikit-learn >= 1.2 (wet_ground/planes.py:43-48);
* particle tables and the channel permutation ``order`` are explicit inputs instead of files and
  the global ``random`` state (sim:324-329, :482-486).
"""
from __future__ import annotations

import ctypes
import json
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_BUILD = _HERE / "_build"
_LIB = None

PI = np.pi
C_LIGHT = 299792458.0  # scipy.constants.speed_of_light, sim:17


class _Item(ctypes.Structure):      # so_item (snow_oracle.c): one run of beams of one (frame, channel)
    pass


class _Laser(ctypes.Structure):
    _fields_ = [("channel", ctypes.c_int32), ("min_intensity", ctypes.c_int32),
                ("max_intensity", ctypes.c_int32), ("focal_slope", ctypes.c_double),
                ("focal_offset", ctypes.c_double)]


_Item._fields_ = [("is_f32", ctypes.c_int32), ("rc", ctypes.c_int32), ("pts_in", ctypes.c_void_p), ("pts_out", ctypes.c_void_p),
                  ("M", ctypes.c_int64), ("table_xyr", ctypes.c_void_p), ("K", ctypes.c_int64), ("beam_div_deg", ctypes.c_double),
                  ("las", _Laser), ("diff_sum", ctypes.c_double)]


def build(force: bool = False) -> Path:
    """Compile snow_oracle.c with gcc into oracle/_build/libsnow_oracle.so."""
    _BUILD.mkdir(exist_ok=True)
    so = _BUILD / "libsnow_oracle.so"
    src = _HERE / "snow_oracle.c"
    if force or not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        cmd = ["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-pthread",
               "-o", str(so), str(src), "-lm"]
        subprocess.check_call(cmd)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        lib = ctypes.CDLL(str(build()))
        i64, dbl = ctypes.c_int64, ctypes.c_double
        p = ctypes.c_void_p
        for name in ("so_process_channel_f32", "so_process_channel_f64"):
            fn = getattr(lib, name)
            fn.restype = ctypes.c_int
            fn.argtypes = [p, i64, p, i64, dbl, ctypes.POINTER(_Laser), p, p, p, p, p, p, p, i64, p]
        lib.so_process_items_mt.restype = ctypes.c_int
        lib.so_process_items_mt.argtypes = [p, i64, p, ctypes.c_int]
        lib.so_get_occlusions.restype = ctypes.c_int
        lib.so_get_occlusions.argtypes = [p, p, i64, p, i64, dbl, p, p, p, p, i64, p, p]
        lib.so_occlusion_dict.restype = ctypes.c_int
        lib.so_occlusion_dict.argtypes = [dbl, dbl, p, i64, dbl, dbl, p, p, p]
        lib.so_flake_table.restype = ctypes.c_int
        lib.so_flake_table.argtypes = [p, i64, p]
        _LIB = lib
    return _LIB


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


# ---------------------------------------------------------------------------------------------------
# data

def load_lasers(path=None):
    """64 x {focal_distance, focal_slope, min_intensity} of calib/20171102_64E_S3.yaml (sim:72-76)."""
    if path is None:
        path = _HERE.parent / "lidar_snow_sim_amd" / "data" / "hdl64e_s3_lasers.json"
    d = json.loads(Path(path).read_text())
    return [{"focal_distance": fd, "focal_slope": fs, **({} if mi is None else {"min_intensity": mi})}
            for fd, fs, mi in zip(d["focal_distance"], d["focal_slope"], d["min_intensity"])]


def range_grid():
    """R of sim:106-116: linspace(0, 120 + c*tau_h, 1230) rounded to 2 decimals (Q3)."""
    lidar_range, per_m, tau_h = 120, 10, 1e-8
    m_ext = int(np.ceil(lidar_range * per_m + C_LIGHT * tau_h * per_m))
    return np.round(np.linspace(0, lidar_range + C_LIGHT * tau_h, m_ext), len(str(per_m)))


def _laser_struct(lasers, channel):
    info = lasers[channel]
    focal_distance = info["focal_distance"] * 100            # sim:74
    las = _Laser()
    las.channel = channel
    las.min_intensity = int(info.get("min_intensity", 0))     # sim:72
    las.max_intensity = 230 if (channel % 64) in (53, 55, 56, 58) else 255   # sim:123-126 (laser tables beyond 64
    #                                 entries are the 64-entry one tiled, SURVEY 8 d: the rule tiles with them)
    las.focal_slope = float(info["focal_slope"])              # sim:75
    las.focal_offset = (1 - focal_distance / 13100) ** 2      # sim:76
    return las


# ---------------------------------------------------------------------------------------------------
# per-channel / per-beam physics (C)

def process_single_channel(pc_channel, table, beam_divergence, lasers, channel, dump=False):
    """sim:50-194 on the rows of ONE channel.  Returns (intensity_diff_sum, pc_out[, dump])."""
    pc_channel = np.ascontiguousarray(pc_channel)
    if pc_channel.dtype not in (np.float32, np.float64):
        pc_channel = pc_channel.astype(np.float64)
    table = np.ascontiguousarray(table, dtype=np.float64)
    m, k = pc_channel.shape[0], table.shape[0]
    assert pc_channel.shape[1] == 5
    out = np.empty_like(pc_channel)
    diff = ctypes.c_double(0.0)
    las = _laser_struct(lasers, channel)
    grid = range_grid()
    cnt = keys = rj = ratio = None
    used = ctypes.c_int64(0)
    cap = 0
    if dump:
        cap = m * 64
        cnt = np.zeros(m, np.int64)
        keys = np.zeros(cap, np.int64)
        rj = np.zeros(cap, np.float64)
        ratio = np.zeros(cap, np.float64)
    fn = _lib().so_process_channel_f32 if pc_channel.dtype == np.float32 else _lib().so_process_channel_f64
    rc = fn(_ptr(pc_channel), m, _ptr(table), k, float(beam_divergence), ctypes.byref(las), _ptr(grid),
            _ptr(out), ctypes.byref(diff), _ptr(cnt), _ptr(keys), _ptr(rj), _ptr(ratio), cap,
            ctypes.byref(used))
    if rc == -2:
        raise IndexError("range >= 120 m: index out of bounds for the 1230-bin grid (sim:149)")
    if rc != 0:
        raise MemoryError("snow_oracle.c allocation failure")
    if dump:
        u = used.value
        return diff.value, out, (cnt, keys[:u], rj[:u], ratio[:u])
    return diff.value, out


def get_occlusions(beam_angles, ranges, table, beam_divergence):
    """sim:298-424.  Returns (count per beam, keys, r_j, ratio, n_intersecting per beam), flattened."""
    beam_angles = np.ascontiguousarray(beam_angles, dtype=np.float64)
    ranges = np.ascontiguousarray(ranges, dtype=np.float64)
    table = np.ascontiguousarray(table, dtype=np.float64)
    m = beam_angles.shape[0]
    cap = m * 64 + 64
    cnt = np.zeros(m, np.int64)
    nint = np.zeros(m, np.int64)
    keys = np.zeros(cap, np.int64)
    rj = np.zeros(cap)
    ratio = np.zeros(cap)
    used = ctypes.c_int64(0)
    rc = _lib().so_get_occlusions(_ptr(beam_angles), _ptr(ranges), m, _ptr(table), table.shape[0],
                                  float(beam_divergence), _ptr(cnt), _ptr(keys), _ptr(rj), _ptr(ratio),
                                  cap, ctypes.byref(used), _ptr(nint))
    assert rc == 0
    u = used.value
    return cnt, keys[:u], rj[:u], ratio[:u], nint


def compute_occlusion_dict(beam_angles, intervals, current_range, beam_divergence):
    """sim:231-295.  Returns the dict as an ordered list of (key, r_j, ratio)."""
    intervals = np.ascontiguousarray(intervals, dtype=np.float64).reshape(-1, 3)
    n = intervals.shape[0]
    keys = np.zeros(n + 1, np.int64)
    rj = np.zeros(n + 1)
    ratio = np.zeros(n + 1)
    cnt = _lib().so_occlusion_dict(float(beam_angles[0]), float(beam_angles[1]), _ptr(intervals), n,
                                   float(current_range), float(beam_divergence),
                                   _ptr(keys), _ptr(rj), _ptr(ratio))
    return [(int(keys[i]), float(rj[i]), float(ratio[i])) for i in range(cnt)]


def flake_table(table):
    """Per-flake (rho, phi, tangent angle right, tangent angle left, bad) -- geometry.py:32-80, :138-190."""
    table = np.ascontiguousarray(table, dtype=np.float64)
    out = np.zeros((table.shape[0], 5))
    assert _lib().so_flake_table(_ptr(table), table.shape[0], _ptr(out)) == 0
    return out


def xsi(r):
    """sim:553-569 (float64)."""
    if r <= 0.9:
        return 0.0
    if r >= 1.0:
        return 1.0
    m = (1 - 0) / (1.0 - 0.9)
    b = 0 - (m * 0.9)
    return m * r + b


def received_power(ca_p0, beta_0, ratio, r, r_j, tau_h):
    """sim:547-551."""
    return ((ca_p0 * beta_0 * ratio * xsi(r_j)) / (r_j ** 2)) * np.sin((PI * (r - r_j)) / (C_LIGHT * tau_h)) ** 2


# ---------------------------------------------------------------------------------------------------
# ground plane + noise-threshold prepass

FLAT_EARTH = ([0, 0, 1], -1.55)   # wet_ground/planes.py:29-32, :43-48


def ground_crop_count(pc):
    """Rows in the front-of-car crop used for plane fitting (wet_ground/planes.py:21-27)."""
    m = ((pc[:, 2] < -1.55) & (pc[:, 2] > -1.86 - 0.01 * pc[:, 0]) & (pc[:, 0] > 10) & (pc[:, 0] < 70)
         & (pc[:, 1] > -3) & (pc[:, 1] < 3))
    return int(m.sum())


def estimate_laser_parameters(ground_pc, angle, power_factor=15, noise_floor=0.7):
    """wet:195-266, 'linear' mode, debug off.  Returns (relative_output_intensity, adaptive_noise_threshold)."""
    from scipy.stats import linregress
    norm_int = ground_pc[:, 3] / np.cos(angle)                          # wet:207
    dist = np.linalg.norm(ground_pc[:, :3], axis=1)                     # wet:208
    if len(norm_int) < 3:                                               # wet:213-214
        return None, None
    reg = linregress(dist, norm_int)                                    # wet:216
    p = [reg[0], reg[1]]
    rel_out = power_factor * (p[0] * dist + p[1])                       # wet:221
    hist, xedges, yedges = np.histogram2d(dist, norm_int, bins=(50, 2555),
                                          range=((10, 70), (5, np.abs(np.max(norm_int)))))   # wet:232-233
    hist[np.where(hist == 0)] = len(ground_pc)                          # wet:234-235
    # wet:236 writes np.argpartition(hist, 2, axis=1)[:, 0].  With kth = 2 NumPy's portable introselect
    # falls into its selection-sort branch, which leaves the FIRST minimum of each row at position 0 --
    # i.e. argmin.  (NumPy's AVX2/AVX-512 x86-simd-sort path returns a different one of the three smallest
    # bins: quirk Q8; the pinned 'portable' flavour is the argmin one.)
    ymins = np.argmin(hist, axis=1)
    min_vals = yedges[ymins]                                            # wet:237
    idx = np.where(min_vals > 5)                                        # wet:238
    min_vals = min_vals[idx]
    x = (xedges[idx] + xedges[idx[0] + 1]) / 2                          # wet:240-241
    if len(min_vals) > 3:                                               # wet:248-251
        pmin = linregress(x, min_vals)
    else:
        pmin = p
    thr = noise_floor * (pmin[0] * dist + pmin[1])                      # wet:252-253
    return rel_out, thr


def noise_threshold_poly(pc_sorted, w, h, noise_floor=0.7):
    """sim:450-467: quadratic (p0, p1, p2) of the per-point noise threshold over range."""
    w = np.asarray(w)
    hog = np.matmul(pc_sorted[:, :3], w) + h
    ground = np.logical_and(hog < 0.5, hog > -0.5)                      # sim:450-451
    g = pc_sorted[ground]
    angle = np.arccos(np.divide(np.matmul(g[:, :3], w),
                                np.linalg.norm(g[:, :3], axis=1) * np.linalg.norm(w)))    # sim:454-455
    _, thr = estimate_laser_parameters(g, angle, noise_floor=noise_floor)                 # sim:457-460
    if thr is None:
        raise TypeError("fewer than 3 ground points (sim:462 multiplies None, Q7)")
    thr = thr * np.cos(angle)                                           # sim:462
    gd = np.linalg.norm(g[:, :3], axis=1)                               # sim:464
    return np.polyfit(gd, thr, 2)                                       # sim:467


# ---------------------------------------------------------------------------------------------------
# frame driver

def augment(pc, tables, beam_divergence, order, noise_floor=0.7, plane=None, lasers=None,
            thr_poly=None, return_full=False, threads=1):
    """sim:427-544 with only_camera_fov=False.

    pc      : N x 5 (x, y, z, intensity, channel), float32 or float64
    tables  : sequence of K_i x 3 float64 arrays; table ``order[channel]`` feeds ``channel`` (sim:70, :78)
    order   : permutation of range(num_lasers) (sim:482-486)
    threads : > 1 runs the channels on a thread pool, as the reference does (sim:498 ThreadPool(cpu_count()));
              the C calls release the GIL.  Results do not depend on it.
    Returns (stats, aug_pc, src) -- ``src[i]`` is the row of ``pc`` that produced output row i.
    """
    lasers = load_lasers() if lasers is None else lasers
    num_channels = len(lasers)
    pc = np.asarray(pc)
    perm = np.argsort(pc[:, 4], kind="stable")                          # sim:447 (canonical: stable)
    pcs = pc[perm]
    if thr_poly is None:
        w, h = FLAT_EARTH if plane is None else plane
        thr_poly = noise_threshold_poly(pcs, w, h, noise_floor)
    distances = np.linalg.norm(pcs[:, :3], axis=1)                      # sim:465
    thr = thr_poly[0] * distances ** 2 + thr_poly[1] * distances + thr_poly[2]   # sim:469
    aug = pcs.copy()                                                    # sim:472
    diff_sum = 0

    def one_channel(ch):
        mask = pcs[:, 4] == ch                                          # sim:80
        return mask, process_single_channel(pcs[mask], tables[order[ch]], beam_divergence, lasers, ch)

    if threads > 1:
        from multiprocessing.pool import ThreadPool
        with ThreadPool(threads) as pool:                               # sim:498-504
            done = pool.map(one_channel, range(num_channels))
    else:
        done = map(one_channel, range(num_channels))                    # sim:488-514
    for mask, (d, out) in done:                                         # channel order: the reference sums in this order (sim:510)
        diff_sum += d
        aug[mask] = out
    aug[:, 3] = np.round(aug[:, 3])                                     # sim:516
    keep = np.logical_or(aug[:, 4] == 2, aug[:, 3] > thr)               # sim:518-520
    num_removed = int(np.logical_not(keep).sum())                       # sim:522
    full = aug
    aug = aug[keep]                                                     # sim:523
    src = perm[keep]
    num_att = int((aug[:, 4] == 1).sum())                               # sim:525
    avg = int(diff_sum / num_att) if num_att > 0 else 0                 # sim:527-530
    stats = (num_att, num_removed, avg)
    if return_full:
        return stats, aug, src, dict(full=full, keep=keep, perm=perm, thr=thr, thr_poly=np.asarray(thr_poly),
                                     diff_sum=diff_sum)
    return stats, aug, src


def augment_many(frames, tables, beam_divergence, orders, noise_floor=0.7, planes=None, lasers=None, thr_polys=None,
                 threads=None, beams_per_item=256):
    """augment() (sim:427-544, only_camera_fov=False) for several frames with the per-beam work spread over `threads` host
    threads by snow_oracle.c's pthread driver: work item = a run of `beams_per_item` beams of one (frame, channel).  Same
    results as one augment() call per frame (beams are independent, and the per-channel intensity-difference sums are sums of
    multiples of 0.5: exact in any order).  Returns ([(stats, aug_pc, src), ...], threads used)."""
    import os
    lasers = load_lasers() if lasers is None else lasers
    num_channels = len(lasers)
    threads = (os.cpu_count() or 1) if threads is None else int(threads)
    grid = range_grid()
    prepared, items, keep_alive = [], [], []
    tabs = [np.ascontiguousarray(t, dtype=np.float64) for t in tables]
    for f, pc in enumerate(frames):
        pc = np.asarray(pc)
        if pc.dtype not in (np.float32, np.float64):
            pc = pc.astype(np.float64)
        perm = np.argsort(pc[:, 4], kind="stable")                      # sim:447 (canonical: stable)
        pcs = np.ascontiguousarray(pc[perm][:, :5])
        poly = None if thr_polys is None else thr_polys[f]
        if poly is None:
            w, h = FLAT_EARTH if planes is None else planes[f]
            poly = noise_threshold_poly(pcs, w, h, noise_floor)
        aug = pcs.copy()
        spans = []
        for ch in range(num_channels):
            rows = np.where(pcs[:, 4] == ch)[0]                         # sim:80; contiguous: pcs is channel-sorted
            if rows.size == 0:
                continue
            lo, hi = int(rows[0]), int(rows[-1]) + 1
            tab = tabs[orders[f][ch]]
            for a in range(lo, hi, beams_per_item):
                b = min(a + beams_per_item, hi)
                it = _Item()
                it.is_f32 = 1 if pcs.dtype == np.float32 else 0
                it.pts_in = pcs[a:b].ctypes.data
                it.pts_out = aug[a:b].ctypes.data
                it.M = b - a
                it.table_xyr = tab.ctypes.data
                it.K = tab.shape[0]
                it.beam_div_deg = float(beam_divergence)
                it.las = _laser_struct(lasers, ch)
                items.append(it)
                spans.append(len(items) - 1)
        keep_alive.append((pcs, aug))
        prepared.append((perm, pcs, aug, np.asarray(poly, np.float64), spans))
    arr = (_Item * len(items))(*items)
    used = _lib().so_process_items_mt(ctypes.byref(arr), len(items), _ptr(grid), threads) if items else 1
    if used < 1:
        raise MemoryError("snow_oracle.c thread pool failure")
    results = []
    for perm, pcs, aug, poly, spans in prepared:
        diff_sum = 0.0
        for i in spans:                                                 # item order = channel order, beams in order (sim:510)
            if arr[i].rc == -2:
                raise IndexError("range >= 120 m: index out of bounds for the 1230-bin grid (sim:149)")
            if arr[i].rc != 0:
                raise MemoryError("snow_oracle.c allocation failure")
            diff_sum += arr[i].diff_sum
        distances = np.linalg.norm(pcs[:, :3], axis=1)                  # sim:465
        thr = poly[0] * distances ** 2 + poly[1] * distances + poly[2]  # sim:469
        aug[:, 3] = np.round(aug[:, 3])                                 # sim:516
        keep = np.logical_or(aug[:, 4] == 2, aug[:, 3] > thr)           # sim:518-520
        num_removed = int(np.logical_not(keep).sum())                   # sim:522
        out = aug[keep]
        num_att = int((out[:, 4] == 1).sum())                           # sim:525
        avg = int(diff_sum / num_att) if num_att > 0 else 0             # sim:527-530
        results.append(((num_att, num_removed, avg), out, perm[keep]))
    return results, used


# ---------------------------------------------------------------------------------------------------
# camera-FOV crop (sim:39-47, :532-540).  PARITY UNPINNED: the reference projects with pcdet's calibration_kitti from
# an un-vendored submodule and a calibration file that is not in its tree (SURVEY 8 c); this is the textbook KITTI
# projection (Tr_velo_to_cam, R0_rect, P2) in float64 with a fixed operation order, which the HIP path restates.

def fov_flag(xyz, v2c, r0, p2, img_shape=(1024, 1920)):
    """get_fov_flag(calib.lidar_to_rect(xyz), img_shape, calib): bool mask of points whose image lies inside the picture
    and in front of the camera.  v2c 3 x 4, r0 3 x 3, p2 3 x 4."""
    v2c, r0, p2 = (np.asarray(m, np.float64) for m in (v2c, r0, p2))
    m = np.zeros((4, 3))                                                # lidar_to_rect: [x y z 1] . (V2C^T . R0^T)
    for i in range(4):
        for j in range(3):
            acc = 0.0
            for k in range(3):
                acc = acc + v2c[k, i] * r0[j, k]
            m[i, j] = acc
    x, y, z = (np.asarray(xyz[:, c], np.float64) for c in range(3))
    rect = [((x * m[0, j] + y * m[1, j]) + z * m[2, j]) + m[3, j] for j in range(3)]
    hom = [((rect[0] * p2[j, 0] + rect[1] * p2[j, 1]) + rect[2] * p2[j, 2]) + p2[j, 3] for j in range(3)]   # rect_to_img
    with np.errstate(divide="ignore", invalid="ignore"):
        u, w = hom[0] / rect[2], hom[1] / rect[2]                       # OpenPCDet calibration_kitti.rect_to_img: by the rectified z
    depth = hom[2] - p2[2, 3]
    return (u >= 0) & (u < img_shape[1]) & (w >= 0) & (w < img_shape[0]) & (depth >= 0)        # sim:42-45


# ---------------------------------------------------------------------------------------------------
# wet ground (wet:25-161; wet_ground/phy_equations.py:35-108)

def fresnel_power(ain, n_in, n_out):
    """phy_equations.py:35-67 -> (rs, ts, rp, tp, aout)."""
    a = np.clip(np.sin(ain) * n_in / n_out, -1, 1)
    aout = np.arcsin(a)
    frac = np.cos(ain) * n_in / n_out / np.cos(aout)
    rs = (n_in * np.cos(ain) - n_out * np.cos(aout)) / (n_in * np.cos(ain) + n_out * np.cos(aout))
    ts = 2 * n_in * np.cos(ain) / (n_in * np.cos(ain) + n_out * np.cos(aout))
    rp = (n_out * np.cos(ain) - n_in * np.cos(aout)) / (n_out * np.cos(ain) + n_in * np.cos(aout))
    tp = 2 * n_in * np.cos(ain) / (n_out * np.cos(ain) + n_in * np.cos(aout))
    return rs ** 2, ts ** 2 / frac, rp ** 2, tp ** 2 / frac, aout


def total_transmittance(ain, rho, nair=1.0003, nw=1.33):
    """phy_equations.py:70-108 -> (rs, ts, rp, tp, aaout)."""
    ras, tas, rap, tap, aaout = fresnel_power(ain, nair, nw)            # :81
    rws, tws, rwp, twp, _ = fresnel_power(aaout, nw, nair)              # :83
    ts = tas * rho * tws / (1 - rho * rws)                              # :86
    tp = tap * rho * twp / (1 - rho * rwp)                              # :89
    return ras, ts, rap, tp, aaout


def ground_water_augmentation(pointcloud, water_height=0.001, pavement_depth=0.0012, noise_floor=0.7,
                              power_factor=15, flat_earth=False, delta=0.5, replace=True, plane=None,
                              return_src=False):
    """wet:25-161, estimation_method='linear', debug off.  Returns float64 N' x C array."""
    w, h = FLAT_EARTH if plane is None else plane
    w = np.asarray(w)
    n = pointcloud.shape[0]
    hog = np.matmul(pointcloud[:, :3], w)
    ground = np.logical_and(hog + h < delta, hog + h > -delta)          # wet:46-47
    gidx = np.where(ground)[0]
    planes = np.hstack((pointcloud[ground, :], hog.reshape(-1, 1)[ground]))   # wet:50
    if planes.shape[0] < 1000:                                          # wet:51-52
        return (pointcloud, np.arange(n)) if return_src else pointcloud
    if not flat_earth:                                                  # wet:56-58
        angle = np.arccos(np.divide(np.matmul(planes[:, :3], w),
                                    np.linalg.norm(planes[:, :3], axis=1) * np.linalg.norm(w)))
    else:                                                               # wet:61-63
        angle = np.arccos(-np.divide(np.matmul(planes[:, :3], np.asarray([0, 0, 1])),
                                     np.linalg.norm(planes[:, :3], axis=1) * np.linalg.norm([0, 0, 1])))
    rel_out, thr = estimate_laser_parameters(planes, angle, power_factor=power_factor, noise_floor=noise_floor)
    refl = planes[:, 3] / np.cos(angle) / rel_out                       # wet:90
    _, ts, _, tp, _ = total_transmittance(angle, rho=np.clip(refl, 0.05, 1))   # wet:108-109
    t = np.maximum(tp, ts)                                              # wet:119
    f = np.clip(water_height / pavement_depth, 0, 1)                    # wet:122
    tw = (1 - f) * refl + f * t / angle                                 # wet:123
    new_i = np.clip(rel_out * np.cos(angle) * tw, 0, planes[:, 3])      # wet:126-127
    new_i[new_i < (thr * np.cos(angle))] = 0                            # wet:128, :131
    keep = np.where(new_i > thr * np.cos(angle))                        # wet:146-147
    planes = planes[:, :pointcloud.shape[1]]                            # wet:148 (5 columns in the reference)
    n_ng = n - gidx.shape[0]
    out = np.zeros((n_ng + keep[0].shape[0], pointcloud.shape[1]))      # wet:150
    out[:n_ng, :] = pointcloud[np.logical_not(ground), :]               # wet:151
    out[n_ng:, :] = planes[keep]                                        # wet:152
    out[n_ng:, 3] = new_i[keep]                                         # wet:153
    if replace:
        out[:, 4] = 0                                                   # wet:155-156
    out[n_ng:, 4] = 1                                                   # wet:159
    if return_src:
        src = np.concatenate((np.where(np.logical_not(ground))[0], gidx[keep]))
        return out, src
    return out


# ---------------------------------------------------------------------------------------------------
# sampling helpers (smp:23-87) and the dart thrower (smp:90-194)

def compute_occupancy(snowfall_rate, terminal_velocity, snow_density=0.1):
    return (1.0 * snowfall_rate) / ((3.6 * 10 ** 6) * (snow_density * terminal_velocity))   # smp:30-32


def rainfall_rate_to_snowfall_rate(rainfall_rate, terminal_velocity, snowflake_density=0.1,
                                   snowflake_diameter=0.003):
    return 487 * snowflake_density * snowflake_diameter * terminal_velocity * (rainfall_rate ** (2 / 3))   # smp:50


def snowfall_rate_to_rainfall_rate(snowfall_rate, terminal_velocity, snowflake_density=0.1,
                                   snowflake_diameter=0.003):
    return np.sqrt((snowfall_rate / (487 * snowflake_density * snowflake_diameter * terminal_velocity)) ** 3)   # smp:67


def sekhon_srivastava(rate):
    return 22.9 * rate ** -0.45     # smp:78


def gunn_marshall(rate):
    return 25.5 * rate ** -0.48     # smp:87


def dart_throwing(occupancy_ratio, precipitation_rate, r_0, rng, distribution):
    """smp:90-194, literal O(K^2) loop (use small r_0)."""
    if distribution == "sekhon":
        lam = sekhon_srivastava(precipitation_rate)
    elif distribution == "gunn":
        lam = gunn_marshall(precipitation_rate)
    else:
        raise NotImplementedError("Distribution model unknown.")       # smp:113 (Q13)
    scale = 1 / lam
    xs, ys, rs = [], [], []
    sx = np.zeros(0)
    sy = np.zeros(0)
    sr = np.zeros(0)
    area, target = 0.0, occupancy_ratio * PI * r_0 ** 2                 # smp:121-124
    while area < target:                                                # smp:142
        length = np.sqrt(rng.uniform(0, r_0 ** 2))                      # smp:145
        angle = rng.uniform(0, 2) * PI                                  # smp:146
        x = length * np.cos(angle)
        y = length * np.sin(angle)
        diam = np.inf
        while diam > 20:                                                # smp:153-154
            diam = rng.exponential(scale * 10)
        diam = diam / 1000                                              # smp:157
        height = rng.uniform(-diam / 2, diam / 2)                       # smp:160
        radius = np.sqrt((diam / 2) ** 2 - height ** 2)                 # smp:163
        if x ** 2 + y ** 2 <= radius ** 2:                              # smp:166
            continue
        if np.any((sx - x) ** 2 + (sy - y) ** 2 <= (sr + radius) ** 2):   # smp:170-173
            continue
        area += PI * radius ** 2                                        # smp:181-182
        xs.append(x); ys.append(y); rs.append(radius)
        sx = np.append(sx, x); sy = np.append(sy, y); sr = np.append(sr, radius)
    return np.column_stack((sx, sy, sr)) if len(xs) else np.zeros((0, 3))
