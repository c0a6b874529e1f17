"""Snowfall augmentation of LiDAR sweeps on MI355X.

Drop-in for tools/snowfall/simulation.py::augment (simulation.py:427-544): same name, positional
arguments, defaults, return shape and exception types, so that

    from lidar_snow_sim_amd.tools.snowfall.simulation import augment

replaces ``from tools.snowfall.simulation import augment`` at the reference's call sites
(pointcloud_viewer.py:2807-2810, :2865-2868; tools/snowfall/precompute.py:103-104).

This module is host logic only: argument handling, the channel permutation drawn from Python's global
``random`` exactly as the reference does (:482-486) and particle-table lookup (:78, :324-329).  The
simulation itself -- channel sort, noise-threshold prepass, per-beam occlusion and received-power
integration, noise-floor filter, camera-FOV crop, compaction, statistics -- runs in
libsnowgpu.so (hand-written HIP for gfx950).  There is no CPU fallback.
"""
from __future__ import annotations

import random
from typing import Sequence, Tuple

import numpy as np

from ... import _native
from ... import engine as _engine
from ..wet_ground.augmentation import noise_polys_from_device_stats, noise_threshold_poly

PI = np.pi


def _as_rows(pc) -> np.ndarray:
    pc = np.asarray(pc)
    if pc.ndim != 2 or pc.shape[1] < 5:
        raise ValueError("pc must be N x 5 (x, y, z, intensity, channel)")
    if pc.dtype not in (np.float32, np.float64):
        pc = pc.astype(np.float64)
    return pc


class _LineIds:
    """Device table id per table line (index = line - 1), resolved on first use; a frame's 64 ids are one fancy index."""

    def __init__(self, eng, n_known=64):
        self.eng = eng
        self.ids = np.full(max(int(n_known), 1), -2, np.int32)               # -2: not resolved yet

    def resolve(self, k):
        raise NotImplementedError

    def __getitem__(self, lines):
        lines = np.asarray(lines, np.int64)
        if lines.size and (lines.min() < 0 or lines.max() >= self.ids.shape[0]):
            if lines.min() < 0:
                raise IndexError("negative table line in `order`")
            grown = np.full(int(lines.max()) + 1, -2, np.int32)
            grown[:self.ids.shape[0]] = self.ids
            self.ids = grown
        out = self.ids[lines]
        if (out == -2).any():
            for k in np.unique(lines[out == -2]):
                self.ids[int(k)] = self.resolve(int(k))
            out = self.ids[lines]
        return out


class _LazyFileIds(_LineIds):
    """<prefix>_<line>.npy, loaded on first use only (a frame touches the lines its permutation names).  sample: 'missing' --
    a table without a file is sampled on the device; 'all' -- every table is, the directory is never looked at."""

    def __init__(self, eng, prefix, root_path, sample=None):
        super().__init__(eng)
        self.prefix, self.root_path, self.sample = prefix, root_path, sample

    def resolve(self, k):
        if self.sample == 'all':
            return self.eng.sampled_table_id(self.prefix, k + 1)
        return self.eng.file_table_id(self.prefix, k + 1, self.root_path, sample_missing=self.sample == 'missing')


class _ArrayIds(_LineIds):
    """Caller-owned tables (index = line - 1), uploaded on first use; a line beyond the sequence is the caller's error and says
    so here rather than as a device status later."""

    def __init__(self, eng, particles):
        super().__init__(eng, len(particles))
        self.particles = particles

    def resolve(self, k):
        if k >= len(self.particles):
            raise IndexError(f"order names line {k + 1}, but only {len(self.particles)} particle tables were given")
        return self.eng.array_table_id(self.particles[k])


def _device_planes(eng, frames, dt, method, seed, trials, ncols):
    """calculate_plane for every frame of a batch, one device call (planes of the host-side polynomial fit)."""
    if method == 'reference':
        from ..wet_ground.planes import flat_earth
        return [flat_earth()] * len(frames)
    off = np.zeros(len(frames) + 1, np.int64)
    off[1:] = np.cumsum([f.shape[0] for f in frames])
    flat = np.concatenate([f[:, :5] for f in frames]).astype(dt, copy=False) if len(frames) else np.zeros((0, 5), dt)
    with eng.batch_lock:
        eng.ctx.set_plane_method(method, seed=seed, trials=trials, min_rows=ncols)
        try:
            pl, _ = eng.ctx.estimate_planes(flat, off)
        finally:
            eng.ctx.set_plane_method('reference')
    return [(pl[i, :3].copy(), float(pl[i, 3])) for i in range(len(frames))]


class FlatBatch:
    """Frames that already lie back to back in ONE N_total x 5 array (ideally page-locked: Context.pinned_empty), with their
    offsets -- what a reader that fills a batch buffer straight from the .bin files hands to augment_batch, which then skips
    its own staging copy (precompute.py:78 np.fromfile -> the upload buffer, no intermediate array)."""

    def __init__(self, rows: np.ndarray, offsets, pinned: bool = True):
        self.rows = rows
        self.offsets = np.ascontiguousarray(offsets, np.int64)
        self.pinned = pinned
        if rows.ndim != 2 or rows.shape[1] != 5 or rows.dtype not in (np.float32, np.float64) or not rows.flags.c_contiguous:
            raise ValueError("a FlatBatch is a C-contiguous float32 / float64 N x 5 array")
        if self.offsets[0] != 0 or self.offsets[-1] > rows.shape[0] or np.any(np.diff(self.offsets) < 0):
            raise ValueError("bad frame offsets")

    def __len__(self):
        return len(self.offsets) - 1

    def frame(self, i):
        return self.rows[int(self.offsets[i]):int(self.offsets[i + 1])]


def _raise_like_reference(err: _native.SnowGPUError):
    """Map library status codes onto the exception types the reference raises (SURVEY 8 b, 'Errors')."""
    if err.code == _native.E_RANGE:
        raise IndexError(str(err)) from err            # simulation.py:149 (quirk Q6)
    if err.code == _native.E_GROUND:
        raise TypeError(str(err)) from err             # simulation.py:462 on None (quirk Q7)
    if err.code == _native.E_TABLE:
        raise ValueError(str(err)) from err
    raise err


def augment_batch(frames: Sequence[np.ndarray], particle_file_prefix: str, beam_divergence: float, shuffle: bool = True,
                  noise_floor: float = 0.7, root_path: str = None, *, planes=None, orders=None, particles=None,
                  thr_polys=None, device: int = 0, return_src: bool = False, device_prepass: bool = True, slot: int = 0,
                  calib=None, pre_crop: bool = False, q8: str = 'first', plane_method: str = 'reference', plane_seed: int = 0,
                  plane_trials: int = 1000, **device_kw):
    """augment() for a list of frames in one launch sequence -- the throughput entry point.

    frames      sequence of N_i x 5 arrays (one dtype for the whole batch); further columns are carried through
    planes      optional per-frame (w, h); default (None): calculate_plane (simulation.py:449) runs ON THE DEVICE for every frame,
                by `plane_method`: 'reference' (default) = what the reference returns today, the flat-earth plane
                ([0, 0, 1], -1.55) (its RANSAC call raises with scikit-learn >= 1.2, planes.py:35-48); 'lsq' = least squares over
                the strip of planes.py:21-27; 'ransac' = RANSAC seeded with `plane_seed` (`plane_trials` trials).  With pre_crop
                the plane comes from the cropped cloud, as in precompute.py:96-104
    orders      optional per-frame channel permutations; default: range(64), shuffled with the global
                `random` module when shuffle=True, one draw per frame in frame order
    particles   optional sequence of K x 3 tables (index = line - 1) instead of <prefix>_<line>.npy files; or 'device': every table
                is sampled and filed ON THE DEVICE from the prefix alone (mode, rain rate, occupancy; R_0 = 80 m; seed = hash of
                (prefix, line)) -- statistically the reference's tables (sampling.py:90-194), not bit for bit, no file read;
                or 'missing': files where they exist, the device sampler where they do not
    thr_polys   optional per-frame (p0, p1, p2) noise-threshold polynomials (skips the prepass)
    calib       optional calibration (V2C, R0, P2): the camera-FOV crop of simulation.py:532-540 is then applied inside the
                compaction kernels, (1024, 1920) image; num_removed counts the cropped rows (:538)
    pre_crop    with calib: also crop every frame to the camera's view BEFORE it is augmented, on the device, as
                tools/snowfall/precompute.py:96-99 does on the host
    q8          'first' (default): the noise-threshold fit takes the FIRST minimum of every histogram row -- what
                np.argpartition(hist, 2)[:, 0] (wet_ground/augmentation.py:236) returns through NumPy's portable selection
                code, on the device.  'numpy': the fit is made on the host with THIS process' NumPy, so the result follows
                whatever the reference itself would compute on this machine (AVX2 / AVX-512 builds of NumPy pick a different
                one of the three smallest bins); everything else still runs on the device
    Returns a list of (stats, aug_pc) -- or (stats, aug_pc, src) with return_src=True.

    torch CUDA tensors (a list of N_i x 5 tensors, an F x N x 5 tensor, or a lidar_snow_sim_amd.tensors.DeviceBatch) take the
    device-resident boundary instead: rows are read where they lie, the call runs on torch's current stream, aug_pc / src come back
    as device tensors and no row crosses the link (lidar_snow_sim_amd/tensors.py, which also documents sync=False and wet=...).
    """
    from ... import tensors as _tensors
    if _tensors.is_device_input(frames):
        if not device_prepass:
            raise ValueError("device_prepass=False fits the threshold on the host: it needs host arrays, not CUDA tensors")
        return _tensors.augment_batch(frames, particle_file_prefix, beam_divergence, shuffle=shuffle, noise_floor=noise_floor,
                                      root_path=root_path, planes=planes, orders=orders, particles=particles, thr_polys=thr_polys,
                                      device=None, return_src=return_src, slot=slot, calib=calib, pre_crop=pre_crop, q8=q8,
                                      plane_method=plane_method, plane_seed=plane_seed, plane_trials=plane_trials, **device_kw)
    if device_kw:
        raise TypeError(f"{sorted(device_kw)}: arguments of the torch-tensor boundary only")
    eng = _engine.get_engine(device, slot)
    flat_in = frames if isinstance(frames, FlatBatch) else None
    rows = [flat_in.frame(i) for i in range(len(flat_in))] if flat_in is not None else [_as_rows(f) for f in frames]
    if not rows:
        return []
    # columns beyond the fifth ride through untouched, as in the reference (it indexes whole rows, simulation.py:447, :508-523)
    extra = any(r.shape[1] > 5 for r in rows)
    want_src = return_src or extra
    dt = rows[0].dtype
    if any(r.dtype != dt for r in rows):
        raise TypeError("all frames of a batch must share one dtype")
    if q8 not in ('first', 'numpy'):
        raise ValueError("q8 must be 'first' or 'numpy'")
    # q8='numpy': the histogram's row minima must come from THIS process' NumPy.  The device still makes everything that is
    # expensive (ground rows, I / cos, regression line, histogram, the sums of the quadratic fit: Context.prepass_stats) unless the
    # caller asked for the host prepass or the batch needs the device pre-crop (whose cropped rows never visit the host).
    q8_device = q8 == 'numpy' and device_prepass and thr_polys is None and not (calib is not None and pre_crop)
    if q8 == 'numpy':
        device_prepass = False
    if plane_method not in _native.PLANE_METHODS:
        raise ValueError("plane_method must be 'reference', 'lsq' or 'ransac'")
    nl = eng.n_lasers
    ncols = rows[0].shape[1]
    host_fit = thr_polys is None and not device_prepass and not q8_device      # the polynomial is fitted here, from the rows
    fit_rows = rows
    if host_fit and calib is not None and pre_crop:
        # precompute.py:96-104 hands augment() the CROPPED cloud: plane and polynomial are fitted on it
        from ...calibration import get_fov_flag
        fit_rows = [r[get_fov_flag(calib.lidar_to_rect(r[:, 0:3]), (1024, 1920), calib)] for r in rows]
    if host_fit and planes is None:
        planes = _device_planes(eng, fit_rows, dt, plane_method, plane_seed, plane_trials, ncols)
    table_ids, polys, plane_rows = [], [], []
    ids_by_line = None                                                      # device table id of line - 1, looked up once per batch
    for i, r in enumerate(rows):
        if orders is not None:
            order = list(orders[i])
        else:
            order = list(range(nl))                                         # simulation.py:483
            if shuffle:
                random.shuffle(order)                                       # simulation.py:485-486
        if ids_by_line is None:
            if isinstance(particles, str):
                if particles not in ('device', 'missing'):
                    raise ValueError("particles must be a sequence of tables, 'device' or 'missing'")
                ids_by_line = _LazyFileIds(eng, particle_file_prefix, root_path, sample='all' if particles == 'device' else 'missing')
            else:
                ids_by_line = _ArrayIds(eng, particles) if particles is not None else _LazyFileIds(eng, particle_file_prefix, root_path)
        table_ids.append(ids_by_line[order[:nl]])                           # channel c reads line order[c] + 1 (simulation.py:78)
        if thr_polys is not None:
            polys.append(np.asarray(thr_polys[i], np.float64))
        elif planes is not None:
            w, h = planes[i]                                                 # simulation.py:449
            plane_rows.append([float(w[0]), float(w[1]), float(w[2]), float(h)])
            if host_fit:
                fr = fit_rows[i]
                srt = fr[np.argsort(fr[:, 4], kind="stable")]
                polys.append(noise_threshold_poly(srt[:, :5], w, h, noise_floor, q8=q8))
    offsets = np.zeros(len(rows) + 1, np.int64)
    offsets[1:] = np.cumsum([r.shape[0] for r in rows])
    with eng.batch_lock:
        if flat_in is not None:
            flat = flat_in.rows[:int(offsets[-1])]                               # the caller's (page-locked) batch buffer as it is
        else:
            flat = eng.staging_in(int(offsets[-1]), dt)                          # page-locked: PCIe speed, no page faults
            if len(rows) > 1:
                np.concatenate([r[:, :5] for r in rows], out=flat)
            else:
                flat[...] = rows[0][:, :5]
        resident = False
        if q8_device:
            # q8='numpy': the library hands the device half of the prepass (histograms, sums) to this callback group by group WHILE the
            # per-beam kernels of the group run (snowgpu_set_threshold_callback); the row minima are taken here, with this process' NumPy
            # (np.argpartition verbatim, quirk Q8), and the polynomials go back for the compaction.  One crossing of the rows, one call.
            eng.ctx.set_threshold_callback(lambda first, hist, rec: noise_polys_from_device_stats(hist, rec, noise_floor))
        out_rows, out_src = eng.result_buffers(int(offsets[-1]), dt)
        # The device counting sort handles integer channel values 0..255 and reports anything else
        # (SNOWGPU_E_CHANNELS); only then is the batch sorted here and run again with the permutation.
        perm = None
        crop_idx = None
        if calib is not None:
            eng.ctx.set_fov(calib, (1024, 1920), pre_crop=pre_crop)              # simulation.py:536
        device_plane = not polys and not plane_rows                              # neither given: calculate_plane on the device
        if device_plane:
            eng.ctx.set_plane_method(plane_method, seed=plane_seed, trials=plane_trials, min_rows=ncols)
        try:
            for attempt in (0, 1):
                try:
                    out, src, counts, stats, _ = eng.ctx.augment_batch(
                        flat, offsets, table_ids, beam_divergence, thr_poly=np.asarray(polys) if polys else None,
                        plane=None if (polys or device_plane) else np.asarray(plane_rows), noise_floor=noise_floor, perm=perm,
                        out_rows=out_rows, out_src=out_src, want_src=want_src, rows_resident=resident and attempt == 0 and crop_idx is None)
                    break
                except _native.SnowGPUError as err:
                    if attempt == 0 and err.code == _native.E_CHANNELS:
                        if calib is not None and pre_crop:
                            # the device pre-crop takes no caller permutation: crop here (precompute.py:96-99), sort the cropped
                            # rows, run without the device pre-crop and map the source rows back afterwards
                            from ...calibration import get_fov_flag
                            crop_idx = [np.where(get_fov_flag(calib.lidar_to_rect(r[:, 0:3]), (1024, 1920), calib))[0] for r in rows]
                            rows_run = [r[ix] for r, ix in zip(rows, crop_idx)]
                            offsets = np.zeros(len(rows) + 1, np.int64)
                            offsets[1:] = np.cumsum([r.shape[0] for r in rows_run])
                            flat = eng.staging_in(int(offsets[-1]), dt)
                            if int(offsets[-1]):
                                np.concatenate([r[:, :5] for r in rows_run], out=flat)
                            eng.ctx.set_fov(calib, (1024, 1920), pre_crop=False)
                            want_src = True
                        else:
                            rows_run = rows
                        perm = np.concatenate([np.argsort(r[:, 4], kind="stable") for r in rows_run]).astype(np.int32)
                        if q8_device and not polys:
                            # (a caller permutation switches the threshold callback off: fit here, with the local NumPy, from the rows)
                            pls = planes if planes is not None else _device_planes(eng, rows_run, dt, plane_method, plane_seed, plane_trials, ncols)
                            polys = [noise_threshold_poly(r[np.argsort(r[:, 4], kind="stable")][:, :5], pls[i][0], pls[i][1], noise_floor, q8='numpy')
                                     for i, r in enumerate(rows_run)]
                        continue
                    _raise_like_reference(err)
        finally:
            if q8_device:
                eng.ctx.set_threshold_callback(None)
            if calib is not None:
                eng.ctx.set_fov(None)
            if device_plane and plane_method != 'reference':
                eng.ctx.set_plane_method('reference')
    results = []
    for i in range(len(rows)):
        a, n = int(offsets[i]), int(counts[i])
        aug = out[a:a + n]
        src_i = None if src is None else src[a:a + n]
        if crop_idx is not None:
            src_i = crop_idx[i][src_i].astype(np.int32)
        if rows[i].shape[1] > 5:
            aug = np.concatenate((aug, rows[i][src_i, 5:]), axis=1)
        st = (np.int64(stats[i, 0]), np.int64(stats[i, 1]), int(stats[i, 2]))
        results.append((st, aug, src_i) if return_src else (st, aug))
    return results


def augment(pc: np.ndarray, particle_file_prefix: str, beam_divergence: float, shuffle: bool = True,
            show_progressbar: bool = False, only_camera_fov: bool = True, noise_floor: float = 0.7,
            root_path: str = None, *, plane=None, order=None, particles=None, thr_poly=None, calib=None,
            device: int = 0, return_src: bool = False, device_prepass: bool = True, q8: str = 'first',
            plane_method: str = 'reference', plane_seed: int = 0, plane_trials: int = 1000) -> Tuple:
    """
    :param pc:                      N-by-5 array containing original pointcloud (x, y, z, intensity, channel).
    :param particle_file_prefix:    Prefix of the particle tables, f'{mode}_{rain_rate}_{occupancy}'.
    :param beam_divergence:         Beam divergence in degrees.
    :param shuffle:                 Flag if order of sampled snowflakes should be shuffled.
    :param show_progressbar:        Accepted for compatibility; there is nothing to show.
    :param only_camera_fov:         Flag if the camera field of view (FOV) filter should be applied
                                    (needs `calib=`, see lidar_snow_sim_amd.calibration).
    :param noise_floor:             Noise floor threshold.
    :param root_path:               Optional root path of <root>/training/snowflakes/npy.

    :return:                        ((num_attenuated, num_removed, avg_intensity_diff), N'-by-5 array)

    `pc` may also be an N-by-5 torch CUDA tensor: it is read in place and the result is a tensor on the same device
    (lidar_snow_sim_amd/tensors.py) -- the training-time boundary (`root_path`, simulation.py:53).

    Keyword-only extras: plane=(w, h), order=<permutation>, particles=<tables>, thr_poly, calib, device,
    return_src (append the source-row index of every output row to the result), q8 ('first' | 'numpy', see augment_batch),
    plane_method / plane_seed / plane_trials (how calculate_plane, simulation.py:449, runs on the device when plane is None).
    """
    cal = None
    if only_camera_fov:                                                     # simulation.py:532-533
        from ...calibration import get_calib
        cal = get_calib() if calib is None else calib                       # AssertionError if the file is missing (:35)
    res = augment_batch([pc], particle_file_prefix, beam_divergence, shuffle=shuffle, noise_floor=noise_floor,
                        root_path=root_path, planes=None if plane is None else [plane],
                        orders=None if order is None else [order], particles=particles,
                        thr_polys=None if thr_poly is None else [thr_poly], device=device, return_src=return_src,
                        device_prepass=device_prepass, calib=cal, q8=q8, plane_method=plane_method, plane_seed=plane_seed,
                        plane_trials=plane_trials)[0]
    return res
