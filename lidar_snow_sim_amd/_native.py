"""ctypes binding of libsnowgpu.so (include/snowgpu.h).  No compute happens in this file.

The library is hand-written HIP for gfx950; it is built in-tree by ``python -m lidar_snow_sim_amd.build``
(or ``__graft_entry__.build()``).  If it is missing every entry point raises -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes
import os
import threading
from pathlib import Path

import numpy as np

_LIB_PATH = Path(__file__).resolve().parent / "libsnowgpu.so"
_lib = None
_lock = threading.Lock()

E_OK, E_INVALID, E_HIP, E_TABLE, E_RANGE, E_CHANNELS, E_OVERFLOW, E_GROUND, E_NO_DEVICE = range(9)

EXPORTS = [
    "snowgpu_create", "snowgpu_destroy", "snowgpu_last_error", "snowgpu_version", "snowgpu_set_lasers",
    "snowgpu_upload_table", "snowgpu_table_count", "snowgpu_range_grid", "snowgpu_augment_batch",
    "snowgpu_augment_batch_device", "snowgpu_debug_occlusions", "snowgpu_wet_ground_batch",
    "snowgpu_profile_begin", "snowgpu_profile_end", "snowgpu_set_exact_math", "snowgpu_augment_wet_batch",
    "snowgpu_sample_table", "snowgpu_host_alloc", "snowgpu_host_free", "snowgpu_set_fov",
    "snowgpu_augment_wet_batch_device", "snowgpu_last_status", "snowgpu_free_table", "snowgpu_debug_table", "snowgpu_file_table_device", "snowgpu_set_fov_precrop",
    "snowgpu_set_pipeline", "snowgpu_set_wet_lines", "snowgpu_set_plane_method", "snowgpu_estimate_planes",
    "snowgpu_estimate_planes_device", "snowgpu_prepass_stats", "snowgpu_set_wet_estimation", "snowgpu_wet_last_fit", "snowgpu_debug_ransac_polyfit", "snowgpu_set_result_transfer", "snowgpu_debug_transfer_times", "snowgpu_status_error", "snowgpu_set_threshold_callback", "snowgpu_augment_batch_compact", "snowgpu_set_serial", "snowgpu_lane_stream", "snowgpu_device_numa_node",
]

WET_ESTIMATION = {"linear": 0, "poly": 1}

PLANE_METHODS = {"reference": 0, "lsq": 1, "ransac": 2}


class SnowGPUError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libsnowgpu error {code}: {msg}")
        self.code = code


def lib():
    """Load libsnowgpu.so (once).  Raises RuntimeError if it has not been built."""
    global _lib
    with _lock:
        if _lib is None:
            if not _LIB_PATH.exists():
                raise RuntimeError(
                    f"{_LIB_PATH} is missing: the HIP extension has not been built. Run "
                    "`python -m lidar_snow_sim_amd.build` (needs hipcc). There is no CPU fallback.")
            L = ctypes.CDLL(str(_LIB_PATH))
            vp, i32, i64, dbl = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_double
            L.snowgpu_create.restype = ctypes.c_int
            L.snowgpu_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
            L.snowgpu_destroy.restype = None
            L.snowgpu_destroy.argtypes = [vp]
            L.snowgpu_last_error.restype = ctypes.c_char_p
            L.snowgpu_last_error.argtypes = [vp]
            L.snowgpu_version.restype = ctypes.c_char_p
            L.snowgpu_version.argtypes = []
            L.snowgpu_set_lasers.restype = ctypes.c_int
            L.snowgpu_set_lasers.argtypes = [vp, ctypes.c_int, vp, vp, vp, vp]
            L.snowgpu_upload_table.restype = ctypes.c_int
            L.snowgpu_upload_table.argtypes = [vp, ctypes.c_int, vp, i64]
            L.snowgpu_file_table_device.restype = ctypes.c_int
            L.snowgpu_file_table_device.argtypes = [vp, ctypes.c_int, vp, i64]
            L.snowgpu_debug_table.restype = ctypes.c_int
            L.snowgpu_debug_table.argtypes = [vp, ctypes.c_int, vp, i64]
            L.snowgpu_free_table.restype = ctypes.c_int
            L.snowgpu_free_table.argtypes = [vp, ctypes.c_int]
            L.snowgpu_table_count.restype = ctypes.c_int
            L.snowgpu_table_count.argtypes = [vp]
            L.snowgpu_range_grid.restype = ctypes.c_int
            L.snowgpu_range_grid.argtypes = [vp]
            L.snowgpu_augment_batch.restype = ctypes.c_int
            L.snowgpu_augment_batch.argtypes = [vp, ctypes.c_int, vp, vp, ctypes.c_int, vp, dbl, vp, vp, dbl, vp,
                                                vp, vp, vp, vp, vp]
            L.snowgpu_augment_batch_compact.restype = ctypes.c_int
            L.snowgpu_augment_batch_compact.argtypes = [vp, ctypes.c_int, vp, vp, vp, vp, dbl, vp, vp, dbl, vp, vp, vp, vp, vp]
            L.snowgpu_augment_batch_device.restype = ctypes.c_int
            L.snowgpu_augment_batch_device.argtypes = [vp, ctypes.c_int, i64, i64, vp, vp, ctypes.c_int, vp, dbl, vp, vp,
                                                       dbl, vp, vp, vp, vp, vp, vp, vp, vp]
            L.snowgpu_debug_occlusions.restype = ctypes.c_int
            L.snowgpu_debug_occlusions.argtypes = [vp, i64, vp, ctypes.c_int, vp, dbl, ctypes.c_int, vp, vp, vp, vp]
            L.snowgpu_wet_ground_batch.restype = ctypes.c_int
            L.snowgpu_wet_ground_batch.argtypes = [vp, ctypes.c_int, vp, vp, ctypes.c_int, vp, dbl, dbl, dbl, dbl,
                                                   ctypes.c_int, dbl, ctypes.c_int, vp, vp, vp, vp]
            L.snowgpu_augment_wet_batch.restype = ctypes.c_int
            L.snowgpu_augment_wet_batch.argtypes = [vp, ctypes.c_int, vp, vp, ctypes.c_int, vp, dbl, vp, vp, dbl, vp, vp,
                                                    dbl, dbl, dbl, dbl, ctypes.c_int, dbl, ctypes.c_int, vp, vp, vp, vp, vp]
            L.snowgpu_augment_wet_batch_device.restype = ctypes.c_int
            L.snowgpu_augment_wet_batch_device.argtypes = [vp, ctypes.c_int, i64, i64, vp, vp, ctypes.c_int, vp, dbl, vp, vp, dbl, vp,
                                                           vp, dbl, dbl, dbl, dbl, ctypes.c_int, dbl, ctypes.c_int, vp, vp, vp, vp, vp,
                                                           vp, vp]
            L.snowgpu_set_fov_precrop.restype = ctypes.c_int
            L.snowgpu_set_fov_precrop.argtypes = [vp, ctypes.c_int]
            L.snowgpu_last_status.restype = ctypes.c_int
            L.snowgpu_last_status.argtypes = [vp, vp]
            L.snowgpu_set_threshold_callback.restype = ctypes.c_int
            L.snowgpu_set_threshold_callback.argtypes = [vp, vp, vp]
            L.snowgpu_status_error.restype = ctypes.c_int
            L.snowgpu_status_error.argtypes = [vp, vp]
            L.snowgpu_set_fov.restype = ctypes.c_int
            L.snowgpu_set_fov.argtypes = [vp, ctypes.c_int, vp, vp, vp, ctypes.c_int, ctypes.c_int]
            L.snowgpu_sample_table.restype = ctypes.c_int
            L.snowgpu_sample_table.argtypes = [vp, ctypes.c_int, dbl, dbl, dbl, ctypes.c_uint64, vp, i64, vp]
            L.snowgpu_set_wet_lines.restype = ctypes.c_int
            L.snowgpu_set_wet_lines.argtypes = [vp, ctypes.c_int, vp]
            L.snowgpu_set_wet_estimation.restype = ctypes.c_int
            L.snowgpu_set_wet_estimation.argtypes = [vp, ctypes.c_int, ctypes.c_uint64]
            L.snowgpu_debug_ransac_polyfit.restype = ctypes.c_int
            L.snowgpu_debug_ransac_polyfit.argtypes = [vp, ctypes.c_int, vp, vp, ctypes.c_uint64, ctypes.c_uint64, vp]
            L.snowgpu_wet_last_fit.restype = ctypes.c_int
            L.snowgpu_wet_last_fit.argtypes = [vp, ctypes.c_int, vp]
            L.snowgpu_set_plane_method.restype = ctypes.c_int
            L.snowgpu_set_plane_method.argtypes = [vp, ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, dbl]
            L.snowgpu_estimate_planes.restype = ctypes.c_int
            L.snowgpu_estimate_planes.argtypes = [vp, ctypes.c_int, vp, vp, ctypes.c_int, vp, vp]
            L.snowgpu_estimate_planes_device.restype = ctypes.c_int
            L.snowgpu_estimate_planes_device.argtypes = [vp, ctypes.c_int, i64, i64, vp, vp, ctypes.c_int, vp, vp, vp]
            L.snowgpu_prepass_stats.restype = ctypes.c_int
            L.snowgpu_prepass_stats.argtypes = [vp, ctypes.c_int, vp, vp, ctypes.c_int, vp, vp, vp]
            L.snowgpu_device_numa_node.restype = ctypes.c_int
            L.snowgpu_device_numa_node.argtypes = [ctypes.c_int]
            L.snowgpu_set_result_transfer.restype = ctypes.c_int
            L.snowgpu_set_result_transfer.argtypes = [vp, ctypes.c_int, ctypes.c_int]
            L.snowgpu_debug_transfer_times.restype = ctypes.c_int
            L.snowgpu_debug_transfer_times.argtypes = [vp, vp]
            L.snowgpu_set_pipeline.restype = ctypes.c_int
            L.snowgpu_set_pipeline.argtypes = [vp, i64]
            L.snowgpu_set_serial.restype = ctypes.c_int
            L.snowgpu_set_serial.argtypes = [vp, ctypes.c_int]
            L.snowgpu_lane_stream.restype = ctypes.c_int
            L.snowgpu_lane_stream.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
            L.snowgpu_set_exact_math.restype = ctypes.c_int
            L.snowgpu_set_exact_math.argtypes = [vp, ctypes.c_int]
            L.snowgpu_profile_begin.restype = ctypes.c_int
            L.snowgpu_profile_begin.argtypes = [vp, ctypes.c_int]
            L.snowgpu_profile_end.restype = ctypes.c_int
            L.snowgpu_profile_end.argtypes = [vp, vp, vp]
            L.snowgpu_host_alloc.restype = ctypes.c_int
            L.snowgpu_host_alloc.argtypes = [vp, ctypes.c_size_t, ctypes.POINTER(vp)]
            L.snowgpu_host_free.restype = ctypes.c_int
            L.snowgpu_host_free.argtypes = [vp, vp]
            _lib = L
    return _lib


def _p(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def _dtype_code(dt):
    if dt == np.float32:
        return 0
    if dt == np.float64:
        return 1
    raise TypeError(f"rows must be float32 or float64, not {dt}")


def range_grid():
    out = np.zeros(1230)
    rc = lib().snowgpu_range_grid(_p(out))
    if rc:
        raise SnowGPUError(rc, "snowgpu_range_grid")
    return out


class Context:
    """One libsnowgpu context = one device + one stream + its uploaded tables."""

    def __init__(self, device: int = 0):
        self._L = lib()
        h = ctypes.c_void_p()
        rc = self._L.snowgpu_create(int(device), ctypes.byref(h))
        self._h = h
        self.device = int(device)
        if rc:
            msg = self._L.snowgpu_last_error(h).decode() if h else "no usable HIP device"
            if h:
                self._L.snowgpu_destroy(h)
                self._h = None
            raise SnowGPUError(rc, msg)
        self.n_lasers = 0
        self._call_lock = threading.Lock()

    def close(self):
        if getattr(self, "_h", None):
            self._L.snowgpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise SnowGPUError(rc, self._L.snowgpu_last_error(self._h).decode())

    @property
    def handle(self):
        return self._h

    def set_lasers(self, focal_slope, focal_offset, min_intensity, max_intensity):
        fs = np.ascontiguousarray(focal_slope, np.float64)
        fo = np.ascontiguousarray(focal_offset, np.float64)
        mi = np.ascontiguousarray(min_intensity, np.int32)
        ma = np.ascontiguousarray(max_intensity, np.int32)
        self._check(self._L.snowgpu_set_lasers(self._h, len(fs), _p(fs), _p(fo), _p(mi), _p(ma)))
        self.n_lasers = len(fs)

    def upload_table(self, table_id: int, xyr):
        t = np.ascontiguousarray(xyr, np.float64)
        if t.ndim != 2 or t.shape[1] != 3:
            raise ValueError("a particle table is K x 3 (x, y, disk radius)")
        with self._call_lock:             # never while a batch of this context is in flight (it reads the table list)
            self._check(self._L.snowgpu_upload_table(self._h, int(table_id), _p(t), t.shape[0]))

    def file_table_device(self, table_id: int, d_xyr: int, n_flakes: int):
        """File a K x 3 float64 table that already lives in device memory (d_xyr: device pointer as int)."""
        with self._call_lock:
            self._check(self._L.snowgpu_file_table_device(self._h, int(table_id), ctypes.c_void_p(d_xyr), int(n_flakes)))

    def free_table(self, table_id: int):
        with self._call_lock:
            self._check(self._L.snowgpu_free_table(self._h, int(table_id)))

    def pinned_empty(self, shape, dtype):
        """An uninitialised NumPy array in page-locked host memory (snowgpu_host_alloc); freed with the array."""
        import weakref
        dtype = np.dtype(dtype)
        shape = (int(shape),) if np.isscalar(shape) else tuple(int(v) for v in shape)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        ptr = ctypes.c_void_p()
        self._check(self._L.snowgpu_host_alloc(self._h, ctypes.c_size_t(nbytes), ctypes.byref(ptr)))
        buf = (ctypes.c_char * max(nbytes, 1)).from_address(ptr.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape, dtype=np.int64))).reshape(shape)
        L, addr = self._L, ptr.value       # page-locked memory is not tied to the context: the array may outlive it
        weakref.finalize(buf, lambda: L.snowgpu_host_free(None, ctypes.c_void_p(addr)))
        return arr

    def augment_batch(self, rows, frame_offsets, table_ids, beam_divergence, thr_poly=None, plane=None,
                      noise_floor=0.7, perm=None, want_thr=False, out_rows=None, out_src=None, want_src=True, rows_resident=False):
        """rows: N_total x 5 (float32/float64), frame_offsets: n_frames + 1, table_ids: n_frames x n_lasers.
        out_rows / out_src: optional caller-owned result buffers (at least N_total rows; pinned_empty() ones move at
        PCIe speed and are not page-faulted in on every call).  want_src=False: the source indices stay on the device
        (out_src is returned as None).  rows_resident: `rows` are the rows of the immediately preceding prepass_stats() call --
        they are still on the device and are not uploaded again (rows = NULL at the C ABI).

        Returns (out_rows [N_total x 5, only the first counts[f] rows of each frame slot are valid],
                 out_src, counts, stats[n_frames x 3], thr_poly or None)."""
        rows = np.ascontiguousarray(rows)
        code = _dtype_code(rows.dtype)
        off = np.ascontiguousarray(frame_offsets, np.int64)
        nf = len(off) - 1
        tids = np.ascontiguousarray(table_ids, np.int32).reshape(nf, -1)
        if tids.shape[1] != self.n_lasers:
            raise ValueError("table_ids must be n_frames x n_lasers")
        n = int(off[-1])
        if out_rows is None:
            out_rows = np.empty((n, 5), rows.dtype)
        elif out_rows.dtype != rows.dtype or out_rows.ndim != 2 or out_rows.shape[1] != 5 or out_rows.shape[0] < n \
                or not out_rows.flags.c_contiguous:
            raise ValueError("out_rows must be a C-contiguous (>= N_total) x 5 array of the input dtype")
        if not want_src:
            out_src = None
        elif out_src is None:
            out_src = np.empty(n, np.int32)
        elif out_src.dtype != np.int32 or out_src.ndim != 1 or out_src.shape[0] < n or not out_src.flags.c_contiguous:
            raise ValueError("out_src must be a C-contiguous int32 array of >= N_total entries")
        counts = np.zeros(nf, np.int64)
        stats = np.zeros((nf, 3), np.int64)
        thr = None if thr_poly is None else np.ascontiguousarray(thr_poly, np.float64).reshape(nf, 3)
        pl = None if plane is None else np.ascontiguousarray(plane, np.float64).reshape(nf, 4)
        pm = None if perm is None else np.ascontiguousarray(perm, np.int32)
        out_thr = np.zeros((nf, 3)) if want_thr else None
        with self._call_lock:
            rc = self._L.snowgpu_augment_batch(self._h, nf, _p(off), None if rows_resident else _p(rows), code, _p(tids), float(beam_divergence),
                                               _p(thr), _p(pl), float(noise_floor), _p(pm), _p(out_rows), _p(out_src),
                                               _p(counts), _p(stats), _p(out_thr))
            err = getattr(self, "_thr_error", None)
            if rc and err is not None:                       # the threshold callback raised: its exception, not the status code
                self._thr_error = None
                raise err
            self._check(rc)
        return out_rows, out_src, counts, stats, out_thr

    def augment_batch_compact(self, xyzi, channels, frame_offsets, table_ids, beam_divergence, thr_poly=None, plane=None, noise_floor=0.7,
                              want_thr=False, out_rows=None, out_src=None, want_src=True):
        """augment_batch for frames given as (x, y, z, intensity) float32 rows + one channel byte per row (17 instead of 20 bytes per point up
        the link: snowgpu_augment_batch_compact).  Same results as augment_batch on the (x, y, z, intensity, channel) rows."""
        xyzi = np.ascontiguousarray(xyzi, np.float32)
        ch = np.ascontiguousarray(channels, np.uint8)
        off = np.ascontiguousarray(frame_offsets, np.int64)
        nf, n = len(off) - 1, int(off[-1])
        if xyzi.ndim != 2 or xyzi.shape[1] != 4 or xyzi.shape[0] < n or ch.shape[0] < n:
            raise ValueError("xyzi must be N x 4 float32 and channels N uint8")
        tids = np.ascontiguousarray(table_ids, np.int32).reshape(nf, -1)
        if tids.shape[1] != self.n_lasers:
            raise ValueError("table_ids must be n_frames x n_lasers")
        if out_rows is None:
            out_rows = np.empty((n, 5), np.float32)
        if not want_src:
            out_src = None
        elif out_src is None:
            out_src = np.empty(n, np.int32)
        counts = np.zeros(nf, np.int64)
        stats = np.zeros((nf, 3), np.int64)
        thr = None if thr_poly is None else np.ascontiguousarray(thr_poly, np.float64).reshape(nf, 3)
        pl = None if plane is None else np.ascontiguousarray(plane, np.float64).reshape(nf, 4)
        out_thr = np.zeros((nf, 3)) if want_thr else None
        with self._call_lock:
            rc = self._L.snowgpu_augment_batch_compact(self._h, nf, _p(off), _p(xyzi), _p(ch), _p(tids), float(beam_divergence), _p(thr), _p(pl),
                                                       float(noise_floor), _p(out_rows), _p(out_src), _p(counts), _p(stats), _p(out_thr))
            err = getattr(self, "_thr_error", None)
            if rc and err is not None:
                self._thr_error = None
                raise err
            self._check(rc)
        return out_rows, out_src, counts, stats, out_thr

    def augment_batch_device(self, n_frames, n_total, max_frame_rows, d_frame_off, d_rows, dtype_code, d_table_ids, beam_divergence,
                             d_thr_poly, d_plane, noise_floor, d_perm, d_out_rows, d_out_src, d_out_counts,
                             d_out_stats, d_out_thr, d_status, stream=0):
        """Raw device-pointer entry (ints from tensor.data_ptr()); asynchronous on `stream`."""
        vp = ctypes.c_void_p
        rc = self._L.snowgpu_augment_batch_device(self._h, int(n_frames), int(n_total), int(max_frame_rows), vp(d_frame_off), vp(d_rows),
                                                  int(dtype_code), vp(d_table_ids), float(beam_divergence),
                                                  vp(d_thr_poly or None), vp(d_plane or None), float(noise_floor),
                                                  vp(d_perm or None), vp(d_out_rows), vp(d_out_src), vp(d_out_counts),
                                                  vp(d_out_stats), vp(d_out_thr or None), vp(d_status), vp(stream or None))
        self._check(rc)

    def augment_wet_batch_device(self, n_frames, n_total, max_frame_rows, d_frame_off, d_rows, dtype_code, d_table_ids,
                                 beam_divergence, d_thr_poly, d_plane, noise_floor, d_perm, d_wet_plane, water_height,
                                 pavement_depth, wet_noise_floor, power_factor, flat_earth, delta, replace, d_out_rows, d_out_src,
                                 d_out_counts, d_out_stats, d_out_flags, d_status, stream=0):
        """Raw device-pointer entry of the fused snowfall + wet-ground chain; asynchronous on `stream`."""
        vp = ctypes.c_void_p
        rc = self._L.snowgpu_augment_wet_batch_device(
            self._h, int(n_frames), int(n_total), int(max_frame_rows), vp(d_frame_off), vp(d_rows), int(dtype_code),
            vp(d_table_ids), float(beam_divergence), vp(d_thr_poly or None), vp(d_plane or None), float(noise_floor),
            vp(d_perm or None), vp(d_wet_plane or None), float(water_height), float(pavement_depth), float(wet_noise_floor),
            float(power_factor), int(bool(flat_earth)), float(delta), int(bool(replace)), vp(d_out_rows), vp(d_out_src),
            vp(d_out_counts), vp(d_out_stats), vp(d_out_flags), vp(d_status), vp(stream or None))
        self._check(rc)

    def set_fov(self, calib=None, img_shape=(1024, 1920), pre_crop=False):
        """Camera-FOV crop inside the compaction of every later batch (None switches it off).  `calib` carries V2C (3 x 4),
        R0 (3 x 3) and P2 (3 x 4), as lidar_snow_sim_amd.calibration.Calibration does.  pre_crop: also crop the input frames
        before the augmentation (precompute.py:96-99)."""
        with self._call_lock:
            self._check(self._L.snowgpu_set_fov_precrop(self._h, int(bool(pre_crop and calib is not None))))
            if calib is None:
                self._check(self._L.snowgpu_set_fov(self._h, 0, None, None, None, 0, 0))
                return
            v2c = np.ascontiguousarray(calib.V2C, np.float64).reshape(3, 4)
            r0 = np.ascontiguousarray(calib.R0, np.float64).reshape(3, 3)
            p2 = np.ascontiguousarray(calib.P2, np.float64).reshape(3, 4)
            self._check(self._L.snowgpu_set_fov(self._h, 1, _p(v2c), _p(r0), _p(p2), int(img_shape[0]), int(img_shape[1])))

    def set_plane_method(self, method="reference", seed=0, trials=1000, min_rows=5, standard_height=-1.55):
        """How this context estimates the ground plane when a call brings none (planes.py:12-50): 'reference' (what the
        reference returns today: the flat-earth plane), 'lsq' or 'ransac' (Philox-seeded)."""
        if method not in PLANE_METHODS:
            raise ValueError("plane method must be 'reference', 'lsq' or 'ransac'")
        with self._call_lock:
            self._check(self._L.snowgpu_set_plane_method(self._h, PLANE_METHODS[method], ctypes.c_uint64(int(seed) & (2 ** 64 - 1)),
                                                         int(trials), int(min_rows), float(standard_height)))

    def estimate_planes(self, rows, frame_offsets):
        """(planes n_frames x 4 (wx, wy, wz, h), info n_frames x 4 int32) by the context's plane method."""
        rows = np.ascontiguousarray(rows)
        code = _dtype_code(rows.dtype)
        off = np.ascontiguousarray(frame_offsets, np.int64)
        nf = len(off) - 1
        planes = np.zeros((nf, 4), np.float64)
        info = np.zeros((nf, 4), np.int32)
        with self._call_lock:
            self._check(self._L.snowgpu_estimate_planes(self._h, nf, _p(off), _p(rows), code, _p(planes), _p(info)))
        return planes, info

    def prepass_stats(self, rows, frame_offsets, plane=None, hist_out=None):
        """(hist n_frames x 50 x 2555 int32, rec n_frames x 18 float64): the device half of the noise-threshold prepass
        (snowgpu_prepass_stats) for a caller that takes the histogram's row minima with its own NumPy (quirk Q8)."""
        rows = np.ascontiguousarray(rows)
        code = _dtype_code(rows.dtype)
        off = np.ascontiguousarray(frame_offsets, np.int64)
        nf = len(off) - 1
        pl = None if plane is None else np.ascontiguousarray(plane, np.float64).reshape(nf, 4)
        if hist_out is not None and hist_out.size >= nf * 50 * 2555 and hist_out.dtype == np.int32 and hist_out.flags.c_contiguous:
            hist = hist_out.reshape(-1)[:nf * 50 * 2555].reshape(nf, 50, 2555)     # e.g. a page-locked buffer kept by the caller
        else:
            hist = np.empty((nf, 50, 2555), np.int32)
        rec = np.empty((nf, 18), np.float64)
        with self._call_lock:
            self._check(self._L.snowgpu_prepass_stats(self._h, nf, _p(off), _p(rows), code, _p(pl), _p(hist), _p(rec)))
        return hist, rec

    def sample_table(self, table_id, occupancy_ratio, diameter_scale_mm, r_0, seed, want_rows=True):
        """Sample a snowflake table on the device (and file it there under table_id >= 0).  Returns the K x 3 rows, or K
        when want_rows is False (the rows then never leave the device)."""
        n = ctypes.c_int64(0)
        args = (float(occupancy_ratio), float(diameter_scale_mm), float(r_0), ctypes.c_uint64(int(seed)))
        with self._call_lock:
            if not want_rows:
                self._check(self._L.snowgpu_sample_table(self._h, int(table_id), *args, None, 0, ctypes.byref(n)))
                return n.value
            # one run: a buffer for twice the expected flake count (E[disk area] = pi s^2 / 3 for Exp(s) sphere diameters)
            s_m = float(diameter_scale_mm) / 1000.0
            cap = int(2.0 * occupancy_ratio * r_0 * r_0 / (s_m * s_m / 3.0)) + 4096
            out = np.empty((cap, 3), np.float64)
            rc = self._L.snowgpu_sample_table(self._h, int(table_id), *args, _p(out), cap, ctypes.byref(n))
            if rc == E_INVALID and n.value > cap:            # buffer too small after all: fetch with the exact size
                out = np.empty((n.value, 3), np.float64)
                rc = self._L.snowgpu_sample_table(self._h, int(table_id), *args, _p(out), n.value, ctypes.byref(n))
            self._check(rc)
        return np.ascontiguousarray(out[:n.value])

    def debug_table(self, table_id, n_flakes):
        """K x 4 per-flake quantities of a filed table by table row: range, azimuth, right and left tangent angle."""
        out = np.zeros((int(n_flakes), 4), np.float64)
        with self._call_lock:
            self._check(self._L.snowgpu_debug_table(self._h, int(table_id), _p(out), int(n_flakes)))
        return out

    def last_status(self):
        """int32[8] status words of the last host-entry batch ([2..5]: beams per later capacity tier)."""
        out = np.zeros(8, np.int32)
        self._check(self._L.snowgpu_last_status(self._h, _p(out)))
        return out

    THRESHOLD_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int32),
                                    ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double))

    def set_threshold_callback(self, fit=None):
        """fit(first_frame, hist[n, 50, 2555] int32, rec[n, 18] float64) -> n x 3 polynomials, called by the following augment_batch calls
        (those without thr_poly) once per group of frames while the per-beam kernels of the group run (snowgpu_set_threshold_callback);
        None: the device fits the threshold itself again.  An exception raised by `fit` fails the call and is re-raised by augment_batch."""
        self._thr_error = None
        if fit is None:
            self._thr_cb = None
            self._check(self._L.snowgpu_set_threshold_callback(self._h, None, None))
            return

        def trampoline(_user, first, n, hist_p, rec_p, out_p):
            try:
                hist = np.ctypeslib.as_array(hist_p, shape=(n, 50, 2555))
                rec = np.ctypeslib.as_array(rec_p, shape=(n, 18))
                out = np.ctypeslib.as_array(out_p, shape=(n, 3))
                out[...] = np.asarray(fit(int(first), hist, rec), np.float64).reshape(n, 3)
                return 0
            except BaseException as ex:      # (no exception may cross the C frames)
                self._thr_error = ex
                return 1

        self._thr_cb = self.THRESHOLD_FN(trampoline)             # kept alive as long as the library may call it
        self._check(self._L.snowgpu_set_threshold_callback(self._h, ctypes.cast(self._thr_cb, ctypes.c_void_p), None))

    def check_status(self, status8):
        """Raise what the int32[8] status words of a device-pointer call say (snowgpu_status_error); no-op for status8[0] == 0."""
        st = np.ascontiguousarray(status8, np.int32)
        if st[0] != 0:
            self._check(self._L.snowgpu_status_error(self._h, _p(st)))

    def set_result_transfer(self, mode="rows", threads=0):
        """How a pipelined host batch's results cross the link: 'rows' (output rows + source indices) or 'packed' (source | label +
        intensity per kept row; `threads` host threads of the library assemble the rows from the caller's input)."""
        if mode not in ("rows", "packed"):
            raise ValueError("mode must be 'rows' or 'packed'")
        with self._call_lock:
            self._check(self._L.snowgpu_set_result_transfer(self._h, 1 if mode == "packed" else 0, int(threads)))

    def transfer_times(self):
        out = np.zeros(4, np.float64)
        self._check(self._L.snowgpu_debug_transfer_times(self._h, _p(out)))
        return {"enqueued_ms": float(out[0]), "downloaded_ms": float(out[1]), "assembled_ms": float(out[2]), "host_threads": int(out[3])}

    def set_pipeline(self, chunk_rows: int):
        """Rows per chunk of the host entry's upload / compute / download pipeline (0: off)."""
        with self._call_lock:
            self._check(self._L.snowgpu_set_pipeline(self._h, int(chunk_rows)))

    def set_serial(self, on: bool):
        """Every kernel of a device-pointer batch on the caller's stream (snowgpu_set_serial): for contexts that run as one of several compute
        lanes, whose batches overlap each other instead of their own side streams."""
        self._check(self._L.snowgpu_set_serial(self._h, int(bool(on))))

    def lane_stream(self, level: int) -> int:
        """hipStream_t (as an integer) of the context's stream at priority level 0 / 1 / 2 = highest / normal / lowest (snowgpu_lane_stream)."""
        h = ctypes.c_void_p(0)
        self._check(self._L.snowgpu_lane_stream(self._h, int(level), ctypes.byref(h)))
        return int(h.value or 0)

    def set_exact_math(self, on: bool):
        self._check(self._L.snowgpu_set_exact_math(self._h, int(bool(on))))

    def profile_begin(self, max_launches):
        self._check(self._L.snowgpu_profile_begin(self._h, int(max_launches)))

    def profile_end(self):
        ms, n = ctypes.c_double(0.0), ctypes.c_int(0)
        self._check(self._L.snowgpu_profile_end(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def debug_occlusions(self, rows, table_ids, beam_divergence, cap=64):
        rows = np.ascontiguousarray(rows)
        code = _dtype_code(rows.dtype)
        n = rows.shape[0]
        tids = np.ascontiguousarray(table_ids, np.int32).reshape(-1)
        count = np.zeros(n, np.int32)
        rj = np.zeros((n, cap))
        ratio = np.zeros((n, cap))
        src = np.zeros(n, np.int32)
        with self._call_lock:
            self._check(self._L.snowgpu_debug_occlusions(self._h, n, _p(rows), code, _p(tids), float(beam_divergence),
                                                         int(cap), _p(count), _p(rj), _p(ratio), _p(src)))
        return count, rj, ratio, src

    def augment_wet_batch(self, rows, frame_offsets, table_ids, beam_divergence, wet_plane, thr_poly=None, plane=None,
                          noise_floor=0.7, perm=None, water_height=0.001, pavement_depth=0.0012, wet_noise_floor=0.7,
                          power_factor=15, flat_earth=False, delta=0.5, replace=False):
        rows = np.ascontiguousarray(rows)
        code = _dtype_code(rows.dtype)
        off = np.ascontiguousarray(frame_offsets, np.int64)
        nf = len(off) - 1
        tids = np.ascontiguousarray(table_ids, np.int32).reshape(nf, -1)
        n = int(off[-1])
        out_rows = np.empty((n, 5), np.float64)
        out_src = np.empty(n, np.int32)
        counts = np.zeros(nf, np.int64)
        stats = np.zeros((nf, 3), np.int64)
        flags = np.zeros(nf, np.int32)
        thr = None if thr_poly is None else np.ascontiguousarray(thr_poly, np.float64).reshape(nf, 3)
        pl = None if plane is None else np.ascontiguousarray(plane, np.float64).reshape(nf, 4)
        wpl = None if wet_plane is None else np.ascontiguousarray(wet_plane, np.float64).reshape(nf, 4)
        pm = None if perm is None else np.ascontiguousarray(perm, np.int32)
        with self._call_lock:
            self._check(self._L.snowgpu_augment_wet_batch(
                self._h, nf, _p(off), _p(rows), code, _p(tids), float(beam_divergence), _p(thr), _p(pl), float(noise_floor),
                _p(pm), _p(wpl), float(water_height), float(pavement_depth), float(wet_noise_floor), float(power_factor),
                int(bool(flat_earth)), float(delta), int(bool(replace)), _p(out_rows), _p(out_src), _p(counts), _p(stats),
                _p(flags)))
        return out_rows, out_src, counts, stats, flags

    def set_wet_estimation(self, method="linear", seed=0):
        """estimation_method of the wet-ground calls of this context: 'linear' or 'poly' (RANSAC draws from Philox keyed by `seed`)."""
        if method not in WET_ESTIMATION:
            raise ValueError("estimation_method must be 'linear' or 'poly'")
        with self._call_lock:
            self._check(self._L.snowgpu_set_wet_estimation(self._h, WET_ESTIMATION[method], int(seed) & 0xFFFFFFFFFFFFFFFF))

    def debug_ransac_polyfit(self, x, y, seed=0, frame=0):
        """The device's ransac_polyfit(x, y, order=2) with the draws of (seed; frame): (coefficients[3], trial kept or -1)."""
        x = np.ascontiguousarray(x, np.float64)
        y = np.ascontiguousarray(y, np.float64)
        out = np.zeros(4, np.float64)
        with self._call_lock:
            self._check(self._L.snowgpu_debug_ransac_polyfit(self._h, int(x.shape[0]), _p(x), _p(y), int(seed), int(frame), _p(out)))
        return out[:3], int(out[3])

    def wet_last_fit(self, n_frames):
        """The curves the last wet-ground call fitted: n_frames x 8 (power c2 c1 c0, noise c2 c1 c0, ground rows, RANSAC trial kept or -1)."""
        out = np.zeros((int(n_frames), 8), np.float64)
        with self._call_lock:
            self._check(self._L.snowgpu_wet_last_fit(self._h, int(n_frames), _p(out)))
        return out

    def wet_ground_batch(self, rows, frame_offsets, plane, water_height, pavement_depth, noise_floor, power_factor,
                         flat_earth, delta, replace, lines=None):
        """lines: optional n_frames x 4 (p slope, p intercept, noise-line slope, intercept) fitted by the caller (quirk Q8)."""
        rows = np.ascontiguousarray(rows)
        code = _dtype_code(rows.dtype)
        off = np.ascontiguousarray(frame_offsets, np.int64)
        nf = len(off) - 1
        n = int(off[-1])
        pl = None if plane is None else np.ascontiguousarray(plane, np.float64).reshape(nf, 4)
        out_rows = np.empty((n, 5), np.float64)
        out_src = np.empty(n, np.int32)
        counts = np.zeros(nf, np.int64)
        flags = np.zeros(nf, np.int32)
        with self._call_lock:
            if lines is not None:
                ln = np.ascontiguousarray(lines, np.float64).reshape(nf, 4)
                self._check(self._L.snowgpu_set_wet_lines(self._h, nf, _p(ln)))
            self._check(self._L.snowgpu_wet_ground_batch(self._h, nf, _p(off), _p(rows), code, _p(pl), float(water_height),
                                                         float(pavement_depth), float(noise_floor), float(power_factor),
                                                         int(bool(flat_earth)), float(delta), int(bool(replace)),
                                                         _p(out_rows), _p(out_src), _p(counts), _p(flags)))
        return out_rows, out_src, counts, flags
