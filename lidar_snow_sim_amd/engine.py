"""Per-device engine: one libsnowgpu context, the HDL-64E S3 laser table and the particle-table cache.

Host logic only -- every number the simulation produces comes out of libsnowgpu.so.
"""
from __future__ import annotations

import json
import threading
from pathlib import Path

import numpy as np

from . import _native

_DATA = Path(__file__).resolve().parent / "data" / "hdl64e_s3_lasers.json"
_engines = {}
_engines_lock = threading.Lock()


def load_lasers(path=None):
    """The per-laser dicts the reference gets from yaml.safe_load(calib/20171102_64E_S3.yaml)['lasers']
    (tools/snowfall/simulation.py:474-480), reduced to the three fields the simulation reads (:72-76)."""
    d = json.loads(Path(path or _DATA).read_text())
    return [{"focal_distance": fd, "focal_slope": fs, **({} if mi is None else {"min_intensity": mi})}
            for fd, fs, mi in zip(d["focal_distance"], d["focal_slope"], d["min_intensity"])]


def laser_constants(lasers):
    """(focal_slope, focal_offset, min_intensity, max_intensity) per channel -- simulation.py:72-76, :123-126."""
    fs, fo, mi, ma = [], [], [], []
    for ch, info in enumerate(lasers):
        focal_distance = info["focal_distance"] * 100                  # :74
        fs.append(float(info["focal_slope"]))                          # :75
        fo.append((1 - focal_distance / 13100) ** 2)                   # :76 (Python float arithmetic, as there)
        mi.append(int(info.get("min_intensity", 0)))                   # :72
        ma.append(230 if (ch % 64) in (53, 55, 56, 58) else 255)       # :123-126 (tiled beyond 64 lasers)
    return fs, fo, mi, ma


class Engine:
    def __init__(self, device: int = 0, lasers=None):
        self.ctx = _native.Context(device)
        self.device = device
        self.lasers = load_lasers() if lasers is None else lasers
        self.ctx.set_lasers(*laser_constants(self.lasers))
        self._tables = {}          # key -> table id (file tables: kept for the life of the engine)
        self._arrays = {}          # id(array) -> (array, table id), least recently used first (array-backed tables)
        self._free_ids = []
        self._next_id = 0
        self._lock = threading.Lock()
        self.batch_lock = threading.RLock()   # one batch at a time per engine: the input staging buffer is reused

    @property
    def n_lasers(self):
        return len(self.lasers)

    # ---- page-locked host buffers -----------------------------------------------------------------------------
    # Host arrays cross PCIe at full speed only from page-locked memory, and a fresh NumPy array is page-faulted in
    # while the copy runs.  So frames are staged in a reused page-locked input buffer, and results land in
    # page-locked buffers that go back to a small pool when the arrays handed to the caller are garbage-collected
    # (the caller still owns what it gets: nothing is overwritten while a result is alive).
    PIN_LIMIT = 3 << 30          # bytes of result buffers that may be out with callers; beyond it: plain np.empty

    def staging_in(self, n_rows, dtype):
        """Reused page-locked n_rows x 5 input staging view (valid until the next call on this engine)."""
        dtype = np.dtype(dtype)
        need = int(n_rows) * 5 * dtype.itemsize
        buf = getattr(self, "_pin_in", None)
        if buf is None or buf.nbytes < need:
            self._pin_in = buf = self.ctx.pinned_empty(max(need, 1 << 20) * 5 // 4, np.uint8)
        return buf[:need].view(dtype).reshape(int(n_rows), 5)

    def result_buffers(self, n_rows, dtype):
        """(out_rows n x 5, out_src n) in page-locked memory from the pool, or pageable arrays past PIN_LIMIT."""
        import ctypes
        import weakref
        dtype = np.dtype(dtype)
        need = int(n_rows) * (5 * dtype.itemsize + 4)
        if need == 0:
            return np.empty((0, 5), dtype), np.empty(0, np.int32)
        with self._lock:
            pool = self.__dict__.setdefault("_pin_pool", [])
            pick = None
            for i, b in enumerate(pool):
                if b.nbytes >= need and (pick is None or b.nbytes < pool[pick].nbytes):
                    pick = i
            base = pool.pop(pick) if pick is not None else None
            if base is None:
                out = self.__dict__.get("_pin_out", 0)
                if out + need > self.PIN_LIMIT:
                    return np.empty((int(n_rows), 5), dtype), np.empty(int(n_rows), np.int32)
            self._pin_out = self.__dict__.get("_pin_out", 0) + (base.nbytes if base is not None else max(need, 1 << 20))
        if base is None:
            base = self.ctx.pinned_empty(max(need, 1 << 20), np.uint8)
        # a fresh wrapper object per hand-out: when the caller's views of it die, the bytes return to the pool
        lease = (ctypes.c_char * base.nbytes).from_address(base.ctypes.data)
        weakref.finalize(lease, self._give_back, base)
        flat = np.frombuffer(lease, np.uint8)
        nb = int(n_rows) * 5 * dtype.itemsize
        return flat[:nb].view(dtype).reshape(int(n_rows), 5), flat[nb:nb + 4 * int(n_rows)].view(np.int32)

    def _give_back(self, base):
        with self._lock:
            self._pin_out = self.__dict__.get("_pin_out", 0) - base.nbytes
            pool = self.__dict__.setdefault("_pin_pool", [])
            if len(pool) < 4:
                pool.append(base)

    def set_lasers(self, lasers):
        self.lasers = lasers
        self.ctx.set_lasers(*laser_constants(lasers))

    ARRAY_TABLES = 1024          # array-backed tables kept on the device; the least recently used one goes first

    def _new_id(self):
        if self._free_ids:
            return self._free_ids.pop()
        self._next_id += 1
        return self._next_id - 1

    def table_id(self, key, loader):
        """Device table id for `key`, uploading `loader()` (K x 3 float64) the first time."""
        with self._lock:
            tid = self._tables.get(key)
            if tid is None:
                table = loader()                                         # may raise (missing file): no id is spent then
                tid = self._new_id()
                self.ctx.upload_table(tid, table)
                self._tables[key] = tid
            return tid

    def array_table_id(self, arr):
        """Device table id of a caller-owned K x 3 array.  The cache holds a reference to the array, so its identity cannot
        be recycled for another array while the entry lives (a table is looked up by `is`, never by address alone); the
        contents are taken as they are at the first use.  Beyond ARRAY_TABLES entries the least recently used table is
        dropped from the device."""
        with self._lock:
            hit = self._arrays.get(id(arr))
            if hit is not None and hit[0] is arr:
                self._arrays[id(arr)] = self._arrays.pop(id(arr))        # most recently used last
                return hit[1]
            tid = self._new_id()
            self.ctx.upload_table(tid, arr)
            self._arrays[id(arr)] = (arr, tid)
            while len(self._arrays) > self.ARRAY_TABLES:
                old_key = next(iter(self._arrays))
                _, old_tid = self._arrays.pop(old_key)
                self.ctx.free_table(old_tid)
                self._free_ids.append(old_tid)
            return tid

    def user_table_id(self):
        """A fresh table id for a table the caller files itself (e.g. sampled on the device): never one of the cache's."""
        with self._lock:
            return self._new_id()

    def file_table_id(self, particle_file_prefix, line, root_path=None, sample_missing=False):
        """Device table id of <prefix>_<line>.npy under the reference's directories (simulation.py:324-329).
        sample_missing: a table whose file does not exist is sampled on the device instead (sampled_table_id)."""
        base = (Path(root_path) / "training" / "snowflakes" / "npy") if root_path else particle_dir()
        path = base / f"{particle_file_prefix}_{line}.npy"
        if sample_missing and not path.is_file():
            return self.sampled_table_id(particle_file_prefix, line)
        return self.table_id(("file", str(path)), lambda p=path: np.load(str(p)))

    # ---- tables made where they are used -----------------------------------------------------------------
    R_0 = 80.0                   # sampling.py:362: the reference's tables cover 80 m
    keep_sampled_rows = False    # tests: keep a host copy of every table sampled on the device (sampled_rows)

    def sampled_table_id(self, particle_file_prefix, line):
        """Device table id of the table the reference would load from <prefix>_<line>.npy, sampled and filed ON THE DEVICE
        (snowgpu_sample_table: the dart-throwing process of sampling.py:90-194 from a Philox stream) -- no file, no download.
        prefix = f'{mode}_{rain_rate}_{occupancy}' as precompute.py:101 / pointcloud_viewer.py:2802 build it; the seed is a
        function of (prefix, line), so every engine, rank and run makes the same table for the same name."""
        key = ("device", particle_file_prefix, int(line))
        with self._lock:
            tid = self._tables.get(key)
            if tid is not None:
                return tid
            mode, rain_rate, occupancy = parse_prefix(particle_file_prefix)
            from .tools.snowfall.sampling import gunn_marshall, sekhon_srivastava
            rate_parameter = gunn_marshall(rain_rate) if mode == "gunn" else sekhon_srivastava(rain_rate)
            tid = self._new_id()
            res = self.ctx.sample_table(tid, occupancy, (1 / rate_parameter) * 10, self.R_0, table_seed(particle_file_prefix, line),
                                        want_rows=self.keep_sampled_rows)
            if self.keep_sampled_rows:
                self.__dict__.setdefault("sampled_rows", {})[(particle_file_prefix, int(line))] = res
                n = res.shape[0]
            else:
                n = int(res)
            self._tables[key] = tid
            self.__dict__.setdefault("sampled_flakes", {})[(particle_file_prefix, int(line))] = n
            return tid

    def table_ids_from_files(self, particle_file_prefix, order, root_path=None):
        """The reference's lookup (simulation.py:78, :324-329): channel c reads <prefix>_<order[c]+1>.npy."""
        return [self.file_table_id(particle_file_prefix, order[ch] + 1, root_path) for ch in range(self.n_lasers)]

    def table_ids_from_arrays(self, particles, order):
        """particles: sequence (index = line - 1) of K x 3 arrays; channel c uses particles[order[c]]."""
        ids = []
        for ch in range(self.n_lasers):
            ids.append(self.array_table_id(particles[order[ch]]))
        return ids


def parse_prefix(particle_file_prefix):
    """(mode, rain_rate, occupancy) of f'{mode}_{rain_rate}_{occupancy}' (precompute.py:101, pointcloud_viewer.py:2802)."""
    parts = str(particle_file_prefix).split("_")
    if len(parts) != 3 or parts[0] not in ("gunn", "sekhon"):
        raise ValueError(f"cannot sample tables for prefix {particle_file_prefix!r}: expected 'gunn_<rain rate>_<occupancy>' or "
                         "'sekhon_<rain rate>_<occupancy>'")
    try:
        return parts[0], float(parts[1]), float(parts[2])
    except ValueError:
        raise ValueError(f"cannot sample tables for prefix {particle_file_prefix!r}: rain rate and occupancy must be numbers") from None


def table_seed(particle_file_prefix, line):
    """64-bit seed of the device-sampled table <prefix>_<line>: a stable hash (not Python's salted hash())."""
    import hashlib
    return int.from_bytes(hashlib.blake2b(f"{particle_file_prefix}|{int(line)}".encode(), digest_size=8).digest(), "little")


_particle_dir = None


def particle_dir() -> Path:
    """Default directory of the <prefix>_<line>.npy tables: <repo root>/npy, the reference's rule
    (Path(__file__).parent.parent.parent / 'npy', simulation.py:327)."""
    return _particle_dir or (Path(__file__).resolve().parent.parent / "npy")


def set_particle_dir(path):
    global _particle_dir
    _particle_dir = None if path is None else Path(path)


def get_engine(device: int = 0, slot: int = 0) -> Engine:
    """The engine of `device`.  `slot` > 0 gives further independent contexts (own stream and scratch) on the same
    device, e.g. two of them driven by two host threads overlap one batch's copies with the other's kernels."""
    with _engines_lock:
        eng = _engines.get((device, slot))
        if eng is None:
            eng = Engine(device)
            _engines[(device, slot)] = eng
        return eng
