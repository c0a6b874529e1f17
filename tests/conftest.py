import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def numpy_is_portable() -> bool:
    """True when this process' NumPy float32 arctan2 is glibc's (SIMD dispatch off or no AVX-512)."""
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.atan2f.restype = ctypes.c_float
    libm.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]
    rng = np.random.default_rng(0)
    x = rng.uniform(-80, 80, 512).astype(np.float32)
    y = rng.uniform(-80, 80, 512).astype(np.float32)
    ref = np.array([libm.atan2f(float(b), float(a)) for a, b in zip(x, y)], np.float32)
    return bool(np.array_equal(np.arctan2(y, x), ref))


@pytest.fixture(scope="session")
def golden():
    def load(name, flavour="portable"):
        return np.load(GOLDEN / f"{name}_{flavour}.npz")
    return load


@pytest.fixture(scope="session")
def tables():
    t = np.load(GOLDEN / "tables.npz")
    return {"t": [t[f"t{i}"] for i in range(4)], "dense": t["dense"]}


def canonical(aug, src):
    """Rows in source-index order (the reference's within-channel order is implementation-defined)."""
    o = np.argsort(src, kind="stable")
    return aug[o], src[o]


def numpy_matches_native_fixtures() -> bool:
    """The `*_native.npz` fixtures were made by the reference on an AVX-512 host with NumPy's default dispatch (x86-simd-sort's
    AVX-512 argpartition, SVML loops).  A host reproduces them only if its NumPy dispatches the same way: SIMD dispatch on
    (not portable) and AVX-512 present.  On anything else q8='numpy' follows THAT host's NumPy, which no committed fixture holds."""
    if numpy_is_portable():
        return False
    try:
        flags = next(line for line in open("/proc/cpuinfo") if line.startswith("flags"))
    except (OSError, StopIteration):
        return False
    return " avx512f" in flags and " avx512dq" in flags
