"""Ground-plane estimate -- the counterpart of tools/wet_ground/planes.py::calculate_plane (planes.py:12-50).

The reference crops the cloud to a strip of road in front of the car, hands the strip to scikit-learn's RANSACRegressor and
falls back to a flat-earth plane when the strip is nearly empty or the regressor raises.  Its RANSAC is unseeded, and with
scikit-learn >= 1.2 the call raises on a misspelt keyword, so today the reference answers every cloud with the flat-earth plane
(parity unpinned).  This module keeps the name, the signature and the return convention -- (w, h) with the
plane normal w and the sensor height h -- and lets the caller choose the estimator; the fits themselves run in libsnowgpu.so
(csrc/snowgpu_plane.hip), never here:

    method='reference'   what the reference returns today: ([0, 0, 1], standart_height).  No row is read, no device needed.
    method='lsq'         least squares z = c0 x + c1 y + b over the strip; w = [c0, c1, -1] / |[c0, c1, -1]|, h = b (planes.py:36-41)
    method='ransac'      seeded RANSAC over the strip (3-point samples from Philox(seed; trial), threshold = MAD of z, refit on
                         the consensus set): the same cloud and seed give the same plane on every run

The augmentation entry points estimate the plane on the device themselves when none is passed (`plane_method=` there); this
function is for callers that want the plane itself, as the reference's own callers of calculate_plane do.
"""
import numpy as np

STANDARD_HEIGHT = -1.55                       # planes.py:12: sensor height of the DENSE car over flat ground
FLAT_EARTH_NORMAL = (0, 0, 1)
# the strip of road the reference fits: below the sensor but not under the road, 10 .. 70 m ahead, 3 m to either side
STRIP = {"x": (10.0, 70.0), "y": (-3.0, 3.0), "z_top": -1.55, "z_floor_at_0": -1.86, "z_floor_slope": -0.01}


def ground_crop(pointcloud):
    """Boolean mask of the rows inside the strip (planes.py:21-26), evaluated in the cloud's own dtype like the device kernel
    k_plane_crop does.  Host-side helper for tests and tools; the estimators crop on the device."""
    pc = np.asarray(pointcloud)
    x, y, z = pc[:, 0], pc[:, 1], pc[:, 2]
    floor = STRIP["z_floor_at_0"] + STRIP["z_floor_slope"] * x
    inside = np.logical_and.reduce((x > STRIP["x"][0], x < STRIP["x"][1], y > STRIP["y"][0], y < STRIP["y"][1]))
    return inside & (z < STRIP["z_top"]) & (z > floor)


def flat_earth(standart_height=STANDARD_HEIGHT):
    return list(FLAT_EARTH_NORMAL), standart_height


def calculate_plane(pointcloud, standart_height=STANDARD_HEIGHT, *, method='reference', seed=0, trials=1000, device=0,
                    return_info=False):
    """(w, h): plane normal and sensor height of `pointcloud` (N x >= 3: x, y, z, ...).

    method / seed / trials: see the module docstring.  return_info=True appends a dict with the strip's row count, the model
    that produced the plane ('flat_earth', 'lsq' or 'ransac') and the rows the final fit used."""
    from ... import _native
    if method not in _native.PLANE_METHODS:
        raise ValueError("method must be 'reference', 'lsq' or 'ransac'")
    pc = np.asarray(pointcloud)
    if method == 'reference':
        w, h = flat_earth(standart_height)
        return (w, h, {"strip_rows": None, "model": "flat_earth", "fit_rows": 0}) if return_info else (w, h)
    if pc.ndim != 2 or pc.shape[1] < 3:
        raise ValueError("pointcloud must be N x >= 3")
    from ... import engine as _engine
    rows = pc if pc.dtype in (np.float32, np.float64) else pc.astype(np.float64)
    xyz5 = np.zeros((rows.shape[0], 5), rows.dtype)
    xyz5[:, :3] = rows[:, :3]
    eng = _engine.get_engine(device)
    with eng.batch_lock:
        # the reference compares the strip's row count with the number of COLUMNS of the array it was given (planes.py:29)
        eng.ctx.set_plane_method(method, seed=seed, trials=trials, min_rows=pc.shape[1], standard_height=standart_height)
        try:
            planes, info = eng.ctx.estimate_planes(xyz5, [0, xyz5.shape[0]])
        finally:
            eng.ctx.set_plane_method('reference')
    model = ("flat_earth", "lsq", "ransac")[int(info[0, 1])]
    if model == "flat_earth":
        w, h = flat_earth(standart_height)
    else:
        w, h = planes[0, :3].copy(), float(planes[0, 3])
    if return_info:
        return w, h, {"strip_rows": int(info[0, 0]), "model": model, "fit_rows": int(info[0, 2]), "valid_trials": int(info[0, 3])}
    return w, h
