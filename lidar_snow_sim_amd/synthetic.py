"""Synthetic LiDAR sweeps used by bench.py, the tests and the golden-vector generator.

This is the generator SURVEY.md section 8(d) specifies ("identical on oracle, CPU baseline and GPU"):
L layers x A azimuths, HDL-64E S3 elevations for L = 64 (``vert_correction`` of
calib/20171102_64E_S3.yaml), a ground plane at z = -1.7 m under a cylinder wall at 40 +- 25 m,
ranges clipped to [3, 119] m (the reference crashes at >= 120 m, simulation.py:146-149),
float32 AoS rows (x, y, z, intensity, channel), channel-major order.
"""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np

_DATA = Path(__file__).resolve().parent / "data" / "hdl64e_s3_lasers.json"


def hdl64_elevations() -> np.ndarray:
    """The 64 per-laser elevation angles (radians) of the HDL-64E S3 calibration."""
    return np.asarray(json.loads(_DATA.read_text())["vert_correction"], dtype=np.float64)


def synthetic_sweep(layers: int = 64, azimuths: int = 2048, seed: int = 1000, intensity: str = "uniform",
                    dtype=np.float32) -> np.ndarray:
    """One clear-weather sweep, ``layers * azimuths`` rows of (x, y, z, intensity, channel).

    intensity='uniform'   : integers U{5..119} (SURVEY 8 d)
    intensity='lambert'   : a distance/incidence-angle falloff on the ground, brighter walls --
                            gives the noise-threshold prepass a realistic fit and a mixed keep mask
    """
    rng = np.random.default_rng(seed)
    if layers == 64:
        elev = hdl64_elevations()
    else:
        elev = np.linspace(np.radians(-25.0), np.radians(15.0), layers)
    az = np.linspace(-np.pi, np.pi, azimuths, endpoint=False) + 1e-4
    el = np.repeat(elev, azimuths)
    aa = np.tile(az, layers)
    wall = 40.0 + rng.uniform(-25.0, 25.0, el.shape[0])
    t_wall = wall / np.cos(el)
    with np.errstate(divide="ignore"):
        t_ground = np.where(el < 0, -1.7 / np.sin(np.minimum(el, -1e-9)), np.inf)
    on_ground = t_ground < t_wall
    rng_m = np.clip(np.minimum(t_ground, t_wall), 3.0, 119.0)
    x = rng_m * np.cos(el) * np.cos(aa)
    y = rng_m * np.cos(el) * np.sin(aa)
    z = rng_m * np.sin(el)
    if intensity == "uniform":
        inten = rng.integers(5, 120, el.shape[0]).astype(np.float64)
    elif intensity == "lambert":
        cos_inc = np.clip(1.7 / rng_m, 0.02, 1.0)
        ground_i = np.clip((6.0 + 0.9 * rng_m) * cos_inc * rng.uniform(0.6, 1.6, el.shape[0]) * 8.0, 1, 250)
        wall_i = rng.integers(20, 200, el.shape[0]).astype(np.float64)
        inten = np.rint(np.where(on_ground, ground_i, wall_i))
    else:
        raise ValueError(intensity)
    ch = np.repeat(np.arange(layers), azimuths).astype(np.float64)
    return np.column_stack((x, y, z, inten, ch)).astype(dtype)


def firing_order(pc: np.ndarray, layers: int, azimuths: int) -> np.ndarray:
    """The rows of a channel-major sweep in the order the sensor fires them: azimuth-major, the ``layers`` channels of one
    azimuth step next to each other -- the row order of an STF ``.bin`` (precompute.py:78).  Same rows, another order."""
    return np.ascontiguousarray(pc.reshape(layers, azimuths, pc.shape[1]).transpose(1, 0, 2).reshape(-1, pc.shape[1]))
