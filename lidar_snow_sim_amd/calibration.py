"""KITTI-style calibration for the optional camera-FOV crop of augment() (simulation.py:32-47, :532-540).

The reference takes this from the un-vendored OpenPCDet fork (pcdet.utils.calibration_kitti.Calibration)
and a data file that is not in its tree (lib/OpenPCDet/data/dense/calib_hdl64.txt), so parity for this
step is UNPINNED: what is here restates that module's published lidar_to_rect / rect_to_img (P2, R0_rect, Tr_velo_to_cam;
image coordinates divided by the rectified z, depth = third homogeneous coordinate - P2[2, 3]).
"""
from pathlib import Path

import numpy as np

_default_calib_file = None


def set_calib_file(path):
    global _default_calib_file
    _default_calib_file = None if path is None else Path(path)


class Calibration:
    def __init__(self, calib_file=None, P2=None, R0=None, V2C=None):
        if calib_file is not None:
            vals = {}
            for line in Path(calib_file).read_text().splitlines():
                if ':' in line:
                    k, v = line.split(':', 1)
                    vals[k.strip()] = np.array(v.split(), dtype=np.float32)
            P2 = vals['P2'].reshape(3, 4)
            R0 = vals['R0_rect'].reshape(3, 3)
            V2C = vals['Tr_velo_to_cam'].reshape(3, 4)
        self.P2, self.R0, self.V2C = np.asarray(P2), np.asarray(R0), np.asarray(V2C)

    def lidar_to_rect(self, pts_lidar):
        hom = np.hstack((pts_lidar, np.ones((pts_lidar.shape[0], 1), dtype=np.float32)))
        return np.dot(hom, np.dot(self.V2C.T, self.R0.T))

    def rect_to_img(self, pts_rect):
        hom = np.hstack((pts_rect, np.ones((pts_rect.shape[0], 1), dtype=np.float32)))
        pts_2d = np.dot(hom, self.P2.T)
        pts_img = (pts_2d[:, 0:2].T / hom[:, 2]).T            # OpenPCDet: by the rectified point's z (not the projected one: P2[2, 3] != 0 in KITTI files)
        depth = pts_2d[:, 2] - self.P2.T[3, 2]
        return pts_img, depth


def get_calib(sensor: str = 'hdl64'):
    """simulation.py:32-36: asserts that the calibration file exists."""
    calib_file = _default_calib_file or (Path(__file__).resolve().parent / 'data' / f'calib_{sensor}.txt')
    assert calib_file.exists(), f'{calib_file} not found'
    return Calibration(calib_file)


def get_fov_flag(pts_rect, img_shape, calib):
    """simulation.py:39-47"""
    pts_img, depth = calib.rect_to_img(pts_rect)
    in_x = np.logical_and(pts_img[:, 0] >= 0, pts_img[:, 0] < img_shape[1])
    in_y = np.logical_and(pts_img[:, 1] >= 0, pts_img[:, 1] < img_shape[0])
    return np.logical_and(np.logical_and(in_x, in_y), depth >= 0)
