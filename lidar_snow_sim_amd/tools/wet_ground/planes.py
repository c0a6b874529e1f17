"""Ground-plane estimate -- mirror of tools/wet_ground/planes.py::calculate_plane (planes.py:12-50).

Host logic.  The reference fits z = c0 x + c1 y + b with scikit-learn's RANSACRegressor on a
front-of-car crop and falls back to the flat-earth plane ([0, 0, 1], -1.55) when the crop is tiny or the
fit raises; with scikit-learn >= 1.2 the `loss='squared_loss'` spelling it passes always raises, so the
fallback is what the reference returns today.  That behaviour is reproduced literally (same call,
same except), which keeps `augment(...)` a drop-in in any environment; pass ``plane=(w, h)`` to the
augmentation entry points for a deterministic plane of your own.
"""
import numpy as np

STANDARD_HEIGHT = -1.55


def ground_crop(pointcloud):
    """planes.py:21-27"""
    return ((pointcloud[:, 2] < -1.55) & (pointcloud[:, 2] > -1.86 - 0.01 * pointcloud[:, 0])
            & (pointcloud[:, 0] > 10) & (pointcloud[:, 0] < 70) & (pointcloud[:, 1] > -3) & (pointcloud[:, 1] < 3))


def calculate_plane(pointcloud, standart_height=STANDARD_HEIGHT):
    """Returns (w, h): plane normal and lidar height (planes.py:12-50)."""
    pc_rect = pointcloud[ground_crop(pointcloud)]
    if pc_rect.shape[0] <= pc_rect.shape[1]:                            # planes.py:29-32
        return [0, 0, 1], standart_height
    try:                                                                # planes.py:34-41
        from sklearn.linear_model import RANSACRegressor
        reg = RANSACRegressor(loss='squared_loss', max_trials=1000).fit(pc_rect[:, [0, 1]], pc_rect[:, 2])
        w = np.zeros(3)
        w[0] = reg.estimator_.coef_[0]
        w[1] = reg.estimator_.coef_[1]
        w[2] = -1.0
        h = reg.estimator_.intercept_
        w = w / np.linalg.norm(w)
    except Exception:                                                   # planes.py:43-48
        print('Was not able to estimate a ground plane. Using default flat earth assumption')
        w = [0, 0, 1]
        h = standart_height
    return w, h
