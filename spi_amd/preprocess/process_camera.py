"""EG3D camera label from a pose / intrinsics pair -- the surface of preprocess/process_camera.py:9-62 (mode 'orig', the one run_total uses).

EG3D's FFHQ convention: the camera sits on a sphere of radius 2.7 around the origin, and the intrinsics are normalised by the 700-pixel crop
(focal 2985.29 / 700, principal point at the centre).  ``process_camera`` returns the 25-float label ``[cam2world (16), K (9)]`` that
``PTIDataset`` reads back as ``c``.
"""
import numpy as np

RADIUS = 2.7
FOCAL = 2985.29 / 700
_YAW_FLIP = ((0, 1), (0, 2), (1, 0), (2, 0), (0, 3))        # entries of a cam2world matrix that change sign under a left-right mirror


def fix_intrinsics(intrinsics):
    K = np.array(intrinsics, copy=True)
    if K.shape != (3, 3):
        raise AssertionError(K)
    off = [K[0, 1], K[1, 0], K[2, 0], K[2, 1]]
    if any(v != 0 for v in off) or K[2, 2] != 1:
        raise AssertionError('expected a skew-free pinhole matrix')
    K[0, 0] = K[1, 1] = FOCAL
    K[0, 2] = K[1, 2] = 0.5
    return K


def _on_sphere(pose, centre):
    out = np.array(pose, copy=True)
    ray = out[:3, 3] - centre
    out[:3, 3] = centre + ray * (RADIUS / np.linalg.norm(ray))
    return out


def fix_pose(pose):
    """variant with the centre of rotation at (0, 0, 0.175) (:23-29; not used by run_total)"""
    return _on_sphere(pose, np.array([0, 0, 0.175]))


def fix_pose_orig(pose):
    out = np.array(pose, copy=True)
    out[:3, 3] = out[:3, 3] / np.linalg.norm(out[:3, 3]) * RADIUS        # (this order of operations: the label is compared bit for bit)
    return out


def flip_yaw(pose_matrix):
    out = np.array(pose_matrix, copy=True)
    for i, j in _YAW_FLIP:
        out[i, j] = -out[i, j]
    return out


def process_camera(pose, intrinsics):
    return np.concatenate([fix_pose_orig(pose).ravel(), fix_intrinsics(intrinsics).ravel()])
