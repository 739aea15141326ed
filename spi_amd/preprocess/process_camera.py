"""EG3D camera label from a pose / intrinsics pair (mirror of preprocess/process_camera.py:9-62, mode 'orig')."""
import numpy as np


def fix_intrinsics(intrinsics):
    intrinsics = np.array(intrinsics).copy()
    assert intrinsics.shape == (3, 3), intrinsics
    intrinsics[0, 0] = 2985.29 / 700
    intrinsics[1, 1] = 2985.29 / 700
    intrinsics[0, 2] = 1 / 2
    intrinsics[1, 2] = 1 / 2
    assert intrinsics[0, 1] == 0 and intrinsics[2, 2] == 1 and intrinsics[1, 0] == 0 and intrinsics[2, 0] == 0 and intrinsics[2, 1] == 0
    return intrinsics


def fix_pose(pose):
    """centre of rotation (0, 0, 0.175) variant (:23-29, not used by run_total)"""
    cor = np.array([0, 0, 0.175])
    pose = np.array(pose).copy()
    location = pose[:3, 3]
    direction = (location - cor) / np.linalg.norm(location - cor)
    pose[:3, 3] = direction * 2.7 + cor
    return pose


def fix_pose_orig(pose):
    pose = np.array(pose).copy()
    location = pose[:3, 3]
    pose[:3, 3] = pose[:3, 3] / np.linalg.norm(location) * 2.7
    return pose


def flip_yaw(pose_matrix):
    flipped = np.array(pose_matrix).copy()
    flipped[0, 1] *= -1
    flipped[0, 2] *= -1
    flipped[1, 0] *= -1
    flipped[2, 0] *= -1
    flipped[0, 3] *= -1
    return flipped


def process_camera(pose, intrinsics):
    """-> 25 floats: the radius-2.7 cam2world matrix (16) and the normalised intrinsics (9)."""
    return np.concatenate([fix_pose_orig(pose).reshape(-1), fix_intrinsics(intrinsics).reshape(-1)])
