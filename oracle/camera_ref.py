"""Oracle (CPU) for the camera helpers the SPI loops call.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Written independently of spi_amd/utils/camera_utils.py (round-1 review: the oracle
loops imported the product's module, so a camera bug would have cancelled in the stage-2 comparison): per-sample, numpy float64 where the
reference uses numpy, following (relative to /root/reference/spi/utils/camera_utils.py)
  LookAtPoseSampler.sample ('uniform')   :70-93      create_cam2world_matrix   :125-144
  sample_camera                          :159-166    angle_to_rotation         :169-193
  sample_surrounding_camera              :196-211    flip_yaw / cal_mirror_c   :336-350
  rotation_to_angle                      :353-364    gauss_function            :385-387
  cal_camera_weight                      :398-411
Pinned against the reference's outputs in tests/golden/geometry.npz (tests/test_oracle_cpu.py).  Random draws are injected (`rand=`).
"""
import math
import numpy as np
import torch


def _unit(v):
    return v / v.norm(dim=-1, keepdim=True)


def _cam2world(forward, origin):
    fwd = _unit(forward)
    up0 = torch.tensor([0.0, 1.0, 0.0]).expand_as(fwd)
    right = -_unit(torch.cross(up0, fwd, dim=-1))
    up = _unit(torch.cross(fwd, right, dim=-1))
    out = []
    for b in range(fwd.shape[0]):
        rot = torch.eye(4)
        rot[:3, 0], rot[:3, 1], rot[:3, 2] = right[b], up[b], fwd[b]
        tr = torch.eye(4)
        tr[:3, 3] = origin[b]
        out.append(tr @ rot)
    return torch.stack(out)


def _look_at(h, v, lookat, radius):
    v = torch.clamp(v, 1e-5, math.pi - 1e-5)
    theta, phi = h, torch.arccos(1 - 2 * (v / math.pi))
    org = torch.zeros(h.shape[0], 3)
    org[:, 0:1] = radius * torch.sin(phi) * torch.cos(math.pi - theta)
    org[:, 2:3] = radius * torch.sin(phi) * torch.sin(math.pi - theta)
    org[:, 1:2] = radius * torch.cos(phi)
    return _cam2world(_unit(lookat - org), org)


_K = [4.2647, 0.0, 0.5, 0.0, 4.2647, 0.5, 0.0, 0.0, 1.0]


def sample_camera(batch_size, yaw_range, pitch_range, rand):
    """'uniform' sampling: mean + U[0,1) * range for both angles (the reference's sample_mode='uniform' branch, :78-80)."""
    rh, rv = rand
    h = rh.reshape(batch_size, 1) * yaw_range + np.pi / 2
    v = rv.reshape(batch_size, 1) * pitch_range + (np.pi / 2 - 0.2)
    ext = _look_at(h.float(), v.float(), torch.tensor([0.0, 0.0, 0.2]), 2.7)
    return torch.cat([ext.reshape(batch_size, 16), torch.tensor(_K).reshape(1, 9).repeat(batch_size, 1)], dim=1)


def _rotation(yaw, pitch):
    """np.matrix product yaw * pitch * roll(0), float64 (:169-193)"""
    ry = np.array([[math.cos(yaw), 0, math.sin(yaw)], [0, 1, 0], [-math.sin(yaw), 0, math.cos(yaw)]])
    rx = np.array([[1, 0, 0], [0, math.cos(pitch), -math.sin(pitch)], [0, math.sin(pitch), math.cos(pitch)]])
    return ry @ rx @ np.eye(3)


def sample_surrounding_camera(middle_camera, batch_size, yaw_range, pitch_range, rand):
    ry, rp = rand
    y = (ry.reshape(batch_size, 1) * 2 - 1) * yaw_range + 0.0
    p = (rp.reshape(batch_size, 1) * 2 - 1) * pitch_range + 0.0
    # float64 rotations rounded to fp32, then ONE batched product with rows 0-2 of the pose (:203-208; the batched product's summation order
    # is what the reference's result carries -- a per-sample `@` differs in the last bit)
    rot = torch.stack([torch.from_numpy(_rotation(float(y[b]), float(p[b]))) for b in range(batch_size)]).float()
    out = middle_camera.reshape(1, 25).repeat(batch_size, 1).clone()
    ext = out[:, :16].reshape(batch_size, 4, 4).clone()
    ext[:, :3] = torch.bmm(rot, ext[:, :3])
    out[:, :16] = ext.reshape(batch_size, 16)
    return out


def cal_mirror_c(camera):
    out = camera.clone().reshape(-1, 25)
    for b in range(out.shape[0]):
        pose = out[b, :16].reshape(4, 4).clone()
        for (i, j) in ((0, 1), (0, 2), (0, 3), (1, 0), (2, 0)):
            pose[i, j] = -pose[i, j]
        out[b, :16] = pose.reshape(16)
    return out


def _yaw_of(c):
    m = c.reshape(25)[:16].reshape(4, 4)[:3, :3]
    pitch = torch.arctan(-m[1, 2] / m[2, 2])
    return torch.arctan(m[0, 2] * torch.cos(pitch) / m[2, 2])


def _gauss(x, std):
    return torch.exp(-0.5 * x * x / std / std) / (std * math.sqrt(2 * math.pi))


def cal_camera_weight(camera):
    ws = []
    for c in camera.reshape(-1, 25):
        y = torch.abs(_yaw_of(c))
        w = (1 - _gauss(y, 0.29) / 2.7) / 2
        ws.append(torch.zeros_like(w) if y < 0.2 else w)
    return torch.stack(ws)
