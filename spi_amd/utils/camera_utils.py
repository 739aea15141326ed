"""Camera pose helpers for the inversion loop (host-side geometry, no kernels).

Mirrors the call surface of the reference's ``spi/utils/camera_utils.py``:
  sample_camera            :159-166   (LookAtPoseSampler 'uniform' mode :70-93, create_cam2world :125-144)
  sample_surrounding_camera :196-211  (angle_to_rotation :169-193)
  cal_canonical_c          :233-240
  flip_yaw / cal_mirror_c  :336-350
  rotation_to_angle        :353-364
  gauss_function           :385-387
  cal_camera_gauss_weight  :389-395
  cal_camera_weight        :398-411

Design differences (MI355X-first): everything is batched tensor math on the camera's own
device -- no per-sample numpy loops and no host syncs inside the stage-2 loop.  Random draws can
be injected (``rand=``) so tests can replay the reference's draws; by default they come from
torch's generator in the reference's order (yaw then pitch, each ``[B,1]``).
"""
import math
import torch

_SQRT_2PI = math.sqrt(2.0 * math.pi)
FOCAL = 4.2647
RADIUS = 2.7
LOOKAT = (0.0, 0.0, 0.2)
PITCH_BIAS = -0.2


def _normalize(v):
    return v / torch.norm(v, dim=-1, keepdim=True)


def create_cam2world_matrix(forward, origin):
    forward = _normalize(forward)
    up = _const([0.0, 1.0, 0.0], origin.device).expand_as(forward)
    right = -_normalize(torch.cross(up, forward, dim=-1))
    up = _normalize(torch.cross(forward, right, dim=-1))
    b = forward.shape[0]
    rot = torch.eye(4, device=origin.device).unsqueeze(0).repeat(b, 1, 1)
    rot[:, :3, :3] = torch.stack((right, up, forward), dim=-1)
    trans = torch.eye(4, device=origin.device).unsqueeze(0).repeat(b, 1, 1)
    trans[:, :3, 3] = origin
    return trans @ rot


_CONSTS = {}


def _const(values, device):
    """Small constant tensor on `device`, uploaded once (a host -> device copy cannot be part of a captured HIP graph, and the samplers run
    inside the stage-2 iteration)."""
    key = (tuple(map(tuple, values)) if isinstance(values[0], (list, tuple)) else tuple(values), str(device))
    t = _CONSTS.get(key)
    if t is None:
        t = _CONSTS[key] = torch.tensor(values, dtype=torch.float32, device=device)
    return t


def look_at_pose(h, v, lookat, radius):
    """h, v: [B,1] azimuth / polar angle (radians)."""
    v = torch.clamp(v, 1e-5, math.pi - 1e-5)
    phi = torch.arccos(1 - 2 * (v / math.pi))
    origins = torch.zeros((h.shape[0], 3), device=h.device)
    origins[:, 0:1] = radius * torch.sin(phi) * torch.cos(math.pi - h)
    origins[:, 2:3] = radius * torch.sin(phi) * torch.sin(math.pi - h)
    origins[:, 1:2] = radius * torch.cos(phi)
    return create_cam2world_matrix(_normalize(lookat - origins), origins)


def _intrinsics(batch_size, device):
    return _const([[FOCAL, 0, 0.5], [0, FOCAL, 0.5], [0, 0, 1]], device).view(1, 9).repeat(batch_size, 1)


def sample_camera(batch_size=1, yaw_range=0.35, pitch_range=0.25, device='cpu', rand=None):
    """Uniform look-at poses: h = U[0,1)*yaw_range + pi/2, v = U[0,1)*pitch_range + pi/2 - 0.2."""
    if rand is None:
        rh = torch.rand((batch_size, 1), device=device)
        rv = torch.rand((batch_size, 1), device=device)
    else:
        rh, rv = rand
    lookat = _const(LOOKAT, device)
    ext = look_at_pose(rh * yaw_range + math.pi / 2, rv * pitch_range + (math.pi / 2 + PITCH_BIAS), lookat, RADIUS)
    return torch.cat([ext.reshape(-1, 16), _intrinsics(batch_size, device)], dim=1)


def cal_canonical_c(yaw_angle=0, pitch_angle=0, batch_size=1, device='cpu'):
    lookat = _const(LOOKAT, device)
    h = torch.full((batch_size, 1), math.pi / 2 + yaw_angle, device=device)
    v = torch.full((batch_size, 1), math.pi / 2 + PITCH_BIAS + pitch_angle, device=device)
    ext = look_at_pose(h, v, lookat, RADIUS)
    return torch.cat([ext.reshape(-1, 16), _intrinsics(batch_size, device)], dim=1)


def angle_to_rotation(yaw, pitch):
    """Batched R = R_y(yaw) @ R_x(pitch) (roll = 0), evaluated in float64 like the reference."""
    y, p = yaw.double().reshape(-1), pitch.double().reshape(-1)
    cy, sy, cp, sp = torch.cos(y), torch.sin(y), torch.cos(p), torch.sin(p)
    z, o = torch.zeros_like(y), torch.ones_like(y)
    ry = torch.stack([cy, z, sy, z, o, z, -sy, z, cy], dim=-1).reshape(-1, 3, 3)
    rx = torch.stack([o, z, z, z, cp, -sp, z, sp, cp], dim=-1).reshape(-1, 3, 3)
    return ry @ rx


def sample_surrounding_camera(middle_camera, batch_size=1, yaw_range=0.1, pitch_range=0.1, rand=None):
    """Left-multiplies a small random yaw/pitch rotation onto rows 0..2 of cam2world (rotation AND
    translation columns, as the reference does)."""
    device = middle_camera.device
    if rand is None:
        ry = torch.rand((batch_size, 1), device=device)
        rp = torch.rand((batch_size, 1), device=device)
    else:
        ry, rp = rand
    y = (ry * 2 - 1) * yaw_range + 0.0
    p = (rp * 2 - 1) * pitch_range + 0.0
    rot = angle_to_rotation(y, p).float().to(device)
    cam = middle_camera.repeat(batch_size, 1).clone()
    ext = cam[:, :16].reshape(-1, 4, 4).clone()
    ext[:, :3] = torch.bmm(rot, ext[:, :3])
    cam[:, :16] = ext.reshape(-1, 16)
    return cam


def flip_yaw(pose):
    flipped = pose.clone()
    flipped[:, 0, 1] *= -1
    flipped[:, 0, 2] *= -1
    flipped[:, 0, 3] *= -1
    flipped[:, 1, 0] *= -1
    flipped[:, 2, 0] *= -1
    return flipped


def cal_mirror_c(camera):
    pose = camera[:, :16].reshape(-1, 4, 4)
    return torch.cat([flip_yaw(pose).reshape(-1, 16), camera[:, 16:].reshape(-1, 9)], dim=1)


def rotation_to_angle(matrix):
    """matrix [..., 3, 3] -> yaw, pitch, roll."""
    pitch = torch.arctan(-matrix[..., 1, 2] / matrix[..., 2, 2])
    yaw = torch.arctan(matrix[..., 0, 2] * torch.cos(pitch) / matrix[..., 2, 2])
    roll = torch.arctan(-matrix[..., 0, 1] / matrix[..., 0, 0])
    return yaw, pitch, roll


def gauss_function(x, mean=0.0, std=0.25):
    return torch.exp(-0.5 * (x - mean) * (x - mean) / std / std) / (std * _SQRT_2PI)


def cal_camera_gauss_weight(camera):
    yaw, _, _ = rotation_to_angle(camera.reshape(-1, 25)[:, :16].reshape(-1, 4, 4)[:, :3, :3])
    w = gauss_function(yaw, std=0.4) / 2.6
    return [wi for wi in w]


def cal_camera_weight(camera):
    """Mirror-view weight: (1 - N(|yaw|; 0, 0.29)/2.7)/2, zero when |yaw| < 0.2.  Returns [B]."""
    yaw, _, _ = rotation_to_angle(camera.reshape(-1, 25)[:, :16].reshape(-1, 4, 4)[:, :3, :3])
    y = yaw.abs()
    w = (1 - gauss_function(y, std=0.29) / 2.7) / 2
    return torch.where(y < 0.2, torch.zeros_like(w), w)
