"""Ray / box geometry helpers of the renderer (surface of eg3d/training/volumetric_rendering/math_utils.py).

Only ``get_ray_limits_box`` and ``linspace`` are used by the renderer -- by its ``ray_start = ray_end = 'auto'`` branch
(renderer.py:91-97), which the FFHQ configuration SPI runs never takes; plain tensor ops on the rays' device.
"""
import torch


def transform_vectors(matrix, vectors4):
    """Left-multiplies MxM @ NxM -> NxM (math_utils.py:26-31)."""
    return vectors4 @ matrix.T


def normalize_vecs(vectors):
    """Unit length along the last axis (math_utils.py:34-38)."""
    return vectors / torch.norm(vectors, dim=-1, keepdim=True)


def torch_dot(x, y):
    return (x * y).sum(-1)


def get_ray_limits_box(rays_o, rays_d, box_side_length):
    """Slab test of rays [..., 3] against the axis-aligned cube of side ``box_side_length`` centred at the origin
    (math_utils.py:44-94) -> (t_near [..., 1], t_far [..., 1]); a ray that misses the cube gets (-1, -2)."""
    shape = rays_o.shape
    o = rays_o.detach().reshape(-1, 3)
    d = rays_d.detach().reshape(-1, 3)
    half = box_side_length / 2
    inv = 1 / d
    neg = inv < 0
    # per axis: entry through the face on the side the ray comes from, exit through the opposite one
    near = (torch.where(neg, half, -half) - o) * inv
    far = (torch.where(neg, -half, half) - o) * inv
    tmin, tmax = near[:, 0], far[:, 0]
    valid = torch.ones_like(tmin, dtype=torch.bool)
    for ax in (1, 2):
        valid &= ~((tmin > far[:, ax]) | (near[:, ax] > tmax))
        tmin = torch.maximum(tmin, near[:, ax])
        tmax = torch.minimum(tmax, far[:, ax])
    tmin = torch.where(valid, tmin, torch.full_like(tmin, -1))
    tmax = torch.where(valid, tmax, torch.full_like(tmax, -2))
    return tmin.reshape(*shape[:-1], 1), tmax.reshape(*shape[:-1], 1)


def linspace(start, stop, num):
    """[num, *start.shape] evenly spaced from start to stop inclusive, per element (math_utils.py:98-118)."""
    steps = torch.arange(num, dtype=torch.float32, device=start.device) / (num - 1)
    steps = steps.reshape(-1, *([1] * start.ndim))
    return start[None] + steps * (stop - start)[None]
