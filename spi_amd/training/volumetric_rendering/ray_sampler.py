"""RaySampler: camera matrices -> ray origins / directions, one HIP launch.

Same surface as the reference's ``RaySampler`` (eg3d/training/volumetric_rendering/ray_sampler.py:24-63):
``forward(cam2world_matrix [N,4,4], intrinsics [N,3,3], resolution) -> (ray_origins, ray_dirs) [N, res^2, 3]``.
No gradients flow to the camera (none are needed on the inversion path).
"""
import torch
from ... import hip


class RaySampler(torch.nn.Module):
    def forward(self, cam2world_matrix, intrinsics, resolution):
        n = cam2world_matrix.shape[0]
        c2w = cam2world_matrix.detach().reshape(n, 16).float().contiguous()
        k = intrinsics.detach().reshape(n, 9).float().contiguous()
        m = resolution * resolution
        ray_o = torch.empty(n, m, 3, device=c2w.device, dtype=torch.float32)
        ray_d = torch.empty_like(ray_o)
        hip.call('spi_ray_sampler', hip.ptr(c2w), hip.ptr(k), n, int(resolution), hip.ptr(ray_o), hip.ptr(ray_d), hip.stream())
        return ray_o, ray_d
