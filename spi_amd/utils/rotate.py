"""Depth-guided warp of a source view into a target view -- one HIP kernel.

Same surface as the reference's ``rotate`` (spi/utils/rotate.py:92-116): returns ``(new_rgb [N,3,R,R],
depth_mask [N,1,R,R])`` where R is the source image resolution; the 128^2 depth maps are resized
bilinearly inside the kernel.  No gradients (the reference calls it under ``no_grad``).
"""
import torch
from .. import hip


@torch.no_grad()
def rotate(target_camera, target_depth, src_image, src_camera, src_depth, src_mask=None, EPS=5e-2, src_cam2world_inv=None):
    """src_cam2world_inv [n,16]: optional precomputed inverse of the source extrinsics (constant per image in SPI's loop; `torch.inverse`
    synchronises with the host and cannot be part of a captured HIP graph)."""
    n = src_image.shape[0]
    res = src_image.shape[-1]
    dres = target_depth.shape[-1]
    dev = src_image.device
    tgt = target_camera.reshape(n, 25).float().contiguous()
    src = src_camera.reshape(n, 25).float().contiguous()
    src_inv = (torch.inverse(src[:, :16].reshape(n, 4, 4)).reshape(n, 16) if src_cam2world_inv is None
               else src_cam2world_inv.reshape(n, 16).float()).contiguous()
    td = target_depth.reshape(n, dres, dres).float().contiguous()
    sd = src_depth.reshape(n, dres, dres).float().contiguous()
    img = src_image.float().contiguous()
    msk = src_mask.reshape(n, res, res).float().contiguous() if src_mask is not None else None
    rgb = torch.empty(n, 3, res, res, device=dev, dtype=torch.float32)
    mask = torch.empty(n, 1, res, res, device=dev, dtype=torch.float32)
    hip.call('spi_rotate_warp', hip.ptr(tgt), hip.ptr(src_inv), hip.ptr(src), hip.ptr(td), hip.ptr(sd), hip.ptr(img), hip.ptr(msk),
             n, res, dres, float(EPS), hip.ptr(rgb), hip.ptr(mask), hip.stream())
    return rgb, mask
