"""Shared by the CPU (oracle) and GPU (HIP) tests of the ImportanceRenderer options: the variants of golden/renderer_options.npz."""
import torch

RENDER_VARIANTS = dict(
    auto=dict(ray_start='auto', ray_end='auto', box_warp=0.45),
    disparity=dict(disparity_space_sampling=True),
    dnoise=dict(density_noise=0.7),
    tiny=dict(),
    tiny_auto_dnoise=dict(ray_start='auto', ray_end='auto', box_warp=0.45, density_noise=0.3))


def tiny_decoder(g, feats, dirs):
    """The golden file's non-OSG decoder (tests/golden/make_golden.py tiny_decoder): uses the ray directions, 4 colour channels."""
    h = torch.tanh(feats.mean(1) @ g['tiny_A'].to(feats.device) + dirs @ g['tiny_B'].to(feats.device))
    y = h @ g['tiny_C'].to(feats.device)
    return torch.sigmoid(y[..., 1:]), y[..., 0:1]
