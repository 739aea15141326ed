"""MipRayMarcher2: alpha-composite ray march as one HIP kernel per direction.

Same surface as the reference's ``MipRayMarcher2`` (eg3d/training/volumetric_rendering/ray_marcher.py:25-62):
``forward(colors [N,M,S,C], densities [N,M,S,1], depths [N,M,S,1], rendering_options)
  -> (composite_rgb [N,M,C], composite_depth [N,M,1], weights [N,M,S-1,1])``.
One wave64 per ray; transmittance is a wave-shuffle product scan (csrc/render.hip).  Gradients flow to
colors and densities (depths are functions of the camera and of no-grad importance samples on this path).
Deviation: for a ray whose total weight is exactly 0 the reference's backward produces NaN
(0/0 in the depth normalisation); this kernel yields 0 there.
"""
import torch
from ... import hip


def depth_range(depths):
    """{min, max} over every depth sample, kept on the device (ray_marcher.py:50)."""
    out = torch.empty(2, device=depths.device, dtype=torch.float32)
    hip.call('spi_minmax', hip.ptr(depths), depths.numel(), hip.ptr(out), hip.stream())
    return out


class _RayMarch(torch.autograd.Function):
    @staticmethod
    def forward(ctx, colors, densities, depths, white_back):
        n, m, s, c = colors.shape
        colors = colors.contiguous().float()
        densities = densities.contiguous().float()
        depths = depths.detach().contiguous().float()
        r = n * m
        clamp2 = depth_range(depths)
        rgb = torch.empty(n, m, c, device=colors.device, dtype=torch.float32)
        depth = torch.empty(n, m, 1, device=colors.device, dtype=torch.float32)
        weights = torch.empty(n, m, s - 1, 1, device=colors.device, dtype=torch.float32)
        hip.call('spi_raymarch_fwd', hip.ptr(colors), hip.ptr(densities), hip.ptr(depths), None, hip.ptr(clamp2), r, s, s, c,
                 int(white_back), hip.ptr(rgb), hip.ptr(depth), hip.ptr(weights), None, hip.stream())
        ctx.save_for_backward(colors, densities, depths, clamp2)
        ctx.white_back = int(white_back)
        return rgb, depth, weights

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_rgb, d_depth, d_weights):
        colors, densities, depths, clamp2 = ctx.saved_tensors
        n, m, s, c = colors.shape
        d_colors = torch.empty_like(colors)
        d_dens = torch.empty_like(densities)
        d_rgb = d_rgb.contiguous().float() if d_rgb is not None else torch.zeros(n, m, c, device=colors.device)
        dd = d_depth.contiguous().float() if d_depth is not None else None
        dw = d_weights.contiguous().float() if d_weights is not None else None
        hip.call('spi_raymarch_bwd', hip.ptr(colors), hip.ptr(densities), hip.ptr(depths), None, hip.ptr(clamp2), hip.ptr(d_rgb),
                 hip.ptr(dd), hip.ptr(dw), n * m, s, s, c, ctx.white_back, hip.ptr(d_colors), None, hip.ptr(d_dens), None, hip.stream())
        return d_colors, d_dens, None, None


class MipRayMarcher2(torch.nn.Module):
    def run_forward(self, colors, densities, depths, rendering_options):
        if rendering_options.get('clamp_mode', 'softplus') != 'softplus':
            raise AssertionError('MipRayMarcher only supports `clamp_mode`=`softplus`!')
        c = colors.shape[-1]
        if c > 32:
            raise NotImplementedError(f'MipRayMarcher2: at most 32 colour channels (one 128-B row per sample), got {c}')
        if c < 32:                                               # a decoder with fewer output channels: zero-pad the rows, drop the padding again
            colors = torch.nn.functional.pad(colors, (0, 32 - c))
        rgb, depth, weights = _RayMarch.apply(colors, densities, depths, bool(rendering_options.get('white_back', False)))
        return (rgb[..., :c] if c < 32 else rgb), depth, weights

    def forward(self, colors, densities, depths, rendering_options):
        return self.run_forward(colors, densities, depths, rendering_options)
