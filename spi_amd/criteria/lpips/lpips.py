"""LPIPS-VGG16 on MI355X kernels.

Same surface as the reference's ``LPIPS`` (spi/criteria/lpips/lpips.py:32-71): ``LPIPS(net_type='vgg')(x, y)``
-> scalar; inputs larger than 256^2 are bilinearly reduced first; sum over the five taps of the
lin-weighted squared difference of channel-normalised features, spatial mean, divided by the batch.
Differences (result-identical): the normalise -> diff^2 -> 1x1 lin -> spatial-mean chain is one HIP
kernel per tap (``spi_lpips_layer_fwd/bwd``) instead of six tensor ops, and the features of a target
that does not change between iterations can be computed once (``features(y)`` / ``y_feats=``) --
the reference recomputes them every step (lpips.py:43).  Only ``x`` receives gradients.
"""
import torch
import torch.nn.functional as F

from ... import hip
from ...torch_utils import zero_arena
from .networks import VGG16, N_CHANNELS


class _LpipsTail(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fx, fy, lin):
        fx = fx.contiguous().float()
        fy = fy.contiguous().float()
        n, c, h, w = fx.shape
        out = zero_arena.zeros(n, fx.device)
        hip.call('spi_lpips_layer_fwd', hip.ptr(fx), hip.ptr(fy), hip.ptr(lin), n, c, h * w, hip.ptr(out), hip.stream())
        ctx.save_for_backward(fx, fy, lin)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        fx, fy, lin = ctx.saved_tensors
        n, c, h, w = fx.shape
        d_fx = torch.empty_like(fx)
        d_out = d_out.contiguous().float()                     # bound to a name: a temporary would be freed before the launch is enqueued
        hip.call('spi_lpips_layer_bwd', hip.ptr(fx), hip.ptr(fy), hip.ptr(lin), hip.ptr(d_out), n, c, h * w,
                 hip.ptr(d_fx), hip.stream())
        return d_fx, None, None


class LPIPS(torch.nn.Module):
    def __init__(self, net_type='vgg', version='0.1', weights=None, seed=0):
        assert version in ['0.1'] and net_type == 'vgg', 'SPI uses LPIPS v0.1 with the VGG16 backbone'
        super().__init__()
        self.net = VGG16(weights=weights, seed=seed)
        if weights is not None:
            lins = [l.reshape(-1) for l in weights['lins']]
        else:
            g = torch.Generator().manual_seed(seed + 1000)
            lins = [torch.rand(c, generator=g) / c for c in N_CHANNELS]
        for i, l in enumerate(lins):
            self.register_buffer(f'lin{i}', l.clone().float().contiguous())

    @staticmethod
    def _resize(x):
        if x.shape[-1] > 256:
            x = F.interpolate(x, size=(256, 256), mode='bilinear', align_corners=False)
        return x

    def features(self, y):
        """Detached tap activations of a (fixed) target image batch."""
        with torch.no_grad():
            return [f.detach() for f in self.net(self._resize(y.float()))]

    def forward(self, x, y=None, y_feats=None, sample_weights=None):
        """``sample_weights`` (extension, [N] tensor): returns sum_i w_i * d(x_i, y_i) instead of the batch mean -- several single-image
        LPIPS terms with their own weights (the mirror projector's `lpips(view) + lpips(mirror view) * weight_m`, mirror_projector.py:104)
        as ONE pass through the VGG (batch 2 fills the matrix cores better than two passes of batch 1)."""
        n = x.shape[0]
        fx = self.net(self._resize(x.float()))
        fy = y_feats if y_feats is not None else self.features(y)
        loss = 0.0
        for i, (a, b) in enumerate(zip(fx, fy)):
            d = _LpipsTail.apply(a, b, getattr(self, f'lin{i}'))
            loss = loss + ((d * sample_weights).sum() if sample_weights is not None else d.sum())
        return loss if sample_weights is not None else loss / n
