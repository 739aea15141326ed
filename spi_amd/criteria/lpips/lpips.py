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


class _LpipsAll(torch.autograd.Function):
    """sum over the tap layers of the per-layer distance (spi_lpips_layer_fwd: normalised feature difference x linear weights, spatial mean) -> [N]: every layer's kernel adds its per-sample distance into the same zeroed accumulator
    (the kernels already finish with one atomic per block), the backward hands the same d_out to every layer's kernel."""

    @staticmethod
    def forward(ctx, nl, *t):
        fx = [a.contiguous().float() for a in t[:nl]]
        fy = [b.contiguous().float() for b in t[nl:2 * nl]]
        lins = list(t[2 * nl:])
        n = fx[0].shape[0]
        out = zero_arena.zeros(n, fx[0].device)
        for a, b, lin in zip(fx, fy, lins):
            _, c, h, w = a.shape
            hip.call('spi_lpips_layer_fwd', hip.ptr(a), hip.ptr(b), hip.ptr(lin), n, c, h * w, hip.ptr(out), hip.stream())
        ctx.nl = nl
        ctx.save_for_backward(*fx, *fy, *lins)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        nl = ctx.nl
        t = ctx.saved_tensors
        d_out = d_out.contiguous().float()                     # bound to a name: a temporary would be freed before the launches are enqueued
        grads = []
        for a, b, lin in zip(t[:nl], t[nl:2 * nl], t[2 * nl:]):
            n, c, h, w = a.shape
            d_fx = torch.empty_like(a)
            hip.call('spi_lpips_layer_bwd', hip.ptr(a), hip.ptr(b), hip.ptr(lin), hip.ptr(d_out), n, c, h * w, hip.ptr(d_fx), hip.stream())
            grads.append(d_fx)
        return (None, *grads, *([None] * (2 * nl)))


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
        # the five tap layers' distances land in ONE per-sample accumulator (round 6: 5 x (weight, sum, add) scalar launches and their backward twins
        # were ~20 launch boundaries per LPIPS call, each ~8 us of an iteration that is bound by them at its two ends)
        d = _LpipsAll.apply(len(fx), *fx, *fy, *[getattr(self, f'lin{i}') for i in range(len(fx))])
        return (d * sample_weights).sum() if sample_weights is not None else d.sum() / n
