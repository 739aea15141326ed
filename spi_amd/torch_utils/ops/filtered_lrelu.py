"""filtered_lrelu on HIP: bias -> up-FIR -> leaky-ReLU * gain, clamp -> down-FIR.

Same surface as the reference's ``filtered_lrelu`` (eg3d/torch_utils/ops/filtered_lrelu.py:58-118).
The op is on the reference's *import* path only (networks_stylegan3, never instantiated by the
StyleGAN2 / 8XDC generator).  Without gradients it is the fused two-pass kernel sequence
``spi_filtered_lrelu`` (no global device state: the reference's constant-memory filter buffer makes it
non-reentrant across streams, filtered_lrelu.cu:81-82).  When a gradient is required -- or the filters
are separable 1-D -- it runs as the reference's own decomposition (filtered_lrelu.py:121-176:
bias -> upfirdn2d(up) -> bias_act(lrelu, gain, clamp) -> upfirdn2d(down)) on the differentiable HIP ops,
so the backward is their adjoint kernels.
"""
import math
import torch
from ... import hip
from .upfirdn2d import _parse_padding


def filtered_lrelu(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=math.sqrt(2), slope=0.2, clamp=None,
                   flip_filter=False, impl='hip'):
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    need_grad = torch.is_grad_enabled() and (x.requires_grad or (b is not None and b.requires_grad))
    separable = (fu is not None and fu.ndim == 1 and fu.numel() > 1) or (fd is not None and fd.ndim == 1 and fd.numel() > 1)
    if need_grad or separable:
        from . import bias_act as _ba, upfirdn2d as _uf
        y = _ba.bias_act(x, b) if b is not None else x
        y = _uf.upfirdn2d(y, fu, up=up, padding=_parse_padding(padding), gain=up ** 2, flip_filter=flip_filter)
        y = _ba.bias_act(y, act='lrelu', alpha=slope, gain=gain, clamp=clamp)
        return _uf.upfirdn2d(y, fd, down=down, flip_filter=flip_filter)
    one = torch.ones(1, 1, device=x.device)
    fu = one if fu is None else fu.to(x.device).float().contiguous()
    fd = one if fd is None else fd.to(x.device).float().contiguous()
    if fu.ndim == 1:
        fu = fu.reshape(1, 1) ** 2 if fu.numel() == 1 else fu
    if fd.ndim == 1:
        fd = fd.reshape(1, 1) ** 2 if fd.numel() == 1 else fd
    px0, px1, py0, py1 = _parse_padding(padding)
    n, c, ih, iw = x.shape
    mid_h = ih * up + py0 + py1 - fu.shape[0] + 1
    mid_w = iw * up + px0 + px1 - fu.shape[1] + 1
    oh = (mid_h - fd.shape[0] + down) // down
    ow = (mid_w - fd.shape[1] + down) // down
    x = x.contiguous().float()
    bb = b.contiguous().float() if b is not None else None
    tmp = torch.empty(n, c, mid_h, mid_w, device=x.device, dtype=torch.float32)
    y = torch.empty(n, c, oh, ow, device=x.device, dtype=torch.float32)
    hip.call('spi_filtered_lrelu', hip.ptr(x), hip.ptr(fu), hip.ptr(fd), hip.ptr(bb), hip.ptr(tmp), hip.ptr(y), n, c, ih, iw,
             fu.shape[0], fu.shape[1], fd.shape[0], fd.shape[1], int(up), int(down), px0, px1, py0, py1, float(gain), float(slope),
             float(-1 if clamp is None else clamp), int(flip_filter), oh, ow, hip.stream())
    return y
