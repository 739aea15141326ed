"""filtered_lrelu on HIP: bias -> up-FIR -> leaky-ReLU * gain, clamp -> down-FIR.

Same surface as the reference's ``filtered_lrelu`` (eg3d/torch_utils/ops/filtered_lrelu.py:58-118).
The op is on the reference's *import* path only (networks_stylegan3, never instantiated by the
StyleGAN2 / 8XDC generator), so this is a forward-only two-pass kernel sequence with no global
device state (the reference's constant-memory filter buffer makes it non-reentrant across streams,
filtered_lrelu.cu:81-82).  2-D filters or None only.
"""
import math
import torch
from ... import hip
from .upfirdn2d import _parse_padding


def filtered_lrelu(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=math.sqrt(2), slope=0.2, clamp=None,
                   flip_filter=False, impl='hip'):
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    if x.requires_grad or (b is not None and b.requires_grad):
        raise NotImplementedError('filtered_lrelu backward is not implemented (the op is never executed on the SPI path)')
    one = torch.ones(1, 1, device=x.device)
    fu = one if fu is None else fu.to(x.device).float().contiguous()
    fd = one if fd is None else fd.to(x.device).float().contiguous()
    if fu.ndim != 2 or fd.ndim != 2:
        raise NotImplementedError('separable (1-D) filters are not supported by the HIP filtered_lrelu')
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
