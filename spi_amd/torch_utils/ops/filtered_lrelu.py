"""filtered_lrelu on HIP: bias -> up-FIR -> leaky-ReLU * gain, clamp -> down-FIR.

Same surface as the reference's ``filtered_lrelu`` (eg3d/torch_utils/ops/filtered_lrelu.py:58-118).
The op is on the reference's *import* path only (networks_stylegan3, never instantiated by the
StyleGAN2 / 8XDC generator).  Since round 3 it is ONE launch either way, ``spi_filtered_lrelu_fused`` (csrc/flrelu.hip: up-FIR, activation and down-FIR through LDS like
the reference's kernel, filtered_lrelu.cu:119-1105; no upsampled tensor in memory; no global device state -- the reference's
constant-memory filter buffer makes it non-reentrant across streams, :81-82).  When a gradient is required the launch also writes the
bit-packed SIGN tensor (2 bits per upsampled sample: negative / clamped), like the reference's plugin op (filtered_lrelu.py:180-270);
only the filters and the signs are kept for the backward, which is the same launch with up / down swapped, flipped filters,
gain * up^2 / down^2, no clamp, and the signs READ at the offset the reference computes (:258-262) -- no full-size activation is stored.
(The two-pass ``spi_filtered_lrelu`` and the stand-alone activation ``spi_filtered_lrelu_act`` stay exported.)
``filtered_lrelu_act_(x, si, sx, sy, gain, slope, clamp, write_signs)`` is exported with the plugin's meaning.
Separable 1-D filters take the reference's decomposition on the differentiable HIP ops (:121-176).
"""
import math
import torch
from ... import hip
from .upfirdn2d import _parse_padding


def filtered_lrelu_act_(x, si=None, sx=0, sy=0, gain=math.sqrt(2), slope=0.2, clamp=None, write_signs=False):
    """Plugin entry point `filtered_lrelu_act_` (filtered_lrelu.cpp:217-296): modifies ``x`` [N,C,H,W] in place; returns the sign tensor it
    wrote (``write_signs``) or an empty uint8 tensor.  ``si``: signs to read (gradient pass) -- exclusive with ``write_signs``."""
    assert x.ndim == 4 and x.is_contiguous() and x.dtype == torch.float32
    n, c, h, w = x.shape
    read = si is not None and si.numel() > 0
    assert not (read and write_signs)
    so = torch.empty(0, dtype=torch.uint8, device=x.device)
    s, mode, sh, sw = None, 0, 0, 0
    if write_signs:
        sw = (w + 15) & ~15                                  # width rounded up to a multiple of 16 elements (filtered_lrelu.cpp:89)
        s = so = torch.empty(n, c, h, sw >> 2, dtype=torch.uint8, device=x.device)
        mode, sh = 1, h
    elif read:
        assert si.dtype == torch.uint8 and si.is_contiguous() and si.ndim == 4 and si.shape[:2] == (n, c)
        s, mode, sh, sw = si, 2, si.shape[2], si.shape[3] << 2
    hip.call('spi_filtered_lrelu_act', hip.ptr(x), hip.ptr(s), n * c, h, w, sh, sw, int(sx), int(sy), float(gain), float(slope),
             float(-1 if clamp is None else clamp), mode, hip.stream())
    return so


def _fused(x, fu, fd, b, up, down, padding, gain, slope, clamp, flip_filter, si, sx, sy, write_signs):
    """One launch of `spi_filtered_lrelu_fused` (csrc/flrelu.hip): -> (y, sign tensor written or an empty uint8 tensor).  Planes go down in
    chunks of 65 535 (the kernel's grid.y)."""
    px0, px1, py0, py1 = padding
    x = x.contiguous().float()
    n, c, ih, iw = x.shape
    fu, fd = fu.to(x.device).float().contiguous(), fd.to(x.device).float().contiguous()
    mid_h = ih * up + py0 + py1 - fu.shape[0] + 1
    mid_w = iw * up + px0 + px1 - fu.shape[1] + 1
    oh = (mid_h - fd.shape[0] + down) // down
    ow = (mid_w - fd.shape[1] + down) // down
    read = si is not None and si.numel() > 0
    assert not (read and write_signs)
    so = torch.empty(0, dtype=torch.uint8, device=x.device)
    s, mode, sh, sw = None, 0, 0, 0
    if write_signs:
        sw = (mid_w + 15) & ~15                               # width rounded up to a multiple of 16 elements (filtered_lrelu.cpp:89)
        s = so = torch.empty(n, c, mid_h, sw >> 2, dtype=torch.uint8, device=x.device)
        mode, sh = 1, mid_h
    elif read:
        assert si.dtype == torch.uint8 and si.is_contiguous() and si.ndim == 4 and si.shape[:2] == (n, c)
        s, mode, sh, sw = si, 2, si.shape[2], si.shape[3] << 2
    bb = b.detach().contiguous().float() if b is not None else None
    y = torch.empty(n, c, oh, ow, device=x.device, dtype=torch.float32)
    assert c <= 65535
    step = max(1, 65535 // c)
    for n0 in range(0, n, step):
        n1 = min(n0 + step, n)
        hip.call('spi_filtered_lrelu_fused', hip.ptr(x[n0:n1]), hip.ptr(fu), hip.ptr(fd), hip.ptr(bb), None if s is None else s[n0:n1].data_ptr(),
                 hip.ptr(y[n0:n1]), n1 - n0, c, ih, iw, fu.shape[0], fu.shape[1], fd.shape[0], fd.shape[1], int(up), int(down), px0, px1, py0, py1,
                 float(gain), float(slope), float(-1 if clamp is None else clamp), int(flip_filter), mode, sh, sw, int(sx), int(sy), oh, ow, hip.stream())
    return y, so


def _half_two_pass(x, fu, fd, b, up, down, padding, gain, slope, clamp, flip_filter):
    """Half tensors (the plugin is dispatched for half as well, filtered_lrelu.cpp:151): `spi_filtered_lrelu_t` -- x, b and y fp16, filters and the
    up-sampled intermediate fp32 (the plugin's internal type for half), one rounding at the store.  Inference path (no gradient); with a
    gradient required the op runs in fp32 like the reference's generic fallback does (filtered_lrelu.py:121-131)."""
    px0, px1, py0, py1 = padding
    x = x.contiguous()
    n, c, ih, iw = x.shape
    one = torch.ones(1, 1, device=x.device)
    fu = one if fu is None else fu.to(x.device).float().contiguous()
    fd = one if fd is None else fd.to(x.device).float().contiguous()
    fu = fu.reshape(1, 1) ** 2 if (fu.ndim == 1 and fu.numel() == 1) else fu
    fd = fd.reshape(1, 1) ** 2 if (fd.ndim == 1 and fd.numel() == 1) else fd
    mid_h, mid_w = ih * up + py0 + py1 - fu.shape[0] + 1, iw * up + px0 + px1 - fu.shape[1] + 1
    oh, ow = (mid_h - fd.shape[0] + down) // down, (mid_w - fd.shape[1] + down) // down
    tmp = torch.empty(n, c, mid_h, mid_w, device=x.device, dtype=torch.float32)
    y = torch.empty(n, c, oh, ow, device=x.device, dtype=torch.float16)
    bb = b.detach().to(torch.float16).contiguous() if b is not None else None
    hip.call('spi_filtered_lrelu_t', hip.ptr(x), hip.ptr(fu), hip.ptr(fd), hip.ptr(bb), hip.ptr(tmp), hip.ptr(y), n, c, ih, iw, fu.shape[0], fu.shape[1],
             fd.shape[0], fd.shape[1], up, down, px0, px1, py0, py1, gain, slope, float(-1 if clamp is None else clamp), int(flip_filter), oh, ow,
             hip.DTYPE_IDS[torch.float16], hip.stream())
    return y


_op_cache = {}


def _filtered_lrelu_op(up, down, padding, gain, slope, clamp, flip_filter):
    """autograd.Function factory keyed like the reference's `_filtered_lrelu_cuda` cache (filtered_lrelu.py:160-176)."""
    from . import upfirdn2d as _uf
    px0, px1, py0, py1 = padding
    key = (up, down, px0, px1, py0, py1, gain, slope, clamp, flip_filter)
    if key in _op_cache:
        return _op_cache[key]

    class FilteredLReluHip(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, fu, fd, b, si, sx, sy):
            x = x.contiguous().float()
            if fu is None:
                fu = torch.ones([1, 1], dtype=torch.float32, device=x.device)
            if fd is None:
                fd = torch.ones([1, 1], dtype=torch.float32, device=x.device)
            write_signs = (si is None or si.numel() == 0) and (x.requires_grad or (b is not None and b.requires_grad))
            y, so = _fused(x, fu, fd, b, up, down, (px0, px1, py0, py1), gain, slope, clamp, flip_filter, si, sx, sy, write_signs)
            ctx.save_for_backward(fu, fd, si if (si is not None and si.numel()) else so)
            ctx.x_shape, ctx.y_shape, ctx.s_ofs, ctx.has_b = x.shape, y.shape, (sx, sy), b is not None
            return y

        @staticmethod
        @torch.autograd.function.once_differentiable
        def backward(ctx, dy):
            fu, fd, si = ctx.saved_tensors
            _, _, xh, xw = ctx.x_shape
            _, _, yh, yw = ctx.y_shape
            sx, sy = ctx.s_ofs
            dx = db = None
            if ctx.needs_input_grad[0] or ctx.needs_input_grad[3]:
                pp = [(fu.shape[-1] - 1) + (fd.shape[-1] - 1) - px0, xw * up - yw * down + px0 - (up - 1),
                      (fu.shape[0] - 1) + (fd.shape[0] - 1) - py0, xh * up - yh * down + py0 - (up - 1)]
                gg = gain * (up ** 2) / (down ** 2)
                sx = sx - (fu.shape[-1] - 1) + px0
                sy = sy - (fu.shape[0] - 1) + py0
                dx = _filtered_lrelu_op(down, up, tuple(pp), gg, slope, None, not flip_filter).apply(dy.contiguous().float(), fd, fu, None, si, sx, sy)
            if ctx.has_b and ctx.needs_input_grad[3]:
                db = dx.sum([0, 2, 3])
            return dx, None, None, db, None, None, None

    _op_cache[key] = FilteredLReluHip
    return FilteredLReluHip


def filtered_lrelu(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=math.sqrt(2), slope=0.2, clamp=None,
                   flip_filter=False, impl='hip'):
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    need_grad = torch.is_grad_enabled() and (x.requires_grad or (b is not None and b.requires_grad))
    separable = (fu is not None and fu.ndim == 1 and fu.numel() > 1) or (fd is not None and fd.ndim == 1 and fd.numel() > 1)
    if need_grad and not separable:
        fu2 = None if fu is None else (fu.float().reshape(1, 1) ** 2 if fu.ndim == 1 else fu.float())
        fd2 = None if fd is None else (fd.float().reshape(1, 1) ** 2 if fd.ndim == 1 else fd.float())
        op = _filtered_lrelu_op(int(up), int(down), tuple(_parse_padding(padding)), float(gain), float(slope),
                                None if clamp is None else float(clamp), bool(flip_filter))
        return op.apply(x, fu2, fd2, b, None, 0, 0)
    if separable:
        from . import bias_act as _ba, upfirdn2d as _uf
        y = _ba.bias_act(x, b) if b is not None else x
        y = _uf.upfirdn2d(y, fu, up=up, padding=_parse_padding(padding), gain=up ** 2, flip_filter=flip_filter)
        y = _ba.bias_act(y, act='lrelu', alpha=slope, gain=gain, clamp=clamp)
        return _uf.upfirdn2d(y, fd, down=down, flip_filter=flip_filter)
    if x.dtype == torch.float16:
        return _half_two_pass(x, fu, fd, b, int(up), int(down), tuple(_parse_padding(padding)), float(gain), float(slope), clamp, bool(flip_filter))
    one = torch.ones(1, 1, device=x.device)
    fu = one if fu is None else fu.to(x.device).float().contiguous()
    fd = one if fd is None else fd.to(x.device).float().contiguous()
    if fu.ndim == 1:
        fu = fu.reshape(1, 1) ** 2 if fu.numel() == 1 else fu
    if fd.ndim == 1:
        fd = fd.reshape(1, 1) ** 2 if fd.numel() == 1 else fd
    return _fused(x, fu, fd, b, int(up), int(down), tuple(_parse_padding(padding)), float(gain), float(slope), clamp, bool(flip_filter), None, 0, 0, False)[0]
