"""upfirdn2d on HIP: pad -> zero-insert upsample -> FIR -> decimate in one pass.

Same surface as the reference's module (eg3d/torch_utils/ops/upfirdn2d.py): ``setup_filter``,
``upfirdn2d``, ``filter2d``, ``upsample2d``, ``downsample2d``.  The backward is the adjoint operator
(up <-> down swapped, filter flipped, padding adjusted), i.e. the same kernel again.
``upfirdn2d_bias_act`` additionally fuses the tail of ``SynthesisLayer.forward`` for up-sampling
layers (networks_stylegan2.py:320-329: + noise, + bias, lrelu, gain, clamp) into the FIR pass so the
(2H)^2 activation is written once.
"""
import numpy as np
import torch
from ... import hip
from . import bias_act as _ba


def _parse_scaling(s):
    if isinstance(s, int):
        s = [s, s]
    sx, sy = s
    assert sx >= 1 and sy >= 1
    return int(sx), int(sy)


def _parse_padding(p):
    if isinstance(p, int):
        p = [p, p]
    p = [int(v) for v in p]
    if len(p) == 2:
        p = [p[0], p[0], p[1], p[1]]
    return tuple(p)


def _get_filter_size(f):
    if f is None:
        return 1, 1
    return int(f.shape[-1]), int(f.shape[0])


def setup_filter(f, device=torch.device('cpu'), normalize=True, flip_filter=False, gain=1, separable=None):
    if f is None:
        f = 1
    f = torch.as_tensor(f, dtype=torch.float32)
    assert f.ndim in [0, 1, 2] and f.numel() > 0
    if f.ndim == 0:
        f = f[np.newaxis]
    if separable is None:
        separable = (f.ndim == 1 and f.numel() >= 8)
    if f.ndim == 1 and not separable:
        f = f.ger(f)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
    f = f * (gain ** (f.ndim / 2))
    return f.to(device=device)


def _launch(x, f2d, up, down, pad, flip, gain, epilogue=None):
    n, c, ih, iw = x.shape
    fh, fw = f2d.shape
    upx, upy = up
    dx, dy = down
    px0, px1, py0, py1 = pad
    oh = (ih * upy + py0 + py1 - fh + dy) // dy
    ow = (iw * upx + px0 + px1 - fw + dx) // dx
    assert oh >= 1 and ow >= 1
    y = torch.empty(n, c, oh, ow, device=x.device, dtype=x.dtype)
    if epilogue is None:
        noise = ng = bias = None
        act, alpha, again, clamp = 0, 0.0, 1.0, -1.0
    else:
        noise, ng, bias, act, alpha, again, clamp = epilogue
    if x.dtype == torch.float16:           # fp16 activation tensors (use_fp16 blocks): the LDS-tiled 4x4 kernel with its fused tail, typed
        hip.call('spi_upfirdn2d_fused_t', hip.ptr(x), hip.ptr(f2d), hip.ptr(y), n, c, ih, iw, fh, fw, upx, upy, dx, dy, px0, px1, py0, py1,
                 int(flip), float(gain), oh, ow, hip.ptr(noise), hip.ptr(ng), hip.ptr(bias), act, alpha, again, clamp, hip.DTYPE_IDS[torch.float16], hip.stream())
        return y
    hip.call('spi_upfirdn2d', hip.ptr(x), hip.ptr(f2d), hip.ptr(y), n, c, ih, iw, fh, fw, upx, upy, dx, dy, px0, px1, py0, py1,
             int(flip), float(gain), oh, ow, hip.ptr(noise), hip.ptr(ng), hip.ptr(bias), act, alpha, again, clamp, hip.stream())
    return y


def _half_tiled_ok(x, f, up, down, pad):
    """fp16 NCHW tensor + the shape spi_upfirdn2d_fused_t serves (4x4 filter, up = down = 1, output >= 100 px): the FIR of the up-sampling layers"""
    if not (x.dtype == torch.float16 and x.is_contiguous() and f is not None and f.ndim == 2 and tuple(f.shape) == (4, 4) and up == (1, 1) and down == (1, 1)):
        return False
    oh = x.shape[2] + pad[2] + pad[3] - 3
    ow = x.shape[3] + pad[0] + pad[1] - 3
    return oh >= 100 and ow >= 100 and x.shape[0] * x.shape[1] <= 65535


def _launch_typed(x, f2d, up, down, pad, flip, gain):
    """fp16 and / or channels_last tensors (the reference's use_fp16 blocks hand the plugin half tensors in channels_last layout,
    networks_stylegan2.py:423-436): spi_upfirdn2d_t with the dtype and both stride sets; the output keeps x's dtype and memory format
    (upfirdn2d.cpp:42 suggest_memory_format)."""
    n, c, ih, iw = x.shape
    fh, fw = f2d.shape
    upx, upy = up
    dx, dy = down
    px0, px1, py0, py1 = pad
    oh = (ih * upy + py0 + py1 - fh + dy) // dy
    ow = (iw * upx + px0 + px1 - fw + dx) // dx
    assert oh >= 1 and ow >= 1
    cl = x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
    if not (cl or x.is_contiguous()):
        x = x.contiguous()
    y = torch.empty(n, c, oh, ow, device=x.device, dtype=x.dtype, memory_format=torch.channels_last if cl else torch.contiguous_format)
    import ctypes
    xs = (ctypes.c_int64 * 4)(*x.stride())
    ys = (ctypes.c_int64 * 4)(*y.stride())
    hip.call('spi_upfirdn2d_t', hip.ptr_any(x), hip.ptr(f2d), hip.ptr_any(y), n, c, ih, iw, ctypes.cast(xs, ctypes.c_void_p), ctypes.cast(ys, ctypes.c_void_p),
             fh, fw, upx, upy, dx, dy, px0, px1, py0, py1, int(flip), float(gain), oh, ow, hip.DTYPE_IDS[x.dtype], hip.stream())
    return y


def _run(x, f, up, down, pad, flip, gain):
    """Non-differentiable core; handles None / separable filters like the reference (upfirdn2d.py:240-250)."""
    if _half_tiled_ok(x, f, up, down, pad):
        return _launch(x, f.to(x.device).float().contiguous(), up, down, pad, flip, gain)
    typed = x.dtype in (torch.float16, torch.float64) or (x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous())
    if typed:
        if f is None:
            f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
        f = f.to(x.device).float().contiguous()
        if f.ndim == 1 and f.shape[0] == 1:
            f = f.square().unsqueeze(0)
        if f.ndim == 2:
            return _launch_typed(x, f, up, down, pad, flip, gain)
        y = _launch_typed(x, f.unsqueeze(0).contiguous(), (up[0], 1), (down[0], 1), (pad[0], pad[1], 0, 0), flip, 1.0)
        return _launch_typed(y, f.unsqueeze(1).contiguous(), (1, up[1]), (1, down[1]), (0, 0, pad[2], pad[3]), flip, gain)
    x = x.contiguous().float()
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
    f = f.to(x.device).float().contiguous()
    if f.ndim == 1 and f.shape[0] == 1:
        f = f.square().unsqueeze(0)
    if f.ndim == 2:
        return _launch(x, f, up, down, pad, flip, gain)
    y = _launch(x, f.unsqueeze(0).contiguous(), (up[0], 1), (down[0], 1), (pad[0], pad[1], 0, 0), flip, 1.0)
    return _launch(y, f.unsqueeze(1).contiguous(), (1, up[1]), (1, down[1]), (0, 0, pad[2], pad[3]), flip, gain)


def _adjoint_padding(x_shape, y_shape, f, up, down, pad):
    _, _, ih, iw = x_shape
    _, _, oh, ow = y_shape
    fw, fh = _get_filter_size(f)
    return (fw - pad[0] - 1, iw * up[0] - ow * down[0] + pad[0] - up[0] + 1,
            fh - pad[2] - 1, ih * up[1] - oh * down[1] + pad[2] - up[1] + 1)


class _Upfirdn2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, f, up, down, pad, flip, gain):
        y = _run(x, f, up, down, pad, flip, gain)
        ctx.f = f
        ctx.cfg = (tuple(x.shape), tuple(y.shape), up, down, pad, flip, gain)
        return y

    @staticmethod
    def backward(ctx, dy):
        xs, ys, up, down, pad, flip, gain = ctx.cfg
        p = _adjoint_padding(xs, ys, ctx.f, up, down, pad)
        dx = _Upfirdn2d.apply(dy, ctx.f, down, up, p, not flip, gain)
        return dx, None, None, None, None, None, None


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, impl='hip'):
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    return _Upfirdn2d.apply(x, f, _parse_scaling(up), _parse_scaling(down), _parse_padding(padding), bool(flip_filter), float(gain))


def filter2d(x, f, padding=0, flip_filter=False, gain=1, impl='hip'):
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [px0 + fw // 2, px1 + (fw - 1) // 2, py0 + fh // 2, py1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl='hip'):
    upx, upy = _parse_scaling(up)
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [px0 + (fw + upx - 1) // 2, px1 + (fw - upx) // 2, py0 + (fh + upy - 1) // 2, py1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl='hip'):
    dx, dy = _parse_scaling(down)
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [px0 + (fw - dx + 1) // 2, px1 + (fw - dx) // 2, py0 + (fh - dy + 1) // 2, py1 + (fh - dy) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain)


class _UpfirdnBiasAct(torch.autograd.Function):
    """y = bias_act(upfirdn2d(x, f, pad, gain) + noise * strength, bias, act) with a 2-D filter, up = down = 1."""
    @staticmethod
    def forward(ctx, x, f, noise, strength, bias, pad, fgain, act_id, alpha, again, clamp):
        if x.dtype == torch.float16:                       # fp16 activation tensors: half in, half out (bias / noise stay fp32)
            x = x.contiguous()
            if not _half_tiled_ok(x, f, (1, 1), (1, 1), pad):
                raise NotImplementedError('upfirdn2d_bias_act on fp16 tensors serves the 4x4 FIR of the up-sampling layers on images >= 100 px')
        else:
            x = x.contiguous().float()
        f = f.to(x.device).float().contiguous()
        nz = noise.contiguous().float() if noise is not None else None
        ng = strength.detach().reshape(1).contiguous().float() if (noise is not None and strength is not None) else None
        bb = bias.contiguous().float() if bias is not None else None
        y = _launch(x, f, (1, 1), (1, 1), pad, False, fgain, epilogue=(nz, ng, bb, act_id, alpha, again, clamp))
        ctx.save_for_backward(y, nz, ng)
        ctx.f = f
        ctx.cfg = (tuple(x.shape), pad, fgain, act_id, alpha, again, clamp)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        y, nz, ng = ctx.saved_tensors
        xs, pad, fgain, act_id, alpha, again, clamp = ctx.cfg
        dz, d_noise, d_strength, d_bias = _ba.tail_backward(dy, y, nz, ng, act_id, alpha, again, clamp, ctx.needs_input_grad[2],
                                                            ctx.needs_input_grad[3], ctx.needs_input_grad[4])
        p = _adjoint_padding(xs, tuple(y.shape), ctx.f, (1, 1), (1, 1), pad)
        dx = _run(dz, ctx.f, (1, 1), (1, 1), p, True, fgain) if ctx.needs_input_grad[0] else None
        return dx, None, d_noise, d_strength, d_bias, None, None, None, None, None, None


def upfirdn2d_bias_act(x, f, noise=None, noise_strength=None, bias=None, padding=0, gain=1, act='lrelu', alpha=None, act_gain=None,
                       clamp=None):
    act_id, d_alpha, d_gain, _ = _ba.activation_funcs[act]
    assert act_id in (1, 2, 3), 'fused epilogue supports linear / relu / lrelu'
    alpha = float(d_alpha if alpha is None else alpha)
    act_gain = float(d_gain if act_gain is None else act_gain)
    clamp = float(-1 if clamp is None else clamp)
    return _UpfirdnBiasAct.apply(x, f, noise, noise_strength, bias, _parse_padding(padding), float(gain), act_id, alpha, act_gain, clamp)
