"""bias_act on HIP: ``y = clamp(act(x + b) * gain)`` in one pass, with its first-order backward.

Same surface as the reference's ``bias_act`` (eg3d/torch_utils/ops/bias_act.py:54-88); ``impl`` is
accepted for signature compatibility but there is only the HIP implementation -- CPU tensors raise.
The backward uses the saved input/output exactly as the reference's plugin does (bias_act.py:159-205,
bias_act.cu: grad = 1).  Second-order gradients are not on the inversion path and are not provided.
"""
import math
import torch
from ... import hip
from .. import zero_arena

_SQRT2 = math.sqrt(2.0)
# name -> (act id, default alpha, default gain, which of x / y the gradient needs)
activation_funcs = {
    'linear': (1, 0.0, 1.0, ''), 'relu': (2, 0.0, _SQRT2, 'y'), 'lrelu': (3, 0.2, _SQRT2, 'y'),
    'tanh': (4, 0.0, 1.0, 'y'), 'sigmoid': (5, 0.0, 1.0, 'y'), 'elu': (6, 0.0, 1.0, 'y'),
    'selu': (7, 0.0, 1.0, 'y'), 'softplus': (8, 0.0, 1.0, 'y'), 'swish': (9, 0.0, _SQRT2, 'x'),
}


def def_gain(act):
    return activation_funcs[act][2]


def _launch(x, b, xref, yref, dy, grad, dim, act_id, alpha, gain, clamp):
    y = torch.empty_like(x)
    size_b = b.numel() if b is not None else 0
    step_b = 1
    if b is not None:
        for d in x.shape[dim + 1:]:
            step_b *= d
    if x.dtype in (torch.float16, torch.float64):        # the reference's plugin is instantiated for half and double too (bias_act.cpp:81): fp32 arithmetic
        # and one rounding at the store for half, double arithmetic for double
        hip.call('spi_bias_act_t', hip.ptr_any(x), hip.ptr_any(b), hip.ptr_any(xref), hip.ptr_any(yref), hip.ptr_any(dy), hip.ptr_any(y), x.numel(),
                 size_b, step_b, grad, act_id, alpha, gain, clamp, hip.DTYPE_IDS[x.dtype], hip.stream())
        return y
    hip.call('spi_bias_act', hip.ptr(x), hip.ptr(b), hip.ptr(xref), hip.ptr(yref), hip.ptr(dy), hip.ptr(y), x.numel(), size_b,
             step_b, grad, act_id, alpha, gain, clamp, hip.stream())
    return y


def tail_zero_elems(dy, noise, need_noise, need_strength, need_bias):
    """floats of zeroed scratch tail_backward needs for this call (its caller may provide them as ``zero_buf``)"""
    want_pix = noise is not None and (need_noise or need_strength)
    return (dy.shape[1] if need_bias else 0) + (dy[0, 0].numel() if want_pix else 0) + (1 if (noise is not None and need_strength) else 0)


def tail_backward(dy, y, noise, strength, act_id, alpha, gain, clamp, need_noise, need_strength, need_bias, zero_buf=None, zdot=None):
    """Backward of  y = clamp(act(z + noise*strength + bias)*gain)  for dy [N,C,H,W] in one launch (spi_tail_bwd).
    ``y`` None: no activation/gain/clamp was applied (dz = dy).  Returns (dz, d_noise, d_strength, d_bias).
    ``zero_buf``: optional zeroed 1-D fp32 tensor whose first ``tail_zero_elems(...)`` entries become d_bias / the pixel sums.
    ``zdot``: optional ``(out [N*C] zeroed fp32, bias [C] or None, noise [H,W] or None, noise_gain [1] or None)`` -- the same launch adds
    ``sum_hw dz * z`` per (n, c) into ``out``, z = the conv result reconstructed from ``y`` (spi_tail_bwd_dot_t; needs ``y``)."""
    # fp16 activation tensors (the reference's use_fp16 blocks): dy / y / dz are half, every sum stays fp32
    half = (y is not None and y.dtype == torch.float16) or (y is None and dy.dtype == torch.float16)
    dy = dy.contiguous().to(torch.float16 if half else torch.float32)
    n, c = dy.shape[0], dy.shape[1]
    hw = dy[0, 0].numel()
    want_pix = noise is not None and (need_noise or need_strength)
    if y is None and not want_pix and not need_bias:
        return dy, None, None, None
    dz = torch.empty_like(dy) if y is not None else None
    want_s = want_pix and need_strength
    if zero_buf is None:
        zero_buf = zero_arena.zeros((c if need_bias else 0) + (hw if want_pix else 0) + (1 if want_s else 0), dy.device)
    nb = c if need_bias else 0
    d_bias = zero_buf[:nb] if need_bias else None
    pix = zero_buf[nb:nb + hw].view(dy.shape[2:]) if want_pix else None
    ds = zero_buf[nb + hw:nb + hw + 1] if want_s else None             # sum_hw pixsum * noise comes out of the same launch
    nzc = noise.contiguous().float() if want_s else None
    if zdot is not None:
        assert y is not None, 'tail_backward: zdot needs the saved output'
        zo, zb, zn, zg = zdot
        hip.call('spi_tail_bwd_dot_t', hip.ptr(dy), hip.ptr(y), hip.ptr(dz), hip.ptr(d_bias), hip.ptr(pix), hip.ptr(nzc), hip.ptr(ds), n, c, hw, act_id, alpha,
                 gain, clamp, hip.ptr(zb), hip.ptr(zn), hip.ptr(zg), hip.ptr(zo), hip.DTYPE_IDS[torch.float16 if half else torch.float32], hip.stream())
    elif half:
        hip.call('spi_tail_bwd_t', hip.ptr(dy), hip.ptr(y), hip.ptr(dz), hip.ptr(d_bias), hip.ptr(pix), hip.ptr(nzc), hip.ptr(ds), n, c, hw, act_id, alpha,
                 gain, clamp, hip.DTYPE_IDS[torch.float16], hip.stream())
    else:
        hip.call('spi_tail_bwd', hip.ptr(dy), hip.ptr(y), hip.ptr(dz), hip.ptr(d_bias), hip.ptr(pix), hip.ptr(nzc), hip.ptr(ds), n, c, hw, act_id, alpha,
                 gain, clamp, hip.stream())
    d_noise = d_strength = None
    if want_pix:
        if need_noise:
            d_noise = pix * strength if strength is not None else pix
        if need_strength:
            d_strength = ds.reshape(())
    return (dz if dz is not None else dy), d_noise, d_strength, d_bias


class _BiasAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, b, dim, act_id, alpha, gain, clamp, ref):
        dt = x.dtype if x.dtype in (torch.float16, torch.float64) else torch.float32          # half / double stay (b follows x, bias_act.py:146-148 of the reference)
        x = x.contiguous().to(dt)
        bb = b.contiguous().to(dt) if b is not None else None
        y = _launch(x, bb, None, None, None, 0, dim, act_id, alpha, gain, clamp)
        # the clamp mask needs the output for every activation (the reference's plugin drops it for
        # 'linear' and so ignores the clamp in that backward; its CPU path -- our oracle -- does not)
        ctx.save_for_backward(x if 'x' in ref else None, bb if 'x' in ref else None, y if ('y' in ref or clamp >= 0) else None)
        ctx.cfg = (dim, act_id, alpha, gain, clamp, b is not None)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, b, y = ctx.saved_tensors
        dim, act_id, alpha, gain, clamp, has_b = ctx.cfg
        ref_t = y if y is not None else x
        dy = dy.contiguous().to(ref_t.dtype if ref_t is not None else (dy.dtype if dy.dtype in (torch.float16, torch.float64) else torch.float32))
        dx = dy
        if act_id != 1 or gain != 1 or clamp >= 0:
            dx = _launch(dy, b, x, y, None, 1, dim, act_id, alpha, gain, clamp)
        db = None
        if has_b and ctx.needs_input_grad[1]:
            db = dx.sum([i for i in range(dx.ndim) if i != dim])
        return dx, db, None, None, None, None, None, None


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None, impl='hip'):
    if act not in activation_funcs:
        raise KeyError(act)
    act_id, d_alpha, d_gain, ref = activation_funcs[act]
    alpha = float(d_alpha if alpha is None else alpha)
    gain = float(d_gain if gain is None else gain)
    clamp = float(-1 if clamp is None else clamp)
    if b is not None:
        assert b.ndim == 1 and 0 <= dim < x.ndim and b.shape[0] == x.shape[dim]
    if act_id == 1 and gain == 1 and clamp < 0 and b is None:
        return x
    return _BiasAct.apply(x, b, dim, act_id, alpha, gain, clamp, ref)
