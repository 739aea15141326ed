"""StyleGAN2 generator half on MI355X kernels (drop-in surface of eg3d/training/networks_stylegan2.py:27-552).

Class names, constructor arguments, attribute names and ``state_dict`` keys follow the reference so
an EG3D checkpoint's ``G_ema`` state loads unchanged and the SPI projectors can find
``backbone.synthesis.named_buffers()`` / ``noise_const`` / ``mapping.num_ws``.

What differs (MI355X-first):
  * every dense conv runs on the fp32 matrix cores through ``ops.conv2d_mfma`` with the sample's own
    modulated weights (``groups = batch`` in the reference, networks_stylegan2.py:85-88) -- no
    [1, N*C, H, W] reshapes, no cuDNN;
  * the layer tail (+ noise, + bias, lrelu, gain, clamp) is fused into the conv epilogue (stride-1
    layers) or into the 4x4 FIR pass that follows the stride-2 transposed conv (up layers);
  * precision is explicit: fp32 everywhere (the reference's rule "fp16 iff use_fp16 and the tensor is
    on 'cuda'", :421-423, would silently pick fp16 on ROCm where tensors also report 'cuda').
Only the fused-modconv path exists (what SPI runs: G.eval() + 'inference_only', :427-428); the
discriminator half of the reference file is out of scope.
"""
import numpy as np
import torch

from .. import hip
from ..configs import global_config
from ..torch_utils.ops import bias_act, upfirdn2d, conv2d_mfma
from ..torch_utils import misc, zero_arena


@misc.profiled_function
def normalize_2nd_moment(x, dim=1, eps=1e-8):
    return x * (x.square().mean(dim=dim, keepdim=True) + eps).rsqrt()


class _Modulate(torch.autograd.Function):
    """weight [O,I,k,k], styles [N,I] -> (de)modulated per-sample weights in the kernels' tap-major layout [N,O,k,k,I]
    (one HIP launch each way instead of ~12 strided elementwise / reduction launches; reference :62-69)."""

    @staticmethod
    def forward(ctx, weight, styles, demodulate, style_gain):
        weight = weight.contiguous().float()
        styles = styles.contiguous().float()
        o, i, kh, kw = weight.shape
        n = styles.shape[0]
        w = torch.empty(n, o, kh, kw, i, device=weight.device, dtype=torch.float32)
        dcoef = torch.empty(n, o, device=weight.device, dtype=torch.float32) if demodulate else None
        hip.call('spi_modulate_fwd', hip.ptr(weight), hip.ptr(styles), hip.ptr(w), hip.ptr(dcoef), n, o, i, kh * kw, int(demodulate), float(style_gain),
                 hip.stream())
        ctx.save_for_backward(weight, styles, dcoef)
        ctx.demodulate, ctx.style_gain = demodulate, float(style_gain)
        return w

    @staticmethod
    def backward(ctx, g):
        weight, styles, dcoef = ctx.saved_tensors
        o, i, kh, kw = weight.shape
        n = styles.shape[0]
        g = g.contiguous().float()
        dw = torch.empty_like(weight) if ctx.needs_input_grad[0] else None
        ds = zero_arena.zeros_like(styles)
        hip.call('spi_modulate_bwd', hip.ptr(weight), hip.ptr(styles), hip.ptr(dcoef), hip.ptr(g), hip.ptr(dw), hip.ptr(ds), n, o, i, kh * kw,
                 int(ctx.demodulate), ctx.style_gain, hip.stream())
        return dw, ds, None, None


def modulate_weights(weight, styles, demodulate=True, style_gain=1.0):
    """``style_gain`` multiplies the styles inside the kernel (ToRGBLayer's weight_gain, reference :303)."""
    return _Modulate.apply(weight, styles, bool(demodulate), float(style_gain))


class _MultiModulate(torch.autograd.Function):
    """`_Modulate` for ALL layers of a network in one launch each way (`spi_modulate_multi_fwd / _bwd`): every layer's styles exist before its first
    convolution runs (`multi_affine`), so the 20 + 6 per-layer modulation launches that head every layer's forward chain -- and the 26 that tail its
    backward -- collapse into one; small layers no longer run on a quarter of the chip.  Same kernel bodies, bit-equal results.
    Inputs: cfg = ((demodulate, style_gain), ...), then the styles [N, I_l] of every layer, then the weights [O_l, I_l, k, k]."""

    @staticmethod
    def forward(ctx, cfg, *sw):
        nl = len(cfg)
        styles = [s.contiguous().float() for s in sw[:nl]]
        weights = [w.contiguous().float() for w in sw[nl:]]
        n = styles[0].shape[0]
        dev = weights[0].device
        wflat = torch.empty(sum(n * w.numel() for w in weights), device=dev, dtype=torch.float32)
        dflat = torch.empty(sum(n * w.shape[0] for w, c in zip(weights, cfg) if c[0]), device=dev, dtype=torch.float32)
        jobs = (hip.ModulateJob * nl)()
        outs, dcoefs, wo, do = [], [], 0, 0
        for l, (w, s, (demod, sgain)) in enumerate(zip(weights, styles, cfg)):
            o, i, kh, kw = w.shape
            assert s.shape == (n, i)
            w2 = wflat[wo:wo + n * w.numel()].view(n, o, kh, kw, i)
            wo += n * w.numel()
            dc = None
            if demod:
                dc = dflat[do:do + n * o].view(n, o)
                do += n * o
            outs.append(w2); dcoefs.append(dc)
            j = jobs[l]
            j.weight, j.styles, j.w_out, j.dcoef = w.data_ptr(), s.data_ptr(), w2.data_ptr(), (dc.data_ptr() if dc is not None else None)
            j.style_gain, j.O, j.I, j.T, j.demodulate = float(sgain), o, i, kh * kw, int(demod)
        hip.call('spi_modulate_multi_fwd', jobs, nl, n, hip.stream())
        ctx.save_for_backward(dflat, *styles, *weights)
        ctx.cfg = cfg
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gs):
        cfg = ctx.cfg
        nl = len(cfg)
        dflat, *sw = ctx.saved_tensors
        styles, weights = sw[:nl], sw[nl:]
        n = styles[0].shape[0]
        dsflat = zero_arena.zeros(sum(s.numel() for s in styles), styles[0].device)
        jobs = (hip.ModulateJob * nl)()
        ds, dws, keep, k, so, do = [None] * nl, [None] * nl, [], 0, 0, 0
        for l, (w, s, (demod, sgain)) in enumerate(zip(weights, styles, cfg)):
            o, i, kh, kw = w.shape
            dsl = dsflat[so:so + n * i].view(n, i)
            so += n * i
            dc = None
            if demod:
                dc = dflat[do:do + n * o]
                do += n * o
            g = gs[l]
            if g is None:
                continue
            g = g.contiguous().float()
            keep.append(g)
            ds[l] = dsl
            if ctx.needs_input_grad[1 + nl + l]:
                dws[l] = torch.empty_like(w)
            j = jobs[k]
            k += 1
            j.weight, j.styles, j.dcoef, j.g = w.data_ptr(), s.data_ptr(), (dc.data_ptr() if dc is not None else None), g.data_ptr()
            j.d_weight, j.d_styles = (dws[l].data_ptr() if dws[l] is not None else None), dsl.data_ptr()
            j.style_gain, j.O, j.I, j.T, j.demodulate = float(sgain), o, i, kh * kw, int(demod)
        if k:
            hip.call('spi_modulate_multi_bwd', jobs, k, n, hip.stream())
        return (None, *ds, *dws)


def multi_modulate(layers, styles):
    """Modulated weights of `layers` (SynthesisLayer / ToRGBLayer modules, execution order) from their `styles` in one launch, or None when the
    fast path does not apply: any layer on the frozen-weight path of stage 1 (it modulates inside its own function and needs no weight gradient),
    non-fp32 / non-GPU tensors, more than 32 layers."""
    if styles is None or not (1 <= len(layers) <= hip.MODULATE_MAX_JOBS):
        return None
    cfg, ws_, frozen = [], [], []
    for m, s in zip(layers, styles):
        w = m.weight
        if not (w.is_cuda and w.dtype == torch.float32 and s.dtype == torch.float32 and w.shape[1] * w.shape[2] * w.shape[3] * 12 <= 64 * 1024):
            return None
        frozen.append(torch.is_grad_enabled() and not w.requires_grad and s.requires_grad)      # stage 1: _ModConvFrozen
        torgb = isinstance(m, ToRGBLayer)
        cfg.append((not torgb, float(m.weight_gain) if torgb else 1.0))
        ws_.append(w)
    if all(frozen):
        return _multi_modulate_frozen(tuple(cfg), styles, ws_)
    if any(frozen):
        return None
    return _MultiModulate.apply(tuple(cfg), *styles, *ws_)


class FrozenMod(tuple):
    """(w2 [N,O,k,k,I], dcoef [N,O] or None) of one frozen-weight layer, modulated ahead of its forward by `_multi_modulate_frozen`."""


def _multi_modulate_frozen(cfg, styles, weights):
    """Round 6: stage 1's frozen-weight layers (`_ModConvFrozen`) modulated in ONE launch ahead of the forward, like `_MultiModulate` does for the
    trainable path: 26 `spi_modulate_fwd` launches per synthesis (each a few us of work behind a ~3 us launch boundary) become one
    `spi_modulate_multi_fwd`.  No autograd here -- `_ModConvFrozen.backward` forms the style gradient from the saved w2 / dcoef as before.
    Same kernel bodies as the per-layer launches: bit-equal weights."""
    with torch.no_grad():
        nl = len(cfg)
        st = [s.detach().contiguous().float() for s in styles]
        wt = [w.detach().contiguous().float() for w in weights]
        n = st[0].shape[0]
        dev = wt[0].device
        wflat = torch.empty(sum(n * w.numel() for w in wt), device=dev, dtype=torch.float32)
        dflat = torch.empty(max(1, sum(n * w.shape[0] for w, c in zip(wt, cfg) if c[0])), device=dev, dtype=torch.float32)
        jobs = (hip.ModulateJob * nl)()
        outs, wo, do = [], 0, 0
        for l, (w, s, (demod, sgain)) in enumerate(zip(wt, st, cfg)):
            o, i, kh, kw = w.shape
            assert s.shape == (n, i)
            w2 = wflat[wo:wo + n * w.numel()].view(n, o, kh, kw, i)
            wo += n * w.numel()
            dc = None
            if demod:
                dc = dflat[do:do + n * o].view(n, o)
                do += n * o
            j = jobs[l]
            j.weight, j.styles, j.w_out, j.dcoef = w.data_ptr(), s.data_ptr(), w2.data_ptr(), (dc.data_ptr() if dc is not None else None)
            j.style_gain, j.O, j.I, j.T, j.demodulate = float(sgain), o, i, kh * kw, int(demod)
            outs.append(FrozenMod((w2, dc)))
        hip.call('spi_modulate_multi_fwd', jobs, nl, n, hip.stream())
    return outs


def _tap_energy(weight):
    """sum_t W[o,i,t]^2 [O, I] of a FROZEN conv weight.  Cached ON the parameter object (so it lives and dies with it) and recomputed
    when the tensor has been written since (``_version``): stage 1 evaluates it 500 times per image on unchanged weights."""
    hit = getattr(weight, '_spi_tap_energy', None)
    if hit is None or hit[0] != weight._version or hit[1].device != weight.device:
        with torch.no_grad():
            hit = (weight._version, weight.detach().float().square().sum(dim=(2, 3)).contiguous())
        weight._spi_tap_energy = hit
    return hit[1]


class _ModConvFrozen(torch.autograd.Function):
    """Modulated conv whose WEIGHTS are frozen but whose styles need a gradient (SPI stage 1: G.requires_grad_(False), W+ is
    optimised).  The reference's autograd graph computes the full O*I*k*k weight gradient only to contract it with dw''/ds;
    algebraically (v = W s g, d = rsqrt(|v|^2), w'' = v d, z = conv(x, w'')):
        d s_i = <x_i, dx_i> / s_i  -  s_i g^2 sum_o d_o^2 <dz_o, z_o> sum_t W[o,i,t]^2
    i.e. two per-channel dot products over activations that already exist (dx comes out of dgrad anyway) and a GEMV --
    the weight-gradient GEMM (20 % of a stage-1 step) disappears.  Same value up to fp32 rounding."""

    @staticmethod
    def forward(ctx, x, weight, styles, bias, noise, strength, pad, transposed, flip, act_id, alpha, gain, clamp, demodulate, style_gain, f16, ww=None, pre=None):
        import ctypes
        from ..torch_utils.ops.conv2d_mfma import _desc, _workspace, _out_tensor, out_size, half_io
        half = half_io(x, f16)                             # fp16 activation tensors (use_fp16 blocks): x, y, dy, dx are half
        x = x.contiguous() if half else x.contiguous().float()
        weight = weight.detach().contiguous().float()
        st = styles.detach().contiguous().float()
        o, i, kh, kw = weight.shape
        n, ns = x.shape[0], st.shape[0]
        if pre is not None:                                # modulated ahead of the forward with every other layer (_multi_modulate_frozen)
            w2, dcoef = pre
            assert w2.shape == (ns, o, kh, kw, i) and (dcoef is not None) == bool(demodulate)
        else:
            w2 = torch.empty(ns, o, kh, kw, i, device=x.device, dtype=torch.float32)
            dcoef = torch.empty(ns, o, device=x.device, dtype=torch.float32) if demodulate else None
            hip.call('spi_modulate_fwd', hip.ptr(weight), hip.ptr(st), hip.ptr(w2), hip.ptr(dcoef), ns, o, i, kh * kw, int(demodulate),
                     float(style_gain), hip.stream())
        h, wd = x.shape[2], x.shape[3]
        wbs = o * i * kh * kw if ns == n and n > 1 else (0 if ns == 1 else o * i * kh * kw)
        if ns == 1:
            wbs = 0
        oh, ow = out_size(h, kh, pad, transposed), out_size(wd, kh, pad, transposed)
        bb = bias.detach().contiguous().float() if bias is not None else None
        nz = noise.detach().contiguous().float() if noise is not None else None
        ng = strength.detach().reshape(1).contiguous().float() if (noise is not None and strength is not None) else None
        d = _desc(n, i, o, h, wd, kh, pad, transposed, flip, wbs, bb, nz, ng, act_id, alpha, gain, clamp, tap_major=1, f16=f16, half=half)
        ws = _workspace(d, 0, x.device)                 # noqa: F841  (Winograd scratch, alive until the launch is enqueued)
        y = _out_tensor(d, 0, (n, o, oh, ow), x.device)
        hip.call('spi_conv2d_fwd', ctypes.byref(d), hip.ptr(x), hip.ptr(w2), hip.ptr(y), hip.stream())
        has_epi = act_id != 0 and (act_id != 1 or gain != 1 or clamp >= 0 or bb is not None or nz is not None)
        ctx.save_for_backward(x, weight, st, w2, dcoef, y, bb, nz, ng, ww)
        ctx.cfg = (pad, transposed, flip, act_id, alpha, gain, clamp, has_epi, wbs, demodulate, float(style_gain), f16)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        import ctypes
        from ..torch_utils.ops.conv2d_mfma import _desc, _workspace, _out_tensor
        x, weight, st, w2, dcoef, y, bb, nz, ng, ww = ctx.saved_tensors
        pad, transposed, flip, act_id, alpha, gain, clamp, has_epi, wbs, demodulate, sgain, f16 = ctx.cfg
        o, i, kh, kw = weight.shape
        n, ns = x.shape[0], st.shape[0]
        h, wd = x.shape[2], x.shape[3]
        # one zeroed buffer for every accumulator of this backward: layer-tail sums | <x_i, dx_i> | <dz_o, z_o>
        n_tail = bias_act.tail_zero_elems(dy, nz, ctx.needs_input_grad[4], ctx.needs_input_grad[5], ctx.needs_input_grad[3])
        zb = zero_arena.zeros(n_tail + n * i + (n * o if demodulate else 0), x.device)
        # <dz_o, z_o> comes out of the layer-tail pass itself when that pass runs (dz and y are in its registers; spi_tail_bwd_dot_t)
        from ..configs import global_config as _gc
        fuse_cv = bool(demodulate and has_epi and _gc.fuse_tail_dot)
        dz, d_noise, d_strength, d_bias = bias_act.tail_backward(dy, y if has_epi else None, nz, ng, act_id, alpha, gain, clamp,
                                                                 ctx.needs_input_grad[4], ctx.needs_input_grad[5], ctx.needs_input_grad[3],
                                                                 zero_buf=zb, zdot=((zb[n_tail + n * i:], bb, nz, ng) if fuse_cv else None))
        half = x.dtype == torch.float16
        d = _desc(n, i, o, h, wd, kh, pad, transposed, flip, wbs, tap_major=1, f16=f16, half=half)
        dd, dzd, w2d = d, dz, w2
        if half and o % 16 != 0:                       # (torgb: 3 output channels; conv2d_mfma.pad_o16)
            from ..torch_utils.ops.conv2d_mfma import pad_o16
            dzd, w2d, o16 = pad_o16(dz, w2, o)
            dd = _desc(n, i, o16, h, wd, kh, pad, transposed, flip, (o16 * i * kh * kw if wbs else 0), tap_major=1, f16=f16, half=True)
        ws = _workspace(dd, 1, x.device)                # noqa: F841
        dx = _out_tensor(dd, 1, tuple(x.shape), x.device)
        hip.call('spi_conv2d_dgrad', ctypes.byref(dd), hip.ptr(dzd), hip.ptr(w2d), hip.ptr(dx), hip.stream())
        a = zb[n_tail:n_tail + n * i]
        dt = (hip.DTYPE_IDS[torch.float16],) if half else ()
        cd = 'spi_chan_dot_t' if half else 'spi_chan_dot'
        hip.call(cd, hip.ptr(x), hip.ptr(dx), hip.ptr(a), n * i, i, h * wd, None, None, None, 0, 0.0, 1.0, *dt, hip.stream())
        cv = None
        if demodulate:
            cv = zb[n_tail + n * i:]
            hw_out = y.shape[2] * y.shape[3]
            if fuse_cv:
                pass
            elif has_epi:
                hip.call(cd, hip.ptr(dz), hip.ptr(y), hip.ptr(cv), n * o, o, hw_out, hip.ptr(bb), hip.ptr(nz), hip.ptr(ng), act_id, alpha,
                         gain, *dt, hip.stream())
            else:
                hip.call(cd, hip.ptr(dz), hip.ptr(y), hip.ptr(cv), n * o, o, hw_out, None, None, None, 0, 0.0, 1.0, *dt, hip.stream())
            if ww is None:
                ww = weight.square().sum(dim=(2, 3)).contiguous()                  # [O, I]
        ds = torch.empty(ns, i, device=x.device, dtype=torch.float32)
        hip.call('spi_style_grad', hip.ptr(a), hip.ptr(cv), hip.ptr(st), hip.ptr(dcoef), hip.ptr(ww), hip.ptr(ds), n, ns, i, o, sgain, hip.stream())
        return (dx if ctx.needs_input_grad[0] else None, None, ds, d_bias, d_noise, d_strength,
                None, None, None, None, None, None, None, None, None, None, None, None)


@misc.profiled_function
def modulated_conv2d(x, weight, styles, noise=None, up=1, down=1, padding=0, resample_filter=None, demodulate=True,
                     flip_weight=True, fused_modconv=True, noise_strength=None, bias=None, act=None, gain=None, clamp=None,
                     style_gain=1.0, fp16=False, w_mod=None):
    """Modulate -> (demodulate) -> conv [-> FIR] [-> + noise -> + bias -> act], reference :34-91 (fused path).

    ``noise`` may be the final noise tensor (reference style) or, with ``noise_strength`` given, the
    raw ``noise_const`` buffer so that both receive gradients from the fused epilogue.
    ``bias``/``act``/``gain``/``clamp`` optionally fuse the bias_act that follows in SynthesisLayer / ToRGBLayer.
    """
    if not fused_modconv:
        raise NotImplementedError('only the fused modulated convolution is implemented (SPI runs the generator in eval mode)')
    assert down == 1 and up in (1, 2)
    n = x.shape[0]
    oc, ic, kh, kw = weight.shape
    # styles [N, I], or [1, I] shared by the whole batch (same w for every view: the weights are modulated once, the
    # conv runs with one weight set and its weight gradient is reduced over the batch inside the kernel)
    assert styles.shape in ((n, ic), (1, ic))
    # modulated (and demodulated) per-sample weights, built directly in the kernels' tap-major layout [N,O,k,k,I]
    frozen = torch.is_grad_enabled() and not weight.requires_grad and styles.requires_grad
    if frozen:                      # stage 1: no weight-gradient GEMM, the style gradient comes from two channel dot products
        from ..torch_utils.ops import bias_act as _ba
        if up == 1:
            if act is None:
                act_id, a_, g_, c_ = (1 if (bias is not None or noise is not None) else 0), 0.0, 1.0, -1.0
            else:
                act_id, d_alpha, d_gain, _ = _ba.activation_funcs[act]
                assert act_id in (1, 3), 'fused epilogue of the frozen-weight path supports linear / lrelu'
                a_, g_, c_ = float(d_alpha), float(d_gain if gain is None else gain), float(-1 if clamp is None else clamp)
            return _ModConvFrozen.apply(x, weight, styles, bias, noise, noise_strength, int(padding), False, not flip_weight, act_id, a_, g_, c_,
                                        bool(demodulate), float(style_gain), conv2d_mfma.precision(fp16), _tap_energy(weight) if demodulate else None,
                                        w_mod if isinstance(w_mod, FrozenMod) else None)
        assert kh == 3 and padding == 1 and resample_filter is not None and resample_filter.ndim == 2
        z = _ModConvFrozen.apply(x, weight, styles, None, None, None, 0, True, flip_weight, 0, 0.0, 1.0, -1.0, bool(demodulate),
                                 float(style_gain), conv2d_mfma.precision(fp16), _tap_energy(weight) if demodulate else None,
                                 w_mod if isinstance(w_mod, FrozenMod) else None)
        return upfirdn2d.upfirdn2d_bias_act(z, resample_filter, noise=noise, noise_strength=noise_strength, bias=bias,
                                            padding=[1, 1, 1, 1], gain=up ** 2, act=(act or 'linear'), act_gain=gain, clamp=clamp)
    assert not isinstance(w_mod, FrozenMod)
    w = w_mod if w_mod is not None else modulate_weights(weight, styles, demodulate, style_gain)   # w_mod: this layer's share of `multi_modulate`
    if styles.shape[0] == 1 and n > 1:
        w = w[0]
    if up == 1:
        return conv2d_mfma.conv2d(x, w, bias=bias, noise=noise, noise_strength=noise_strength, padding=padding,
                                  flip=not flip_weight, act=act, gain=gain, clamp=clamp, tap_major=True, fp16=fp16, sparse_grad=True)
    # up = 2: stride-2 transposed conv, then the 4x4 low-pass (gain up^2) with the layer tail fused in
    assert kh == 3 and padding == 1 and resample_filter is not None and resample_filter.ndim == 2
    z = conv2d_mfma.conv2d(x, w, transposed=True, flip=flip_weight, tap_major=True, fp16=fp16, sparse_grad=True)
    return upfirdn2d.upfirdn2d_bias_act(z, resample_filter, noise=noise, noise_strength=noise_strength, bias=bias,
                                        padding=[1, 1, 1, 1], gain=up ** 2, act=(act or 'linear'), act_gain=gain, clamp=clamp)


class _LinearGain(torch.autograd.Function):
    """y = bias + gain * x W^T.  At inversion batch sizes (<= 8 rows: one image, or the four pseudo-views) these are matrix-VECTOR products: a
    library GEMM runs them on one workgroup (18 us forward, 2 x 7.5 us backward; 29 affine layers per generator pass = 2 % of the step's GPU
    time), `spi_affine_fwd / _bwd` read W once with the whole chip and produce dx and dW in one pass.  Larger batches keep the library GEMM,
    with the gain riding as alpha in both directions (torch's own addmm backward scales the weight gradient with a separate [out, in]
    elementwise launch and materialises the broadcast bias gradient through a fill)."""

    @staticmethod
    def _small(x, weight):
        return x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 2 and 1 <= x.shape[0] <= 8 and x.shape[1] % 4 == 0

    @staticmethod
    def forward(ctx, x, weight, bias, gain):
        ctx.gain = float(gain)
        if _LinearGain._small(x, weight):
            x, weight, bias = x.contiguous(), weight.contiguous(), bias.contiguous()
            ctx.save_for_backward(x, weight)
            y = torch.empty(x.shape[0], weight.shape[0], device=x.device, dtype=torch.float32)
            hip.call('spi_affine_fwd', hip.ptr(x), hip.ptr(weight), hip.ptr(bias), ctx.gain, hip.ptr(y), x.shape[0], x.shape[1], weight.shape[0], hip.stream())
            return y
        ctx.save_for_backward(x, weight)
        return torch.addmm(bias.unsqueeze(0), x, weight.t(), alpha=ctx.gain)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        dx = dw = db = None
        if _LinearGain._small(x, weight) and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            g = g.contiguous()
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
            if ctx.needs_input_grad[1]:
                dw = torch.empty_like(weight)
            hip.call('spi_affine_bwd', hip.ptr(g), hip.ptr(x), hip.ptr(weight), ctx.gain, hip.ptr(dx), hip.ptr(dw), x.shape[0], x.shape[1], weight.shape[0], hip.stream())
        else:
            if ctx.needs_input_grad[0]:
                dx = torch.addmm(g.new_empty(x.shape), g, weight, beta=0, alpha=ctx.gain)            # beta = 0: the input tensor is ignored
            if ctx.needs_input_grad[1]:
                dw = torch.addmm(g.new_empty(weight.shape), g.t(), x, beta=0, alpha=ctx.gain)
        if ctx.needs_input_grad[2]:
            db = g[0] if g.shape[0] == 1 else g.sum(0)
        return dx, dw, db, None


class _MultiAffine(torch.autograd.Function):
    """Every affine (style) layer of a synthesis network in ONE launch each way (`spi_affine_multi_fwd / _bwd`): layer l maps row idx[l] of
    ws [N, L, 512] through its FullyConnectedLayer (networks_stylegan2.py:95-127 of the reference, activation 'linear').  The backbone has 20 of them
    and a super-resolution call 6 -- 5 us matrix-vector launches that head (forward) and tail (backward) every layer's chain; the same arithmetic
    per element as the per-layer `_LinearGain` (bit-equal forward), the ws gradient is summed over the layers that share a row inside the launch.
    Inputs: ws, then (weight, bias) per layer.  Outputs: one style tensor [N, O_l] per layer (views of one buffer)."""

    @staticmethod
    def forward(ctx, ws, idx, gains, *wb):
        n, L, I = ws.shape
        nl = len(idx)
        weights, biases = wb[0::2], wb[1::2]
        outs = [w_.shape[0] for w_ in weights]
        flat = torch.empty(n * sum(outs), device=ws.device, dtype=torch.float32)
        ys, off = [], 0
        for o in outs:
            ys.append(flat[off:off + n * o].view(n, o))
            off += n * o
        jobs = (hip.AffineJob * nl)()
        base = ws.data_ptr()
        for l in range(nl):
            jobs[l].x, jobs[l].w, jobs[l].b, jobs[l].y = base + 4 * I * idx[l], weights[l].data_ptr(), biases[l].data_ptr(), ys[l].data_ptr()
            jobs[l].gain, jobs[l].O = gains[l], outs[l]
        hip.call('spi_affine_multi_fwd', jobs, nl, n, I, L * I, hip.stream())
        ctx.save_for_backward(ws, *weights)
        ctx.idx, ctx.gains, ctx.outs = idx, gains, outs
        ctx.set_materialize_grads(False)                         # a layer whose styles reach no loss arrives as None in backward and is skipped
        return tuple(ys)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gs):
        ws, *weights = ctx.saved_tensors
        n, L, I = ws.shape
        nl = len(ctx.idx)
        need_ws = ctx.needs_input_grad[0]
        d_ws = zero_arena.zeros((n, L, I), ws.device) if need_ws else None
        jobs = (hip.AffineJob * nl)()
        base, dbase = ws.data_ptr(), (d_ws.data_ptr() if need_ws else 0)
        grads = [None] * (2 * nl)
        keep, k = [], 0
        for l in range(nl):
            g = gs[l]
            if g is None:                                        # the layer's styles reached no loss
                continue
            g = g.contiguous().float()
            keep.append(g)
            need_w, need_b = ctx.needs_input_grad[3 + 2 * l], ctx.needs_input_grad[4 + 2 * l]
            j = jobs[k]
            k += 1
            j.x, j.w, j.g = base + 4 * I * ctx.idx[l], weights[l].data_ptr(), g.data_ptr()
            j.gain, j.O = ctx.gains[l], ctx.outs[l]
            if need_ws:
                j.dx_acc = dbase + 4 * I * ctx.idx[l]
            if need_w:
                grads[2 * l] = torch.empty_like(weights[l])
                j.dw = grads[2 * l].data_ptr()
            if need_b:
                grads[2 * l + 1] = g[0] if n == 1 else g.sum(0)
        if k:
            hip.call('spi_affine_multi_bwd', jobs, k, n, I, L * I, hip.stream())
        return (d_ws, None, None, *grads)


def multi_affine(ws, layers):
    """Styles of `layers` = [(module with .affine, row of ws)] in one launch, or None when the fast path does not apply (then every layer runs its own
    affine): fp32 ws [N <= 8, L, 512] on the GPU, linear FullyConnectedLayers with bias, at most 32 of them."""
    if not (ws.is_cuda and ws.dtype == torch.float32 and ws.dim() == 3 and 1 <= ws.shape[0] <= 8 and ws.shape[2] % 4 == 0 and 1 <= len(layers) <= hip.AFFINE_MAX_JOBS):
        return None
    wb, gains, idx = [], [], []
    for mod, row in layers:
        a = mod.affine
        if a.activation != 'linear' or a.bias is None or a.weight.dtype != torch.float32 or a.bias_gain != 1 or a.weight.shape[1] != ws.shape[2]:
            return None
        wb += [a.weight.contiguous(), a.bias.contiguous()]
        gains.append(float(a.weight_gain))
        idx.append(int(row))
    return _MultiAffine.apply(ws.contiguous(), tuple(idx), tuple(gains), *wb)


class FullyConnectedLayer(torch.nn.Module):
    def __init__(self, in_features, out_features, bias=True, activation='linear', lr_multiplier=1, bias_init=0):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.activation = activation
        self.weight = torch.nn.Parameter(torch.randn([out_features, in_features]) / lr_multiplier)
        self.bias = torch.nn.Parameter(torch.full([out_features], np.float32(bias_init))) if bias else None
        self.weight_gain = lr_multiplier / np.sqrt(in_features)
        self.bias_gain = lr_multiplier

    def forward(self, x):
        b = self.bias
        if b is not None and self.bias_gain != 1:
            b = b * self.bias_gain
        if self.activation == 'linear' and b is not None:
            # plain library GEMM; the weight gain rides along as alpha instead of a [out, in] elementwise pass each way
            return _LinearGain.apply(x, self.weight.to(x.dtype), b, self.weight_gain)
        w = self.weight.to(x.dtype) * self.weight_gain
        return bias_act.bias_act(x.matmul(w.t()), b, act=self.activation)

    def extra_repr(self):
        return f'in_features={self.in_features:d}, out_features={self.out_features:d}, activation={self.activation:s}'


class MappingNetwork(torch.nn.Module):
    def __init__(self, z_dim, c_dim, w_dim, num_ws, num_layers=8, embed_features=None, layer_features=None,
                 activation='lrelu', lr_multiplier=0.01, w_avg_beta=0.998):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim, self.num_ws = z_dim, c_dim, w_dim, num_ws
        self.num_layers, self.w_avg_beta = num_layers, w_avg_beta
        if embed_features is None:
            embed_features = w_dim
        if c_dim == 0:
            embed_features = 0
        if layer_features is None:
            layer_features = w_dim
        feats = [z_dim + embed_features] + [layer_features] * (num_layers - 1) + [w_dim]
        if c_dim > 0:
            self.embed = FullyConnectedLayer(c_dim, embed_features)
        for idx in range(num_layers):
            setattr(self, f'fc{idx}', FullyConnectedLayer(feats[idx], feats[idx + 1], activation=activation, lr_multiplier=lr_multiplier))
        if num_ws is not None and w_avg_beta is not None:
            self.register_buffer('w_avg', torch.zeros([w_dim]))

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False):
        x = None
        if self.z_dim > 0:
            x = normalize_2nd_moment(z.to(torch.float32))
        if self.c_dim > 0:
            y = normalize_2nd_moment(self.embed(c.to(torch.float32)))
            x = torch.cat([x, y], dim=1) if x is not None else y
        for idx in range(self.num_layers):
            x = getattr(self, f'fc{idx}')(x)
        if update_emas and self.w_avg_beta is not None:
            self.w_avg.copy_(x.detach().mean(dim=0).lerp(self.w_avg, self.w_avg_beta))
        if self.num_ws is not None:
            x = x.unsqueeze(1).repeat([1, self.num_ws, 1])
        if truncation_psi != 1:
            if self.num_ws is None or truncation_cutoff is None:
                x = self.w_avg.lerp(x, truncation_psi)
            else:
                x[:, :truncation_cutoff] = self.w_avg.lerp(x[:, :truncation_cutoff], truncation_psi)
        return x


class SynthesisLayer(torch.nn.Module):
    def __init__(self, in_channels, out_channels, w_dim, resolution, kernel_size=3, up=1, use_noise=True, activation='lrelu',
                 resample_filter=[1, 3, 3, 1], conv_clamp=None, channels_last=False):
        super().__init__()
        self.in_channels, self.out_channels, self.w_dim, self.resolution = in_channels, out_channels, w_dim, resolution
        self.up, self.use_noise, self.activation, self.conv_clamp = up, use_noise, activation, conv_clamp
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.padding = kernel_size // 2
        self.act_gain = bias_act.def_gain(activation)
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]))
        if use_noise:
            self.register_buffer('noise_const', torch.randn([resolution, resolution]))
            self.noise_strength = torch.nn.Parameter(torch.zeros([]))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))

    def forward(self, x, w, noise_mode='random', fused_modconv=True, gain=1, fp16=False, noise_rng=None, styles=None, w_mod=None):
        """noise_rng (extension): draw source with ``randn(*shape)`` for noise_mode='random' (tests replay recorded draws).
        styles (extension): this layer's ``self.affine(w)``, already computed (all layers of a network in one launch: `multi_affine`)."""
        assert noise_mode in ['random', 'const', 'none']
        if styles is None:
            styles = self.affine(w)
        noise = strength = None
        if self.use_noise and noise_mode == 'random':
            shape = [1, 1, self.resolution, self.resolution]                           # the reference's draw shape (:317) at N = 1
            noise = (noise_rng.randn(*shape) if noise_rng is not None else torch.randn(shape, device=x.device)).reshape(shape[2:])
            if x.shape[0] != 1:
                raise NotImplementedError("noise_mode='random' with batch > 1 is not on the SPI path")
            strength = self.noise_strength
        if self.use_noise and noise_mode == 'const':
            noise, strength = self.noise_const, self.noise_strength
        clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        return modulated_conv2d(x=x, weight=self.weight, styles=styles, noise=noise, noise_strength=strength, up=self.up,
                                padding=self.padding, resample_filter=self.resample_filter, flip_weight=(self.up == 1),
                                fused_modconv=fused_modconv, bias=self.bias, act=self.activation, gain=self.act_gain * gain,
                                clamp=clamp, fp16=fp16, w_mod=w_mod)

    def extra_repr(self):
        return f'in_channels={self.in_channels:d}, out_channels={self.out_channels:d}, resolution={self.resolution:d}, up={self.up}'


class ToRGBLayer(torch.nn.Module):
    def __init__(self, in_channels, out_channels, w_dim, kernel_size=1, conv_clamp=None, channels_last=False):
        super().__init__()
        self.in_channels, self.out_channels, self.w_dim, self.conv_clamp = in_channels, out_channels, w_dim, conv_clamp
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))
        self.weight_gain = 1 / np.sqrt(in_channels * (kernel_size ** 2))

    def forward(self, x, w, fused_modconv=True, fp16=False, styles=None, w_mod=None):
        if styles is None:
            styles = self.affine(w)                                # * weight_gain happens inside the modulation kernel
        return modulated_conv2d(x=x, weight=self.weight, styles=styles, demodulate=False, fused_modconv=fused_modconv,
                                bias=self.bias, act='linear', gain=1, clamp=self.conv_clamp, style_gain=self.weight_gain, fp16=fp16, w_mod=w_mod)


class SynthesisBlock(torch.nn.Module):
    def __init__(self, in_channels, out_channels, w_dim, resolution, img_channels, is_last, architecture='skip',
                 resample_filter=[1, 3, 3, 1], conv_clamp=256, use_fp16=False, fp16_channels_last=False,
                 fused_modconv_default=True, **layer_kwargs):
        assert architecture == 'skip', "only the 'skip' architecture is on the SPI path"
        super().__init__()
        self.in_channels, self.w_dim, self.resolution, self.img_channels = in_channels, w_dim, resolution, img_channels
        self.is_last, self.architecture, self.use_fp16 = is_last, architecture, use_fp16
        self.fused_modconv_default = fused_modconv_default
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.num_conv = 0
        self.num_torgb = 0
        if in_channels == 0:
            self.const = torch.nn.Parameter(torch.randn([out_channels, resolution, resolution]))
        else:
            self.conv0 = SynthesisLayer(in_channels, out_channels, w_dim=w_dim, resolution=resolution, up=2,
                                        resample_filter=resample_filter, conv_clamp=conv_clamp, **layer_kwargs)
            self.num_conv += 1
        self.conv1 = SynthesisLayer(out_channels, out_channels, w_dim=w_dim, resolution=resolution, conv_clamp=conv_clamp, **layer_kwargs)
        self.num_conv += 1
        self.torgb = ToRGBLayer(out_channels, img_channels, w_dim=w_dim, conv_clamp=conv_clamp)
        self.num_torgb += 1

    def affine_layers(self):
        """the block's style layers in execution order"""
        return ([self.conv0] if self.in_channels != 0 else []) + [self.conv1, self.torgb]

    def forward(self, x, img, ws, force_fp32=False, fused_modconv=None, update_emas=False, styles=None, w_mods=None, **layer_kwargs):
        rows = ws if isinstance(ws, (tuple, list)) else None          # SynthesisNetwork hands over the rows of ONE unbind (see there)
        if styles is None and rows is None:                           # called on its own (super-resolution module): the block's 2-3 affines in one launch
            styles = multi_affine(ws, [(m, k) for k, m in enumerate(self.affine_layers())])
            w_mods = multi_modulate(self.affine_layers(), styles)
        s_iter = iter(zip(styles, w_mods if w_mods is not None else [None] * len(styles))) if styles is not None else None
        nxt = (lambda: dict(zip(('styles', 'w_mod'), next(s_iter)))) if s_iter is not None else (lambda: {})
        if rows is None:
            assert ws.shape[1] == self.num_conv + self.num_torgb and ws.shape[2] == self.w_dim
        else:
            assert len(rows) == self.num_conv + self.num_torgb
        if fused_modconv is None:
            fused_modconv = self.fused_modconv_default
        if fused_modconv == 'inference_only':
            fused_modconv = not self.training
        # reference rule (:421-423): fp16 iff use_fp16 and not force_fp32 and the tensor is on 'cuda'.  Here it is opt-in
        # (global_config.enable_fp16_blocks, `--sr_fp16`): default fp32 everywhere = the reference's CPU path, which is the
        # parity target.  Round 5: like the reference's (:423-436: `x = x.to(dtype=torch.float16)`, every conv / FIR / bias_act of the block on half
        # tensors, `y.to(torch.float32)` for the skip image) the block's ACTIVATIONS are fp16 tensors in HBM -- NCHW, not channels_last: the kernels'
        # pixel-major loaders stay coalesced -- with fp32 weights, bias, noise and accumulators (global_config.fp16_storage; off = rounds 3-4:
        # fp32 tensors, conv operands rounded to fp16 on their way into the MFMAs).
        f16 = bool(self.use_fp16 and not force_fp32 and global_config.enable_fp16_blocks)
        half = f16 and global_config.fp16_storage and x is not None and x.is_cuda and self.in_channels % 16 == 0 and self.conv1.out_channels % 16 == 0
        w_iter = iter(rows if rows is not None else ws.unbind(dim=1))   # (:432) one unbind -> one stack in the backward instead of a zero-fill + copy per row
        if self.in_channels == 0:
            x = self.const.unsqueeze(0).repeat([(rows[0] if rows is not None else ws).shape[0], 1, 1, 1])
        else:
            x = self.conv0(x.to(torch.float16) if half else x.float(), next(w_iter), fused_modconv=fused_modconv, fp16=f16, **nxt(), **layer_kwargs)
        x = self.conv1(x, next(w_iter), fused_modconv=fused_modconv, fp16=f16, **nxt(), **layer_kwargs)
        if img is not None:
            img = upfirdn2d.upsample2d(img, self.resample_filter)
        y = self.torgb(x, next(w_iter), fused_modconv=fused_modconv, fp16=f16, **nxt())
        if y.dtype != torch.float32:
            y = y.float()                                          # (:443) the skip image is accumulated in fp32
        img = img + y if img is not None else y
        return x, img


class SynthesisNetwork(torch.nn.Module):
    def __init__(self, w_dim, img_resolution, img_channels, channel_base=32768, channel_max=512, num_fp16_res=4, **block_kwargs):
        assert img_resolution >= 4 and img_resolution & (img_resolution - 1) == 0
        super().__init__()
        self.w_dim, self.img_resolution, self.img_channels, self.num_fp16_res = w_dim, img_resolution, img_channels, num_fp16_res
        self.img_resolution_log2 = int(np.log2(img_resolution))
        self.block_resolutions = [2 ** i for i in range(2, self.img_resolution_log2 + 1)]
        channels = {res: min(channel_base // res, channel_max) for res in self.block_resolutions}
        fp16_resolution = max(2 ** (self.img_resolution_log2 + 1 - num_fp16_res), 8)
        self.num_ws = 0
        for res in self.block_resolutions:
            block = SynthesisBlock(channels[res // 2] if res > 4 else 0, channels[res], w_dim=w_dim, resolution=res,
                                   img_channels=img_channels, is_last=(res == self.img_resolution),
                                   use_fp16=(res >= fp16_resolution), **block_kwargs)
            self.num_ws += block.num_conv
            if res == self.img_resolution:
                self.num_ws += block.num_torgb
            setattr(self, f'b{res}', block)

    def forward(self, ws, **block_kwargs):
        assert ws.shape[1] == self.num_ws and ws.shape[2] == self.w_dim
        ws = ws.to(torch.float32)
        x = img = None
        w_idx = 0
        # the reference narrows ws per block (:509-511) and unbinds inside it: per block a zero-filled [N, num_ws, 512] gradient, a slice copy and an
        # accumulation in the backward.  One unbind for the whole network gives the same rows and ONE stack in the backward.
        rows = ws.unbind(dim=1)
        # every style layer of the network reads its row of ws through its own affine: all of them in one launch (and one in the backward)
        layers, k = [], 0
        for res in self.block_resolutions:
            block = getattr(self, f'b{res}')
            layers += [(m, k + j) for j, m in enumerate(block.affine_layers())]
            k += block.num_conv
        styles = multi_affine(ws, layers)
        w_mods = multi_modulate([m for m, _ in layers], styles)       # ... and so is every layer's weight modulation (not on stage 1's frozen-weight path)
        s_pos = 0
        for res in self.block_resolutions:
            block = getattr(self, f'b{res}')
            nb = block.num_conv + block.num_torgb
            x, img = block(x, img, rows[w_idx:w_idx + nb], styles=(styles[s_pos:s_pos + nb] if styles is not None else None),
                           w_mods=(w_mods[s_pos:s_pos + nb] if w_mods is not None else None), **block_kwargs)
            w_idx += block.num_conv
            s_pos += nb
        return img


class Generator(torch.nn.Module):
    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels, mapping_kwargs={}, **synthesis_kwargs):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim = z_dim, c_dim, w_dim
        self.img_resolution, self.img_channels = img_resolution, img_channels
        self.synthesis = SynthesisNetwork(w_dim=w_dim, img_resolution=img_resolution, img_channels=img_channels, **synthesis_kwargs)
        self.num_ws = self.synthesis.num_ws
        self.mapping = MappingNetwork(z_dim=z_dim, c_dim=c_dim, w_dim=w_dim, num_ws=self.num_ws, **mapping_kwargs)

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False, **synthesis_kwargs):
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff, update_emas=update_emas)
        return self.synthesis(ws, update_emas=update_emas, **synthesis_kwargs)
