"""GPU parity of the MFMA convolution kernels and the modulated-conv layer against the oracle / golden vectors."""
import os
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close, rel_err
from oracle import stylegan_ref as osg

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _ref_conv(x, w, pad, transposed, flip):
    """per-sample weights via a python loop over the batch (CPU oracle)."""
    outs = []
    for n in range(x.shape[0]):
        wn = w[n] if w.ndim == 5 else w
        if flip:
            wn = wn.flip([2, 3])
        if transposed:
            outs.append(F.conv_transpose2d(x[n:n + 1], wn.transpose(0, 1), stride=2))
        else:
            outs.append(F.conv2d(x[n:n + 1], wn, padding=pad))
    return torch.cat(outs)


CASES = [  # N, I, O, H, k, pad, transposed, flip, per_sample
    (1, 8, 12, 6, 3, 1, False, False, True), (2, 8, 12, 6, 3, 1, False, True, True), (2, 8, 5, 6, 1, 0, False, False, True),
    (2, 8, 12, 5, 3, 0, True, False, True), (1, 8, 12, 5, 3, 0, True, True, True), (2, 3, 64, 20, 3, 1, False, False, False),
    (1, 40, 130, 33, 3, 1, False, False, True), (2, 130, 40, 17, 3, 0, True, False, True), (1, 128, 3, 40, 1, 0, False, False, True),
    (1, 64, 64, 4, 3, 1, False, False, True), (3, 32, 96, 16, 1, 0, False, False, True), (1, 70, 200, 64, 3, 1, False, False, True),
    # narrow-generator shapes with a batch of per-sample weights (few channels, many pixels)
    (2, 32, 16, 64, 3, 0, True, False, True), (2, 8, 8, 256, 3, 1, False, True, True), (2, 16, 8, 128, 3, 0, True, False, True),
    (2, 16, 16, 128, 3, 1, False, True, True), (3, 8, 96, 256, 1, 0, False, False, True),
]
# the real layer shapes of the ffhqrebalanced512-128 generator (the instances bench.py times): 512-channel split-K layer,
# stride-2 transposed 256 -> 128 at 256^2, the 128 -> 128 conv at 512^2, an N = 4 batch sharing one weight set (rot / depth branches),
# the 512 -> 512 transposed conv at 16^2, torgb 128 -> 96 at 256^2, the VGG16 3 -> 64 stem at 256^2
FULL_CASES = [
    (1, 512, 512, 16, 3, 1, False, True, True), (1, 256, 128, 256, 3, 0, True, False, True), (1, 128, 128, 512, 3, 1, False, True, True),
    (4, 128, 128, 256, 3, 1, False, True, False), (1, 512, 512, 16, 3, 0, True, False, True), (1, 128, 96, 256, 1, 0, False, False, True),
    (2, 3, 64, 256, 3, 1, False, False, False), (2, 512, 512, 4, 3, 1, False, True, True),
]


@pytest.mark.parametrize('case', CASES + FULL_CASES)
def test_conv_fwd_dgrad_wgrad_vs_oracle(case):
    from spi_amd.torch_utils.ops import conv2d_mfma
    N, I, O, H, k, pad, tr, flip, per = case
    big = case in FULL_CASES                        # K up to 4608 terms, weight gradients reduce over up to 1 M pixels: fp32 summation order shows
    torch.set_num_threads(min(__import__('os').cpu_count() or 1, 32))
    gen = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(N, I, H, H + 1, generator=gen, requires_grad=True)          # non-square on purpose
    w = (torch.randn(*((N,) if per else ()), O, I, k, k, generator=gen) / (I * k * k) ** 0.5).requires_grad_(True)
    ref = _ref_conv(x, w, pad, tr, flip)
    dy = torch.randn(ref.shape, generator=gen)
    gx, gw = torch.autograd.grad(ref, [x, w], dy)
    xg, wg = x.detach().to(DEV).requires_grad_(True), w.detach().to(DEV).requires_grad_(True)
    y = conv2d_mfma.conv2d(xg, wg, padding=pad, transposed=tr, flip=flip)
    assert y.shape == ref.shape
    assert_close(y, ref, 1e-5 if big else 3e-6, 'conv fwd')
    hx, hw = torch.autograd.grad(y, [xg, wg], dy.to(DEV))
    assert_close(hx, gx, 1e-5 if big else 3e-6, 'conv dgrad')
    assert_close(hw, gw, 1e-4 if big else 2e-5, 'conv wgrad')


TWGRAD_CASES = [  # N, I, O, H, W, per_sample, masked      (stride-2 transposed 3x3, channels in whole 64-blocks, W % 16 == 0: twgrad_kernel)
    (1, 256, 128, 256, 256, True, False),      # SR block1.conv0 / b256.conv0: the most expensive conv launch of the loop
    (2, 64, 128, 40, 48, True, False),         # per-sample weights, rows that do not fill a chunk evenly
    (3, 128, 64, 32, 32, False, False),        # batch summed into ONE weight set
    (2, 64, 64, 64, 64, False, True),          # masked gradient (dy_seg_flags): steps without a flagged segment are skipped
]


@pytest.mark.parametrize('case', TWGRAD_CASES)
def test_conv_transposed_wgrad_direct_kernel_vs_oracle(case):
    """Round 6: the weight gradient of the stride-2 transposed 3x3 convolutions on its own kernel (twgrad_kernel: 64 x 64 channels x all nine taps per
    block, dY rows de-interleaved into the three kx operand streams on their way into LDS) against autograd of F.conv_transpose2d on the CPU.  The sums run
    over up to 65 536 pixels with fp32 atomics between blocks: 1e-4 of the tensor's range on the real layer size like the other weight-gradient tests."""
    from spi_amd.torch_utils.ops import conv2d_mfma
    N, I, O, H, W, per, masked = case
    torch.set_num_threads(min(__import__('os').cpu_count() or 1, 32))
    gen = torch.Generator().manual_seed(N * 10 + O)
    x = torch.randn(N, I, H, W, generator=gen)
    w = (torch.randn(*((N,) if per else ()), O, I, 3, 3, generator=gen) / (I * 9) ** 0.5).requires_grad_(True)
    ref = _ref_conv(x, w, 0, True, False)
    dy = torch.randn(ref.shape, generator=gen)
    if masked:
        m = torch.zeros(ref.shape[-2:])
        m[20:60, 70:110] = 1                       # a box: most 16-pixel segments of dY are exactly zero
        dy = dy * m
    gw, = torch.autograd.grad(ref, [w], dy)
    xg, wg = x.to(DEV).requires_grad_(True), w.detach().to(DEV).requires_grad_(True)
    y = conv2d_mfma.conv2d(xg, wg, padding=0, transposed=True, flip=False, sparse_grad=masked)
    if masked:
        with conv2d_mfma.sparse_gradients():
            hw, = torch.autograd.grad(y, [wg], dy.to(DEV))
    else:
        hw, = torch.autograd.grad(y, [wg], dy.to(DEV))
    assert_close(hw, gw, 1e-4 if H * W >= 65536 else 2e-5, 'transposed conv wgrad (direct kernel)')


def test_conv_fused_epilogue_vs_oracle():
    from spi_amd.torch_utils.ops import conv2d_mfma
    gen = torch.Generator().manual_seed(8)
    x = torch.randn(2, 16, 12, 12, generator=gen, requires_grad=True)
    w = (torch.randn(2, 24, 16, 3, 3, generator=gen) / 12).requires_grad_(True)
    b = torch.randn(24, generator=gen, requires_grad=True)
    nz = torch.randn(12, 12, generator=gen, requires_grad=True)
    st = torch.tensor(0.4, requires_grad=True)
    ref = osg.bias_act(_ref_conv(x, w, 1, False, False) + nz * st, b, act='lrelu', gain=1.2, clamp=1.0)
    dy = torch.randn(ref.shape, generator=gen)
    gref = torch.autograd.grad(ref, [x, w, b, nz, st], dy)
    t = [v.detach().to(DEV).requires_grad_(True) for v in (x, w, b, nz, st)]
    y = conv2d_mfma.conv2d(t[0], t[1], bias=t[2], noise=t[3], noise_strength=t[4], padding=1, act='lrelu', gain=1.2, clamp=1.0)
    assert_close(y, ref, 3e-6, 'fused fwd')
    for a, bb, nm in zip(torch.autograd.grad(y, t, dy.to(DEV)), gref, ('dx', 'dw', 'db', 'dnoise', 'dstrength')):
        assert_close(a, bb, 2e-5, 'fused ' + nm)


def test_modulated_conv2d_golden(golden):
    from spi_amd.training.networks_stylegan2 import modulated_conv2d
    g = golden('ops')
    f = g['fir'].to(DEV)
    for tag, (k, up, demod) in dict(c1=(3, 1, True), c0=(3, 2, True), rgb=(1, 1, False)).items():
        x = g[f'mc_{tag}_x'].to(DEV).requires_grad_(True)
        w = g[f'mc_{tag}_w'].to(DEV).requires_grad_(True)
        s = g[f'mc_{tag}_s'].to(DEV).requires_grad_(True)
        noise = g[f'mc_{tag}_noise'].to(DEV) if f'mc_{tag}_noise' in g else None
        y = modulated_conv2d(x=x, weight=w, styles=s, noise=noise, up=up, padding=k // 2, resample_filter=f, demodulate=demod,
                             flip_weight=(up == 1), fused_modconv=True)
        assert_close(y, g[f'mc_{tag}_y'], 5e-6, f'modconv {tag} fwd')
        gx, gw, gs = torch.autograd.grad(y, [x, w, s], g[f'mc_{tag}_dy'].to(DEV))
        assert_close(gx, g[f'mc_{tag}_gx'], 1e-5, f'modconv {tag} dx')
        assert_close(gw, g[f'mc_{tag}_gw'], 2e-5, f'modconv {tag} dw')
        assert_close(gs, g[f'mc_{tag}_gs'], 2e-5, f'modconv {tag} dstyles')


@pytest.mark.parametrize('n,o,i,k,demod', [(2, 24, 40, 3, True), (1, 512, 512, 3, True), (3, 3, 128, 1, False), (2, 96, 32, 1, False), (1, 7, 5, 3, True)])
def test_modulate_weights_vs_torch(n, o, i, k, demod):
    """spi_modulate_fwd/bwd against the reference expression (networks_stylegan2.py:62-69), output in tap-major layout."""
    from spi_amd.training.networks_stylegan2 import modulate_weights
    gen = torch.Generator().manual_seed(n * 1000 + o)
    w = torch.randn(o, i, k, k, generator=gen, dtype=torch.float64, requires_grad=True)
    s = (torch.randn(n, i, generator=gen, dtype=torch.float64) + 1).requires_grad_(True)
    g = torch.randn(n, o, k, k, i, generator=gen, dtype=torch.float64)
    sg = 1.0 if demod else 0.37                      # ToRGB folds its weight_gain into the kernel
    ref = w.unsqueeze(0) * (s * sg).reshape(n, 1, i, 1, 1)
    if demod:
        ref = ref * (ref.square().sum(dim=[2, 3, 4]) + 1e-8).rsqrt().reshape(n, o, 1, 1, 1)
    ref = ref.permute(0, 1, 3, 4, 2)
    gw, gs = torch.autograd.grad(ref, [w, s], g)
    wd, sd = w.detach().float().to(DEV).requires_grad_(True), s.detach().float().to(DEV).requires_grad_(True)
    out = modulate_weights(wd, sd, demod, sg)
    assert out.shape == (n, o, k, k, i) and out.is_contiguous()
    assert_close(out, ref.float(), 2e-5, 'modulate fwd')
    a, b = torch.autograd.grad(out, [wd, sd], g.float().to(DEV))
    assert_close(a, gw.float(), 5e-5, 'modulate dW')
    assert_close(b, gs.float(), 5e-5, 'modulate ds')
    # frozen weights (stage 1): only d_styles
    out = modulate_weights(wd.detach(), sd, demod, sg)
    (b2,) = torch.autograd.grad(out, [sd], g.float().to(DEV))
    assert_close(b2, gs.float(), 5e-5, 'modulate ds (weights frozen)')


@pytest.mark.parametrize('n,i,o,h,k,transposed', [(2, 32, 48, 24, 3, False), (1, 64, 128, 40, 3, False), (2, 16, 32, 33, 1, False),
                                                  (1, 32, 64, 12, 3, True)])
def test_conv2d_fp16_operands_vs_rounded_reference(n, i, o, h, k, transposed):
    """fp16 MFMA mode (BASELINE config 5 / the reference's use_fp16 SR blocks): every pass equals the fp32-accumulated
    convolution of fp16-ROUNDED operands (products of two fp16 values are exact in fp32)."""
    import torch.nn.functional as F
    from spi_amd.torch_utils.ops import conv2d_mfma
    gen = torch.Generator().manual_seed(n * 100 + i)
    x = torch.randn(n, i, h, h, generator=gen)
    w = torch.randn(n, o, i, k, k, generator=gen) * 0.2
    r = lambda t: t.half().double()
    pad = 0 if transposed else k // 2
    outs, dxs, dws = [], [], []
    for b in range(n):
        xb, wb = r(x[b:b + 1]), r(w[b])
        if transposed:
            yb = F.conv_transpose2d(xb, wb.transpose(0, 1), stride=2)                       # out[o,2y+ky,2x+kx] += x[i,y,x] w[o,i,ky,kx]
        else:
            yb = F.conv2d(xb, wb.flip([2, 3]), padding=pad)                                  # flip=True: true convolution
        outs.append(yb)
    ref = torch.cat(outs)
    dy = torch.randn(ref.shape, generator=gen)
    for b in range(n):
        xb, wb, db = r(x[b:b + 1]), r(w[b]), r(dy[b:b + 1])
        if transposed:
            xx, ww = x[b:b + 1].double().requires_grad_(True), w[b].double().requires_grad_(True)
            (gx,) = torch.autograd.grad(F.conv_transpose2d(xx, wb.transpose(0, 1), stride=2), xx, db)
            (gw,) = torch.autograd.grad(F.conv_transpose2d(xb, ww.transpose(0, 1), stride=2), ww, db)
        else:
            xx, ww = x[b:b + 1].double().requires_grad_(True), w[b].double().requires_grad_(True)
            (gx,) = torch.autograd.grad(F.conv2d(xx, wb.flip([2, 3]), padding=pad), xx, db)
            (gw,) = torch.autograd.grad(F.conv2d(xb, ww.flip([2, 3]), padding=pad), ww, db)
        dxs.append(gx); dws.append(gw)
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    y = conv2d_mfma.conv2d(xd, wd, padding=pad, transposed=transposed, flip=not transposed, fp16=True)
    assert_close(y, ref.float(), 2e-5, 'fp16-operand conv fwd')
    gx, gw = torch.autograd.grad(y, [xd, wd], dy.to(DEV))
    assert_close(gx, torch.cat(dxs).float(), 2e-5, 'fp16-operand conv dgrad')
    assert_close(gw, torch.stack(dws).float(), 2e-5, 'fp16-operand conv wgrad')
    # and it differs from the fp32 path by about fp16 rounding, not more
    y32 = conv2d_mfma.conv2d(xd, wd, padding=pad, transposed=transposed, flip=not transposed)
    e = rel_err(y, y32)
    assert 1e-5 < e < 5e-3, e


@pytest.mark.parametrize('n,i,o,h,k,transposed', [(2, 32, 48, 24, 3, False), (1, 64, 128, 40, 3, False), (2, 16, 3, 33, 1, False), (1, 32, 64, 12, 3, True),
                                                  (1, 128, 128, 128, 3, False), (2, 32, 256, 125, 3, False), (2, 128, 256, 128, 3, False), (2, 128, 32, 128, 3, True),
                                                  (2, 32, 128, 120, 3, True)])
# (the last four: hconv.hip forward (+ dgrad); the transposed ones: the stride-2 data-gradient kernel / the transposed forward kernel)
def test_conv2d_fp16_tensors_vs_rounded_reference(n, i, o, h, k, transposed):
    """fp16 ACTIVATION TENSORS (round 5, spi_conv_desc.act_dtype; the reference's use_fp16 blocks, networks_stylegan2.py:421-436): x, y, dy, dx are
    half tensors in HBM, weights / weight gradients fp32, fp32 accumulation.  Every pass equals the fp64 convolution of the fp16-rounded
    operands ROUNDED ONCE to fp16 (outputs; half an ulp + the fp32 accumulation error) -- and the weight gradient, an fp32 tensor, to 2e-5.
    Includes the 3-channel torgb shape, whose data gradient pads the output channels to a 16-channel slab."""
    import torch.nn.functional as F
    from spi_amd.torch_utils.ops import conv2d_mfma
    gen = torch.Generator().manual_seed(n * 100 + i + 7)
    x = torch.randn(n, i, h, h, generator=gen).half()
    w = torch.randn(n, o, i, k, k, generator=gen) * 0.2
    pad = 0 if transposed else k // 2
    r = lambda t: t.half().double()

    def conv(xb, wb):
        return F.conv_transpose2d(xb, wb.transpose(0, 1), stride=2) if transposed else F.conv2d(xb, wb.flip([2, 3]), padding=pad)
    ref = torch.cat([conv(x[b:b + 1].double(), r(w[b])) for b in range(n)])
    dy = torch.randn(ref.shape, generator=gen).half()
    dxs, dws = [], []
    for b in range(n):
        xx, ww = x[b:b + 1].double().requires_grad_(True), w[b].double().requires_grad_(True)
        (gx,) = torch.autograd.grad(conv(xx, r(w[b])), xx, dy[b:b + 1].double())
        (gw,) = torch.autograd.grad(conv(x[b:b + 1].double(), ww), ww, dy[b:b + 1].double())
        dxs.append(gx); dws.append(gw)
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    y = conv2d_mfma.conv2d(xd, wd, padding=pad, transposed=transposed, flip=not transposed, fp16=True)
    assert y.dtype == torch.float16 and xd.dtype == torch.float16

    def ulp_close(a, b64, what):
        # a: half tensor from the kernel; b64: exact value.  |a - b| <= half an fp16 ulp of b (+ fp32 accumulation slack), element by element
        a, b64 = a.detach().double().cpu(), b64.double()
        ulp = torch.maximum(torch.ldexp(torch.ones_like(b64), torch.floor(torch.log2(b64.abs().clamp_min(6.1e-5))).to(torch.int32) - 10), torch.full_like(b64, 2.0 ** -24))
        bad = ((a - b64).abs() > 0.5 * ulp + 2e-5 * b64.abs().max()).float().mean().item()
        assert bad == 0.0, f'{what}: {bad:.2e} of the elements are further than half an fp16 ulp from the exact value'
    ulp_close(y, ref, 'fp16-tensor conv fwd')
    gx, gw = torch.autograd.grad(y, [xd, wd], dy.to(DEV))
    assert gx.dtype == torch.float16 and gw.dtype == torch.float32
    ulp_close(gx, torch.cat(dxs), 'fp16-tensor conv dgrad')
    assert_close(gw, torch.stack(dws).float(), 2e-5, 'fp16-tensor conv wgrad')
    # the fp32-tensor / fp16-operand mode of rounds 3-4 on the same values: same numbers before the final rounding
    y32 = conv2d_mfma.conv2d(x.float().to(DEV), wd, padding=pad, transposed=transposed, flip=not transposed, fp16=True)
    assert y32.dtype == torch.float32
    ulp_close(y, y32.detach().cpu(), 'fp16-tensor vs fp32-tensor fp16-operand conv')


@pytest.mark.parametrize('n,i,o,h,wd,shared', [(2, 32, 256, 120, 136, False), (1, 128, 128, 256, 256, True), (2, 128, 256, 128, 128, True), (1, 16, 128, 250, 270, True)])
def test_direct_fp16_conv_vs_implicit_gemm_epilogue_and_skipping(n, i, o, h, wd, shared):
    """hconv.hip (fp16 activation tensors, 3x3 / stride 1, output channels in blocks of 128): the plan names it, its results equal the implicit
    GEMM's (same fp16 products, fp32 sums in another order -> at most one fp16 ulp on a few elements) with the fused epilogue (noise, bias,
    lrelu, gain, clamp), ragged tiles, per-sample and shared weights; needed-output maps and sparse gradients skip tiles without changing
    the flagged / non-zero results."""
    import ctypes
    from spi_amd import hip
    from spi_amd.configs import global_config
    from spi_amd.torch_utils.ops import conv2d_mfma
    gen = torch.Generator().manual_seed(n * 1000 + i + h)
    x = torch.randn(n, i, h, wd, generator=gen).half().to(DEV).requires_grad_(True)
    w = (torch.randn(*(() if shared else (n,)), o, i, 3, 3, generator=gen) / (i * 9) ** 0.5).to(DEV).requires_grad_(True)
    bias = torch.randn(o, generator=gen).to(DEV)
    noise = torch.randn(h, wd, generator=gen).to(DEV)
    strength = torch.tensor(0.3, device=DEV)
    kw = dict(bias=bias, noise=noise, noise_strength=strength, padding=1, flip=True, act='lrelu', gain=1.2, clamp=2.0, fp16=True, sparse_grad=True)
    d = conv2d_mfma._desc(n, i, o, h, wd, 3, 1, False, True, 0 if shared else o * i * 9, tap_major=1, f16=True, half=True)
    ws = conv2d_mfma._workspace(d, 0, x.device)
    plan = (ctypes.c_int32 * 8)()
    hip.call('spi_conv2d_plan', ctypes.byref(d), 0, plan)
    assert ws is not None and plan[0] == 2 and plan[5] == 512, list(plan)

    def ulps(a, b):
        a, b = a.detach().float(), b.detach().float()
        ulp = torch.ldexp(torch.ones_like(b), torch.floor(torch.log2(b.abs().clamp_min(6.1e-5))).to(torch.int32) - 10)
        e = ((a - b).abs() - 2e-5 * b.abs().max()).clamp_min(0) / ulp        # (fp32 sums in another order: absolute slack for results that cancel to ~0)
        return float(e.max()), float((e > 0).float().mean())

    def run(direct, dy, sparse=False, needed=None):
        old = global_config.conv_direct_fp16
        global_config.conv_direct_fp16 = direct
        try:
            with conv2d_mfma.needed_output(needed):
                y = conv2d_mfma.conv2d(x, w, **kw)
            with conv2d_mfma.sparse_gradients(sparse):
                gx, gw = torch.autograd.grad(y, [x, w], dy)
        finally:
            global_config.conv_direct_fp16 = old
        return y.detach(), gx, gw
    dy = torch.randn(n, o, h, wd, generator=gen).half().to(DEV)
    ya, gxa, gwa = run(True, dy)
    yb, gxb, gwb = run(False, dy)
    assert ya.dtype == torch.float16 and gxa.dtype == torch.float16
    for a, b, what in ((ya, yb, 'fwd'), (gxa, gxb, 'dgrad')):
        mx, frac = ulps(a, b)
        assert mx <= 1.0 and frac < 2e-2, (what, mx, frac)
    assert_close(gwa, gwb, 1e-5, 'wgrad after the direct forward')
    # sparse gradient operand: tiles whose receptive field holds no flagged segment are zeros, everything else unchanged
    if True:
        old_min = conv2d_mfma.SPARSE_MIN_PIXELS
        conv2d_mfma.SPARSE_MIN_PIXELS = 1
        try:
            for kind in ('box', 'blobs', 'pixel', 'empty'):
                dym = _masked_gradient((n, o, h, wd), gen, kind).half().to(DEV)
                _, gd, _ = run(True, dym)
                _, gs, _ = run(True, dym, sparse=True)
                assert torch.equal(gs, gd), kind
                if kind == 'empty':
                    assert float(gs.float().abs().max()) == 0
        finally:
            conv2d_mfma.SPARSE_MIN_PIXELS = old_min
    # needed-output map: flagged pixels bit-identical, an empty map gives zeros
    for kind in ('box', 'blobs', 'pixel', 'empty'):
        m = (_masked_gradient((n, 1, h, wd), gen, kind) != 0).to(DEV)
        flags = conv2d_mfma.seg_flags(m.float())
        yn, _, _ = run(True, dy, needed={(h, wd): flags})
        assert torch.equal(yn * m, ya * m), kind
        if kind == 'empty':
            assert float(yn.float().abs().max()) == 0
        if kind == 'box':
            assert float((yn == 0).float().mean()) > 0.3


@pytest.mark.parametrize('n,i,o,h,wd,shared', [(2, 128, 48, 128, 128, False), (1, 256, 32, 120, 160, True)])
def test_direct_fp16_stride2_dgrad_of_transposed_conv(n, i, o, h, wd, shared):
    """hconv_s2_kernel: the data gradient of a stride-2 transposed 3x3 conv on fp16 tensors (a stride-2 conv of the gradient).  The plan names it;
    its result equals the implicit GEMM's up to one fp16 ulp on a few elements (same products, another summation order), also on ragged tiles;
    a sparse gradient (dy_seg_flags) skips tiles without changing anything."""
    import ctypes
    from spi_amd import hip
    from spi_amd.configs import global_config
    from spi_amd.torch_utils.ops import conv2d_mfma
    gen = torch.Generator().manual_seed(n * 100 + i)
    x = torch.randn(n, i, h, wd, generator=gen).half().to(DEV).requires_grad_(True)
    w = (torch.randn(*(() if shared else (n,)), o, i, 3, 3, generator=gen) / (i * 9) ** 0.5).to(DEV)
    d = conv2d_mfma._desc(n, i, o, h, wd, 3, 0, True, False, 0 if shared else o * i * 9, tap_major=1, f16=True, half=True)
    ws = conv2d_mfma._workspace(d, 1, x.device)
    plan = (ctypes.c_int32 * 8)()
    hip.call('spi_conv2d_plan', ctypes.byref(d), 1, plan)
    assert ws is not None and plan[0] == 2 and plan[2] == 256, list(plan)
    assert conv2d_mfma._workspace(d, 0, x.device) is None                           # (the forward keeps the implicit GEMM)

    def run(direct, dy, sparse=False):
        old = global_config.conv_direct_fp16
        global_config.conv_direct_fp16 = direct
        try:
            y = conv2d_mfma.conv2d(x, w, transposed=True, fp16=True, sparse_grad=True)
            with conv2d_mfma.sparse_gradients(sparse):
                (gx,) = torch.autograd.grad(y, [x], dy)
        finally:
            global_config.conv_direct_fp16 = old
        return gx
    oh, ow = 2 * h + 1, 2 * wd + 1
    dy = torch.randn(n, o, oh, ow, generator=gen).half().to(DEV)
    a, b = run(True, dy).float(), run(False, dy).float()
    ulp = torch.ldexp(torch.ones_like(b), torch.floor(torch.log2(b.abs().clamp_min(6.1e-5))).to(torch.int32) - 10)
    e = ((a - b).abs() - 2e-5 * b.abs().max()).clamp_min(0) / ulp
    assert float(e.max()) <= 1.0 and float((e > 0).float().mean()) < 2e-2, (float(e.max()), float((e > 0).float().mean()))
    old_min = conv2d_mfma.SPARSE_MIN_PIXELS
    conv2d_mfma.SPARSE_MIN_PIXELS = 1
    try:
        for kind in ('box', 'blobs', 'pixel', 'empty'):
            dym = _masked_gradient((n, o, oh, ow), gen, kind).half().to(DEV)
            gd, gs = run(True, dym), run(True, dym, sparse=True)
            assert torch.equal(gs, gd), kind
            if kind == 'empty':
                assert float(gs.float().abs().max()) == 0
    finally:
        conv2d_mfma.SPARSE_MIN_PIXELS = old_min


@pytest.mark.parametrize('n,i,o,h,wd,shared', [(2, 32, 128, 120, 136, False), (1, 64, 256, 128, 128, True)])
def test_direct_fp16_transposed_forward(n, i, o, h, wd, shared):
    """hconv_t2_kernel: the forward of a stride-2 transposed 3x3 conv on fp16 tensors (both x-parity classes in one block, whole output rows, 16-byte
    stores at 2-byte alignment).  The plan names it; equal to the implicit GEMM up to one fp16 ulp on a few elements, ragged class grids included
    (2H+1 is odd); needed-output maps skip tiles and leave the flagged pixels bit-identical."""
    import ctypes
    from spi_amd import hip
    from spi_amd.configs import global_config
    from spi_amd.torch_utils.ops import conv2d_mfma
    gen = torch.Generator().manual_seed(n * 10 + i)
    x = torch.randn(n, i, h, wd, generator=gen).half().to(DEV)
    w = (torch.randn(*(() if shared else (n,)), o, i, 3, 3, generator=gen) / (i * 9) ** 0.5).to(DEV)
    d = conv2d_mfma._desc(n, i, o, h, wd, 3, 0, True, False, 0 if shared else o * i * 9, tap_major=1, f16=True, half=True)
    ws = conv2d_mfma._workspace(d, 0, x.device)
    plan = (ctypes.c_int32 * 8)()
    hip.call('spi_conv2d_plan', ctypes.byref(d), 0, plan)
    assert ws is not None and plan[0] == 2, list(plan)

    def run(direct, needed=None):
        old = global_config.conv_direct_fp16
        global_config.conv_direct_fp16 = direct
        try:
            with conv2d_mfma.needed_output(needed):
                return conv2d_mfma.conv2d(x, w, transposed=True, fp16=True, sparse_grad=True)
        finally:
            global_config.conv_direct_fp16 = old
    ya, yb = run(True), run(False)
    assert ya.dtype == torch.float16 and tuple(ya.shape) == (n, o, 2 * h + 1, 2 * wd + 1)
    a, b = ya.float(), yb.float()
    ulp = torch.ldexp(torch.ones_like(b), torch.floor(torch.log2(b.abs().clamp_min(6.1e-5))).to(torch.int32) - 10)
    e = ((a - b).abs() - 2e-5 * b.abs().max()).clamp_min(0) / ulp
    assert float(e.max()) <= 1.0 and float((e > 0).float().mean()) < 2e-2, (float(e.max()), float((e > 0).float().mean()))
    oh, ow = 2 * h + 1, 2 * wd + 1
    for kind in ('box', 'blobs', 'pixel', 'empty'):
        m = (_masked_gradient((n, 1, oh, ow), gen, kind) != 0).to(DEV)
        yn = run(True, needed={(oh, ow): conv2d_mfma.seg_flags(m.float())})
        assert torch.equal(yn * m, ya * m), kind
        if kind == 'empty':
            assert float(yn.float().abs().max()) == 0
        if kind == 'box':
            assert float((yn == 0).float().mean()) > 0.3


def test_direct_fp16_wgrad_partial_sums_and_atomics_agree_through_the_c_abi():
    """spi_conv2d_wgrad with fp16 activation tensors: a workspace of spi_conv2d_workspace_bytes(d, 2) bytes selects the partial-sum form of hwgrad_kernel
    (dw is OVERWRITTEN -- garbage in dw does not matter, two runs are bit-identical), a smaller non-null workspace the atomic form (adds into the
    zeroed dw); both equal the fp64 weight gradient of the fp16 operands; dy_seg_flags sends the call back to the implicit GEMM's slab list."""
    import ctypes
    import torch.nn.functional as F
    from spi_amd import hip
    from spi_amd.torch_utils.ops import conv2d_mfma as cm
    gen = torch.Generator().manual_seed(3)
    n, i, o, h, wd = 2, 64, 128, 64, 96
    x = torch.randn(n, i, h, wd, generator=gen).half()
    dy = (torch.randn(n, o, h, wd, generator=gen) * 0.1).half()
    w64 = torch.zeros(o, i, 3, 3, dtype=torch.float64, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(x.double(), w64, padding=1), w64, dy.double())
    ref = ref.permute(0, 2, 3, 1).float()                                         # tap-major [O, k, k, I]
    xd, dyd = x.to(DEV), dy.to(DEV)

    def run(ws_bytes, poison):
        d = cm._desc(n, i, o, h, wd, 3, 1, False, False, 0, tap_major=1, f16=True, half=True)
        need = hip.lib().spi_conv2d_workspace_bytes(ctypes.byref(d), 2)
        assert need > 16
        ws = torch.empty(need if ws_bytes is None else ws_bytes, device=DEV, dtype=torch.uint8)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
        plan = (ctypes.c_int32 * 8)()
        hip.call('spi_conv2d_plan', ctypes.byref(d), 2, plan)
        assert plan[0] == 2, list(plan)
        dw = torch.full((o, 3, 3, i), float(poison), device=DEV)
        if ws_bytes is not None:
            dw.zero_(); d.dw_zeroed = 1
        hip.call('spi_conv2d_wgrad', ctypes.byref(d), hip.ptr(xd), hip.ptr(dyd), hip.ptr(dw), hip.stream())
        return dw.cpu()
    a1, a2 = run(None, 7.0), run(None, float('nan'))
    assert torch.equal(a1, a2)                                                    # partial sums: deterministic, dw overwritten
    assert_close(a1, ref, 2e-5, 'direct fp16 wgrad (partial sums)')
    assert_close(run(16, 0.0), ref, 2e-5, 'direct fp16 wgrad (atomics)')
    # a masked gradient keeps the implicit GEMM (plan path 0)
    d = cm._desc(n, i, o, h, wd, 3, 1, False, False, 0, tap_major=1, f16=True, half=True, dy_flags=cm.seg_flags(dyd))
    assert hip.lib().spi_conv2d_workspace_bytes(ctypes.byref(d), 2) == 0


@pytest.mark.parametrize('up,demod,shared,n', [(1, True, False, 2), (1, True, True, 3), (2, True, False, 1), (2, True, True, 2), (1, False, False, 2)])
def test_frozen_weight_modconv_style_gradient(up, demod, shared, n):
    """Stage-1 path (weights frozen): d styles from  <x,dx>/s - s g^2 sum_o d_o^2 <dz_o,z_o> sum_t W^2  equals the autograd
    gradient of the ordinary path (weight-gradient GEMM -> modulate backward); dx / d noise unchanged."""
    from spi_amd.training.networks_stylegan2 import modulated_conv2d
    from spi_amd.torch_utils.ops import upfirdn2d
    gen = torch.Generator().manual_seed(up * 7 + n)
    i, o, h, k = 32, 48, 20, (3 if demod else 1)
    x = torch.randn(n, i, h, h, generator=gen).to(DEV).requires_grad_(True)
    st = (torch.randn(1 if shared else n, i, generator=gen) * 0.5 + 1).to(DEV).requires_grad_(True)
    w = torch.randn(o, i, k, k, generator=gen).to(DEV)
    oh = h * up
    noise = torch.randn(oh, oh, generator=gen).to(DEV).requires_grad_(True) if demod else None
    strength = torch.tensor(0.4, device=DEV) if demod else None
    bias = torch.randn(o, generator=gen).to(DEV)
    f = upfirdn2d.setup_filter([1, 3, 3, 1]).to(DEV)
    kw = dict(noise=noise, noise_strength=strength, up=up, padding=k // 2, resample_filter=f, demodulate=demod, flip_weight=(up == 1),
              bias=bias, act=('lrelu' if demod else 'linear'), gain=(1.3 if demod else 1), clamp=(2.5 if demod else 256), style_gain=(1.0 if demod else 0.2))
    res = []
    for frozen in (False, True):
        wt = w.clone().requires_grad_(not frozen)
        y = modulated_conv2d(x=x, weight=wt, styles=st, **kw)
        dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(99)).to(DEV)
        wrt = [x, st] + ([noise] if noise is not None else [])
        res.append((y, torch.autograd.grad(y, wrt, dy)))
    (ya, ga), (yb, gb) = res
    assert_close(yb, ya, 1e-6, 'frozen-weight fwd')
    for a, b, nm in zip(gb, ga, ('dx', 'dstyles', 'dnoise')):
        assert_close(a, b, 2e-4, 'frozen-weight ' + nm)


def _masked_gradient(shape, gen, kind):
    """dy with exactly-zero regions like the masked pseudo-view losses leave: blobs / a box / everything / nothing."""
    n, o, h, w = shape
    dy = torch.randn(shape, generator=gen)
    m = torch.zeros(n, 1, h, w)
    if kind == 'box':
        m[:, :, h // 3: h // 3 + max(h // 5, 2), w // 4: w // 4 + max(w // 3, 2)] = 1
    elif kind == 'blobs':
        m = (F.avg_pool2d(torch.rand(n, 1, h, w, generator=gen), 9, 1, 4) > 0.56).float()
    elif kind == 'pixel':
        m[0, 0, h - 1, w - 1] = 1
        m[-1, 0, 0, 0] = 1
    elif kind == 'dense':
        m[:] = 1
    return dy * m                                                                     # kind == 'empty': all zero


def test_seg_flags_vs_torch():
    from spi_amd.torch_utils.ops import conv2d_mfma
    gen = torch.Generator().manual_seed(5)
    for shape in [(2, 5, 37, 41), (1, 3, 128, 128), (3, 17, 33, 64)]:
        x = _masked_gradient(shape, gen, 'blobs')
        x[0, 0, 0, 3] = -0.0                                                          # a negative zero is still zero
        fl = conv2d_mfma.seg_flags(x.to(DEV)).cpu()
        hw = shape[2] * shape[3]
        nz = (x != 0).any(1).reshape(shape[0], hw).float()
        nz = F.pad(nz, (0, (-hw) % 16)).reshape(shape[0], -1, 16).amax(2).int()
        assert torch.equal(fl, nz)


@pytest.mark.parametrize('kind', ['box', 'blobs', 'pixel', 'dense', 'empty'])
@pytest.mark.parametrize('n,i,o,h,k,transposed,fp16', [(2, 32, 48, 128, 3, False, False), (1, 128, 3, 130, 1, False, False),
                                                      (2, 32, 128, 64, 3, True, False), (1, 128, 128, 128, 3, False, True),
                                                      (1, 16, 16, 131, 3, False, False), (1, 64, 64, 256, 3, False, False)])   # last: Winograd dgrad
def test_conv_backward_sparse_gradient_equals_dense(kind, n, i, o, h, k, transposed, fp16):
    """`with sparse_gradients()`: dgrad / wgrad skip the all-zero 16-pixel segments of dy -> same gradients as the dense kernels
    (up to the order of the fp32 sums), and exact zeros where the dense result is exactly zero."""
    from spi_amd.torch_utils.ops import conv2d_mfma
    gen = torch.Generator().manual_seed(h + k)
    wd = h + 16 if not transposed else h
    x = torch.randn(n, i, h, wd, generator=gen).to(DEV).requires_grad_(True)
    w = (torch.randn(n, o, i, k, k, generator=gen) / (i * k * k) ** 0.5).to(DEV).requires_grad_(True)
    bias = torch.randn(o, generator=gen).to(DEV).requires_grad_(True) if not transposed else None
    kw = dict(padding=(0 if transposed else k // 2), transposed=transposed, flip=not transposed, fp16=fp16)
    if not transposed:
        kw.update(bias=bias, act='lrelu', clamp=256)
    y = conv2d_mfma.conv2d(x, w, sparse_grad=True, **kw)
    dy = _masked_gradient(tuple(y.shape), gen, kind).to(DEV)
    wrt = [x, w] + ([bias] if bias is not None else [])
    dense = torch.autograd.grad(y, wrt, dy, retain_graph=True)
    with conv2d_mfma.sparse_gradients():
        sparse = torch.autograd.grad(y, wrt, dy)
    assert_close(sparse[0], dense[0], 1e-6 if not fp16 else 1e-5, 'sparse dgrad')
    assert torch.equal(sparse[0] == 0, dense[0] == 0)
    assert_close(sparse[1], dense[1], 2e-6 if not fp16 else 1e-5, 'sparse wgrad')
    if kind == 'empty':
        assert float(sparse[1].abs().max()) == 0
    if bias is not None:
        assert_close(sparse[2], dense[2], 1e-6, 'bias gradient')


@pytest.mark.parametrize('n,i,o,h,k,transposed', [(2, 32, 48, 128, 3, False), (1, 128, 3, 130, 1, False), (2, 32, 128, 64, 3, True), (1, 16, 16, 131, 3, False),
                                                  (1, 32, 64, 256, 3, False)])                                    # last: Winograd forward
def test_conv_forward_needed_output_region(n, i, o, h, k, transposed):
    """`with needed_output({(OH, OW): flags})`: output tiles without a flagged segment may be skipped (zeros); every flagged pixel is
    bit-identical to the dense forward, and the backward still works on the region."""
    from spi_amd.torch_utils.ops import conv2d_mfma
    gen = torch.Generator().manual_seed(h + k + 1)
    wd = h + 16 if not transposed else h
    x = torch.randn(n, i, h, wd, generator=gen).to(DEV).requires_grad_(True)
    w = (torch.randn(o, i, k, k, generator=gen) / (i * k * k) ** 0.5).to(DEV).requires_grad_(True)       # shared weights (one w, several views)
    kw = dict(padding=(0 if transposed else k // 2), transposed=transposed, flip=not transposed, sparse_grad=True)
    dense = conv2d_mfma.conv2d(x, w, **kw)
    oh, ow = dense.shape[2:]
    for kind in ('box', 'blobs', 'pixel', 'empty', 'dense'):
        m = (_masked_gradient((n, 1, oh, ow), gen, kind) != 0).float().to(DEV)
        flags = conv2d_mfma.seg_flags(m)
        with conv2d_mfma.needed_output({(oh, ow): flags}):
            y = conv2d_mfma.conv2d(x, w, **kw)
        assert torch.equal(y * m, dense * m), kind
        big = k == 3 and not transposed and h in (128, 256)             # small problems run split-K: dense, flags ignored
        if kind == 'empty' and big:
            assert float(y.detach().abs().max()) == 0
        if kind == 'box':
            if big:
                assert float((y.detach() == 0).float().mean()) > 0.3     # something was actually skipped (small problems run split-K, dense)
            gx, gw = torch.autograd.grad((y * m).square().sum(), [x, w])
            hx, hw = torch.autograd.grad((dense * m).square().sum(), [x, w], retain_graph=True)
            assert_close(gx, hx, 1e-6, 'dgrad through a region forward')
            assert_close(gw, hw, 2e-6, 'wgrad through a region forward')
    with conv2d_mfma.needed_output({(oh + 1, ow): flags}):              # other resolutions are untouched
        assert torch.equal(conv2d_mfma.conv2d(x, w, **kw), dense)


def test_needed_output_with_channel_split_sized_winograd_layer():
    """a 512-channel 64^2 layer at N = 1 is the one Winograd problem that runs channel-split (128..255 blocks); its separate epilogue
    pass wrote act(bias) into the tiles the kernel had skipped.  With an output map the split is not taken: skipped tiles hold exact zeros,
    needed pixels equal the dense forward bit for bit, with the fused bias + lrelu epilogue on."""
    from spi_amd.torch_utils.ops import conv2d_mfma
    gen = torch.Generator().manual_seed(77)
    x = torch.randn(1, 512, 64, 64, generator=gen).to(DEV)
    w = (torch.randn(512, 512, 3, 3, generator=gen) / (512 * 9) ** 0.5).to(DEV)
    b = torch.randn(512, generator=gen).to(DEV)
    kw = dict(padding=1, flip=True, bias=b, act='lrelu', sparse_grad=True)
    dense = conv2d_mfma.conv2d(x, w, **kw)
    m = torch.zeros(1, 1, 64, 64, device=DEV)
    m[:, :, 8:24, 16:48] = 1
    with conv2d_mfma.needed_output({(64, 64): conv2d_mfma.seg_flags(m)}):
        y = conv2d_mfma.conv2d(x, w, **kw)
    # (the dense forward of this layer IS channel-split -- two partial sums meeting through atomics -- so the two runs differ by fp32 round-off)
    assert_close(y * m, dense * m, 1e-6, 'needed pixels of the unsplit region forward vs the channel-split dense forward')
    assert float(y[:, :, 40:, :].abs().max()) == 0              # rows far from the flagged box: exact zeros, not act(bias)


WINO_CASES = [  # N, I, O, H, W, flip, per_sample, epilogue
    (1, 16, 64, 256, 256, True, True, True),        # interior fast path (even sizes, whole channel chunks) with the fused epilogue
    (2, 24, 96, 250, 246, True, True, False),       # ragged edges, 96 channels = one and a half chunks, per-sample weights
    (1, 64, 48, 272, 301, False, False, True),      # odd row length (scalar stores), shared weights, 48 of 64 channel rows
    (4, 8, 64, 128, 128, True, False, False),       # batch sharing one weight set
    # 128 .. 255 blocks: the channel reduction is split in two ranges that meet through atomics, the epilogue runs as its own kernel (round 3)
    (2, 256, 256, 64, 64, True, True, True),
    (2, 256, 256, 60, 52, False, False, False),     # ... ragged tiles, shared weights
]


@pytest.mark.parametrize('case', WINO_CASES)
def test_conv_winograd_vs_oracle_and_implicit_gemm(case):
    """The Winograd F(2x2, 3x3) path (winograd.hip) of the forward and data-gradient passes against the CPU oracle, and against the
    implicit-GEMM kernels on the same inputs.  fp32 throughout: tolerance 1e-5 of the tensor's max (direct kernels: 3e-6 .. 1e-5); the
    two HIP paths agree to 6e-6."""
    import ctypes
    from spi_amd import hip
    from spi_amd.configs import global_config
    from spi_amd.torch_utils.ops import conv2d_mfma
    N, I, O, H, W, flip, per, epi = case
    torch.set_num_threads(min(__import__('os').cpu_count() or 1, 32))
    gen = torch.Generator().manual_seed(N * 100 + O)
    x = torch.randn(N, I, H, W, generator=gen, requires_grad=True)
    w = (torch.randn(*((N,) if per else ()), O, I, 3, 3, generator=gen) / (I * 9) ** 0.5).requires_grad_(True)
    b = torch.randn(O, generator=gen) if epi else None
    nz = torch.randn(H, W, generator=gen) if epi else None
    st = torch.tensor(0.3) if epi else None
    ref = _ref_conv(x, w, 1, False, flip)
    if epi:
        ref = osg.bias_act(ref + nz * st, b, act='lrelu', gain=1.3, clamp=2.0)
    dy = torch.randn(ref.shape, generator=gen)
    gx, = torch.autograd.grad(ref, [x], dy)
    kw = dict(padding=1, flip=flip)
    if epi:
        kw.update(bias=b.to(DEV), noise=nz.to(DEV), noise_strength=st.to(DEV), act='lrelu', gain=1.3, clamp=2.0)
    # the shape must actually take the Winograd path (both passes)
    d = conv2d_mfma._desc(N, I, O, H, W, 3, 1, False, flip, O * I * 9 if per else 0, tap_major=1)
    assert hip.lib().spi_conv2d_workspace_bytes(ctypes.byref(d), 0) == (N if per else 1) * 16 * I * ((O + 63) // 64 * 64) * 4
    assert (hip.lib().spi_conv2d_workspace_bytes(ctypes.byref(d), 1) > 0) == (I >= 48)
    outs = {}
    for wino in (True, False):
        old = global_config.conv_winograd
        global_config.conv_winograd = wino
        try:
            xg, wg = x.detach().to(DEV).requires_grad_(True), w.detach().to(DEV).requires_grad_(True)
            y = conv2d_mfma.conv2d(xg, wg, **kw)
            hx, hw = torch.autograd.grad(y, [xg, wg], dy.to(DEV))
        finally:
            global_config.conv_winograd = old
        outs[wino] = (y.detach(), hx, hw)
    assert_close(outs[True][0], ref, 1e-5, 'winograd fwd')
    assert_close(outs[True][0], outs[False][0], 6e-6, 'winograd vs implicit GEMM fwd')
    if not epi:
        assert_close(outs[True][1], gx, 1e-5, 'winograd dgrad')
        assert_close(outs[True][1], outs[False][1], 6e-6, 'winograd vs implicit GEMM dgrad')
        assert_close(outs[True][2], outs[False][2], 1e-5, 'wgrad behind either forward')
    else:
        # behind lrelu + clamp a pre-activation within 1e-6 of the kink or of the clamp bound may land on the other side of it in another
        # summation order: its gradient flips between slope values (a few of the 4 M pre-activations), and each flip reaches the 9 x I
        # gradient entries under its taps
        for got, want, what in ((outs[True][1], gx, 'dgrad vs oracle'), (outs[True][1], outs[False][1], 'dgrad vs implicit GEMM')):
            diff = (got.cpu() - want.cpu()).abs() / want.cpu().abs().max()
            assert float((diff > 1e-5).float().mean()) < 5e-3 and float(diff.median()) < 1e-6, what


WINO_F4_CASES = [  # N, I, O, H, W, flip, per_sample, epilogue      (>= 256 blocks of 16 x 32 pixels x 64 channels: the F(4x4, 3x3) kernel)
    (1, 128, 128, 256, 256, True, True, True),      # b256.conv1: two channel blocks, per-sample weights, fused noise / bias / lrelu / clamp epilogue
    (2, 128, 64, 256, 512, False, False, False),    # batch sharing one weight set, un-flipped taps, plain output
    (1, 136, 48, 512, 512, True, False, True),      # 48 of 64 channel rows, 34 slabs of four channels
]


@pytest.mark.parametrize('case', WINO_F4_CASES)
def test_conv_winograd_f4_vs_fp64_reference_and_f2(case):
    """Round 6: the F(4x4, 3x3) Winograd kernel (wino4_conv_kernel) the >= 256^2 layers take, forward and data gradient, against an fp64 convolution of the
    same fp32 inputs, next to F(2x2, 3x3) (`global_config.conv_winograd_f4 = False` -> `spi_conv_wino_f4_set(0)`) and the implicit GEMM on the same inputs.  fp32 operands and accumulation in all
    three; the minimal-filtering transforms of F(4x4) multiply by up to 8 and divide by up to 24, so its rounding error is a few times F(2x2)'s: the bar
    is 6e-5 of the tensor's range (measured 2e-5 .. 4e-5 at 128 channels behind a clamping epilogue, 6e-6 on plain outputs: profiles/r06_wino_f4_precision.txt; the
    other fp32 conv tests hold 1e-5), far inside the north star's 1e-3 on the rendered image, which the full-size parity tests check with this kernel in the path."""
    import ctypes
    from spi_amd import hip
    from spi_amd.configs import global_config
    from spi_amd.torch_utils.ops import conv2d_mfma
    N, I, O, H, W, flip, per, epi = case
    torch.set_num_threads(min(__import__('os').cpu_count() or 1, 32))
    gen = torch.Generator().manual_seed(N * 100 + O + 7)
    x = torch.randn(N, I, H, W, generator=gen)
    w = torch.randn(*((N,) if per else ()), O, I, 3, 3, generator=gen) / (I * 9) ** 0.5
    b = torch.randn(O, generator=gen) if epi else None
    nz = torch.randn(H, W, generator=gen) if epi else None
    st = torch.tensor(0.3) if epi else None
    x64, w64 = x.double().requires_grad_(True), w.double()
    ref = _ref_conv(x64, w64, 1, False, flip)
    if epi:
        ref = osg.bias_act(ref + (nz * st).double(), b.double(), act='lrelu', gain=1.3, clamp=2.0)
    dy = torch.randn(ref.shape, generator=gen)
    gx, = torch.autograd.grad(ref, [x64], dy.double())
    kw = dict(padding=1, flip=flip)
    if epi:
        kw.update(bias=b.to(DEV), noise=nz.to(DEV), noise_strength=st.to(DEV), act='lrelu', gain=1.3, clamp=2.0)
    L = hip.lib()
    ocp = (O + 63) // 64 * 64
    outs = {}
    try:
        for mode in ('f4', 'f2', 'igemm'):
            global_config.conv_winograd_f4 = mode == 'f4'
            conv2d_mfma._sync_wino_f4()
            d = conv2d_mfma._desc(N, I, O, H, W, 3, 1, False, flip, O * I * 9 if per else 0, tap_major=1)
            # the shape must actually take the kernel under test (forward; the data gradient where its reduction is wide enough)
            assert L.spi_conv2d_workspace_bytes(ctypes.byref(d), 0) == (N if per else 1) * (36 if mode == 'f4' else 16) * I * ocp * 4
            old = global_config.conv_winograd
            global_config.conv_winograd = mode != 'igemm'
            conv2d_mfma._frozen_ws.clear()
            try:
                xg, wg = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
                y = conv2d_mfma.conv2d(xg, wg, **kw)
                hx, = torch.autograd.grad(y, [xg], dy.to(DEV))
            finally:
                global_config.conv_winograd = old
            outs[mode] = (y.detach(), hx)
    finally:
        global_config.conv_winograd_f4 = True
        conv2d_mfma._sync_wino_f4()
        conv2d_mfma._frozen_ws.clear()
    e = {m: rel_err(outs[m][0], ref) for m in outs}
    print(f'  F(4x4,3x3) {case}: forward vs fp64  f4 {e["f4"]:.2e}  f2 {e["f2"]:.2e}  implicit GEMM {e["igemm"]:.2e};  f4 vs f2 {rel_err(outs["f4"][0], outs["f2"][0]):.2e}')
    assert_close(outs['f4'][0], ref, 6e-5, 'F(4x4,3x3) forward vs the fp64 convolution')
    assert_close(outs['f2'][0], ref, 1e-5, 'F(2x2,3x3) forward vs the fp64 convolution')
    if not epi:
        eg = {m: rel_err(outs[m][1], gx) for m in outs}
        print(f'      data gradient vs fp64  f4 {eg["f4"]:.2e}  f2 {eg["f2"]:.2e}  implicit GEMM {eg["igemm"]:.2e}')
        assert_close(outs['f4'][1], gx, 6e-5, 'F(4x4,3x3) dgrad vs the fp64 convolution')
        assert_close(outs['f2'][1], gx, 1e-5, 'F(2x2,3x3) dgrad vs the fp64 convolution')
    else:
        # behind lrelu + clamp: kink flips (see test_conv_winograd_vs_oracle_and_implicit_gemm)
        for got, want, what in ((outs['f4'][1], gx, 'dgrad vs fp64'), (outs['f4'][1], outs['f2'][1], 'dgrad vs F(2x2,3x3)')):
            diff = (got.double().cpu() - want.double().cpu()).abs() / want.double().cpu().abs().max()
            assert float((diff > 6e-5).float().mean()) < 5e-3 and float(diff.median()) < 1e-5, what


WINO_WGRAD_CASES = [  # N, I, O, H, W, flip, per_sample, tap_major
    (1, 64, 64, 64, 64, False, True, 1),            # one 64 x 64 channel block, two strips, whole tiles
    (2, 96, 40, 50, 70, True, True, 1),             # ragged: 40 of 64 output rows, 1.5 input blocks, a partial third strip, per-sample weights
    (3, 32, 32, 33, 45, False, False, 1),           # odd height and width (tiles hanging over both edges), batch summed into ONE weight set
    (1, 128, 128, 256, 256, True, True, 0),         # a real layer size (b256.conv1), [O,I,3,3] weight layout
    (4, 64, 128, 128, 128, True, False, 1),         # the pseudo-view branches' shape: 4 images sharing one weight set
]


@pytest.mark.parametrize('case', WINO_WGRAD_CASES)
def test_conv_winograd_wgrad_vs_oracle_and_implicit_gemm(case):
    """The F(3x3, 2x2) weight-gradient kernel (winograd.hip: wino_wgrad_kernel) against the CPU oracle (autograd of F.conv2d) and against the
    implicit-GEMM weight gradient on the same inputs, through the C ABI (spi_conv2d_wgrad with / without the opt-in workspace).  fp32
    throughout; both kernels reduce 10^3..10^5 products per weight with atomics, so the bar is 1e-5 of the tensor's max vs the oracle
    (2e-5 for the 65 536-pixel layer) and the two HIP paths agree as closely."""
    import ctypes
    from spi_amd import hip
    from spi_amd.torch_utils.ops import conv2d_mfma
    N, I, O, H, W, flip, per, tap_major = case
    torch.set_num_threads(min(__import__('os').cpu_count() or 1, 32))
    gen = torch.Generator().manual_seed(N * 1000 + O + H)
    x = torch.randn(N, I, H, W, generator=gen)
    w = (torch.randn(*((N,) if per else ()), O, I, 3, 3, generator=gen) / (I * 9) ** 0.5).requires_grad_(True)
    dy = torch.randn(N, O, H, W, generator=gen)
    gw, = torch.autograd.grad(_ref_conv(x.double(), w.double(), 1, False, flip), [w], dy.double())
    xg, dyg = x.to(DEV), dy.to(DEV)
    got = {}
    for wino in (True, False):
        d = conv2d_mfma._desc(N, I, O, H, W, 3, 1, False, flip, O * I * 9 if per else 0, tap_major=tap_major)
        if wino:
            nbytes = hip.lib().spi_conv2d_workspace_bytes(ctypes.byref(d), 2)
            assert nbytes > 0, 'the shape must take the Winograd weight-gradient path'
            ws = torch.empty(nbytes, device=DEV, dtype=torch.uint8)
            d.workspace, d.workspace_bytes = ws.data_ptr(), nbytes
        dw = torch.full((*((N,) if per else ()), O, 3, 3, I) if tap_major else w.shape, 7.0, device=DEV)       # dirty: the call zeroes it
        hip.call('spi_conv2d_wgrad', ctypes.byref(d), hip.ptr(xg), hip.ptr(dyg), hip.ptr(dw), hip.stream())
        got[wino] = dw.movedim(-1, -3) if tap_major else dw
    tol = 2e-5 if H * W * (1 if per else N) >= 65536 else 1e-5
    assert_close(got[True], gw.float(), tol, 'winograd wgrad vs oracle')
    assert_close(got[False], gw.float(), tol, 'implicit-GEMM wgrad vs oracle')
    assert_close(got[True], got[False], tol, 'winograd vs implicit-GEMM wgrad')
    # masked gradient (the pseudo-view branches): dy is zero outside a blob, the zero-segment map lets the kernel skip the tile rows of a strip
    # that hold nothing -- the pipeline restarts after every gap (here: two blobs per image, one touching the border, plus a lone pixel)
    mask = torch.zeros(N, 1, H, W)
    mask[:, :, H // 5:H // 2, W // 6:W // 2] = 1
    mask[:, :, H - H // 4:, W // 3:] = 1
    mask[:, :, 1, W - 1] = 1
    dym = dy * mask
    gwm, = torch.autograd.grad(_ref_conv(x.double(), w.double(), 1, False, flip), [w], dym.double())
    dymg = dym.to(DEV)
    flags = conv2d_mfma.seg_flags(dymg)
    for wino in (True, False):
        d = conv2d_mfma._desc(N, I, O, H, W, 3, 1, False, flip, O * I * 9 if per else 0, tap_major=tap_major, dy_flags=flags)
        if wino:
            assert hip.lib().spi_conv2d_workspace_bytes(ctypes.byref(d), 2) > 0
            d.workspace, d.workspace_bytes = ws.data_ptr(), nbytes
        dw = torch.full((*((N,) if per else ()), O, 3, 3, I) if tap_major else w.shape, 7.0, device=DEV)
        hip.call('spi_conv2d_wgrad', ctypes.byref(d), hip.ptr(xg), hip.ptr(dymg), hip.ptr(dw), hip.stream())
        assert_close(dw.movedim(-1, -3) if tap_major else dw, gwm.float(), tol, f'masked wgrad vs oracle (winograd={wino})')
    # not offered where it does not apply: 1x1 / transposed convs, rows shorter than a strip
    d = conv2d_mfma._desc(1, 64, 64, 16, 16, 3, 1, False, False, 0, tap_major=1)
    assert hip.lib().spi_conv2d_workspace_bytes(ctypes.byref(d), 2) == 0
    d = conv2d_mfma._desc(1, 64, 64, 64, 64, 3, 0, True, False, 0, tap_major=1)
    assert hip.lib().spi_conv2d_workspace_bytes(ctypes.byref(d), 2) == 0


def test_conv_wgrad_zeroes_a_dirty_buffer_itself():
    """spi_conv2d_wgrad with dw_zeroed = 0 clears dw with the library's own kernel (spi_zero_async; not hipMemsetAsync, whose captured
    form misbehaves for odd byte counts on ROCm 7.0): a dirty, odd-sized (135-float) buffer ends up holding exactly the gradient."""
    import ctypes
    from spi_amd import hip
    from spi_amd.torch_utils.ops import conv2d_mfma
    gen = torch.Generator().manual_seed(12)
    N, I, O, H = 2, 5, 3, 20
    x = torch.randn(N, I, H, H, generator=gen)
    dy = torch.randn(N, O, H, H, generator=gen)
    w = torch.randn(O, I, 3, 3, generator=gen, requires_grad=True)
    ref, = torch.autograd.grad(F.conv2d(x, w, padding=1), [w], dy)
    d = conv2d_mfma._desc(N, I, O, H, H, 3, 1, False, False, 0, tap_major=1, dw_zeroed=0)
    dw = torch.full((O, 3, 3, I), 7.5, device=DEV)                       # tap-major layout, dirty
    xg, dyg = x.to(DEV), dy.to(DEV)
    hip.call('spi_conv2d_wgrad', ctypes.byref(d), hip.ptr(xg), hip.ptr(dyg), hip.ptr(dw), hip.stream())
    assert_close(dw.permute(0, 3, 1, 2), ref, 2e-5, 'wgrad into a dirty buffer')


def test_conv_winograd_random_shapes_vs_implicit_gemm():
    """24 seeded random problems that take the Winograd path (ragged image sizes, 8 .. 128 input channels, output channels that are not a
    multiple of the 64-channel chunk, shared / per-sample weights, flipped taps, fused epilogues): forward and data gradient equal the
    implicit-GEMM kernels' to 1e-5 of the tensor's maximum (median for gradients behind an activation, see the kink-flip note above)."""
    import ctypes
    import random
    from spi_amd import hip
    from spi_amd.configs import global_config
    from spi_amd.torch_utils.ops import conv2d_mfma
    rnd = random.Random(3)
    done = 0
    old = global_config.conv_winograd
    try:
        while done < 24:
            N, I, O = rnd.choice([1, 1, 2, 3]), rnd.choice([8, 16, 24, 40, 64, 72, 128]), rnd.choice([48, 64, 80, 100, 128, 130, 192])
            H, W = rnd.randint(40, 300), rnd.randint(40, 300)
            flip, per, epi = rnd.random() < 0.5, rnd.random() < 0.6, rnd.random() < 0.5
            d = conv2d_mfma._desc(N, I, O, H, W, 3, 1, False, flip, O * I * 9 if per else 0, tap_major=1)
            if hip.lib().spi_conv2d_workspace_bytes(ctypes.byref(d), 0) == 0:
                continue
            done += 1
            g = torch.Generator().manual_seed(done)
            x = torch.randn(N, I, H, W, generator=g).to(DEV).requires_grad_(True)
            w = (torch.randn(*((N,) if per else ()), O, I, 3, 3, generator=g) / (I * 9) ** 0.5).to(DEV).requires_grad_(True)
            kw = dict(padding=1, flip=flip)
            if epi:
                kw.update(bias=torch.randn(O, generator=g).to(DEV), noise=torch.randn(H, W, generator=g).to(DEV), noise_strength=torch.tensor(0.3, device=DEV),
                          act=rnd.choice(['lrelu', 'linear', 'relu']), gain=1.3, clamp=rnd.choice([None, 2.0]))
            dy = torch.randn(N, O, H, W, generator=g).to(DEV)
            outs = []
            for wino in (True, False):
                global_config.conv_winograd = wino
                y = conv2d_mfma.conv2d(x, w, **kw)
                outs.append((y.detach(), torch.autograd.grad(y, [x], dy)[0]))
            what = f'case {done}: {(N, I, O, H, W, flip, per, epi)}'
            assert_close(outs[0][0], outs[1][0], 1e-5, what + ' fwd')
            diff = (outs[0][1] - outs[1][1]).abs() / outs[1][1].abs().max()
            assert float(diff.median() if epi else diff.max()) < 1e-5, what + ' dgrad'
    finally:
        global_config.conv_winograd = old


def test_conv_winograd_is_not_offered_where_it_does_not_apply():
    import ctypes
    from spi_amd import hip
    from spi_amd.torch_utils.ops import conv2d_mfma
    for args, kw in [((1, 64, 64, 256, 256, 1, 0, False, False, 0), {}),                  # 1x1
                     ((1, 64, 64, 128, 128, 3, 0, True, False, 0), {}),                   # stride-2 transposed
                     ((1, 512, 512, 16, 16, 3, 1, False, False, 0), {}),                  # too few blocks: split-K implicit GEMM fills the chip
                     ((1, 12, 64, 256, 256, 3, 1, False, False, 0), {}),                  # no whole 8-channel slabs
                     ((1, 64, 64, 256, 256, 3, 1, False, False, 0), dict(f16=2)),         # 3-product bf16 split requested: its own (faster) kernels
                     ((1, 64, 64, 256, 256, 3, 1, False, False, 0), dict(f16=1))]:        # fp16 operands requested
        d = conv2d_mfma._desc(*args, tap_major=1, **kw)
        assert hip.lib().spi_conv2d_workspace_bytes(ctypes.byref(d), 0) == 0, args
        assert hip.lib().spi_conv2d_workspace_bytes(ctypes.byref(d), 2) == 0, args                 # (< 32 input channels / rows shorter than a strip for the weight gradient)
    d = conv2d_mfma._desc(1, 64, 64, 256, 256, 3, 1, False, False, 0, tap_major=1)
    assert hip.lib().spi_conv2d_workspace_bytes(ctypes.byref(d), 2) == 16                 # the weight-gradient pass (round 3): a nominal opt-in workspace
    assert hip.lib().spi_conv2d_workspace_bytes(ctypes.byref(d), 3) == 0                  # no such pass


SPLIT_CASES = [(1, 64, 128, 40, 3, 1, False, True, True), (2, 32, 48, 24, 3, 0, True, False, True), (1, 128, 128, 96, 3, 1, False, True, True),
               (2, 16, 32, 33, 1, 0, False, False, True), (4, 128, 128, 64, 3, 1, False, True, False), (1, 256, 128, 64, 3, 0, True, False, True),
               (1, 64, 64, 128, 3, 1, False, False, False)]


@pytest.mark.parametrize('prec,tol', [(3, 2e-6), (2, 6e-5)])
@pytest.mark.parametrize('case', SPLIT_CASES)
def test_conv_split_bf16_modes_vs_fp32_reference(case, prec, tol):
    """global_config.conv_precision = 3 / 2: fp32 operands cut into 3 / 2 bf16 pieces while they are staged in LDS, the 6 / 3 significant piece
    products on the bf16 matrix cores with fp32 accumulation.  Against the plain fp32 convolution (CPU): the 3-piece mode holds the SAME
    tolerance as the exact-fp32 MFMA kernels (products carry ~2^-23 relative error), the 2-piece mode ~2^-16 per product.
    All three passes; operands scaled over 12 orders of magnitude (bf16 keeps fp32's exponent range: no underflow on tiny gradients)."""
    from spi_amd.torch_utils.ops import conv2d_mfma
    from spi_amd.configs import global_config
    N, I, O, H, k, pad, tr, flip, per = case
    gen = torch.Generator().manual_seed(hash(case) % 1000 + prec)
    for xs, ws, ds in ((1.0, 1.0, 1.0), (1e-6, 30.0, 1e-7)):
        x = (torch.randn(N, I, H, H + 1, generator=gen) * xs).requires_grad_(True)
        w = (torch.randn(*((N,) if per else ()), O, I, k, k, generator=gen) * ws / (I * k * k) ** 0.5).requires_grad_(True)
        ref = _ref_conv(x, w, pad, tr, flip)
        dy = torch.randn(ref.shape, generator=gen) * ds
        gx, gw = torch.autograd.grad(ref, [x, w], dy)
        xg, wg = x.detach().to(DEV).requires_grad_(True), w.detach().to(DEV).requires_grad_(True)
        old = global_config.conv_precision
        global_config.conv_precision = prec
        try:
            y = conv2d_mfma.conv2d(xg, wg, padding=pad, transposed=tr, flip=flip)
            hx, hw = torch.autograd.grad(y, [xg, wg], dy.to(DEV))
            global_config.conv_precision = 0
            y0 = conv2d_mfma.conv2d(xg, wg, padding=pad, transposed=tr, flip=flip)
        finally:
            global_config.conv_precision = old
        assert_close(y, ref, tol, f'split x{prec} fwd')
        assert_close(hx, gx, tol, f'split x{prec} dgrad')
        assert_close(hw, gw, 10 * tol, f'split x{prec} wgrad')
        if prec == 2 and H >= 40 and not tr:       # (grids that need split-K run the exact fp32 kernels whatever was asked for)
            assert not torch.equal(y, y0), 'the 2-piece mode must really run different arithmetic'


def test_conv_out_zeroed_flag_and_zero_arena():
    """ABI 9: `spi_conv2d_out_accumulates` tells which forward / dgrad launches add into their output (split-K at 16^2, channel-split
    Winograd at 64^2 x 512) and `out_zeroed = 1` makes exactly those skip their fill launch; inside an iteration (`zero_arena.begin`) the
    wrapper hands such launches a view of the iteration's one cleared buffer.  Same values as the self-clearing call."""
    import ctypes
    from spi_amd import hip
    from spi_amd.torch_utils import zero_arena
    from spi_amd.torch_utils.ops import conv2d_mfma
    gen = torch.Generator().manual_seed(21)
    lib = hip.lib()
    for (I, O, H, expect) in ((512, 512, 16, 1), (512, 512, 64, 1), (128, 128, 256, 0)):
        x = torch.randn(1, I, H, H, generator=gen).to(DEV)
        w = (torch.randn(1, O, 3, 3, I, generator=gen) / (3 * I ** 0.5)).to(DEV)        # tap-major, per sample
        d = conv2d_mfma._desc(1, I, O, H, H, 3, 1, False, True, O * I * 9, tap_major=1)
        ws = conv2d_mfma._workspace(d, 0, x.device)                                    # noqa: F841
        assert lib.spi_conv2d_out_accumulates(ctypes.byref(d), 0) == expect, (I, O, H)
        y_ref = torch.full((1, O, H, H), 3.25, device=DEV)                             # dirty: the library clears it itself
        hip.call('spi_conv2d_fwd', ctypes.byref(d), hip.ptr(x), hip.ptr(w), hip.ptr(y_ref), hip.stream())
        d.out_zeroed = 1
        y_dirty = torch.full((1, O, H, H), 3.25, device=DEV)
        hip.call('spi_conv2d_fwd', ctypes.byref(d), hip.ptr(x), hip.ptr(w), hip.ptr(y_dirty), hip.stream())
        if expect:                                                                     # the flag is honoured: the launch ADDED to what was there
            assert_close(y_dirty - 3.25, y_ref, 1e-5, f'out_zeroed on a dirty buffer {I}x{H}')
        else:
            assert torch.equal(y_dirty, y_ref)
        # through the wrapper, inside an iteration: the second iteration of a kind serves the output from the arena
        zero_arena.reset()
        try:
            outs = []
            for it in range(2):
                zero_arena.begin(x.device, key='test')
                xr = x.clone().requires_grad_(True)
                y = conv2d_mfma.conv2d(xr, w, padding=1, flip=True, tap_major=True)
                gx, = torch.autograd.grad(y, [xr], torch.ones_like(y))
                outs.append((y.detach().clone(), gx.clone()))
                if it == 1 and expect:
                    buf = zero_arena._s.buf
                    assert buf is not None and buf.data_ptr() <= y.data_ptr() < buf.data_ptr() + 4 * buf.numel(), 'forward output not from the arena'
            zero_arena.finish()
            assert (zero_arena._s.peaks.get('test', 0) > 0) == bool(expect)
            assert_close(outs[1][0], y_ref, 1e-5, 'arena forward')
            assert_close(outs[1][0], outs[0][0], 1e-5, 'arena vs own fill: forward')
            assert_close(outs[1][1], outs[0][1], 1e-5, 'arena vs own fill: dgrad')
        finally:
            zero_arena.reset()


@pytest.mark.parametrize('n', [1, 2])
def test_multi_modulate_equals_the_per_layer_modulation(n):
    """`multi_modulate` (spi_modulate_multi_fwd / _bwd): the weight modulation of all layers of a network in one launch each way -- the same kernel
    bodies over a job table: modulated weights bit-equal to `modulate_weights` per layer (3x3 demodulated convs and 1x1 ToRGB layers with their
    weight gain), style and weight gradients equal, a layer without an incoming gradient skipped; on stage 1's frozen-weight path (round 6) the same launch hands `_ModConvFrozen` its (w2, dcoef) pairs."""
    from spi_amd.training.networks_stylegan2 import SynthesisLayer, ToRGBLayer, modulate_weights, multi_modulate
    gen = torch.Generator().manual_seed(60 + n)
    layers = [SynthesisLayer(64, 128, 512, 16), SynthesisLayer(128, 128, 512, 16), ToRGBLayer(128, 96, 512), SynthesisLayer(128, 32, 512, 32, up=2),
              ToRGBLayer(32, 3, 512)]
    layers = [m.to(DEV) for m in layers]
    with torch.no_grad():
        for m in layers:
            m.weight.copy_(torch.randn(m.weight.shape, generator=gen))
    styles = [(torch.randn(n, m.weight.shape[1], generator=gen) + 1).to(DEV).requires_grad_(True) for m in layers]
    ref = [modulate_weights(m.weight, s, not isinstance(m, ToRGBLayer), m.weight_gain if isinstance(m, ToRGBLayer) else 1.0) for m, s in zip(layers, styles)]
    got = multi_modulate(layers, styles)
    assert got is not None and len(got) == len(layers)
    for a, b in zip(got, ref):
        assert a.shape == b.shape and torch.equal(a, b)
    gws = [torch.randn(r.shape, generator=gen).to(DEV) for r in ref]
    use = [0, 1, 2, 4]                                           # layer 3 gets no gradient
    params = [m.weight for m in layers]
    g_ref = torch.autograd.grad([ref[i] for i in use], styles + params, [gws[i] for i in use], allow_unused=True)
    g_got = torch.autograd.grad([got[i] for i in use], styles + params, [gws[i] for i in use], allow_unused=True)
    for k, (a, b) in enumerate(zip(g_got, g_ref)):
        assert (a is None) == (b is None), k
        if a is not None:
            assert_close(a, b, 1e-6, f'multi-modulate gradient {k}')
    for m in layers:
        m.requires_grad_(False)
    # frozen weights + styles that need a gradient (stage 1): the layers' weights still come out of ONE launch, as (w2, dcoef) pairs for _ModConvFrozen
    from spi_amd.training.networks_stylegan2 import FrozenMod
    fz = multi_modulate(layers, styles)
    assert fz is not None and all(isinstance(p, FrozenMod) for p in fz)
    for (w2, dc), b, m in zip(fz, ref, layers):
        assert torch.equal(w2, b.detach()) and (dc is None) == isinstance(m, ToRGBLayer) and not w2.requires_grad
    # ... and a layer run with its pair gives the per-layer launch's output and style gradient
    xin = torch.randn(n, 64, 16, 16, generator=gen).to(DEV)
    outs = []
    for wm in (None, fz[0]):
        s0 = styles[0].detach().clone().requires_grad_(True)
        y0 = layers[0](xin, None, noise_mode='none', styles=s0, w_mod=wm)
        gs, = torch.autograd.grad(y0.square().sum(), [s0])
        outs.append((y0.detach(), gs))
    # (the 16^2 layer runs split-K with atomics: two launches of the SAME weights differ in the last bits)
    assert_close(outs[0][0], outs[1][0], 1e-6, 'frozen layer output, weights modulated ahead') and None
    assert_close(outs[0][1], outs[1][1], 1e-5, 'frozen layer style gradient, weights modulated ahead')
    layers[2].requires_grad_(True)
    assert multi_modulate(layers, styles) is None                # a mix of frozen and trainable layers keeps the per-layer paths
    layers[2].requires_grad_(False)
    with torch.no_grad():
        got = multi_modulate(layers, styles)                     # inference: fine
    assert got is not None and torch.equal(got[2], ref[2])


def test_frozen_weight_winograd_transform_is_cached_and_follows_the_weights():
    """Round 5: shared weights that take no gradient (the VGG extractors of the losses) keep their Winograd-transformed weights per (tensor, pass)
    -- `spi_conv_desc.workspace_ready` skips the transform launch of later calls.  Same numbers as the uncached path (weights that DO take a
    gradient), forward and data gradient; an in-place update of the weights is seen (`_version`); the cache holds one entry per pass."""
    from spi_amd.torch_utils.ops import conv2d_mfma as cm
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(1, 64, 256, 256, generator=gen).to(DEV).requires_grad_(True)
    w = (torch.randn(64, 3, 3, 64, generator=gen) * 0.05).to(DEV)                     # tap-major [O, k, k, I], frozen
    dy = torch.randn(1, 64, 256, 256, generator=gen).to(DEV)
    cm._frozen_ws.clear()

    def run(weights):
        y = cm.conv2d(x, weights, padding=1, tap_major=True)
        (gx,) = torch.autograd.grad(y, [x], dy)
        return y, gx
    y_ref, gx_ref = run(w.clone().requires_grad_(True))                              # takes a gradient: never cached
    assert len(cm._frozen_ws) == 0
    y1, gx1 = run(w)
    assert len(cm._frozen_ws) == 2                                                   # forward + dgrad transforms of this tensor
    y2, gx2 = run(w)                                                                 # served from the cache
    assert len(cm._frozen_ws) == 2
    assert torch.equal(y1, y_ref) and torch.equal(gx1, gx_ref) and torch.equal(y2, y_ref) and torch.equal(gx2, gx_ref)
    addrs = sorted(e[1].data_ptr() for e in cm._frozen_ws.values())
    w.mul_(2.0)                                                                      # in-place update: the cached transform is stale
    y3, gx3 = run(w)
    # ... and is redone INTO the same workspace: a graph that baked the address with workspace_ready = 1 keeps reading live memory
    assert sorted(e[1].data_ptr() for e in cm._frozen_ws.values()) == addrs
    assert_close(y3, 2 * y_ref, 1e-6, 'cached transform after an in-place weight update') 
    assert_close(gx3, 2 * gx_ref, 1e-6, 'cached dgrad transform after an in-place weight update')
    y4, _ = run(w)
    assert torch.equal(y4, y3)
    # a TEMPORARY weight tensor (e.g. freshly modulated weights under no_grad) dies; the next temporary of that shape lands on the same address
    # with other values and must not be served the dead tensor's transform (the entry holds a weak reference to the tensor object)
    cm._frozen_ws.clear()
    t = w * 0.5
    addr = t.data_ptr()
    ya, _ = run(t)
    del t
    t = w * 3.0
    assert t.data_ptr() == addr or os.environ.get('SPI_EFENCE') == '1', 'the caching allocator was expected to reuse the block (test premise)'
    yb, _ = run(t)
    assert_close(yb, 6 * ya, 1e-6, 'a new tensor at a dead tensor\'s address')
    # the sweep past the size limit drops dead tensors' entries only -- never a live tensor's, never one a captured launch was handed (pinned)
    cm._frozen_ws.clear()
    run(w)
    t = w * 0.25
    run(t)
    assert len(cm._frozen_ws) == 4
    live = {k: e[1].data_ptr() for k, e in cm._frozen_ws.items() if e[2]() is w}
    next(e for e in cm._frozen_ws.values() if e[2]() is t)[3] = True                 # as if a capture had been handed this workspace
    del t
    cm._sweep_frozen()
    assert len(cm._frozen_ws) == 3 and all(cm._frozen_ws[k][1].data_ptr() == a for k, a in live.items())
    assert sum(1 for e in cm._frozen_ws.values() if e[3]) == 1
    cm._frozen_ws.clear()


@pytest.mark.parametrize('half', [False, True])
def test_conv2d_takes_channels_last_tensors_by_relayout(half):
    """The reference's use_fp16 blocks hand conv2d_resample channels_last tensors (networks_stylegan2.py:424, conv2d_resample.py:31-43).  `spi_conv2d_*`
    take dense NCHW (include/spi_hip.h at spi_conv_desc; INTEGRATION.md sections 1c / 4): the binding relayouts -- a channels_last input or output
    gradient gives the NCHW call's results (up to the summation order of split-K atomics), fp32 and fp16 activation tensors."""
    from spi_amd.torch_utils.ops import conv2d_mfma as cm
    gen = torch.Generator().manual_seed(5)
    dt = torch.float16 if half else torch.float32
    x = torch.randn(2, 32, 40, 44, generator=gen).to(DEV).to(dt)
    w = (torch.randn(2, 48, 32, 3, 3, generator=gen) * 0.06).to(DEV)
    dy = torch.randn(2, 48, 40, 44, generator=gen).to(DEV).to(dt)

    def run(xin, dyin):
        xin = xin.detach().requires_grad_(True)
        wg = w.detach().requires_grad_(True)
        y = cm.conv2d(xin, wg, padding=1, fp16=half)
        gx, gw = torch.autograd.grad(y, [xin, wg], dyin)
        return y, gx, gw
    ref = run(x, dy)
    xcl, dycl = x.contiguous(memory_format=torch.channels_last), dy.contiguous(memory_format=torch.channels_last)
    assert not xcl.is_contiguous()
    got = run(xcl, dycl)
    assert ref[0].dtype == dt and got[0].dtype == dt
    # (small grids take split-K with atomics: the summation order differs run to run, so the NCHW call is reproduced to a few fp32 roundings -- resp.
    #  one fp16 rounding of them -- not bit for bit)
    tol = 1e-3 if half else 1e-6
    assert_close(got[0].float(), ref[0].float(), tol, 'y')
    assert_close(got[1].float(), ref[1].float(), tol, 'dx')
    assert_close(got[2], ref[2], 1e-6, 'dw')
