"""GPU parity of the StyleGAN2 operator kernels (bias_act, upfirdn2d, filtered_lrelu) through the C ABI."""
import json
import math
import pytest
import torch

from conftest import assert_close
from oracle import stylegan_ref as osg

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def test_bias_act_golden(golden):
    from spi_amd.torch_utils.ops import bias_act
    g = golden('ops')
    cases = json.loads(str(g.z['ba_cases'][0]))
    for i, (act, alpha, gain, clamp) in enumerate(cases):
        x = g['ba_x'].to(DEV).requires_grad_(True)
        b = g['ba_b'].to(DEV).requires_grad_(True)
        y = bias_act.bias_act(x, b, act=act, alpha=alpha, gain=gain, clamp=clamp)
        assert_close(y, g[f'ba_y{i}'], 2e-6, f'bias_act {act} fwd')
        gx, gb = torch.autograd.grad(y, [x, b], g['ba_dy'].to(DEV))
        assert_close(gx, g[f'ba_gx{i}'], 5e-6, f'bias_act {act} dx')
        assert_close(gb, g[f'ba_gb{i}'], 5e-6, f'bias_act {act} db')


@pytest.mark.parametrize('shape,dim', [((3, 5), 1), ((2, 7, 3), 1), ((1, 128, 64, 64), 1), ((2, 3, 5, 7), 3), ((5,), 0)])
def test_bias_act_shapes_vs_oracle(shape, dim):
    from spi_amd.torch_utils.ops import bias_act
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(*shape, generator=gen) * 2
    b = torch.randn(shape[dim], generator=gen)
    for act, clamp in (('lrelu', None), ('lrelu', 0.7), ('linear', 0.5)):
        y = bias_act.bias_act(x.to(DEV), b.to(DEV), dim=dim, act=act, clamp=clamp)
        assert_close(y, osg.bias_act(x, b, dim=dim, act=act, clamp=clamp), 1e-6, f'{shape} {act}')


def test_bias_act_refuses_cpu():
    from spi_amd.torch_utils.ops import bias_act
    with pytest.raises(RuntimeError):
        bias_act.bias_act(torch.randn(4, 4), torch.randn(4), act='lrelu')


def test_upfirdn2d_golden(golden):
    from spi_amd.torch_utils.ops import upfirdn2d
    g = golden('ops')
    f = g['fir'].to(DEV)
    cases = json.loads(str(g.z['uf_cases'][0]))
    for i, kw in enumerate(cases):
        x = g['uf_x'].to(DEV).requires_grad_(True)
        y = upfirdn2d.upfirdn2d(x, f, **kw)
        assert_close(y, g[f'uf_y{i}'], 2e-6, f'upfirdn2d[{i}] fwd')
        gx, = torch.autograd.grad(y, x, g[f'uf_dy{i}'].to(DEV))
        assert_close(gx, g[f'uf_gx{i}'], 2e-6, f'upfirdn2d[{i}] dx')


def test_upfirdn2d_layer_shapes_vs_oracle():
    """The two hot instances: FIR after the transposed conv (up=1, pad 1, gain 4) and skip-image upsample (up=2)."""
    from spi_amd.torch_utils.ops import upfirdn2d
    gen = torch.Generator().manual_seed(2)
    f = osg.fir_filter()
    x = torch.randn(2, 16, 65, 65, generator=gen)
    assert_close(upfirdn2d.upfirdn2d(x.to(DEV), f.to(DEV), padding=[1, 1, 1, 1], gain=4), osg.upfirdn2d(x, f, padding=(1, 1, 1, 1), gain=4), 1e-6, 'fir')
    x = torch.randn(1, 96, 32, 32, generator=gen)
    assert_close(upfirdn2d.upsample2d(x.to(DEV), f.to(DEV)), osg.upsample2d(x, f), 1e-6, 'upsample2d')
    # separable 1-D filter and identity filter
    f1 = torch.tensor([1., 2., 4., 8., 8., 4., 2., 1.]) / 30
    x = torch.randn(1, 3, 20, 20, generator=gen)
    assert_close(upfirdn2d.upfirdn2d(x.to(DEV), f1.to(DEV), up=2, padding=[4, 3, 4, 3], gain=4), osg.upfirdn2d(x, f1, up=2, padding=(4, 3, 4, 3), gain=4), 2e-6, 'separable')
    assert_close(upfirdn2d.upfirdn2d(x.to(DEV), None, down=2), osg.upfirdn2d(x, None, down=2), 1e-7, 'identity filter')


def test_upfirdn2d_bias_act_fused_vs_oracle():
    from spi_amd.torch_utils.ops import upfirdn2d
    gen = torch.Generator().manual_seed(4)
    f = osg.fir_filter()
    x = torch.randn(2, 6, 17, 17, generator=gen, requires_grad=True)
    noise = torch.randn(16, 16, generator=gen, requires_grad=True)
    strength = torch.tensor(0.3, requires_grad=True)
    bias = torch.randn(6, generator=gen, requires_grad=True)
    dy = torch.randn(2, 6, 16, 16, generator=gen)
    ref = osg.bias_act(osg.upfirdn2d(x, f, padding=(1, 1, 1, 1), gain=4) + noise * strength, bias, act='lrelu', clamp=1.5)
    gref = torch.autograd.grad(ref, [x, noise, strength, bias], dy)
    xs = [t.detach().to(DEV).requires_grad_(True) for t in (x, noise, strength, bias)]
    y = upfirdn2d.upfirdn2d_bias_act(xs[0], f.to(DEV), noise=xs[1], noise_strength=xs[2], bias=xs[3], padding=[1, 1, 1, 1], gain=4,
                                     act='lrelu', clamp=1.5)
    assert_close(y, ref, 2e-6, 'fused fwd')
    for a, b, nm in zip(torch.autograd.grad(y, xs, dy.to(DEV)), gref, ('dx', 'dnoise', 'dstrength', 'dbias')):
        assert_close(a, b, 1e-5, 'fused ' + nm)


def test_filtered_lrelu_golden(golden):
    from spi_amd.torch_utils.ops import filtered_lrelu
    g = golden('ops')
    cases = json.loads(str(g.z['fl_cases'][0]))
    for i, kw in enumerate(cases):
        y = filtered_lrelu.filtered_lrelu(g['fl_x'].to(DEV), fu=g['fl_fu'].to(DEV), fd=g['fl_fd'].to(DEV), b=g['fl_b'].to(DEV), **kw)
        assert_close(y, g[f'fl_y{i}'], 2e-6, f'filtered_lrelu[{i}]')


@pytest.mark.parametrize('up,down,pad', [(2, 2, [5, 4, 5, 4]), (1, 2, [3, 3, 3, 3]), (2, 1, [2, 1, 2, 1])])
def test_filtered_lrelu_gradients_vs_oracle(up, down, pad):
    """filtered_lrelu with gradients (the differentiable HIP decomposition) against the oracle's autograd, and the fused
    forward-only kernel against the same values."""
    from spi_amd.torch_utils.ops import filtered_lrelu
    gen = torch.Generator().manual_seed(up * 10 + down)
    x = torch.randn(2, 6, 18, 20, generator=gen, requires_grad=True)
    b = torch.randn(6, generator=gen, requires_grad=True)
    fu, fd = osg.fir_filter() * 1.0, osg.fir_filter().flip(0) * 0.9 + 0.01
    ref = osg.filtered_lrelu(x, fu, fd, b, up=up, down=down, padding=pad, gain=1.3, slope=0.15, clamp=0.9)
    dy = torch.randn(ref.shape, generator=gen)
    gx, gb = torch.autograd.grad(ref, [x, b], dy)
    xd, bd = x.detach().to(DEV).requires_grad_(True), b.detach().to(DEV).requires_grad_(True)
    y = filtered_lrelu.filtered_lrelu(xd, fu.to(DEV), fd.to(DEV), bd, up=up, down=down, padding=pad, gain=1.3, slope=0.15, clamp=0.9)
    assert_close(y, ref, 3e-6, 'filtered_lrelu fwd (differentiable path)')
    ax, ab = torch.autograd.grad(y, [xd, bd], dy.to(DEV))
    assert_close(ax, gx, 1e-5, 'filtered_lrelu dx'); assert_close(ab, gb, 1e-5, 'filtered_lrelu db')
    with torch.no_grad():
        y2 = filtered_lrelu.filtered_lrelu(xd, fu.to(DEV), fd.to(DEV), bd, up=up, down=down, padding=pad, gain=1.3, slope=0.15, clamp=0.9)
    assert_close(y2, ref, 3e-6, 'filtered_lrelu fwd (fused kernels)')


@pytest.mark.parametrize('up,down,fu_t,fd_t,pad,hw', [(2, 2, 12, 12, [10, 10, 10, 10], (36, 40)), (4, 2, 12, 6, [9, 8, 7, 6], (17, 23)),
                                                      (2, 4, 8, 16, [6, 5, 6, 5], (40, 33)), (1, 1, 1, 1, [0, 0, 0, 0], (19, 21)), (2, 1, 6, 1, [3, 2, 3, 2], (70, 9))])
def test_filtered_lrelu_single_pass_kernel(up, down, fu_t, fd_t, pad, hw):
    """`spi_filtered_lrelu_fused` (csrc/flrelu.hip: up-FIR, activation, down-FIR in one launch through LDS, like filtered_lrelu.cu:119-1105) at
    StyleGAN3-like filter sizes (12 taps, up / down 2 and 4), ragged tiles, asymmetric padding and flip: the forward against the oracle, the sign tensor
    it writes against the one the stand-alone activation kernel writes on the materialised upsampled tensor, and the gradients (the same launch
    with up / down swapped, reading the signs) against the oracle's autograd."""
    from spi_amd.torch_utils.ops import filtered_lrelu as fl, upfirdn2d as uf
    gen = torch.Generator().manual_seed(up * 100 + down * 10 + fu_t)
    x = torch.randn(2, 5, *hw, generator=gen, requires_grad=True)
    b = torch.randn(5, generator=gen, requires_grad=True)
    k1 = torch.rand(fu_t, generator=gen) + 0.1
    k2 = torch.rand(fd_t, generator=gen) + 0.1
    fu = torch.outer(k1, k1.flip(0)) / k1.sum() ** 2 + 0.01 * torch.rand(fu_t, fu_t, generator=gen) / fu_t ** 2       # 2-D, not symmetric
    fd = torch.outer(k2.flip(0), k2) / k2.sum() ** 2
    for flip in (False, True):
        kw = dict(up=up, down=down, padding=pad, gain=1.4, slope=0.2, clamp=0.8, flip_filter=flip)
        ref = osg.filtered_lrelu(x, fu, fd, b, **kw)
        dy = torch.randn(ref.shape, generator=gen)
        gx, gb = torch.autograd.grad(ref, [x, b], dy)
        xd, bd = x.detach().to(DEV).requires_grad_(True), b.detach().to(DEV).requires_grad_(True)
        with torch.no_grad():
            assert_close(fl.filtered_lrelu(xd, fu.to(DEV), fd.to(DEV), bd, **kw), ref, 3e-6, 'single-pass forward')
        y = fl.filtered_lrelu(xd, fu.to(DEV), fd.to(DEV), bd, **kw)
        assert_close(y, ref, 3e-6, 'single-pass forward (writing signs)')
        ax, ab = torch.autograd.grad(y, [xd, bd], dy.to(DEV))
        assert_close(ax, gx, 1e-5, 'single-pass dx')
        assert_close(ab, gb, 1e-5, 'single-pass db')
        # the sign tensor: fused launch == activation kernel on the materialised upsampled tensor
        mid = uf.upfirdn2d(xd.detach() + bd.detach().reshape(1, -1, 1, 1), fu.to(DEV), up=up, padding=pad, gain=up ** 2, flip_filter=flip).contiguous()
        want = fl.filtered_lrelu_act_(mid.clone(), None, 0, 0, 1.4, 0.2, 0.8, write_signs=True)
        _, got = fl._fused(xd.detach(), fu.to(DEV), fd.to(DEV), bd.detach(), up, down, tuple(pad), 1.4, 0.2, 0.8, flip, None, 0, 0, True)
        assert got.shape == want.shape
        # samples under no output's down-filter footprint are never evaluated by the single-pass kernel (their sign is irrelevant: no gradient reaches
        # them) -- compare where an output depends on the sample
        need_h = (ref.shape[2] - 1) * down + fd_t
        need_w = (ref.shape[3] - 1) * down + fd_t
        wb = (min(need_w, mid.shape[3]) // 4)
        assert torch.equal(got[:, :, :min(need_h, mid.shape[2]), :wb], want[:, :, :min(need_h, mid.shape[2]), :wb])


@pytest.mark.parametrize('inh,pad', [(201, 1), (199, 2), (130, 1), (113, 2)])
def test_upfirdn2d_tiled_ragged_vs_oracle(inh, pad):
    """LDS-tiled 4x4 FIR (outputs >= 100 px): ragged tile edges, unaligned rows (scalar stores) and the fused tail + gradients."""
    from spi_amd.torch_utils.ops import upfirdn2d
    gen = torch.Generator().manual_seed(inh)
    f = osg.fir_filter()
    oh = inh + 2 * pad - 3
    x = torch.randn(2, 5, inh, inh - 7, generator=gen, requires_grad=True)
    ow = oh - 7
    noise = torch.randn(oh, ow, generator=gen, requires_grad=True)
    strength = torch.tensor(0.3, requires_grad=True)
    bias = torch.randn(5, generator=gen, requires_grad=True)
    dy = torch.randn(2, 5, oh, ow, generator=gen)
    ref = osg.bias_act(osg.upfirdn2d(x, f, padding=(pad,) * 4, gain=4) + noise * strength, bias, act='lrelu', clamp=1.5)
    gref = torch.autograd.grad(ref, [x, noise, strength, bias], dy)
    xs = [t.detach().to(DEV).requires_grad_(True) for t in (x, noise, strength, bias)]
    y = upfirdn2d.upfirdn2d_bias_act(xs[0], f.to(DEV), noise=xs[1], noise_strength=xs[2], bias=xs[3], padding=[pad] * 4, gain=4,
                                     act='lrelu', clamp=1.5)
    assert_close(y, ref, 2e-6, 'tiled fused fwd')
    for a, b, nm in zip(torch.autograd.grad(y, xs, dy.to(DEV)), gref, ('dx', 'dnoise', 'dstrength', 'dbias')):
        assert_close(a, b, 2e-5, 'tiled fused ' + nm)
    assert_close(upfirdn2d.upfirdn2d(x.detach().to(DEV), f.to(DEV), padding=[pad] * 4, flip_filter=True),
                 osg.upfirdn2d(x.detach(), f, padding=(pad,) * 4, flip_filter=True), 2e-6, 'tiled plain')


def test_filtered_lrelu_act_sign_tensor_modes():
    """`filtered_lrelu_act_` (filtered_lrelu.cpp:217-296, .cu:1109-1215): in-place gain / lrelu / clamp, the bit-packed sign tensor it writes
    (2 bits per element: 1 = negative, 2 = clamped; 4 per byte; width rounded up to 16 elements) and the gradient pass that reads it at an offset."""
    from spi_amd.torch_utils.ops.filtered_lrelu import filtered_lrelu_act_
    gen = torch.Generator().manual_seed(4)
    n, c, h, w = 2, 3, 9, 37
    x = torch.randn(n, c, h, w, generator=gen) * 2
    gain, slope, clamp = 1.3, 0.2, 1.7
    v = x * gain
    neg = v < 0
    v = torch.where(neg, v * slope, v)
    clamped = v.abs() > clamp
    ref = v.clamp(-clamp, clamp)
    code = torch.where(clamped, torch.full_like(x, 2), neg.float()).long()               # clamp overrides sign
    for write in (False, True):
        y = x.clone().to(DEV)
        so = filtered_lrelu_act_(y, None, 0, 0, gain, slope, clamp, write_signs=write)
        assert_close(y, ref, 1e-6, 'act forward')
        if not write:
            assert so.numel() == 0
            continue
        sw = (w + 15) & ~15
        assert so.dtype == torch.uint8 and so.shape == (n, c, h, sw // 4)
        s = so.cpu().long()
        unpacked = torch.stack([(s >> (2 * j)) & 3 for j in range(4)], dim=-1).reshape(n, c, h, sw)
        assert torch.equal(unpacked[..., :w], code) and int(unpacked[..., w:].sum()) == 0
        # gradient pass: read the signs at an offset (sx, sy); outside the sign tensor the element is just scaled by the gain
        sx, sy = 3, -2
        g = torch.randn(n, c, h + 1, w - 5, generator=gen)
        gd = g.clone().to(DEV).contiguous()
        filtered_lrelu_act_(gd, so, sx, sy, 0.7, slope, None, write_signs=False)
        yy, xx = torch.meshgrid(torch.arange(h + 1) + sy, torch.arange(w - 5) + sx, indexing='ij')
        inside = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < sw)
        cc = torch.zeros(n, c, h + 1, w - 5, dtype=torch.long)
        cc[:, :, inside] = unpacked[:, :, yy[inside], xx[inside]]
        expect = g * 0.7 * torch.where(cc == 1, torch.tensor(slope), torch.tensor(1.0)) * (cc != 2).float()
        assert_close(gd, expect, 1e-6, 'act gradient pass (read signs)')
    # no clamp: code 2 never appears
    y = x.clone().to(DEV)
    so = filtered_lrelu_act_(y, None, 0, 0, gain, slope, None, write_signs=True)
    assert int(((so.cpu().long() >> 1) & 0x55).sum()) == 0


@pytest.mark.parametrize('n,i,o', [(1, 512, 512), (4, 512, 256), (8, 512, 32), (1, 512, 3), (3, 24, 7), (4, 516, 70)])
def test_affine_layer_matrix_vector_kernels(n, i, o):
    """FullyConnectedLayer (activation linear) at inversion batch sizes runs on spi_affine_fwd / _bwd: forward, dx, dW and db against fp64
    (networks_stylegan2.py:95-127: y = x (W * weight_gain)^T + b * bias_gain); nine rows fall back to the library GEMM with the same results."""
    from spi_amd.training.networks_stylegan2 import FullyConnectedLayer
    gen = torch.Generator().manual_seed(n * 1000 + i + o)
    fc = FullyConnectedLayer(i, o, bias_init=1).to(DEV)
    with torch.no_grad():
        fc.weight.copy_(torch.randn(o, i, generator=gen)); fc.bias.copy_(torch.randn(o, generator=gen))
    for rows in (n, 9):
        x = torch.randn(rows, i, generator=gen).to(DEV).requires_grad_(True)
        gy = torch.randn(rows, o, generator=gen).to(DEV)
        y = fc(x)
        dx, dw, db = torch.autograd.grad(y, [x, fc.weight, fc.bias], gy)
        xr = x.detach().double().cpu().requires_grad_(True); wr = fc.weight.detach().double().cpu().requires_grad_(True); br = fc.bias.detach().double().cpu().requires_grad_(True)
        yr = xr @ (wr * fc.weight_gain).t() + br * fc.bias_gain
        dxr, dwr, dbr = torch.autograd.grad(yr, [xr, wr, br], gy.double().cpu())
        assert_close(y, yr.float(), 2e-6, f'affine y rows={rows}'); assert_close(dx, dxr.float(), 2e-6, 'affine dx')
        assert_close(dw, dwr.float(), 2e-6, 'affine dW'); assert_close(db, dbr.float(), 2e-6, 'affine db')
    # frozen weights (stage 1): only dx is asked for
    x = torch.randn(n, i, generator=gen).to(DEV).requires_grad_(True)
    fc.requires_grad_(False)
    (dx,) = torch.autograd.grad(fc(x).square().sum(), [x])
    xr = x.detach().double().cpu().requires_grad_(True)
    (dxr,) = torch.autograd.grad((xr @ (fc.weight.double().cpu() * fc.weight_gain).t() + fc.bias.double().cpu() * fc.bias_gain).square().sum(), [xr])
    assert_close(dx, dxr.float(), 3e-6, 'affine dx (frozen)')


@pytest.mark.parametrize('n', [1, 3])
def test_multi_affine_equals_the_per_layer_affines(n):
    """`multi_affine` (spi_affine_multi_fwd / _bwd): all style layers of a network in one launch each way -- outputs bit-equal to the per-layer
    FullyConnectedLayers, ws / weight / bias gradients equal to theirs (the ws gradient sums the layers that share a row: a + b, commutative),
    layers with no incoming gradient skipped, frozen weights (stage 1) ask for d ws only."""
    from spi_amd.training.networks_stylegan2 import FullyConnectedLayer, multi_affine
    gen = torch.Generator().manual_seed(40 + n)
    outs, rows = [512, 512, 96 * 0 + 256, 64, 32, 512], [0, 1, 1, 2, 3, 3]

    class L(torch.nn.Module):
        def __init__(self, o):
            super().__init__()
            self.affine = FullyConnectedLayer(512, o, bias_init=1)
    mods = [L(o).to(DEV) for o in outs]
    with torch.no_grad():
        for m in mods:
            m.affine.weight.copy_(torch.randn(m.affine.weight.shape, generator=gen)); m.affine.bias.copy_(torch.randn(m.affine.bias.shape, generator=gen))
    ws = torch.randn(n, 5, 512, generator=gen).to(DEV).requires_grad_(True)
    gys = [torch.randn(n, o, generator=gen).to(DEV) for o in outs]
    params = [p for m in mods for p in (m.affine.weight, m.affine.bias)]
    ref = [m.affine(ws[:, r]) for m, r in zip(mods, rows)]
    got = multi_affine(ws, list(zip(mods, rows)))
    assert got is not None and len(got) == len(mods)
    for a, b in zip(got, ref):
        assert torch.equal(a, b)
    use = [0, 1, 2, 4, 5]                                        # layer 3 gets no gradient
    g_ref = torch.autograd.grad([ref[i] for i in use], [ws] + params, [gys[i] for i in use], allow_unused=True)
    g_got = torch.autograd.grad([got[i] for i in use], [ws] + params, [gys[i] for i in use], allow_unused=True)
    for k, (a, b) in enumerate(zip(g_got, g_ref)):
        assert (a is None) == (b is None), k
        if a is not None:
            assert_close(a, b, 1e-6, f'multi-affine gradient {k}')
    for m in mods:
        m.requires_grad_(False)
    got = multi_affine(ws, list(zip(mods, rows)))
    (d1,) = torch.autograd.grad(sum(g.square().sum() for g in got), [ws])
    (d2,) = torch.autograd.grad(sum(m.affine(ws[:, r]).square().sum() for m, r in zip(mods, rows)), [ws])
    assert_close(d1, d2, 2e-6, 'multi-affine d ws (frozen weights)')
    assert multi_affine(ws.double(), list(zip(mods, rows))) is None and multi_affine(ws.cpu(), list(zip(mods, rows))) is None


# ---- the typed plugin boundary (round 4): fp16 tensors and channels_last strides, as the reference's plugins are instantiated ----------------
def _half_ulps(a, b):
    """largest difference of two fp16 tensors in units of b's ulp (0 = bit-equal)"""
    a, b = a.detach().cpu().float(), b.detach().cpu().float()
    ulp = torch.maximum(b.abs(), torch.full_like(b, 2.0 ** -14)).log2().floor().exp2() * 2.0 ** -10
    return ((a - b).abs() / ulp).max().item()


def test_bias_act_fp16_golden_cases(golden):
    """bias_act.cpp:81 dispatches the plugin over half / float / double; half tensors are computed in fp32 and rounded once at the store
    (bias_act.cu:14-16 InternalType).  The 11 golden activation / clamp cases with fp16 x, b, dy: the result must be the fp32 oracle of the
    fp16-rounded inputs, rounded to fp16 (<= 1 fp16 ulp: transcendental functions differ in the last fp32 bit); outputs stay fp16."""
    from spi_amd.torch_utils.ops import bias_act
    g = golden('ops')
    cases = json.loads(str(g.z['ba_cases'][0]))
    x16, b16, dy16 = g['ba_x'].half(), g['ba_b'].half(), g['ba_dy'].half()
    for i, (act, alpha, gain, clamp) in enumerate(cases):
        x = x16.to(DEV).requires_grad_(True)
        b = b16.to(DEV).requires_grad_(True)
        y = bias_act.bias_act(x, b, act=act, alpha=alpha, gain=gain, clamp=clamp)
        assert y.dtype == torch.float16
        xr, br = x16.float().requires_grad_(True), b16.float().requires_grad_(True)
        yr = osg.bias_act(xr, br, act=act, alpha=alpha, gain=gain, clamp=clamp)
        assert _half_ulps(y, yr.half()) <= 1.0, (act, clamp, _half_ulps(y, yr.half()))
        gx, gb = torch.autograd.grad(y, [x, b], dy16.to(DEV))
        assert gx.dtype == torch.float16
        # the backward reads the SAVED fp16 output (the reference plugin's yref): its oracle is the fp32 formula on that rounded tensor
        gxr, gbr = torch.autograd.grad(yr, [xr, br], dy16.float())
        assert_close(gx, gxr, 4e-3, f'fp16 bias_act {act} dx')
        assert_close(gb, gbr, 4e-3, f'fp16 bias_act {act} db')
    # odd sizes take the scalar kernel, multiples of 4 the 8-byte vector kernel: same numbers
    xo = torch.randn(3, 5, 7, generator=torch.Generator().manual_seed(3)).half()
    bo = torch.randn(5, generator=torch.Generator().manual_seed(4)).half()
    y = bias_act.bias_act(xo.to(DEV), bo.to(DEV), act='lrelu', clamp=0.9)
    assert _half_ulps(y, osg.bias_act(xo.float(), bo.float(), act='lrelu', clamp=0.9).half()) <= 1.0


def test_upfirdn2d_fp16_and_channels_last(golden):
    """upfirdn2d.cpp:67 dispatches over half / float / double and keeps the input's memory format (:42): the six golden pad / up / down cases
    with (a) fp16 NCHW, (b) fp32 channels_last, (c) fp16 channels_last -- what the reference's fp16 super-resolution blocks hand the plugin."""
    from spi_amd.torch_utils.ops import upfirdn2d
    g = golden('ops')
    f = g['fir'].to(DEV)
    cases = json.loads(str(g.z['uf_cases'][0]))
    for i, kw in enumerate(cases):
        x32 = g['uf_x']
        ref = g[f'uf_y{i}']
        xcl = x32.to(DEV).contiguous(memory_format=torch.channels_last)
        ycl = upfirdn2d.upfirdn2d(xcl, f, **kw)
        assert ycl.dtype == torch.float32 and ycl.is_contiguous(memory_format=torch.channels_last)
        assert_close(ycl, ref, 2e-6, f'upfirdn2d[{i}] fp32 channels_last')
        x16 = x32.half()
        ref16 = osg.upfirdn2d(x16.float(), g['fir'], **kw).half()
        for tag, xin in (('nchw', x16.to(DEV)), ('channels_last', x16.to(DEV).contiguous(memory_format=torch.channels_last))):
            xin = xin.requires_grad_(True)
            y = upfirdn2d.upfirdn2d(xin, f, **kw)
            assert y.dtype == torch.float16 and y.shape == ref.shape
            assert y.is_contiguous(memory_format=torch.channels_last) == (tag == 'channels_last')
            assert _half_ulps(y, ref16) <= 1.0, (i, tag, _half_ulps(y, ref16))
            gx, = torch.autograd.grad(y, xin, g[f'uf_dy{i}'].half().to(DEV))
            assert gx.dtype == torch.float16
            assert_close(gx, g[f'uf_gx{i}'], 3e-3, f'upfirdn2d[{i}] fp16 {tag} dx')


def test_typed_entry_points_through_the_c_abi():
    """spi_bias_act_t / spi_upfirdn2d_t called directly: dtype 0 forwards to the fp32 kernels, dtype 1 is fp16, anything else returns
    SPI_ERR_UNSUPPORTED (-2) with a message -- the caller's cue to fall back, like filtered_lrelu's rc = -1 in the reference."""
    import ctypes
    from spi_amd import hip
    x = torch.randn(2, 4, 6, 6, device=DEV)
    b = torch.randn(4, device=DEV)
    y0, y1 = torch.empty_like(x), torch.empty_like(x)
    hip.call('spi_bias_act', hip.ptr(x), hip.ptr(b), None, None, None, hip.ptr(y0), x.numel(), 4, 36, 0, 3, 0.2, 1.4142, -1.0, hip.stream())
    hip.call('spi_bias_act_t', hip.ptr(x), hip.ptr(b), None, None, None, hip.ptr(y1), x.numel(), 4, 36, 0, 3, 0.2, 1.4142, -1.0, 0, hip.stream())
    assert torch.equal(y0, y1)
    xh, bh, yh = x.half(), b.half(), torch.empty_like(x, dtype=torch.float16)
    hip.call('spi_bias_act_t', xh.data_ptr(), bh.data_ptr(), None, None, None, yh.data_ptr(), x.numel(), 4, 36, 0, 3, 0.2, 1.4142, -1.0, 1, hip.stream())
    assert _half_ulps(yh, osg.bias_act(xh.float().cpu(), bh.float().cpu(), act='lrelu', gain=1.4142).half()) <= 1.0
    rc = hip.lib().spi_bias_act_t(hip.ptr(x), None, None, None, None, hip.ptr(y1), x.numel(), 0, 0, 0, 3, 0.2, 1.0, -1.0, 3, hip.stream())
    assert rc == -2 and b'dtype 3' in hip.lib().spi_last_error()      # (2 = fp64 is served since round 5)
    f = torch.tensor([[1., 2.], [3., 4.]], device=DEV) / 10
    yo = torch.empty(2, 4, 5, 5, device=DEV)
    rc = hip.lib().spi_upfirdn2d_t(hip.ptr(x), hip.ptr(f), hip.ptr(yo), 2, 4, 6, 6, None, None, 2, 2, 1, 1, 1, 1, 0, 0, 0, 0, 0, 1.0, 5, 5, 7, hip.stream())
    assert rc == -2
    bad = (ctypes.c_int64 * 4)(144, 36, 12, 2)                     # overlapping / non-dense strides are refused, like the plugin (upfirdn2d.cpp:23)
    rc = hip.lib().spi_upfirdn2d_t(hip.ptr(x), hip.ptr(f), hip.ptr(yo), 2, 4, 6, 6, ctypes.cast(bad, ctypes.c_void_p), None, 2, 2, 1, 1, 1, 1, 0, 0, 0, 0, 0, 1.0,
                                   5, 5, 0, hip.stream())
    assert rc == -1 and b'dense' in hip.lib().spi_last_error()


def test_plugin_boundary_fp64_and_half_filtered_lrelu(golden):
    """the rest of the plugins' dtype dispatch.  (a) bias_act.cpp:81 / upfirdn2d.cpp:67 instantiate the plugins for double:
    `spi_bias_act_t` / `spi_upfirdn2d_t` with SPI_DTYPE_F64 compute in double (all golden activation / clamp cases forward and backward, all
    golden pad / up / down cases, against the oracle run in fp64: 1e-7 -- the C ABI carries alpha / gain / clamp / the FIR taps as fp32 scalars, so
    sqrt(2) arrives rounded to fp32; the arithmetic on the tensors is double).  (b) filtered_lrelu.cpp:151 dispatches half: `spi_filtered_lrelu_t`
    on fp16 x / b / y with fp32 filters and intermediate -- the golden cases on fp16-rounded inputs within one fp16 ulp of the fp32 oracle."""
    from spi_amd.torch_utils.ops import bias_act, upfirdn2d, filtered_lrelu
    g = golden('ops')
    cases = json.loads(str(g.z['ba_cases'][0]))
    x64, b64, dy64 = g['ba_x'].double(), g['ba_b'].double(), g['ba_dy'].double()
    for act, alpha, gain, clamp in cases:
        x, b = x64.to(DEV).requires_grad_(True), b64.to(DEV).requires_grad_(True)
        y = bias_act.bias_act(x, b, act=act, alpha=alpha, gain=gain, clamp=clamp)
        assert y.dtype == torch.float64
        xr, br = x64.clone().requires_grad_(True), b64.clone().requires_grad_(True)
        yr = osg.bias_act(xr, br, act=act, alpha=alpha, gain=gain, clamp=clamp)
        assert_close(y, yr, 1e-7, f'fp64 bias_act {act}')
        gx, gb = torch.autograd.grad(y, [x, b], dy64.to(DEV))
        gxr, gbr = torch.autograd.grad(yr, [xr, br], dy64)
        assert gx.dtype == torch.float64
        assert_close(gx, gxr, 1e-7, f'fp64 bias_act {act} dx')
        assert_close(gb, gbr, 1e-7, f'fp64 bias_act {act} db')
    f = g['fir']
    for i, kw in enumerate(json.loads(str(g.z['uf_cases'][0]))):
        y = upfirdn2d.upfirdn2d(g['uf_x'].double().to(DEV), f.to(DEV), **kw)
        assert y.dtype == torch.float64
        assert_close(y, osg.upfirdn2d(g['uf_x'].double(), f.double(), **kw), 1e-7, f'fp64 upfirdn2d[{i}]')
    # (b) filtered_lrelu on half tensors
    x16, b16 = g['fl_x'].half(), g['fl_b'].half()
    for i, kw in enumerate(json.loads(str(g.z['fl_cases'][0]))):
        with torch.no_grad():
            y = filtered_lrelu.filtered_lrelu(x16.to(DEV), fu=g['fl_fu'].to(DEV), fd=g['fl_fd'].to(DEV), b=b16.to(DEV), **kw)
        assert y.dtype == torch.float16
        yr = osg.filtered_lrelu(x16.float(), fu=g['fl_fu'], fd=g['fl_fd'], b=b16.float(), **kw)
        assert _half_ulps(y, yr.half()) <= 1.0, (i, _half_ulps(y, yr.half()))


@pytest.mark.gpu
@pytest.mark.parametrize('n,c,h,w,act,half', [(1, 20, 33, 35, 'lrelu', False), (3, 130, 32, 48, 'lrelu', False), (2, 64, 64, 64, 'linear', False),
                                              (2, 48, 40, 40, 'relu', False), (2, 128, 64, 96, 'lrelu', True), (4, 600, 8, 8, 'lrelu', False)])
def test_tail_backward_and_fused_channel_dot_vs_autograd(n, c, h, w, act, half):
    """spi_tail_bwd(_t) / spi_tail_bwd_dot_t (ABI 12) against autograd of  y = clamp(act(z + noise * s + b) * gain):  dz, d bias, d noise, d strength --
    and the fused per-(n, c) dot product  sum_hw dz * z  with z reconstructed from y (the frozen-weight style gradient's second dot product),
    which also equals the separate spi_chan_dot pass it replaces.  fp32 and fp16 tensors, ragged pixel counts (scalar path), more than one channel
    chunk, N * cchunk beyond one block's LDS partials."""
    from spi_amd import hip
    from spi_amd.torch_utils.ops import bias_act
    gen = torch.Generator().manual_seed(n * 7 + c)
    act_id, alpha, gain, clamp = {'linear': (1, 0.0, 1.3, 1.5), 'relu': (2, 0.0, 1.41, 2.0), 'lrelu': (3, 0.2, 1.41, 1.8)}[act]
    z = torch.randn(n, c, h, w, generator=gen, dtype=torch.float64, requires_grad=True)
    b = torch.randn(c, generator=gen, dtype=torch.float64, requires_grad=True)
    noise = torch.randn(h, w, generator=gen, dtype=torch.float64, requires_grad=True)
    s = torch.tensor(0.4, dtype=torch.float64, requires_grad=True)
    pre = z + noise * s + b.view(1, -1, 1, 1)
    a = pre if act == 'linear' else (torch.relu(pre) if act == 'relu' else torch.nn.functional.leaky_relu(pre, alpha))
    y = (a * gain).clamp(-clamp, clamp)
    dy = torch.randn(n, c, h, w, generator=gen, dtype=torch.float64)
    if half:
        dy = dy.half().double()
    gz, gb, gn, gs = torch.autograd.grad(y, [z, b, noise, s], dy)
    dt = torch.float16 if half else torch.float32
    yd, dyd = y.detach().to(DEV, dt), dy.to(DEV, dt)
    nzd, sd, bd = noise.detach().float().to(DEV), s.detach().float().to(DEV).reshape(1), b.detach().float().to(DEV)
    zo = torch.zeros(n * c, device=DEV)
    dz, d_noise, d_strength, d_bias = bias_act.tail_backward(dyd, yd, nzd, sd, act_id, alpha, gain, clamp, True, True, True, zdot=(zo, bd, nzd, sd))
    tol = 2e-3 if half else 1e-5
    # (an element that sits exactly on the clamp / at the kink after rounding y to the tensor type is a different element: compare where |y| is clear of both)
    clear = ((y.detach().abs() - clamp).abs() > 1e-2) & (y.detach().abs() > 1e-2)
    assert_close(dz.double().cpu() * clear, gz * clear, tol, 'dz')
    if not half:
        assert_close(d_bias.cpu().double(), gb, 1e-4, 'd bias')
        assert_close(d_noise.cpu().double(), gn, 1e-4, 'd noise')
        assert_close(d_strength.cpu().double().reshape(()), gs, 1e-4, 'd strength')
    # the fused dot product equals the separate pass on the same tensors (same reconstruction), and the exact sum up to the rounding of y
    if act != 'relu':                                  # (spi_chan_dot refuses relu outputs; the fused form is exact there too: dz == 0 wherever y cannot be inverted)
        zc = torch.zeros(n * c, device=DEV)
        name = 'spi_chan_dot_t' if half else 'spi_chan_dot'
        extra = (hip.DTYPE_IDS[torch.float16],) if half else ()
        hip.call(name, hip.ptr(dz), hip.ptr(yd), hip.ptr(zc), n * c, c, h * w, hip.ptr(bd), hip.ptr(nzd), hip.ptr(sd), act_id, alpha, gain, *extra, hip.stream())
        assert_close(zo, zc, 1e-3 if half else 1e-5, 'fused <dz, z> vs spi_chan_dot')       # (half: the separate pass reads dz AFTER its rounding to fp16)
    if not half:
        exact = (gz * z.detach()).sum(dim=(2, 3)).reshape(-1)
        assert_close(zo.cpu().double(), exact, 1e-4, 'fused <dz, z> vs the exact sum')
