#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by importing the *reference* (FeiiYin/SPI at
/root/reference) on CPU.  Runs only in the build container (the GPU box has no reference);
the committed ``*.npz`` / ``*.json`` files are data: inputs + the reference's outputs.

    python tests/golden/make_golden.py [ops renderer synthesis geometry schedule trajectory manifest]

Every section also prints max |reference - oracle| so the oracle is pinned at generation time.
"""
import json
import math
import os
import sys

os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), REF, REF + '/eg3d', REF + '/spi']

import numpy as np
import torch
import torch.nn.functional as F

from synth_weights import synth_tensor, synth_state_dict  # noqa: E402
from oracle import stylegan_ref as osg, renderer_ref as orr, losses_ref as olo  # noqa: E402

torch.set_num_threads(8)


def npy(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def save(name, **arrs):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **{k: npy(v) for k, v in arrs.items()})
    print(f'  wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} KiB)')


def diff(tag, a, b):
    a, b = torch.as_tensor(npy(a)).double(), torch.as_tensor(npy(b)).double()
    d = (a - b).abs().max().item()
    rel = d / max(a.abs().max().item(), 1e-30)
    print(f'    pin {tag:34s} max|ref-oracle| = {d:.3e}  (rel {rel:.2e})')
    return d


RK = dict(superresolution_module='training.superresolution.SuperresolutionHybrid8XDC', sr_antialias=True,
          superresolution_noise_mode='none', c_gen_conditioning_zero=False, c_scale=1.0, clamp_mode='softplus',
          disparity_space_sampling=False, decoder_lr_mul=1.0, box_warp=1, ray_start=2.25, ray_end=3.3,
          depth_resolution=12, depth_resolution_importance=12, white_back=False)


def build_ref_generator(narrow=True, rk=None):
    from training.triplane import TriPlaneGenerator
    rk = dict(RK if rk is None else rk)
    cb, cm = (2048, 32) if narrow else (32768, 512)
    G = TriPlaneGenerator(z_dim=512, c_dim=25, w_dim=512, img_resolution=512, img_channels=3,
                          mapping_kwargs=dict(num_layers=2), channel_base=cb, channel_max=cm,
                          fused_modconv_default='inference_only', num_fp16_res=0, conv_clamp=None, sr_num_fp16_res=4,
                          sr_kwargs=dict(channel_base=cb, channel_max=cm, fused_modconv_default='inference_only'),
                          rendering_kwargs=rk).eval()
    sd = G.state_dict()
    G.load_state_dict({k: synth_tensor(k, v.shape) for k, v in sd.items()})
    return G


class Recorder:
    """Record torch.rand / rand_like / randn_like draws made by the reference."""
    def __init__(self, randn=False):
        self.draws = []
        self.randn = randn               # also record torch.randn (StyleGAN2 noise_mode='random' draws)

    def __enter__(self):
        self._o = (torch.rand, torch.rand_like, torch.randn_like)
        self._randn = torch.randn
        rec = self

        def wrap(fn):
            def inner(*a, **k):
                out = fn(*a, **k)
                rec.draws.append(out.detach().clone())
                return out
            return inner
        torch.rand, torch.rand_like, torch.randn_like = [wrap(f) for f in self._o]
        if self.randn:
            torch.randn = wrap(self._randn)
        return self

    def __exit__(self, *a):
        torch.rand, torch.rand_like, torch.randn_like = self._o
        torch.randn = self._randn


class SeededRecorder:
    """Replace the reference's random stream by a counter-seeded one (tests/loss_inputs.indexed_draw): draw j of torch.rand / rand_like /
    randn_like comes from its own generator seeded with base + j.  Which uniform numbers the loop consumes is not arithmetic of the path;
    making them reproducible from (index, shape) keeps the stage-2 fixtures small.  Records shapes + kinds."""
    def __init__(self):
        self.draws, self.shapes, self.kinds = [], [], []

    def __enter__(self):
        import loss_inputs as li
        self._o = (torch.rand, torch.rand_like, torch.randn_like)
        rec = self

        def make(kind, like):
            def inner(*a, **k):
                shape = tuple(a[0].shape) if like else (tuple(a[0]) if len(a) == 1 and isinstance(a[0], (tuple, list, torch.Size)) else tuple(a))
                assert k.get('generator') is None
                t = li.indexed_draw(len(rec.draws), shape, kind)
                rec.draws.append(t)
                rec.shapes.append(list(shape))
                rec.kinds.append(kind)
                return t.clone()
            return inner
        torch.rand, torch.rand_like, torch.randn_like = make('rand', False), make('rand', True), make('randn', True)
        return self

    def __exit__(self, *a):
        torch.rand, torch.rand_like, torch.randn_like = self._o


# ------------------------------------------------------------------------------------------------
def sec_manifest():
    for kind, narrow in (('narrow', True), ('full', False)):
        G = build_ref_generator(narrow)
        man = {k: list(v.shape) for k, v in G.state_dict().items()}
        with open(os.path.join(HERE, f'manifest_{kind}.json'), 'w') as f:
            json.dump(man, f, indent=0)
        nparam = sum(p.numel() for p in G.parameters())
        print(f'  manifest_{kind}.json: {len(man)} entries, {nparam} params')


def sec_ops():
    from torch_utils.ops import bias_act, upfirdn2d, filtered_lrelu, conv2d_resample
    from training.networks_stylegan2 import modulated_conv2d
    g = torch.Generator().manual_seed(11)
    out = {}
    # bias_act: forward + grad wrt x and b
    x = torch.randn(2, 6, 5, 7, generator=g) * 3
    b = torch.randn(6, generator=g)
    dy = torch.randn(2, 6, 5, 7, generator=g)
    out['ba_x'], out['ba_b'], out['ba_dy'] = x, b, dy
    cases = [('lrelu', None, None, None), ('lrelu', 0.2, math.sqrt(2), 2.0), ('linear', None, None, 1.5),
             ('linear', None, None, None), ('relu', None, None, None), ('sigmoid', None, None, None),
             ('tanh', None, None, None), ('softplus', None, None, None), ('swish', None, None, None),
             ('elu', None, None, None), ('selu', None, None, None)]
    for i, (act, alpha, gain, clamp) in enumerate(cases):
        xr, br = x.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = bias_act.bias_act(xr, br, act=act, alpha=alpha, gain=gain, clamp=clamp)
        gx, gb = torch.autograd.grad(y, [xr, br], dy)
        out[f'ba_y{i}'], out[f'ba_gx{i}'], out[f'ba_gb{i}'] = y, gx, gb
        diff(f'bias_act[{act},clamp={clamp}]', y, osg.bias_act(x, b, act=act, alpha=alpha, gain=gain, clamp=clamp))
    out['ba_cases'] = np.array([json.dumps(cases)])
    # upfirdn2d
    f = upfirdn2d.setup_filter([1, 3, 3, 1])
    out['fir'] = f
    diff('setup_filter', f, osg.fir_filter())
    x = torch.randn(2, 3, 9, 9, generator=g)
    out['uf_x'] = x
    ucases = [dict(up=1, down=1, padding=[1, 1, 1, 1], gain=4.0, flip_filter=False),
              dict(up=2, down=1, padding=[2, 1, 2, 1], gain=4.0, flip_filter=False),
              dict(up=1, down=2, padding=[1, 1, 1, 1], gain=1.0, flip_filter=True),
              dict(up=1, down=2, padding=[2, 2, 2, 2], gain=4.0, flip_filter=True),
              dict(up=1, down=1, padding=[2, 1, 2, 1], gain=1.0, flip_filter=True),
              dict(up=2, down=2, padding=[3, 2, 1, 0], gain=2.0, flip_filter=False)]
    for i, kw in enumerate(ucases):
        xr = x.clone().requires_grad_(True)
        y = upfirdn2d.upfirdn2d(xr, f, **kw)
        dyu = torch.randn(y.shape, generator=g)
        gx, = torch.autograd.grad(y, xr, dyu)
        out[f'uf_y{i}'], out[f'uf_dy{i}'], out[f'uf_gx{i}'] = y, dyu, gx
        diff(f'upfirdn2d[{i}]', y, osg.upfirdn2d(x, f, **kw))
    out['uf_cases'] = np.array([json.dumps(ucases)])
    diff('upsample2d', upfirdn2d.upsample2d(x, f), osg.upsample2d(x, f))
    # filtered_lrelu (ref path)
    fu = upfirdn2d.setup_filter([1, 3, 3, 1])
    fd = upfirdn2d.setup_filter([1, 2, 1])
    xb = torch.randn(2, 3, 8, 8, generator=g)
    bb = torch.randn(3, generator=g)
    flc = [dict(up=2, down=2, padding=[3, 2, 3, 2], gain=math.sqrt(2), slope=0.2, clamp=None, flip_filter=False),
           dict(up=2, down=1, padding=[2, 1, 2, 1], gain=1.3, slope=0.1, clamp=0.8, flip_filter=False),
           dict(up=1, down=2, padding=[1, 1, 1, 1], gain=1.0, slope=0.2, clamp=None, flip_filter=True)]
    out['fl_x'], out['fl_b'], out['fl_fu'], out['fl_fd'] = xb, bb, fu, fd
    for i, kw in enumerate(flc):
        xr = xb.clone().requires_grad_(True)
        y = filtered_lrelu.filtered_lrelu(xr, fu=fu, fd=fd, b=bb, impl='ref', **kw)
        dyf = torch.randn(y.shape, generator=g)
        gx, = torch.autograd.grad(y, xr, dyf)
        out[f'fl_y{i}'], out[f'fl_dy{i}'], out[f'fl_gx{i}'] = y, dyf, gx
        diff(f'filtered_lrelu[{i}]', y, osg.filtered_lrelu(xb, fu, fd, bb, **kw))
    out['fl_cases'] = np.array([json.dumps(flc)])
    # modulated conv: up=1 3x3, up=2 3x3, 1x1 no-demod; batch 2; grads wrt x, weight, styles
    for tag, (ic, oc, k, up, demod, hw) in dict(c1=(8, 12, 3, 1, True, 6), c0=(8, 12, 3, 2, True, 5),
                                                 rgb=(8, 5, 1, 1, False, 6)).items():
        x = torch.randn(2, ic, hw, hw, generator=g).requires_grad_(True)
        w = torch.randn(oc, ic, k, k, generator=g).requires_grad_(True)
        s = (1 + 0.3 * torch.randn(2, ic, generator=g)).requires_grad_(True)
        res = hw * up
        noise = torch.randn(res, res, generator=g) * 0.1 if demod else None
        y = modulated_conv2d(x=x, weight=w, styles=s, noise=noise, up=up, padding=k // 2, resample_filter=f,
                             demodulate=demod, flip_weight=(up == 1), fused_modconv=True)
        dym = torch.randn(y.shape, generator=g)
        gx, gw, gs = torch.autograd.grad(y, [x, w, s], dym)
        for nm, v in dict(x=x, w=w, s=s, y=y, dy=dym, gx=gx, gw=gw, gs=gs).items():
            out[f'mc_{tag}_{nm}'] = v
        if noise is not None:
            out[f'mc_{tag}_noise'] = noise
        diff(f'modulated_conv2d[{tag}]', y, osg.modulated_conv2d(x, w, s, noise=noise, up=up, padding=k // 2, f=f,
                                                                 demodulate=demod, flip_weight=(up == 1)))
    save('ops', **out)


def sec_renderer():
    from training.volumetric_rendering.renderer import ImportanceRenderer, sample_from_planes
    from training.volumetric_rendering.ray_marcher import MipRayMarcher2
    from training.volumetric_rendering.ray_sampler import RaySampler
    from training.triplane import OSGDecoder
    sys.path.insert(0, ROOT)
    from spi_amd.utils import camera_utils as cu
    g = torch.Generator().manual_seed(5)
    out = {}
    # rays
    c = torch.cat([cu.cal_canonical_c(0.4, -0.1), cu.cal_canonical_c(-0.3, 0.2)], 0)
    c[1, 17] = 0.03     # exercise the skew term
    ro, rd = RaySampler()(c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3), 8)
    out['cam'], out['ray_o'], out['ray_d'] = c, ro, rd
    oo, od = orr.ray_sampler(c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3), 8)
    diff('ray origins', ro, oo)
    diff('ray dirs', rd, od)
    # decoder + planes
    dec = OSGDecoder(32, {'decoder_lr_mul': 1, 'decoder_output_dim': 32})
    dsd = {k: synth_tensor('decoder.' + k, v.shape) for k, v in dec.state_dict().items()}
    dsd['net.0.bias'] = dsd['net.0.bias'] * 3
    dsd['net.2.bias'] = dsd['net.2.bias'] * 3
    dec.load_state_dict(dsd)
    P = {'decoder.' + k: v for k, v in dsd.items()}
    for k, v in P.items():
        out['P_' + k] = v
    planes = torch.randn(2, 3, 32, 16, 16, generator=g)
    out['planes'] = planes
    # gather + decode on arbitrary coords (some outside the box)
    coords = (torch.rand(2, 300, 3, generator=g) - 0.5) * 1.3
    out['coords'] = coords
    pr, dr = planes.clone().requires_grad_(True), [p.requires_grad_(True) for p in dec.parameters()]
    feats = sample_from_planes(ImportanceRenderer().plane_axes, pr, coords, padding_mode='zeros', box_warp=1)
    o = dec(feats, None)
    out['gd_feats'], out['gd_rgb'], out['gd_sigma'] = feats, o['rgb'], o['sigma']
    d_rgb = torch.randn(o['rgb'].shape, generator=g)
    d_sig = torch.randn(o['sigma'].shape, generator=g)
    grads = torch.autograd.grad([o['rgb'], o['sigma']], [pr] + list(dec.parameters()), [d_rgb, d_sig])
    out['gd_drgb'], out['gd_dsigma'], out['gd_gplanes'] = d_rgb, d_sig, grads[0]
    for (k, _), gv in zip(dec.named_parameters(), grads[1:]):
        out['gd_g_' + k] = gv
    f2 = orr.sample_planes(planes, coords)
    diff('sample_planes', feats, f2)
    r2, s2 = orr.osg_decoder(P, f2)
    diff('decoder rgb', o['rgb'], r2)
    diff('decoder sigma', o['sigma'], s2)
    # ray marcher (S = 24 sorted depths), with grads
    n, m, s = 2, 40, 24
    col = torch.rand(n, m, s, 32, generator=g).requires_grad_(True)
    den = (torch.randn(n, m, s, 1, generator=g) * 3 + 1).requires_grad_(True)
    dep = (torch.sort(torch.rand(n, m, s, 1, generator=g) * 1.05 + 2.25, dim=2)[0]).requires_grad_(True)
    den.data[0, 0] = -50.0      # an empty ray: weight_total == 0 -> nan -> inf -> clamp to max depth
    for wb in (False, True):
        rgb, depth, w = MipRayMarcher2()(col, den, dep, {'clamp_mode': 'softplus', 'white_back': wb})
        d1, d2, d3 = (torch.randn(rgb.shape, generator=g), torch.randn(depth.shape, generator=g),
                      torch.randn(w.shape, generator=g))
        gc, gs_, gd_ = torch.autograd.grad([rgb, depth, w], [col, den, dep], [d1, d2, d3])
        t = f'rm{int(wb)}_'
        for nm, v in dict(rgb=rgb, depth=depth, w=w, drgb=d1, ddepth=d2, dw=d3, gcol=gc, gden=gs_, gdep=gd_).items():
            out[t + nm] = v
        a, b_, c_ = orr.ray_march(col, den, dep, white_back=wb)
        diff(f'ray_march rgb wb={wb}', rgb, a)
        diff(f'ray_march depth wb={wb}', depth, b_)
        diff(f'ray_march weights wb={wb}', w, c_)
    out['rm_col'], out['rm_den'], out['rm_dep'] = col, den, dep
    # importance sampling with recorded u
    R = ImportanceRenderer()
    wts = torch.rand(n, m, s - 1, 1, generator=g) ** 4
    wts[0, 1] = 0.0
    with Recorder() as rec:
        torch.manual_seed(3)
        fine = R.sample_importance(dep.detach(), wts, 20)
    out['is_w'], out['is_u'], out['is_fine'] = wts, rec.draws[0], fine
    diff('importance depths', fine, orr.importance_depths(dep.detach(), wts, 20, u=rec.draws[0]))
    # full renderer forward + grads (planes & decoder), recorded draws
    opts = dict(RK)
    n, res = 2, 8
    ro, rd = RaySampler()(c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3), res)
    pr = planes.clone().requires_grad_(True)
    for p in dec.parameters():
        p.grad = None
    with Recorder() as rec:
        torch.manual_seed(7)
        rgb, depth, wsum = R(pr, dec, ro, rd, opts)
    xi, u = rec.draws
    d1, d2 = torch.randn(rgb.shape, generator=g), torch.randn(depth.shape, generator=g)
    grads = torch.autograd.grad([rgb, depth], [pr] + list(dec.parameters()), [d1, d2])
    out['fr_xi'], out['fr_u'], out['fr_rgb'], out['fr_depth'], out['fr_wsum'] = xi, u, rgb, depth, wsum
    out['fr_drgb'], out['fr_ddepth'], out['fr_gplanes'] = d1, d2, grads[0]
    for (k, _), gv in zip(dec.named_parameters(), grads[1:]):
        out['fr_g_' + k] = gv
    a, b_, c_ = orr.render(P, planes, ro, rd, opts, xi=xi, u=u)
    diff('render rgb', rgb, a)
    diff('render depth', depth, b_)
    diff('render weight sum', wsum, c_)
    save('renderer', **out)


def tiny_decoder_weights():
    """A decoder that is NOT the OSG MLP (ImportanceRenderer takes any callable, renderer.py:88,142-148): uses the ray directions, 4 colours."""
    return dict(A=synth_tensor('tiny.A', (32, 16)) * 0.3, B=synth_tensor('tiny.B', (3, 16)), C=synth_tensor('tiny.C', (16, 5)))


def tiny_decoder(Wt, feats, dirs):
    h = torch.tanh(feats.mean(1) @ Wt['A'] + dirs @ Wt['B'])
    y = h @ Wt['C']
    return torch.sigmoid(y[..., 1:]), y[..., 0:1]


def sec_renderer_options():
    """ImportanceRenderer options off the SPI path (VERDICT r02 missing #3): 'auto' ray limits, disparity-space sampling, density noise,
    a decoder callable that is not the OSG MLP.  Reference forward + plane gradients with every draw recorded; oracle pinned."""
    from training.volumetric_rendering.renderer import ImportanceRenderer
    from training.volumetric_rendering.ray_sampler import RaySampler
    from training.triplane import OSGDecoder
    sys.path.insert(0, ROOT)
    from spi_amd.utils import camera_utils as cu
    g = torch.Generator().manual_seed(11)
    out = {}
    c = torch.cat([cu.cal_canonical_c(0.4, -0.1), cu.cal_canonical_c(-0.3, 0.2)], 0)
    ro, rd = RaySampler()(c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3), 8)
    dec = OSGDecoder(32, {'decoder_lr_mul': 1, 'decoder_output_dim': 32})
    dsd = {k: synth_tensor('decoder.' + k, v.shape) for k, v in dec.state_dict().items()}
    dsd['net.0.bias'] = dsd['net.0.bias'] * 3
    dsd['net.2.bias'] = dsd['net.2.bias'] * 3
    dec.load_state_dict(dsd)
    P = {'decoder.' + k: v for k, v in dsd.items()}
    planes = torch.randn(2, 3, 32, 16, 16, generator=g)
    out.update(cam=c, ray_o=ro, ray_d=rd, planes=planes, **{'P_' + k: v for k, v in P.items()})
    Wt = tiny_decoder_weights()
    out.update({'tiny_' + k: v for k, v in Wt.items()})

    class Tiny(torch.nn.Module):
        def forward(self, sampled_features, ray_directions):
            rgb, sigma = tiny_decoder(Wt, sampled_features, ray_directions)
            return {'rgb': rgb, 'sigma': sigma}
    R = ImportanceRenderer()
    variants = dict(
        auto=(dict(RK, ray_start='auto', ray_end='auto', box_warp=0.45), dec),          # 0.45: the outer rays of the 8x8 image miss the box
        disparity=(dict(RK, disparity_space_sampling=True), dec),
        dnoise=(dict(RK, density_noise=0.7), dec),
        tiny=(dict(RK), Tiny()),
        tiny_auto_dnoise=(dict(RK, ray_start='auto', ray_end='auto', box_warp=0.45, density_noise=0.3), Tiny()))
    for tag, (opts, d) in variants.items():
        pr = planes.clone().requires_grad_(True)
        with Recorder() as rec:
            torch.manual_seed(21)
            rgb, depth, wsum = R(pr, d, ro, rd, opts)
        d1, d2 = torch.randn(rgb.shape, generator=g), torch.randn(depth.shape, generator=g)
        gp, = torch.autograd.grad([rgb, depth], [pr], [d1, d2])
        draws = rec.draws
        if opts.get('density_noise', 0) > 0:
            xi, e0, u, e1 = draws
            out[tag + '_eps0'], out[tag + '_eps1'] = e0, e1
        else:
            (xi, u), e0, e1 = draws, None, None
        out.update({tag + '_xi': xi, tag + '_u': u, tag + '_rgb': rgb, tag + '_depth': depth, tag + '_wsum': wsum,
                    tag + '_drgb': d1, tag + '_ddepth': d2, tag + '_gplanes': gp})
        if tag.startswith('auto'):
            from training.volumetric_rendering import math_utils
            s0, e0_ = math_utils.get_ray_limits_box(ro, rd, box_side_length=opts['box_warp'])
            out['auto_miss_fraction'] = (e0_ <= s0).float().mean()
            print(f'    auto: {float(out["auto_miss_fraction"]) * 100:.0f} % of the rays miss the box')
        fn = (lambda f, dd: tiny_decoder(Wt, f, dd)) if tag.startswith('tiny') else None
        oopts = {k: v for k, v in opts.items() if k in ('depth_resolution', 'depth_resolution_importance', 'ray_start', 'ray_end', 'box_warp',
                                                        'white_back', 'disparity_space_sampling', 'density_noise')}
        a, b_, c_ = orr.render(P, planes, ro, rd, oopts, xi=xi, u=u, eps=(e0, e1), decoder_fn=fn)
        diff(f'{tag}: render rgb', rgb, a)
        diff(f'{tag}: render depth', depth, b_)
        diff(f'{tag}: render weight sum', wsum, c_)
    save('renderer_options', **out)


def sec_synthesis():
    """Narrow generator (2.84 M params, rendering res 32, 12+12 samples): outputs + grad wrt ws."""
    sys.path.insert(0, ROOT)
    from spi_amd.utils import camera_utils as cu
    G = build_ref_generator(True)
    G.neural_rendering_resolution = 32
    g = torch.Generator().manual_seed(9)
    ws = (torch.randn(2, 14, 512, generator=g)).requires_grad_(True)
    c = torch.cat([cu.cal_canonical_c(0.4, 0.0), cu.cal_mirror_c(cu.cal_canonical_c(0.4, 0.0))], 0)
    with Recorder() as rec:
        torch.manual_seed(0)
        o = G.synthesis(ws, c, noise_mode='const')
    xi, u = rec.draws
    planes = G.backbone.synthesis(ws, noise_mode='const')
    d_img = torch.randn(o['image'].shape, generator=g)
    d_dep = torch.randn(o['image_depth'].shape, generator=g)
    loss = (o['image'] * d_img).sum() / 1000 + (o['image_depth'] * d_dep).sum()
    gws, = torch.autograd.grad(loss, ws)
    P = {k: v for k, v in G.state_dict().items()}
    opts = dict(RK)
    wso = ws.detach().clone().requires_grad_(True)
    oo = orr.synthesis(P, wso, c, opts, neural_rendering_resolution=32, xi=xi, u=u)
    lo = (oo['image'] * d_img).sum() / 1000 + (oo['image_depth'] * d_dep).sum()
    gwo, = torch.autograd.grad(lo, wso)
    diff('synthesis planes', planes, oo['planes'].reshape(planes.shape))
    diff('synthesis image', o['image'], oo['image'])
    diff('synthesis image_raw', o['image_raw'], oo['image_raw'])
    diff('synthesis image_depth', o['image_depth'], oo['image_depth'])
    diff('synthesis grad ws', gws, gwo)
    z = torch.randn(4, 512, generator=g)
    wmap = G.mapping(z, c[:1].repeat(4, 1))
    diff('mapping', wmap, osg.mapping(P, z, c[:1].repeat(4, 1) * 1.0))
    save('synthesis_narrow', ws=ws, c=c, xi=xi, u=u, planes_sub=planes[:, ::7, ::5, ::5], image_sub=o['image'][:, :, ::8, ::8],
         image_raw=o['image_raw'], image_depth=o['image_depth'], d_img_seed=np.array([9]), d_img=d_img[:, :, ::8, ::8],
         d_dep=d_dep, gws=gws, image_mean=o['image'].mean(), image_absmean=o['image'].abs().mean(),
         map_z=z, map_w=wmap[:, 0])


def sec_geometry():
    import spi.utils.camera_utils as rcu
    rcu.GAUSS_CONST = torch.sqrt(torch.tensor(2 * torch.pi))
    from spi.utils.rotate import rotate as rrotate
    from spi.utils.mask_utils import calculate_face_mask
    sys.path.insert(0, ROOT)
    from spi_amd.utils import camera_utils as cu
    torch.Tensor.cuda = lambda self, *a, **k: self       # rotate.py hard-codes .cuda()
    g = torch.Generator().manual_seed(21)
    out = {}
    cams = []
    for i, (yw, pt) in enumerate(((0.0, 0.0), (0.4, 0.0), (-0.55, 0.2), (0.1, -0.3))):
        cr = rcu.cal_canonical_c(yw, pt)
        cams.append(cr)
        diff(f'cal_canonical_c[{i}]', cr, cu.cal_canonical_c(yw, pt))
    cams = torch.cat(cams, 0)
    out['canon'] = cams
    out['mirror'] = rcu.cal_mirror_c(cams)
    diff('cal_mirror_c', out['mirror'], cu.cal_mirror_c(cams))
    out['weight'] = rcu.cal_camera_weight(cams)
    diff('cal_camera_weight', out['weight'], cu.cal_camera_weight(cams))
    out['weight_m'] = rcu.cal_camera_weight(out['mirror'])
    out['gauss_weight'] = torch.stack(rcu.cal_camera_gauss_weight(cams))
    diff('cal_camera_gauss_weight', out['gauss_weight'], torch.stack(cu.cal_camera_gauss_weight(cams)))
    with Recorder() as rec:
        torch.manual_seed(4)
        sur = rcu.sample_surrounding_camera(cams[1:2].clone(), batch_size=4, yaw_range=0.2, pitch_range=0.1)
    out['sur'], out['sur_r0'], out['sur_r1'] = sur, rec.draws[0], rec.draws[1]
    diff('sample_surrounding_camera', sur, cu.sample_surrounding_camera(cams[1:2], 4, 0.2, 0.1, rand=rec.draws))
    with Recorder() as rec:
        torch.manual_seed(5)
        sc = rcu.sample_camera(batch_size=4, yaw_range=0.7, pitch_range=0.4)
    out['sc'], out['sc_r0'], out['sc_r1'] = sc, rec.draws[0], rec.draws[1]
    diff('sample_camera', sc, cu.sample_camera(4, 0.7, 0.4, rand=rec.draws))
    # face mask
    parsing = torch.randint(0, 19, (1, 1, 32, 32), generator=g)
    out['parsing'], out['face_mask'] = parsing, calculate_face_mask(parsing)
    # rotate: smooth synthetic depth maps at 128^2, images at 64^2 (resolution = src_image.shape[-1])
    n, res = 2, 64
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, 128), torch.linspace(-1, 1, 128), indexing='ij')
    bump = torch.exp(-(xx ** 2 + yy ** 2) * 2)
    tdepth = (2.7 - 0.25 * bump)[None, None].repeat(n, 1, 1, 1) + 0.002 * torch.randn(n, 1, 128, 128, generator=g)
    sdepth = (2.7 - 0.25 * bump)[None, None].repeat(n, 1, 1, 1) + 0.002 * torch.randn(n, 1, 128, 128, generator=g)
    img = torch.rand(n, 3, res, res, generator=g) * 2 - 1
    msk = (torch.rand(n, 1, res, res, generator=g) > 0.3).float()
    src_cam = cams[1:2].repeat(n, 1)
    rgb, m = rrotate(target_camera=sur[:n], target_depth=tdepth, src_image=img, src_camera=src_cam, src_depth=sdepth,
                     src_mask=msk, EPS=5e-2)
    out.update(rot_tdepth=tdepth, rot_sdepth=sdepth, rot_img=img, rot_msk=msk, rot_rgb=rgb, rot_mask=m)
    r2, m2 = olo.rotate(sur[:n], tdepth, img, src_cam, sdepth, msk, EPS=5e-2)
    diff('rotate rgb', rgb, r2)
    diff('rotate mask', m, m2)
    print(f'    rotate mask coverage = {m.mean().item():.3f}')
    save('geometry', **out)


def sec_schedule():
    """Stage-1 lr / w-noise schedules (mirror_projector.py:84-91) restated as the host math the reference runs."""
    out = {}
    for num_steps in (10, 500):
        lrs, ns = [], []
        for step in range(num_steps):
            t = step / num_steps
            ns.append(0.05 * max(0.0, 1.0 - t / 0.75) ** 2)
            r = min(1.0, (1.0 - t) / 0.25)
            r = 0.5 - 0.5 * np.cos(r * np.pi)
            r = r * min(1.0, t / 0.05)
            lrs.append(0.01 * r)
        out[f'lr_{num_steps}'] = np.array(lrs)
        out[f'noise_{num_steps}'] = np.array(ns)
    save('schedule', **out)


def sec_trajectory():
    """3 steps of the reference's mirror_projector.project and w_plus_projector.project on the narrow
    generator, with a seeded-weight LPIPS (oracle's) injected as lpips_func."""
    import spi.utils.camera_utils as rcu
    rcu.GAUSS_CONST = torch.sqrt(torch.tensor(2 * torch.pi))
    from spi.configs import global_config, paths_config
    global_config.device = 'cpu'
    import spi.training.projectors.mirror_projector as mp
    import spi.training.projectors.w_plus_projector as wp
    mp.log_image = lambda *a, **k: None
    wp.log_image = lambda *a, **k: None
    mp.tqdm = lambda x: x
    wp.tqdm = lambda x: x
    sys.path.insert(0, ROOT)
    from spi_amd.utils import camera_utils as cu
    G = build_ref_generator(True)
    G.neural_rendering_resolution = 128   # mirror_projector.py:74 hard-codes a 128^2 bg mask
    W = olo.make_vgg16_weights(seed=0)
    lp = lambda a, b: olo.lpips(W, a, b)
    g = torch.Generator().manual_seed(31)
    target = torch.rand(1, 3, 512, 512, generator=g) * 2 - 1
    c = cu.cal_canonical_c(0.4, 0.0)
    fg = torch.zeros(1, 1, 512, 512)
    fg[:, :, 100:400, 120:380] = 1
    steps_log = []
    orig_step = torch.optim.Adam.step

    def logging_step(self, *a, **k):
        r = orig_step(self, *a, **k)
        steps_log.append(self.param_groups[0]['params'][0].detach().clone())
        return r
    torch.optim.Adam.step = logging_step
    try:
        torch.manual_seed(0)
        np.random.seed(0)
        w = mp.project(G, target, c, lp, fg, num_steps=3, w_avg_samples=64, device=torch.device('cpu'), w_name='t')
        w_mir = torch.stack(steps_log)
        steps_log.clear()
        torch.manual_seed(0)
        np.random.seed(0)
        w2 = wp.project(G, target, c, lp, num_steps=3, w_avg_samples=64, device=torch.device('cpu'), w_name='t')
        w_plus = torch.stack(steps_log)
    finally:
        torch.optim.Adam.step = orig_step
    # pin the oracle loops against the reference trajectories
    from oracle import loops_ref as olp
    P = {k: v for k, v in G.state_dict().items()}
    opts = dict(RK)
    for mirror, ref_w in ((True, w_mir), (False, w_plus)):
        torch.manual_seed(0)
        np.random.seed(0)
        log = []
        olp.project_w_plus(P, target, c, lp, opts, mirror=mirror, num_steps=3, w_avg_samples=64, nrr=128, log=log)
        diff(f'stage-1 trajectory mirror={mirror}', ref_w, torch.stack([l['w'] for l in log]))
    save('trajectory', target_seed=np.array([31]), c=c, w_mir=w_mir[:, 0], w_mir_final=w[0], w_plus=w_plus[:, 0],
         w_plus_final=w2[0])

def sec_trajectory_sg():
    """3 steps of the reference's w_projector.project (`first_inv_type=sg`, BASELINE configs[0]/[2]) on the narrow generator with
    a seeded stand-in for NVIDIA's vgg16.pt (oracle.losses_ref.sg_vgg_features) injected as `vgg16`; w and dL/dw per step."""
    from spi.configs import global_config
    global_config.device = 'cpu'
    import spi.training.projectors.w_projector as wproj
    wproj.tqdm = lambda x: x
    sys.path.insert(0, ROOT)
    from spi_amd.utils import camera_utils as cu
    G = build_ref_generator(True)
    G.neural_rendering_resolution = 64
    W = olo.make_vgg16_weights(seed=0)
    vgg = lambda img, resize_images=False, return_lpips=True: olo.sg_vgg_features(W, img, resize_images, return_lpips)
    g = torch.Generator().manual_seed(41)
    target = torch.rand(1, 3, 512, 512, generator=g) * 2 - 1
    c = cu.cal_canonical_c(-0.3, 0.05)
    w_log, g_log = [], []
    orig_step = torch.optim.Adam.step

    def logging_step(self, *a, **k):
        g_log.append(self.param_groups[0]['params'][0].grad.detach().clone())
        r = orig_step(self, *a, **k)
        w_log.append(self.param_groups[0]['params'][0].detach().clone())
        return r
    torch.optim.Adam.step = logging_step
    try:
        torch.manual_seed(0)
        np.random.seed(0)
        w = wproj.project(G, target, c, vgg, num_steps=3, w_avg_samples=64, device=torch.device('cpu'), w_name='t')
    finally:
        torch.optim.Adam.step = orig_step
    from oracle import loops_ref as olp
    P = {k: v for k, v in G.state_dict().items()}
    torch.manual_seed(0)
    np.random.seed(0)
    log = []
    w2 = olp.project_w(P, target, c, vgg, dict(RK), num_steps=3, w_avg_samples=64, nrr=64, log=log)
    diff('sg trajectory w', torch.stack(w_log), torch.stack([l['w'] for l in log]))
    diff('sg trajectory dL/dw', torch.stack(g_log), torch.stack([l['grad_w'] for l in log]))
    diff('sg final w [1,14,512]', w, w2)
    save('trajectory_sg', target_seed=np.array([41]), c=c, w_sg=torch.stack(w_log)[:, 0], gw_sg=torch.stack(g_log)[:, 0], w_sg_final=w[0])


def sec_tv():
    """TriPlaneGenerator.sample_mixed (triplane.py:98-102) and cal_tv_loss (spi/criteria/tv_loss.py:9-19) on the
    narrow generator: densities / colours at explicit coordinates, the TV value and its gradient wrt W+, with the draws recorded."""
    from spi.criteria.tv_loss import cal_tv_loss
    G = build_ref_generator(True)
    g = torch.Generator().manual_seed(51)
    ws = torch.randn(1, 14, 512, generator=g)
    coords = torch.rand(1, 3000, 3, generator=g) * 1.2 - 0.6          # some points outside the box (zero-padded gather)
    with torch.no_grad():
        sm = G.sample_mixed(coords, torch.randn(1, 3000, 3, generator=g), ws, noise_mode='const')
    wr = ws.clone().requires_grad_(True)
    with Recorder(randn=True) as rec:
        torch.manual_seed(7)
        tv = cal_tv_loss(wr, G)                                       # NB: backbone noise_mode defaults to 'random' here (tv_loss.py:14)
    gw, = torch.autograd.grad(tv, wr)
    # rand coords, randn perturbation, randn directions (unused by the decoder), then one [1,1,res,res] noise draw per synthesis layer
    assert len(rec.draws) == 3 + 13, [d.shape for d in rec.draws]
    # pin the oracle's restatement
    P = {k: v for k, v in G.state_dict().items()}
    wo = ws.clone().requires_grad_(True)

    class Replay:
        def __init__(self, d):
            self.d = list(d)

        def randn(self, *shape):
            t = self.d.pop(0)
            assert tuple(t.shape) == tuple(shape)
            return t
    planes = osg.backbone_synthesis(P, wo, noise_mode='random', noise_rng=Replay(rec.draws[3:])).reshape(1, 3, 32, 256, 256)
    init = rec.draws[0] * 2 - 1
    allc = torch.cat([init, init + rec.draws[1] * 0.004], 1)
    rgb_o, sig_o = orr.run_model(P, planes, allc, dict(RK))
    tv_o = F.l1_loss(sig_o[:, :1000], sig_o[:, 1000:])
    diff('cal_tv_loss', tv, tv_o)
    diff('cal_tv_loss dL/dws', gw, torch.autograd.grad(tv_o, wo)[0])
    rgb2, sig2 = orr.run_model(P, osg.backbone_synthesis(P, ws).reshape(1, 3, 32, 256, 256), coords, dict(RK))
    diff('sample_mixed sigma', sm['sigma'], sig2)
    diff('sample_mixed rgb', sm['rgb'], rgb2)
    save('tv', ws=ws, coords=coords, sigma=sm['sigma'], rgb=sm['rgb'], tv=tv, gws=gw,
         **{f'r{i}': d for i, d in enumerate(rec.draws)})


def sec_orbit():
    """the camera path of spi/utils/video_utils.py:155-160 (that module itself needs imageio / mrcfile and cannot be
    imported): the reference's own LookAtPoseSampler (eg3d/camera_utils.py) driven with the orbit expressions of those lines, plus the
    query grid of create_samples (:41-70), restated from the cited lines with the reference's numpy / torch calls."""
    from camera_utils import LookAtPoseSampler
    sys.path.insert(0, ROOT)
    from spi_amd.utils import video_utils as vu
    out = {}
    for frames in (120, 7):
        poses = []
        for frame_idx in range(frames):
            pose = LookAtPoseSampler.sample(3.14 / 2 + 0.7 * np.sin(2 * 3.14 * frame_idx / frames),
                                            3.14 / 2 - 0.05 + 0.4 * np.cos(2 * 3.14 * frame_idx / frames),
                                            torch.tensor([0, 0, 0.2]), radius=2.7, device='cpu')
            poses.append(pose.reshape(16))
        intr = torch.tensor([[4.2647, 0, 0.5], [0, 4.2647, 0.5], [0, 0, 1]]).reshape(1, 9).repeat(frames, 1)
        cams = torch.cat([torch.stack(poses), intr], 1)
        out[f'cams_{frames}'] = cams
        diff(f'orbit cameras ({frames} frames)', cams, vu.orbit_cameras(frames))
    N, cube = 6, 1.0
    origin = np.array([0, 0, 0]) - cube / 2
    vs = cube / (N - 1)
    idx = torch.arange(0, N ** 3, 1, out=torch.LongTensor())
    smp = torch.zeros(N ** 3, 3)
    smp[:, 2] = idx % N
    smp[:, 1] = (idx.float() / N) % N
    smp[:, 0] = ((idx.float() / N) / N) % N
    smp[:, 0] = (smp[:, 0] * vs) + origin[2]
    smp[:, 1] = (smp[:, 1] * vs) + origin[1]
    smp[:, 2] = (smp[:, 2] * vs) + origin[0]
    out['samples_6'] = smp
    print('    (create_samples in the reference uses float division: columns 0/1 are fractional positions, kept as data)')
    save('orbit', **out)

def mesh_stats(verts, faces):
    """vertex / face count, area, signed volume, Euler characteristic and closedness of a triangle mesh (numpy)."""
    p = verts[faces]
    nrm = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
    e = np.sort(np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]]), axis=1)
    ue, cnt = np.unique(e[:, 0] * (len(verts) + 1) + e[:, 1], return_counts=True)
    return dict(verts=len(verts), faces=len(faces), area=0.5 * np.linalg.norm(nrm, axis=1).sum(),
                volume=np.einsum('ij,ij->i', p[:, 0], nrm).sum() / 6, euler=len(verts) - len(ue) + len(faces), closed=bool((cnt == 2).all()))


def sec_orbit_frames():
    """SURVEY 8f-1, what BaseCoach.post_process produces (spi/utils/video_utils.py:74-230 -- the module needs imageio / mrcfile and
    cannot be imported; its loop body is driven here with the reference's own generator): frames 0, 37 and 119 of the 120-frame
    orbit = G.synthesis(w, c_k, noise_mode='const') (:170-172) on the narrow reference generator with the renderer's draws recorded,
    and the density grid of :185-207 = G.sample_mixed on create_samples' points, flipped and border-cleaned, at 32^3.  The iso-surface
    itself comes from skimage in the reference (absent here): the grid is the pinned quantity, the mesh statistics below are those of
    the build's marching cubes on the REFERENCE's grid (regression values + topological invariants)."""
    sys.path.insert(0, ROOT)
    from spi_amd.utils import video_utils as vu, shape_utils as su
    G = build_ref_generator(True)
    G.neural_rendering_resolution = 32
    g = torch.Generator().manual_seed(77)
    ws = torch.randn(1, 14, 512, generator=g) * 0.8
    cams = torch.from_numpy(np.load(os.path.join(HERE, 'orbit.npz'))['cams_120'])
    out = dict(ws=ws, frame_ids=np.array([0, 37, 119]))
    P = {k: v for k, v in G.state_dict().items()}
    with torch.no_grad():
        for k in (0, 37, 119):
            with Recorder() as rec:
                torch.manual_seed(0)
                o = G.synthesis(ws=ws, c=cams[k:k + 1], noise_mode='const')
            xi, u = rec.draws
            oo = orr.synthesis(P, ws, cams[k:k + 1], dict(RK), neural_rendering_resolution=32, xi=xi, u=u)
            for nm in ('image', 'image_raw', 'image_depth'):
                diff(f'frame {k}: {nm}', o[nm], oo[nm])
            out.update({f'f{k}_xi': xi, f'f{k}_u': u, f'f{k}_image_sub': o['image'][:, :, ::4, ::4], f'f{k}_image_raw': o['image_raw'],
                        f'f{k}_image_depth': o['image_depth'], f'f{k}_image_mean': o['image'].mean(), f'f{k}_image_absmean': o['image'].abs().mean()})
        N = 32
        samples, _, _ = vu.create_samples(N=N, voxel_origin=[0, 0, 0], cube_length=G.rendering_kwargs['box_warp'])   # (pinned by orbit.npz)
        dirs = torch.zeros(1, samples.shape[1], 3)
        dirs[..., -1] = -1
        torch.manual_seed(0)
        sigma = G.sample_mixed(samples, dirs, ws, truncation_psi=1, noise_mode='const')['sigma']
        _, so = orr.run_model(P, osg.backbone_synthesis(P, ws).reshape(1, 3, 32, 256, 256), samples, dict(RK))
        diff('sigma grid (sample_mixed on create_samples)', sigma, so)
    raw = sigma.reshape(N, N, N).numpy()
    sig = np.flip(raw, 0).copy()                                  # video_utils.py:198-207
    pad, pad_top = int(30 * N / 256), int(38 * N / 256)
    sig[:pad] = 0; sig[-pad:] = 0; sig[:, :pad] = 0; sig[:, -pad_top:] = 0; sig[:, :, :pad] = 0; sig[:, :, -pad:] = 0
    level = float(np.percentile(sig[sig != 0], 60))               # a level that cuts this random-init field (the reference uses 10 on trained weights)
    verts, faces = su.marching_cubes(np.transpose(sig, (2, 1, 0)), level=level)
    st = mesh_stats(verts, faces)
    print('    sigma grid mesh at level %.4f:' % level, st)
    out.update(sigma_raw=raw, sigma_grid=sig, mesh_level=np.array(level), mesh_verts=np.array(st['verts']), mesh_faces=np.array(st['faces']),
               mesh_area=np.array(st['area']), mesh_volume=np.array(st['volume']), mesh_euler=np.array(st['euler']), mesh_closed=np.array(st['closed']))
    save('orbit_frames', **out)


def sec_bisenet():
    """the reference's own BiSeNet (third_part/bisenet) on seeded synthetic weights (no bisenet.pth offline), eval mode.
    Two import-time obstacles, both outside the arithmetic: `import torchvision` at the top of bisenet.py (never used in the file; the
    package does not exist here -> an empty placeholder module is registered for the import) and Resnet18.init_weight's download of the
    ImageNet weights (resnet.py:79-85; no network -> model_zoo.load_url is pointed at an empty dict, every weight is overwritten anyway)."""
    import types
    import torch.utils.model_zoo as modelzoo
    sys.modules.setdefault('torchvision', types.ModuleType('torchvision'))
    orig = modelzoo.load_url
    modelzoo.load_url = lambda *a, **k: {}
    try:
        from third_part.bisenet.bisenet import BiSeNet
        net = BiSeNet(19)
    finally:
        modelzoo.load_url = orig
    from oracle import bisenet_ref as obr
    man = {k: list(v.shape) for k, v in net.state_dict().items()}
    with open(os.path.join(HERE, 'manifest_bisenet.json'), 'w') as f:
        json.dump(man, f, indent=0)
    sd = obr.synthetic_state_dict(man, seed=0)
    net.load_state_dict(sd)
    net.eval()
    g = torch.Generator().manual_seed(61)
    # a smooth synthetic "photo" (low-pass noise) in [-1, 1] like extract_mask.py:58-59 feeds it, plus a second, non-square-friendly size
    img = F.interpolate(torch.rand(1, 3, 24, 24, generator=g), size=(512, 512), mode='bicubic', align_corners=False).clamp(0, 1) * 2 - 1
    img2 = torch.rand(2, 3, 96, 160, generator=torch.Generator().manual_seed(62)) * 2 - 1
    with torch.no_grad():
        out, out16, out32 = net(img)
        o2 = net(img2)[0]
        ref = obr.bisenet_forward(sd, img)
        ref2 = obr.bisenet_forward(sd, img2)[0]
    for tag, a, b in (('out', out, ref[0]), ('out16', out16, ref[1]), ('out32', out32, ref[2]), ('out (2x3x96x160)', o2, ref2)):
        diff('bisenet ' + tag, a, b)
    parsing = torch.argmax(out, dim=1, keepdim=True)
    top2 = out.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1])
    print(f'    classes present: {parsing.unique().tolist()}; pixels with a top-2 margin < 1e-3: {(margin < 1e-3).float().mean().item():.2e}')
    save('bisenet', seed=np.array([0]), img_seed=np.array([61]), out_sub=out[:, :, ::8, ::8], out16_sub=out16[:, :, ::16, ::16],
         out32_sub=out32[:, :, ::16, ::16], parsing=parsing.to(torch.uint8), margin_sub=margin[:, ::8, ::8], out2_sub=o2[:, :, ::4, ::4], out2_mean=o2.mean(dim=(2, 3)),
         out_mean=out.mean(dim=(2, 3)), out_absmax=out.abs().amax())


def sec_recon():
    """SURVEY 8f-4, crop + camera producer: the reference's 3DMM regressor (third_part/Deep3DFaceRecon_pytorch/models/networks.py:
    ReconNetWrapper('resnet50')) on seeded synthetic weights (the trained epoch_20.pth is not available offline), eval mode, and
    preprocess/process_camera.py.  One import-time obstacle outside the arithmetic: networks.py does `from kornia.geometry import warp_affine`
    for RecogNetWrapper.resize_n_crop, which this path never calls; the package does not exist here -> an empty placeholder module is
    registered for the import (the same treatment as `torchvision` in sec_bisenet).  preprocess/extract_3dmm.py / extract_camera.py are NOT
    importable here (face_alignment, cv2, torchvision; the detector is instantiated at import): align_img / cal_camera stay unpinned by
    import, oracle/recon_ref.py says so."""
    import types
    kg = types.ModuleType('kornia.geometry')
    kg.warp_affine = None
    sys.modules.setdefault('kornia', types.ModuleType('kornia'))
    sys.modules.setdefault('kornia.geometry', kg)
    from third_part.Deep3DFaceRecon_pytorch.models.networks import ReconNetWrapper
    from preprocess.process_camera import process_camera as ref_process_camera
    from oracle import recon_ref as orr2
    net = ReconNetWrapper('resnet50', use_last_fc=False)
    man = {k: list(v.shape) for k, v in net.state_dict().items()}
    with open(os.path.join(HERE, 'manifest_recon.json'), 'w') as f:
        json.dump(man, f, indent=0)
    sd = orr2.synthetic_state_dict(man, seed=0)
    net.load_state_dict(sd)
    net.eval()
    g = torch.Generator().manual_seed(71)
    img = F.interpolate(torch.rand(2, 3, 28, 28, generator=g), size=(224, 224), mode='bicubic', align_corners=False).clamp(0, 1)
    with torch.no_grad():
        out = net(img)
        ref = orr2.recon_net(sd, img)
    diff('recon coefficients', out, ref)
    print('    coefficient scale: |angle| %.3f |trans| %.3f' % (out[:, 224:227].abs().max(), out[:, 254:].abs().max()))
    # camera label from a pose / intrinsics pair
    rng = np.random.default_rng(5)
    pose = np.eye(4)
    pose[:3, :3] = np.linalg.qr(rng.standard_normal((3, 3)))[0]
    pose[:3, 3] = rng.standard_normal(3)
    K = np.eye(3)
    K[0, 0] = K[1, 1] = 2985.29
    K[0, 2] = K[1, 2] = 512.0
    cam = ref_process_camera(pose.tolist(), K.tolist())
    mine = orr2.process_camera(pose.tolist(), K.tolist())
    print('    pin process_camera: max|ref-oracle| = %.3e' % np.abs(cam - mine).max())
    assert np.array_equal(cam, mine)
    save('recon', seed=np.array([0]), img_seed=np.array([71]), coeffs=out, pose=pose, K=K, camera=cam)



# ------------------------------------------------------------------------------------------------
# Sections that need the reference's loss / coach modules.  Those import `torchvision` (absent here), hard-code `.to("cuda")` / `.cuda()`
# and download weights; none of that is arithmetic of the path.  `reference_loss_env()` registers a placeholder `torchvision` whose three
# entry points the reference uses return (a) structure-identical VGG16 / VGG19 `features` Sequentials carrying the oracle's SEEDED weights
# (no pretrained weights exist offline) and (b) `ops.roi_align` = the oracle's restatement of torchvision's published definition (the op
# itself is third-party code outside /root/reference: that edge stays "parity unpinned"), maps every "cuda" device request onto the CPU,
# and points the LPIPS lin-layer download at the seeded lin weights.  Everything else -- LPIPS.forward, BoxCXLoss.forward, get_bbox,
# get_landmark_bbox, compute_cosine_distance / relative_distance / cx, RotBboxCoach.train, SingleIDCoach.train, BaseCoach -- is the
# reference's own code, executed.

def _vgg16_features_module(W):
    layers, ci = [], 0
    cin = 3
    for v in olo.VGG16_CFG:
        if v == 'M':
            layers.append(torch.nn.MaxPool2d(kernel_size=2, stride=2))
            continue
        conv = torch.nn.Conv2d(cin, v, 3, padding=1)
        conv.weight.data.copy_(W['convs'][ci][0])
        conv.bias.data.copy_(W['convs'][ci][1])
        layers += [conv, torch.nn.ReLU(inplace=True)]
        cin = v
        ci += 1
    layers.append(torch.nn.MaxPool2d(kernel_size=2, stride=2))      # torchvision's vgg16.features has 31 entries; the last pool is never reached
    return torch.nn.Sequential(*layers)


def _vgg19_features_module(W19):
    layers = []
    for i, (cin, cout) in enumerate(((3, 64), (64, 64), (64, 128))):
        conv = torch.nn.Conv2d(cin, cout, 3, padding=1)
        conv.weight.data.copy_(W19[i][0])
        conv.bias.data.copy_(W19[i][1])
        layers.append(conv)
        if i < 2:
            layers.append(torch.nn.ReLU(inplace=True))
        if i == 1:
            layers.append(torch.nn.MaxPool2d(kernel_size=2, stride=2))
    # features[0:6] = conv relu conv relu pool conv -- all bbox_cx_loss.py:80-82 takes; the remaining 31 entries are never touched
    return torch.nn.Sequential(*layers)


_ENV = {}


def reference_loss_env(seed16=0, seed19=1):
    """Install the placeholders (idempotent).  -> dict(W16, W19)."""
    if _ENV:
        return _ENV
    import types
    W16, W19 = olo.make_vgg16_weights(seed=seed16), olo.make_vgg19_head_weights(seed=seed19)
    tv = types.ModuleType('torchvision')
    tv.ops = types.ModuleType('torchvision.ops')
    tv.models = types.ModuleType('torchvision.models')
    tv.models.vgg = types.ModuleType('torchvision.models.vgg')
    tv.transforms = types.ModuleType('torchvision.transforms')

    def roi_align(input, boxes, output_size, spatial_scale=1.0, sampling_ratio=-1, aligned=False):
        assert spatial_scale == 1.0 and sampling_ratio == -1 and not aligned                   # the defaults bbox_cx_loss.py:50-59 relies on
        assert boxes.shape == (input.shape[0], 5) and torch.equal(boxes[:, 0], torch.arange(input.shape[0]).float())
        return olo.roi_align(input, boxes[:, 1:], out=output_size)
    tv.ops.roi_align = roi_align

    class _Net:
        def __init__(self, features):
            self.features = features
    tv.models.vgg16 = lambda pretrained=False, **k: _Net(_vgg16_features_module(W16))
    tv.models.vgg.vgg19 = lambda pretrained=False, **k: _Net(_vgg19_features_module(W19))
    for name, mod in (('torchvision', tv), ('torchvision.ops', tv.ops), ('torchvision.models', tv.models),
                      ('torchvision.models.vgg', tv.models.vgg), ('torchvision.transforms', tv.transforms)):
        sys.modules[name] = mod
    for name in ('imageio', 'mrcfile', 'wandb'):                                                # imported at module level by video_utils / coaches, unused by train()
        sys.modules.setdefault(name, types.ModuleType(name))
    # "cuda" -> this CPU
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    orig_to = torch.nn.Module.to

    def to(self, *a, **k):
        a = tuple('cpu' if (isinstance(v, str) and v.startswith('cuda')) or (isinstance(v, torch.device) and v.type == 'cuda') else v for v in a)
        if isinstance(k.get('device'), str) and k['device'].startswith('cuda'):
            k['device'] = 'cpu'
        return orig_to(self, *a, **k)
    torch.nn.Module.to = to
    orig_tto = torch.Tensor.to

    def tto(self, *a, **k):
        a = tuple('cpu' if (isinstance(v, str) and v.startswith('cuda')) else v for v in a)
        return orig_tto(self, *a, **k)
    torch.Tensor.to = tto
    import spi.criteria.lpips.lpips as rlp
    lin_sd = {f'{i}.1.weight': w.clone() for i, w in enumerate(W16['lins'])}
    rlp.get_state_dict = lambda net_type='alex', version='0.1': lin_sd                          # lpips/utils.py:13-21 fetches these by URL
    _ENV.update(W16=W16, W19=W19)
    return _ENV


def sec_losses():
    """VERDICT r03 next-#1a: the reference's own LPIPS.forward (spi/criteria/lpips/lpips.py:32-71 + networks.py:53-63,88-96 + utils.py:6-8) and
    BoxCXLoss.forward (+ get_bbox, get_landmark_bbox, compute_cosine_distance / relative_distance / cx; spi/criteria/bbox_cx_loss.py:20-61,
    93-129,141-182) on seeded inputs: values and input gradients, at 512^2 (the bilinear reduction to 256^2 is part of both) and at batch 4
    with a visibility mask multiplied in (the stage-2 `rot` / `mirror-rot` usage, rot_bbox_cx_coach.py:101,127)."""
    env = reference_loss_env()
    from spi.criteria.lpips.lpips import LPIPS
    from spi.criteria.bbox_cx_loss import BoxCXLoss, get_landmark_bbox
    sys.path.insert(0, ROOT)
    from spi_amd.data.images_dataset import synthetic_landmarks
    W16, W19 = env['W16'], env['W19']
    lp = LPIPS(net_type='vgg').to('cuda').eval()
    bx = BoxCXLoss().cuda().eval()
    out = dict(seed16=np.array([0]), seed19=np.array([1]))
    import loss_inputs as li
    g = torch.Generator().manual_seed(91)
    for tag, (x, y, m) in li.lpips_cases().items():
        xr = x.clone().requires_grad_(True)
        val = lp(xr * m if m is not None else xr, y)
        gx, = torch.autograd.grad(val, xr)
        xo = x.clone().requires_grad_(True)
        vo = olo.lpips(W16, xo * m if m is not None else xo, y)
        go, = torch.autograd.grad(vo, xo)
        diff(f'LPIPS[{tag}] value', val, vo)
        diff(f'LPIPS[{tag}] d/dx', gx, go)
        out.update({tag + '_val': val, tag + '_gx_sub': li.grad_sub(gx), tag + '_gx_sum': gx.double().sum(), tag + '_gx_abssum': gx.double().abs().sum(),
                    tag + '_x_sum': x.double().sum(), tag + '_y_sum': y.double().sum()})
    # LPIPS feature taps of the reference's BaseNet.forward (what the build caches for the target image)
    x64 = li.lpips_cases()['lp64'][0]
    with torch.no_grad():
        feats = lp.net(x64)
    for i, f in enumerate(feats):
        out[f'lp64_feat{i}'] = f
        diff(f'LPIPS net tap {i}', f, olo.vgg16_features(W16, x64)[i])
    # BoxCX: boxes differ per batch element; one reaches outside the 256^2 frame so roi_align's out-of-range rule is exercised
    lm = li.landmarks(911, 4)
    bb = get_landmark_bbox(lm)
    for i, b in enumerate(bb):
        out[f'bx_box{i}'] = b
    mine = olo.landmark_boxes(lm)
    for i in range(3):
        assert torch.equal(bb[i].float(), mine[i]), i
    print('    pin get_landmark_bbox: mouth / l_eye / r_eye boxes bit-equal;', [b.tolist()[3] for b in bb[:3]])
    for tag, (x, m, y, l) in li.boxcx_cases().items():
        xr = x.clone().requires_grad_(True)
        val = bx(xr * m if m is not None else xr, y, l)
        gx, = torch.autograd.grad(val, xr)
        xo = x.clone().requires_grad_(True)
        vo = olo.box_cx_loss(W19, xo * m if m is not None else xo, y, l)
        go, = torch.autograd.grad(vo, xo)
        diff(f'BoxCX[{tag}] value', val, vo)
        diff(f'BoxCX[{tag}] d/dx', gx, go)
        out.update({tag + '_val': val, tag + '_gx_sub': li.grad_sub(gx), tag + '_gx_sum': gx.double().sum(), tag + '_gx_abssum': gx.double().abs().sum(),
                    tag + '_x_sum': x.double().sum(), tag + '_y_sum': y.double().sum(), tag + '_lm': l})
    # the contextual chain alone on small feature maps (compute_cosine_distance -> relative_distance -> cx -> max / mean / -log)
    from spi.criteria import bbox_cx_loss as rbx
    fx, fy = torch.randn(2, 16, 6, 6, generator=g), torch.randn(2, 16, 6, 6, generator=g)
    fxr = fx.clone().requires_grad_(True)
    d = rbx.compute_cosine_distance(fxr, fy)
    cx = rbx.compute_cx(rbx.compute_relative_distance(d), 0.5)
    cxl = torch.mean(-torch.log(torch.mean(torch.max(cx, dim=1)[0], dim=1) + 1e-5))
    gfx, = torch.autograd.grad(cxl, fxr)
    fxo = fx.clone().requires_grad_(True)
    co = olo.contextual_loss(fxo, fy)
    diff('contextual chain value', cxl, co)
    diff('contextual chain d/dfx', gfx, torch.autograd.grad(co, fxo)[0])
    out.update(cx_fx=fx, cx_fy=fy, cx_val=cxl, cx_gfx=gfx, cx_dist=d)
    # storage: the 512^2 inputs are regenerated by tests/loss_inputs.py (their sums are stored as a guard); gradients are stored subsampled + their sums
    save('losses', **out)


from loss_inputs import STAGE2_KEYS, stage2_sub, ReplayDraws, oracle_stage2_run  # noqa: E402


def _reference_coach(kind, g1_steps, threshold):
    """Instantiate the reference's RotBboxCoach / SingleIDCoach on the narrow reference generator and run ITS train() on the synthetic image.
    Obstacles removed, none of them arithmetic: load_eg3d (needs the 380 MB pickle) returns the seeded narrow generator, set up exactly as
    load_utils.py:26-32 leaves it (requires_grad False, eval, neural_rendering_resolution 128); load_sg_vgg / Metric (checkpoints) are stubs
    (unused by train() without wandb); post_process (jpg / mp4 writers) is a no-op; the pivot comes through the reference's own
    load_inversions cache (base_coach.py:91-93) so stage 1 need not run first.
    -> dict(losses per iteration, pre-Adam gradients and post-step parameters of STAGE2_KEYS, every random draw in order)."""
    import tempfile
    import importlib
    env = reference_loss_env()
    import spi.utils.camera_utils as rcu
    rcu.GAUSS_CONST = torch.sqrt(torch.tensor(2 * torch.pi))
    from spi.configs import global_config, paths_config, hyperparameters
    global_config.device = 'cpu'
    tmp = tempfile.mkdtemp()
    for k in ('checkpoints_dir', 'embedding_base_dir', 'experiments_output_dir', 'images_output_dir', 'mirror_images_output_dir', 'video_output_dir'):
        setattr(paths_config, k, f'{tmp}/{k}/')
    import spi.training.coaches.base_coach as bc
    mod = importlib.import_module('spi.training.coaches.' + ('rot_bbox_cx_coach' if kind == 'RotBbox' else 'pti_coach'))

    def load_narrow():
        G = build_ref_generator(True).requires_grad_(False)
        G.neural_rendering_resolution = 128
        return G.eval()
    bc.load_old_G = load_narrow
    bc.load_modules.load_sg_vgg = lambda: torch.nn.Identity()
    bc.Metric = lambda: None
    mod.tqdm = lambda x: x
    hyperparameters.first_inv_type, hyperparameters.first_inv_steps = 'mir', 500
    hyperparameters.G_1_type, hyperparameters.G_1_step = ('RotBbox' if kind == 'RotBbox' else 'pti'), g1_steps
    hyperparameters.pt_rot_lambda, hyperparameters.pt_mirror_rot_lambda, hyperparameters.pt_depth_lambda, hyperparameters.pt_tv_lambda = 0.1, 0.05, 1.0, 0.0
    hyperparameters.LPIPS_value_threshold = threshold
    hyperparameters.load_embedding_coach_name = 'preloaded'
    sys.path.insert(0, ROOT)
    from spi_amd.data.images_dataset import SyntheticDataset
    item = SyntheticDataset(1)[0]
    data = dict(name=[item['name']], img=item['img'][None], c=torch.as_tensor(item['c'])[None], mask=item['mask'][None, None], lm=item['lm'][None])
    w_pivot = torch.randn(1, 14, 512, generator=torch.Generator().manual_seed(16))
    coach = (mod.RotBboxCoach if kind == 'RotBbox' else mod.SingleIDCoach)([data], False)
    coach.w_pivots[item['name']] = w_pivot.clone()
    coach.post_process = lambda *a, **k: None
    log = dict(l2=[], lpips=[], boxcx=[], grads=[], params=[])

    def rec(fn, key):
        def inner(*a, **k):
            v = fn(*a, **k)
            log[key].append(v.detach().clone())
            return v
        return inner
    mod.l2_loss = rec(mod.l2_loss, 'l2')
    lp_mod = coach.lpips_loss
    coach.lpips_loss = rec(lambda *a, **k: lp_mod(*a, **k), 'lpips')
    if kind == 'RotBbox':
        bx_mod = coach.box_cx_loss
        coach.box_cx_loss = rec(lambda *a, **k: bx_mod(*a, **k), 'boxcx')
    orig_step = torch.optim.Adam.step

    def logging_step(self, *a, **k):
        named = dict(coach.G.named_parameters())
        log['grads'].append({k_: named[k_].grad.detach().clone() for k_ in STAGE2_KEYS})
        r = orig_step(self, *a, **k)
        log['params'].append({k_: named[k_].detach().clone() for k_ in STAGE2_KEYS})
        return r
    torch.optim.Adam.step = logging_step
    try:
        with SeededRecorder() as recd:
            coach.train()
    finally:
        torch.optim.Adam.step = orig_step
    log.update(draws=recd.draws, draw_shapes=recd.shapes, draw_kinds=recd.kinds, data=data, w_pivot=w_pivot, coach_name=coach.coach_name, image_counter=coach.image_counter,
               final={k_: v.detach().clone() for k_, v in coach.G.named_parameters() if k_ in STAGE2_KEYS},
               P0={k_: v.detach().clone() for k_, v in coach.original_G.state_dict().items()},
               pnames=[k_ for k_, _ in coach.G.named_parameters()], W16=env['W16'], W19=env['W19'])
    return log


def _oracle_stage2(ref, n_iters, threshold, pti_only):
    """the oracle's loop on the reference's recorded draws -> per-iteration dicts (losses, pre-Adam grads, post-step params)"""
    d = ref['data']
    data = dict(img=d['img'], c=d['c'], lm=d['lm'], mask=d['mask'])
    res, rd, _ = oracle_stage2_run(ref['P0'], ref['pnames'], data, ref['w_pivot'], ref['draws'], n_iters, threshold, pti_only, ref['W16'], ref['W19'], dict(RK))
    assert rd.pos == len(rd.d), (rd.pos, len(rd.d))
    return res


def sec_stage2():
    """VERDICT r03 next-#1b: the reference's own RotBboxCoach.train() (spi/training/coaches/rot_bbox_cx_coach.py:24-171, BaseCoach
    base_coach.py:36-135) for 5 iterations -- iterations 0 and 4 carry the rot / mirror-rot / depth branches (`i % 4 == 0`) -- on the narrow
    reference generator with every random draw recorded; then once more with the early-stop threshold above the loss (the `break` before
    optimizer.step(), :148-149)."""
    ref = _reference_coach('RotBbox', 5, -1.0)
    n = len(ref['params'])
    assert n == 5 and len(ref['l2']) == 5 + 2 and len(ref['lpips']) == 5 + 2 and len(ref['boxcx']) == 2, (n, len(ref['l2']), len(ref['lpips']), len(ref['boxcx']))
    print(f"    reference coach '{ref['coach_name']}': {n} optimiser steps, {len(ref['draws'])} random draws")
    orc = _oracle_stage2(ref, 5, -1.0, False)
    out = dict(w_pivot=ref['w_pivot'], n_draws=np.array(len(ref['draws'])))
    il2 = ilp = ibx = 0
    for i in range(5):
        vals = dict(l2=ref['l2'][il2], lpips=ref['lpips'][ilp])
        il2, ilp = il2 + 1, ilp + 1
        if i % 4 == 0:
            vals['rot'] = ref['lpips'][ilp] * 0.1 * 4            # (:104) loss_rot * pt_rot_lambda * rot_bs
            vals['mirror_rot'] = ref['boxcx'][ibx] * 0.05 * 4    # (:130)
            vals['depth'] = ref['l2'][il2] * 1.0                 # (:139-140)
            il2, ilp, ibx = il2 + 1, ilp + 1, ibx + 1
        for k, v in vals.items():
            out[f'it{i}_{k}'] = v
            diff(f'iteration {i}: {k}', v, torch.tensor(orc[i][k]))
        for k in STAGE2_KEYS:
            out[f'it{i}_grad/{k}'] = stage2_sub(k, ref['grads'][i][k])
            out[f'it{i}_param/{k}'] = stage2_sub(k, ref['params'][i][k])
        dg = max(diff(f'iteration {i}: pre-Adam grad {k.split("synthesis.")[-1]}', ref['grads'][i][k], orc[i]['grads'][k]) for k in STAGE2_KEYS)
        dp = max((ref['params'][i][k].double() - orc[i]['params'][k].double()).abs().max().item() for k in STAGE2_KEYS)
        print(f'    pin iteration {i}: post-step parameters max|ref-oracle| = {dp:.3e}   (pre-Adam gradients {dg:.3e})')
    out['draw_shapes'] = np.array([json.dumps(ref['draw_shapes'])])
    out['draw_kinds'] = np.array([json.dumps(ref['draw_kinds'])])
    out['draw_checksum'] = torch.stack([d.double().sum() for d in ref['draws']])
    # early stop: threshold above every loss -> the loop breaks in iteration 0 after the backward passes, before optimizer.step()
    ref2 = _reference_coach('RotBbox', 3, 1e9)
    assert len(ref2['params']) == 0 and len(ref2['lpips']) == 2 and ref2['image_counter'] == 1
    for k in STAGE2_KEYS:
        assert torch.equal(ref2['final'][k], ref2['P0'][k]), k
    orc2 = _oracle_stage2(ref2, 3, 1e9, False)
    assert len(orc2) == 1 and orc2[0].get('stopped')
    out['stop_n_draws'] = np.array(len(ref2['draws']))
    out['stop_steps'] = np.array(0)
    print(f"    early stop: reference broke in iteration 0 with {len(ref2['draws'])} draws made and no optimiser step; oracle agrees")
    save('trajectory_stage2', **out)


def sec_pti():
    """VERDICT r03 next-#1b: the reference's own SingleIDCoach.train() (spi/training/coaches/pti_coach.py:34-98) for 3 iterations, then with the
    early-stop threshold above the loss (the `break` BEFORE backward / step, :75-76)."""
    ref = _reference_coach('pti', 3, -1.0)
    assert len(ref['params']) == 3 and len(ref['l2']) == 3 and len(ref['lpips']) == 3
    print(f"    reference coach '{ref['coach_name']}': 3 optimiser steps, {len(ref['draws'])} random draws")
    orc = _oracle_stage2(ref, 3, -1.0, True)
    out = dict(w_pivot=ref['w_pivot'], n_draws=np.array(len(ref['draws'])))
    for i in range(3):
        for k, v in dict(l2=ref['l2'][i], lpips=ref['lpips'][i]).items():
            out[f'it{i}_{k}'] = v
            diff(f'iteration {i}: {k}', v, torch.tensor(orc[i][k]))
        for k in STAGE2_KEYS:
            out[f'it{i}_grad/{k}'] = stage2_sub(k, ref['grads'][i][k])
            out[f'it{i}_param/{k}'] = stage2_sub(k, ref['params'][i][k])
        dg = max(diff(f'iteration {i}: pre-Adam grad {k.split("synthesis.")[-1]}', ref['grads'][i][k], orc[i]['grads'][k]) for k in STAGE2_KEYS)
        dp = max((ref['params'][i][k].double() - orc[i]['params'][k].double()).abs().max().item() for k in STAGE2_KEYS)
        print(f'    pin iteration {i}: post-step parameters max|ref-oracle| = {dp:.3e}   (pre-Adam gradients {dg:.3e})')
    out['draw_shapes'] = np.array([json.dumps(ref['draw_shapes'])])
    out['draw_kinds'] = np.array([json.dumps(ref['draw_kinds'])])
    out['draw_checksum'] = torch.stack([d.double().sum() for d in ref['draws']])
    ref2 = _reference_coach('pti', 3, 1e9)
    assert len(ref2['params']) == 0 and len(ref2['lpips']) == 1
    for k in STAGE2_KEYS:
        assert torch.equal(ref2['final'][k], ref2['P0'][k]), k
    out['stop_n_draws'] = np.array(len(ref2['draws']))
    print(f"    early stop: reference broke in iteration 0 with {len(ref2['draws'])} draws made and no optimiser step")
    save('trajectory_pti', **out)


class StubGenerator:
    """a generator whose "image" is a cheap deterministic function of (w, c): what gen_interp_video feeds the generator per frame and cell
    is the pinned quantity of sec_interp_video, not the rendering (that is golden/orbit_frames.npz)"""
    neural_rendering_resolution = 8
    rendering_kwargs = dict(depth_resolution=2, depth_resolution_importance=2, box_warp=1)

    def __init__(self):
        self.calls = []

    def synthesis(self, ws, c, noise_mode='const', **kw):
        self.calls.append((ws.detach().clone().float(), c.detach().clone().float()))
        n = ws.shape[0]
        base = ws.reshape(n, -1)[:, :192].reshape(n, 3, 8, 8).float()
        img = torch.tanh(base + c[:, :16].sum(dim=1).view(n, 1, 1, 1) * 0.05 + c[:, 3].view(n, 1, 1, 1))
        return {'image': img, 'image_depth': img[:, :1] + 2.5, 'image_raw': img}


def sec_interp_video():
    """SURVEY 8f-1 / VERDICT r03 missing #5: the reference's own gen_interp_video (spi/utils/video_utils.py:74-230) over SEVERAL latents -- keyframe
    interpolation (scipy interp1d over the tiled keyframes), one camera orbit over num_keyframes * w_frames frames, layout_grid -- driven with a stub
    generator and a placeholder imageio writer that keeps the frames (imageio / mrcfile are import-time dependencies of that module only)."""
    import types
    reference_loss_env()
    captured = []

    class Writer:
        def append_data(self, f):
            captured.append(np.asarray(f).copy())

        def close(self):
            pass
    sys.modules['imageio'].get_writer = lambda *a, **k: Writer()
    import spi.utils.video_utils as rvu
    rvu.tqdm = lambda x, **k: x
    out = {}
    g = torch.Generator().manual_seed(81)
    for tag, (nk, gw, gh, wf) in dict(k3=(3, 1, 1, 5), grid=(2, 2, 1, 4), one=(1, 1, 1, 6)).items():
        ws = torch.randn(nk * gw * gh, 14, 16, generator=g)
        G = StubGenerator()
        captured.clear()
        rvu.gen_interp_video(G, {'w': ws}, mp4='/tmp/unused.mp4', w_frames=wf, grid_dims=(gw, gh), device=torch.device('cpu'))
        calls = G.calls[1:]                                       # (the first call is the reference's warm-up, :113)
        assert len(calls) == nk * wf * gw * gh and len(captured) == nk * wf
        out.update({f'{tag}_ws': ws, f'{tag}_w_per_call': torch.cat([a for a, _ in calls]), f'{tag}_c_per_call': torch.cat([b for _, b in calls]),
                    f'{tag}_frames': np.stack(captured), f'{tag}_cfg': np.array([nk, gw, gh, wf])})
        print(f'    {tag}: {nk} keyframes, grid {gw}x{gh}, {wf} frames per keyframe -> {len(captured)} frames of shape {captured[0].shape}')
    save('interp_video', **out)


def preprocess_inputs():
    """synthetic photo + 68 landmarks (image coordinates, y down) + a 5 x 3 stand-in for the BFM's standard landmarks (similarity_Lm3D_all.mat
    is not available offline); shared with the tests through this function's recipe (seeded numpy)"""
    from PIL import Image
    rng = np.random.default_rng(7)
    base = rng.random((30, 28, 3))
    photo = Image.fromarray((base * 255).astype(np.uint8)).resize((280, 300), resample=Image.BICUBIC)          # W 280, H 300, smooth
    a = np.linspace(0, 2 * np.pi, 68, endpoint=False)
    lm = np.stack([140 + 55 * np.cos(a) + 3 * np.sin(5 * a), 150 + 65 * np.sin(a) + 2 * np.cos(7 * a)], axis=1)
    lm3d = np.array([[-0.31, 0.29, 0.05], [0.31, 0.30, 0.04], [0.0, 0.02, 0.35], [-0.25, -0.33, 0.07], [0.24, -0.34, 0.06]])
    return photo, lm, lm3d


def sec_preprocess():
    """SURVEY 8f-4 / VERDICT r03 missing #4: the crop + camera producer's own arithmetic, executed from the reference:
    preprocess/extract_3dmm.py POS / extract_5p / resize_n_crop_img / align_img / Extract3dmm.image_transform (:16-138),
    preprocess/extract_camera.py compute_rotation / CameraExtractor.crop / cal_camera / flip_yaw / _cal_mirror_c (:14-176).
    Import-time obstacles, none of them arithmetic: `face_alignment`, `cv2`, `skimage`, `kornia`, `torchvision` are imported at module level
    (placeholders; the detector instantiated by extract_landmark.py:10 is a placeholder call) and Deep3DFaceRecon's util/preprocess.py:12 names
    `np.VisibleDeprecationWarning`, which numpy 2 removed (aliased).  One run-time obstacle: align_img builds `np.array([w0, h0, s, t[0], t[1]])`
    from scalars and shape-(1,) arrays (:98) -- a ragged array, an error since numpy 1.24 -- so that module sees a numpy proxy whose `array`
    takes the first element of such entries (the five numbers that are meant); every other numpy call is forwarded untouched."""
    import types
    import tempfile

    class Lazy(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith('__'):
                raise AttributeError(k)
            m = Lazy(self.__name__ + '.' + k)
            setattr(self, k, m)
            return m

        def __call__(self, *a, **k):
            return Lazy('called')
    reference_loss_env()
    for name in ('face_alignment', 'cv2', 'skimage', 'skimage.transform', 'kornia', 'kornia.geometry'):
        sys.modules[name] = Lazy(name)
    sys.modules['torchvision'] = Lazy('torchvision')
    if not hasattr(np, 'VisibleDeprecationWarning'):
        np.VisibleDeprecationWarning = DeprecationWarning
    import preprocess.extract_3dmm as e3
    import preprocess.extract_camera as ec
    from PIL import Image

    class NpProxy:
        def __getattr__(self, k):
            return getattr(np, k)

        @staticmethod
        def array(obj, *a, **k):
            try:
                return np.array(obj, *a, **k)
            except ValueError:
                return np.array([float(np.ravel(v)[0]) for v in obj], *a, **k)
    e3.np = NpProxy()
    from oracle import recon_ref as orr2
    photo, lm, lm3d = preprocess_inputs()
    out = dict(photo=np.array(photo), lm=lm, lm3d=lm3d)
    lm_up = lm.copy()
    lm_up[:, -1] = photo.size[1] - 1 - lm_up[:, -1]
    # POS / extract_5p
    lm5 = e3.extract_5p(lm_up)
    t, sc = e3.POS(lm5.transpose(), lm3d.transpose())
    out.update(lm5=lm5, pos_t=np.ravel(t), pos_s=np.array(sc))
    to, so = orr2.pos(lm5.transpose(), lm3d.transpose())
    print('    pin POS: max|ref-oracle| t %.3e  s %.3e' % (np.abs(np.ravel(t) - np.ravel(to)).max(), abs(sc - so)))
    # align_img at both rescale factors the pipeline uses (466.285 for the regressor, 300 for the training crop)
    for tag, rf in (('a466', 466.285), ('a300', 300)):
        tp, im_low, lm_new, _, im_high = e3.align_img(photo, lm_up.copy(), lm3d, rescale_factor=rf)
        tpo, im_low_o, lm_new_o, im_high_o = orr2.align_img(photo, lm_up.copy(), lm3d, rescale_factor=rf)
        print(f'    pin align_img[{rf}]: trans_params %.3e  landmarks %.3e  224^2 image %d  1024^2 image %d grey levels' % (
            np.abs(tp - tpo).max(), np.abs(lm_new - lm_new_o).max(), np.abs(np.array(im_low).astype(int) - np.array(im_low_o).astype(int)).max(),
            np.abs(np.array(im_high).astype(int) - np.array(im_high_o).astype(int)).max()))
        out.update({tag + '_tp': tp, tag + '_lm': lm_new, tag + '_low': np.array(im_low), tag + '_high_sub': np.array(im_high)[::8, ::8],
                    tag + '_high_sum': np.array(im_high).astype(np.int64).sum()})
    # Extract3dmm.image_transform (flips the caller's landmarks in place, :134) and CameraExtractor.crop on dummy instances (no checkpoints)
    ex = object.__new__(e3.Extract3dmm)
    ex.lm3d_std = lm3d
    lm_arg = lm.copy()
    img_t, lm_t = ex.image_transform(photo, lm_arg)
    out.update(it_img=img_t, it_lm=lm_t, it_lm_after=lm_arg)
    tmp = tempfile.mkdtemp()
    cex = object.__new__(ec.CameraExtractor)
    cex.lm3d_std, cex.crop_outdir, cex.c_outdir, cex.mode = lm3d, tmp, tmp, 'png'
    cex.crop(photo, lm_arg, 'x')                                      # (sees the flipped landmarks, as in __extract :142-146)
    crop = np.array(Image.open(os.path.join(tmp, 'x.png')))
    out.update(crop_sub=crop[::4, ::4], crop_sum=crop.astype(np.int64).sum(), crop_shape=np.array(crop.shape))
    # rotation / camera
    ang = torch.tensor([[0.3, -0.5, 0.2], [0.1, 0.2, -0.3]])
    R = ec.compute_rotation(ang)
    print('    pin compute_rotation: %.3e' % (R - orr2.compute_rotation(ang)).abs().max().item())
    coeff = {'angle': ang[:1].clone(), 'trans': torch.tensor([[0.1, -0.2, 0.3]])}
    cam = cex.cal_camera(coeff)
    camo = orr2.cal_camera(ang[:1].clone(), torch.tensor([0.1, -0.2, 0.3]))
    print('    pin cal_camera: pose %.3e  intrinsics %.3e' % (np.abs(np.array(cam['pose']) - np.array(camo['pose'])).max(),
                                                            np.abs(np.array(cam['intrinsics']) - np.array(camo['intrinsics'])).max()))
    from preprocess.process_camera import process_camera
    c25 = process_camera(cam['pose'], cam['intrinsics'])
    c25m = cex._cal_mirror_c(c25)
    print('    pin _cal_mirror_c: %.3e' % np.abs(c25m - orr2.mirror_camera(c25)).max())
    out.update(rot_ang=ang, rot=R, cam_pose=np.array(cam['pose']), cam_K=np.array(cam['intrinsics']), cam_angle=np.array(cam['angle']),
               cam_trans_after=coeff['trans'], c25=c25, c25_mirror=c25m)
    save('preprocess', **out)

SECTIONS = dict(manifest=sec_manifest, ops=sec_ops, renderer=sec_renderer, renderer_options=sec_renderer_options, synthesis=sec_synthesis,
                geometry=sec_geometry, schedule=sec_schedule, trajectory=sec_trajectory, trajectory_sg=sec_trajectory_sg,
                tv=sec_tv, orbit=sec_orbit, orbit_frames=sec_orbit_frames, bisenet=sec_bisenet, recon=sec_recon, losses=sec_losses, stage2=sec_stage2, pti=sec_pti, interp_video=sec_interp_video, preprocess=sec_preprocess)

if __name__ == '__main__':
    todo = sys.argv[1:] or list(SECTIONS)
    for s in todo:
        print(f'[{s}]')
        SECTIONS[s]()
