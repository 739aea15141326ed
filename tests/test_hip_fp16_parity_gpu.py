"""BASELINE configs[4]'s fp16 super-resolution arithmetic, BACKWARD side (round-5 review, weak #1 / next #3).

The reference's use_fp16 blocks run their whole backward in half (networks_stylegan2.py:421-461: the block's activations are half tensors, so are the
gradients autograd hands between its layers).  The oracle restates that with `fp16_operands` + `fp16_storage` (oracle/stylegan_ref.py): every tensor
boundary of an SR block is `x.half().float()`, whose autograd backward rounds the GRADIENT to fp16 at the same boundary -- so the oracle's backward
carries fp16 gradient tensors exactly where the HIP path does (conv2d_mfma `act_dtype = 1`, tail_bwd / FIR adjoint on half tensors).

What is compared, and how:
  * layer by layer: every SR layer (transposed conv0 + FIR, conv1, torgb of both blocks) fed with the ORACLE's fp16-path input and a
    fp16-representable output cotangent.  The data gradient (an fp16 tensor) agrees with the oracle's to a quarter of the fp32 <-> fp16 distance (rms),
    99 % of its elements within 5e-4 of the range; what is left are isolated 3 x 3 spots behind leaky-ReLU SIGN flips (two fp32 summation orders put
    ~1e-6 of the pre-activations on different sides of zero -- located and counted on the GPU, profiles/r06_fp16_layer_backward.txt; never at tile or
    image borders).  Weight / style / bias gradients within half their own fp32 <-> fp16 gap;
  * network level, statistical (fp16 rounding and those flips decorrelate two implementations after a few layers -- see
    test_fp16_sr_full_size_128_vs_fp16_rounding_oracle): the gradients of one synthesis wrt W+ and 9 generator tensors, HIP fp16 vs oracle fp16, within
    3 x the fp32 <-> fp16 gap of the oracle itself, with the fp32 path of the same launches held to 2e-3 next to it;
  * the BRANCH iteration (i = 0: rot, mirror-rot, depth -- what configs[4] turns on; rot_bbox_cx_coach.py:86-146) in fp16 against
    `oracle.loops_ref.stage2_iteration(..., opts16)`: the five loss values within 1e-2 (north star), the pre-Adam gradients like the item above.
"""
import os
import tempfile
import pytest
import torch

from conftest import assert_close, rel_err
from synth_weights import load_manifest
from oracle import renderer_ref as orr
from test_hip_fullsize_gpu import _setup

pytestmark = pytest.mark.gpu
DEV = 'cuda'

FP16_GRAD_NAMES = [
    'superresolution.block0.conv0.weight',            # 32 -> 256 transposed at 128^2 -> 256^2, fp16 block
    'superresolution.block0.conv1.weight',            # 256 -> 256 at 256^2
    'superresolution.block1.conv0.weight',            # 256 -> 128 transposed
    'superresolution.block1.conv1.weight',            # 128 -> 128 at 512^2
    'superresolution.block1.torgb.weight',
    'superresolution.block1.conv0.affine.weight',     # style gradient through the modulation adjoint of an fp16 layer
    'superresolution.block0.conv1.bias',
    'backbone.synthesis.b256.conv0.weight',           # upstream of the fp16 blocks: receives their fp16 data gradient
    'decoder.net.0.weight',
]


def _rms(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).square().mean().sqrt() / b.square().mean().sqrt().clamp_min(1e-30)).item()


def _half_ulps(a, b):
    """largest |a - b| in units of the fp16 spacing at |b| (both already fp16-representable)"""
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    ulp = torch.maximum(b.abs(), torch.tensor(6.1e-5)).log2().floor().exp2() * 2.0 ** -10
    return ((a - b).abs() / ulp).max().item()


@pytest.mark.timeout(3000)
def test_fp16_sr_layer_backward_on_the_oracles_inputs():
    """Every SR layer's backward in the fp16 arithmetic, on the oracle's own fp16-path input and one shared cotangent (no flips can build up inside
    one layer): dx is an fp16 tensor whose distance from the oracle's is a quarter of the fp32 <-> fp16 distance or less (rms), 99 % of its elements within
    5e-4 of the range, <= 2e-3 of them in the isolated spots leaky-ReLU sign flips leave; parameter gradients within half their fp32 <-> fp16 gap."""
    from oracle import stylegan_ref as sg
    from spi_amd.configs import global_config
    P, G, ws, c, xi, u, opts, gen = _setup(96, seed=3)
    opts16 = dict(opts, sr_fp16_operands=True, sr_fp16_storage=True)
    with torch.no_grad():
        ref16 = orr.synthesis(P, ws, c, opts16, neural_rendering_resolution=128, xi=xi, u=u)
    global_config.enable_fp16_blocks = True
    w_last = ws[:, -1]
    kw16 = dict(fp16_operands=True, fp16_storage=True)
    x_in = ref16['feature_image']
    checks = []
    for bname, block in (('block0', G.superresolution.block0), ('block1', G.superresolution.block1)):
        pfx = f'superresolution.{bname}.'
        x_in = x_in.half().float()                                # the block entry's cast (networks_stylegan2.py:436)
        layers = (('conv0', block.conv0, dict(up=2, noise_mode='none', conv_clamp=256), dict(noise_mode='none', fp16=True), sg.synthesis_layer),
                  ('conv1', block.conv1, dict(noise_mode='none', conv_clamp=256), dict(noise_mode='none', fp16=True), sg.synthesis_layer),
                  ('torgb', block.torgb, dict(conv_clamp=256), dict(fp16=True), sg.torgb_layer))
        x_layer = x_in
        for nm, mod, okw, hkw, ofn in layers:
            names = [pfx + nm + s for s in ('.weight', '.affine.weight', '.bias')]
            Pl = dict(P)
            for k in names:
                Pl[k] = P[k].clone().requires_grad_(True)
            xr = x_layer.clone().requires_grad_(True)
            y_ref = ofn(Pl, pfx + nm + '.', xr, w_last, **okw, **kw16)
            dy = (torch.randn(y_ref.shape, generator=gen) * 0.5).half().float()          # an fp16 gradient tensor arrives from the next layer
            g_ref = torch.autograd.grad(y_ref, [xr] + [Pl[k] for k in names], dy)
            params = dict(mod.named_parameters())
            for p in params.values():
                p.requires_grad_(True)
            xg = x_layer.to(DEV).half().requires_grad_(True)
            y = mod(xg, w_last.to(DEV), **hkw)
            assert y.dtype == torch.float16
            g = torch.autograd.grad(y, [xg, params['weight'], params['affine.weight'], params['bias']], dy.to(DEV).half())
            for p in params.values():
                p.requires_grad_(False)
            assert g[0].dtype == torch.float16, (bname, nm, g[0].dtype)                   # the data gradient IS an fp16 tensor
            d_ = (g[0].float().cpu() - g_ref[0]).abs()
            e, flips, ulps = (d_.max() / g_ref[0].abs().max()).item(), (d_ > 0).float().mean().item(), _half_ulps(g[0], g_ref[0].half())
            print(f'  fp16 SR {bname}.{nm} backward on the oracle input: dx max {e:.2e}, {flips:.2e} of the elements differ, worst {ulps:.2f} ulp;',
                  ' '.join(f'{k.split(".", 2)[2]} {rel_err(a, b):.1e}' for k, a, b in zip(names, g[1:], g_ref[1:])))
            # What "the same arithmetic" can and cannot mean here (measured, profiles/r06_fp16_layer_backward.txt): both sides round the same fp32 sums to
            # fp16, so most elements agree exactly or to one ulp of the element -- but the leaky-ReLU derivative is decided by the SIGN of a
            # pre-activation, and the two fp32 summation orders put ~1e-6 of the pre-activations on different sides of zero (10-30 elements of a
            # 16-33 M-element layer).  Each such flip changes dz at ONE (channel, pixel) and reaches its 3 x 3 neighbourhood in every input channel of
            # dx: isolated spots of 1e-2 of the tensor's range (never at the borders: not a tiling effect), 2e-4 .. 7e-4 of the elements beyond 1e-3.
            # cuDNN and the reference's CPU kernels differ from each other in the same way.  So: quantiles and rms, measured against the arithmetic's own
            # size -- the distance between the oracle's fp32 and fp16 backward of the same layer.
            xr32 = x_layer.clone().requires_grad_(True)
            g32 = torch.autograd.grad(ofn(Pl, pfx + nm + '.', xr32, w_last, **okw), [xr32] + [Pl[k] for k in names], dy)
            gap_rms, e_rms = _rms(g32[0], g_ref[0]), _rms(g[0].float(), g_ref[0])
            q99 = torch.quantile((d_.flatten()[:: max(1, d_.numel() // 4000000)] / g_ref[0].abs().max()), 0.99).item()
            spots = (d_ > 1e-3 * g_ref[0].abs().max()).float().mean().item()
            yh = y.detach().float().cpu()
            nflip = int(((yh > 0) != (y_ref.detach() > 0)).sum().item())          # outputs on different sides of zero = different leaky-ReLU branches in the backward
            print(f'     dx rms {e_rms:.2e} (oracle fp32 <-> fp16: {gap_rms:.2e}), 99 % of the elements within {q99:.1e} of the range, {spots:.1e} beyond 1e-3; '
                  f'{nflip} of {yh.numel()} forward outputs differ in SIGN between the two implementations')
            checks.append((f'{bname}.{nm} dx rms vs the fp32 <-> fp16 gap', e_rms, 0.25 * gap_rms + 1e-5))
            checks.append((f'{bname}.{nm} dx 99 % quantile', q99, 5e-4))
            checks.append((f'{bname}.{nm} dx share beyond 1e-3 (kink flips)', spots, 2e-3))
            checks.append((f'{bname}.{nm} dx max', e, 6e-2))
            # Parameter gradients (fp32 sums of fp16 products on the HIP side; the oracle, like cuDNN, rounds the modulated-weight gradient to fp16):
            # within half the fp32 <-> fp16 gap of the same quantity, and 6e-3 absolute
            for k, a, b, c32 in zip(names, g[1:], g_ref[1:], g32[1:]):
                checks.append((f'{k} gradient, fp16 layer backward', rel_err(a, b), min(6e-3, 0.5 * rel_err(c32, b) + 3e-4)))
            with torch.no_grad():
                x_layer = y_ref.detach() if nm != 'torgb' else x_layer
        x_in = x_layer
    bad = [(w_, v, b_) for w_, v, b_ in checks if not v <= b_]
    assert not bad, bad


@pytest.mark.timeout(3000)
def test_fp16_sr_network_gradients_vs_fp16_rounding_oracle():
    """One synthesis at full width with fp16 SR blocks: gradients wrt W+ and nine generator tensors, HIP vs the oracle in the same arithmetic.
    Bars RELATIVE to the arithmetic's own size: the HIP fp16 gradient is no further from the fp16 oracle than 3 x the distance between the oracle's
    fp32 and fp16 gradients (rms and max-normalised; two implementations are two independent draws of the rounding noise), 4e-2 absolute; the fp32
    path of the same launch sequence is held to the north star's 2e-3 next to it (measured: 3e-6)."""
    from spi_amd.configs import global_config
    P, G, ws, c, xi, u, opts, gen = _setup(96, seed=4)
    opts16 = dict(opts, sr_fp16_operands=True, sr_fp16_storage=True)
    d_img = None

    def oracle_grads(o):
        nonlocal d_img
        Pl = {k: v.clone() for k, v in P.items()}
        for k in FP16_GRAD_NAMES:
            Pl[k].requires_grad_(True)
        wr = ws.clone().requires_grad_(True)
        ref = orr.synthesis(Pl, wr, c, o, neural_rendering_resolution=128, xi=xi, u=u)
        if d_img is None:
            d_img = torch.randn(ref['image'].shape, generator=gen)
        loss = (ref['image'] * d_img).mean() + (ref['image'] ** 2).mean() * 0.1
        return loss.item(), torch.autograd.grad(loss, [wr] + [Pl[k] for k in FP16_GRAD_NAMES])

    def hip_grads(fp16):
        global_config.enable_fp16_blocks = fp16
        wg = ws.to(DEV).requires_grad_(True)
        params = dict(G.named_parameters())
        out = G.synthesis(wg, c.to(DEV), noise_mode='const', render_noise=(xi, u))
        loss = (out['image'] * d_img.to(DEV)).mean() + (out['image'] ** 2).mean() * 0.1
        g = torch.autograd.grad(loss, [wg] + [params[k] for k in FP16_GRAD_NAMES])
        global_config.enable_fp16_blocks = False
        return loss.item(), g
    l16, g16 = oracle_grads(opts16)
    l32, g32 = oracle_grads(opts)
    lh16, h16 = hip_grads(True)
    lh32, h32 = hip_grads(False)
    assert abs(lh16 - l16) <= 1e-2 * abs(l16) + 1e-6 and abs(lh32 - l32) <= 1e-2 * abs(l32) + 1e-6, (lh16, l16, lh32, l32)
    rows = []
    for nm, a16, b16, a32, b32 in zip(['ws'] + FP16_GRAD_NAMES, h16, g16, h32, g32):
        e_max, e_rms, gap = rel_err(a16, b16), _rms(a16, b16), _rms(b32, b16)
        print(f'  fp16 gradient {nm}: hip16 vs oracle16 max {e_max:.2e} rms {e_rms:.2e} | oracle32 vs oracle16 rms {gap:.2e} | hip32 vs oracle32 max {rel_err(a32, b32):.2e}')
        assert_close(a32, b32, 2e-3, f'fp32 gradient {nm} (control)')
        rows.append((nm, e_max, e_rms, gap, rel_err(b32, b16)))
    # Two implementations of the fp16 arithmetic are two independent draws of its rounding noise and of its leaky-ReLU sign flips (see the layer test):
    # their distance is ~sqrt(2) x the distance of either from the fp32 result, more where every flip of ten layers ends up (measured, rms: 1.3 .. 2.7 x).
    # Bars: 3 x the oracle's own fp32 <-> fp16 distance in rms, 5 x max-normalised (a maximum is one element: measured up to 3.4 x), absolute caps of
    # 4e-2 -- the fp32 path next to it is at 1e-5.
    bad = [r for r in rows if not (r[2] <= 3.0 * r[3] + 2e-4 and r[1] <= 5.0 * r[4] + 5e-4 and r[1] <= 4e-2 and r[2] <= 4e-2)]
    assert not bad, bad


@pytest.mark.timeout(3000)
def test_fp16_stage2_branch_iteration_vs_oracle():
    """configs[4]'s arithmetic in the iteration that matters for it: ONE RotBbox iteration with the rot / mirror-rot / depth branches (i = 0) and
    fp16 super-resolution blocks, against the oracle's iteration in the same arithmetic with every draw replayed: five loss values within 1e-2
    (north star); pre-Adam gradients of tensors inside and upstream of the fp16 blocks within the fp16 path's statistical bars."""
    from oracle import losses_ref as olo, loops_ref as olp
    from spi_amd.configs import global_config, hyperparameters, paths_config
    from spi_amd.criteria.lpips.lpips import LPIPS
    from spi_amd.criteria.bbox_cx_loss import BoxCXLoss
    from spi_amd.training.coaches.rot_bbox_cx_coach import RotBboxCoach
    from spi_amd.data.images_dataset import SyntheticDataset
    from spi_amd.utils.rng import ReplayRNG
    P, G, _, _, _, _, opts, _ = _setup(96)
    opts16 = dict(opts, sr_fp16_operands=True, sr_fp16_storage=True)
    G = G.requires_grad_(False)
    W, W19 = olo.make_vgg16_weights(seed=0), olo.make_vgg19_head_weights(seed=1)
    data = SyntheticDataset(1)[0]
    data = {k: (v[None] if torch.is_tensor(v) else v) for k, v in data.items()}
    w_pivot = torch.randn(1, 14, 512, generator=torch.Generator().manual_seed(6)) * 0.7
    man = load_manifest('full')
    pnames = [k for k in man if not (k.endswith('noise_const') or k.endswith('resample_filter') or k.endswith('w_avg'))]
    keys = ('superresolution.block1.conv1.weight', 'superresolution.block0.conv0.weight', 'superresolution.block0.conv1.affine.weight',
            'superresolution.block1.torgb.weight', 'backbone.synthesis.b256.conv0.weight', 'backbone.synthesis.b64.conv1.weight',
            'decoder.net.0.weight', 'decoder.net.2.weight')
    mask = data['mask'].reshape(1, 1, 512, 512)
    od = dict(img=data['img'], c=torch.as_tensor(data['c']).reshape(1, 25), lm=data['lm'].reshape(1, 68, 2),
              face_mask=olp.face_mask_from_parsing(mask).float())
    hp = dict(olp.HP, LPIPS_value_threshold=-1.0)
    st = olp.Stage2State(P, pnames)
    draws = olp.Draws()
    torch.manual_seed(0)
    ref = olp.stage2_iteration(st, 0, od, w_pivot, opts16, lambda a, b: olo.lpips(W, a, b), lambda a, b, l: olo.box_cx_loss(W19, a, b, l), hp=hp, draws=draws)
    ref_grads = {k: st.P[k].grad.detach().clone() for k in keys}
    assert {'l2', 'lpips', 'rot', 'mirror_rot', 'depth'} <= set(ref)
    tmp = tempfile.mkdtemp()
    for k in ('checkpoints_dir', 'embedding_base_dir', 'experiments_output_dir', 'images_output_dir', 'mirror_images_output_dir'):
        setattr(paths_config, k, f'{tmp}/{k}/')                    # (the autouse fixture of conftest.py restores the config modules)
    hyperparameters.first_inv_type, hyperparameters.G_1_type = 'mir', 'RotBbox'
    hyperparameters.pt_rot_lambda, hyperparameters.pt_mirror_rot_lambda, hyperparameters.pt_depth_lambda = 0.1, 0.05, 1.0
    hyperparameters.LPIPS_value_threshold = -1.0
    global_config.enable_fp16_blocks = True
    coach = RotBboxCoach(None, False, G=G, lpips_loss=LPIPS(weights=W), box_cx_loss=BoxCXLoss(weights=W19))
    ctx = coach.prepare_image(data)
    rng = ReplayRNG(draws.log, DEV)
    got = coach.train_step(0, ctx, w_pivot.to(DEV), rng=rng)[1]
    assert rng.pos == len(draws.log)
    print('fp16 branch iteration losses (hip, oracle16):', {k: (round(got[k].item(), 6), round(float(ref[k]), 6)) for k in ('l2', 'lpips', 'rot', 'mirror_rot', 'depth')})
    for k in ('l2', 'lpips', 'rot', 'mirror_rot', 'depth'):
        assert abs(got[k].item() - ref[k]) <= 1e-2 * abs(ref[k]) + 1e-7, (k, got[k].item(), ref[k])
    params = dict(coach.G.named_parameters())
    rows = []
    for k in keys:
        e_max, e_rms = rel_err(params[k].grad, ref_grads[k]), _rms(params[k].grad, ref_grads[k])
        print(f'  fp16 branch iteration, pre-Adam gradient {k}: max {e_max:.2e} rms {e_rms:.2e}')
        # (the relative bar -- against the oracle's own fp32 <-> fp16 gap -- is test_fp16_sr_network_gradients_vs_fp16_rounding_oracle's; a second
        #  oracle iteration here would cost another ~100 s of host time)
        rows.append((k, e_max, e_rms))
    # (fp16 gap of these tensors: 0.6 - 3.5e-2, test_fp16_sr_network_gradients_vs_fp16_rounding_oracle prints it per tensor)
    bad = [r for r in rows if not (r[1] <= 4e-2 and r[2] <= 4e-2)]
    assert not bad, bad
