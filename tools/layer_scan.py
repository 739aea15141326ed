#!/usr/bin/env python3
"""Diagnostic: every SynthesisLayer / ToRGB shape of a generator, HIP module vs CPU oracle layer, with white-noise inputs and output gradients.
    python tools/layer_scan.py [narrow|full]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
from oracle import stylegan_ref as osg
from spi_amd.training.networks_stylegan2 import SynthesisLayer, ToRGBLayer

kind = sys.argv[1] if len(sys.argv) > 1 else 'narrow'
cm = 32 if kind == 'narrow' else 512
cb = 2048 if kind == 'narrow' else 32768
torch.set_num_threads(32)
g = torch.Generator().manual_seed(0)
def rel(a, b):
    return ((a.detach().cpu().double() - b.detach().double()).abs().max() / b.detach().double().abs().max().clamp_min(1e-30)).item()
chans = {r: min(cb // r, cm) for r in (4, 8, 16, 32, 64, 128, 256)}
cfgs = []
for r in (4, 8, 16, 32, 64, 128, 256):
    if r > 4:
        cfgs.append(('conv0', chans[r // 2], chans[r], r, 2))
    cfgs.append(('conv1', chans[r], chans[r], r, 1))
    cfgs.append(('torgb', chans[r], 96, r, 0))
for n in (1, 2):
  for name, ci, co, res, up in cfgs:
    if name == 'torgb':
        L = ToRGBLayer(ci, co, w_dim=512)
    else:
        L = SynthesisLayer(ci, co, w_dim=512, resolution=res, up=max(up, 1))
    sd = {k: torch.randn(v.shape, generator=g) * (0.1 if k.endswith('bias') and 'affine' not in k else 1.0) for k, v in L.state_dict().items() if k != 'resample_filter'}
    if 'affine.bias' in sd:
        sd['affine.bias'] = 1 + 0.1 * sd['affine.bias']
    if 'noise_strength' in sd:
        sd['noise_strength'] = torch.tensor(0.3)
    sd['resample_filter'] = L.state_dict().get('resample_filter', None)
    if sd['resample_filter'] is None:
        del sd['resample_filter']
    L.load_state_dict(sd)
    L = L.cuda()
    rin = res // 2 if up == 2 else res
    x = torch.randn(n, ci, rin, rin, generator=g)
    w = torch.randn(n, 512, generator=g)
    P = {('L.' + k): v.clone().requires_grad_(v.dtype.is_floating_point and k not in ('resample_filter', 'noise_const')) for k, v in sd.items()}
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    if name == 'torgb':
        yr = osg.torgb_layer(P, 'L.', xr, wr)
    else:
        yr = osg.synthesis_layer(P, 'L.', xr, wr, up=max(up, 1), noise_mode='const')
    dy = torch.randn(yr.shape, generator=g)
    pk = [k for k in P if P[k].requires_grad]
    gr = torch.autograd.grad(yr, [xr, wr] + [P[k] for k in pk], dy)
    xg, wg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    yg = L(xg, wg) if name == 'torgb' else L(xg, wg, noise_mode='const')
    params = dict(L.named_parameters())
    gg = torch.autograd.grad(yg, [xg, wg] + [params[k[2:]] for k in pk], dy.cuda())
    errs = {'y': rel(yg, yr), 'dx': rel(gg[0], gr[0]), 'dw': rel(gg[1], gr[1])}
    for k, a, b in zip(pk, gg[2:], gr[2:]):
        errs[k[2:]] = rel(a, b)
    bad = max(errs.values())
    print(f'N={n} {name:5s} {ci:3d}->{co:3d} res {res:3d}: ' + ' '.join(f'{k}={v:.1e}' for k, v in errs.items()) + ('   <<<<<<' if bad > 1e-4 else ''))
