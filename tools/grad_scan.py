#!/usr/bin/env python3
"""Diagnostic: gradient of EVERY generator parameter, HIP path vs the CPU oracle, for one synthesis + a random-projection loss.
    python tools/grad_scan.py [narrow|full] [nrr] [depth]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
from synth_weights import load_manifest, synth_state_dict
from oracle import renderer_ref as orr
from spi_amd.training.triplane import TriPlaneGenerator, ffhq512_kwargs
from spi_amd.utils import camera_utils as cu

kind = sys.argv[1] if len(sys.argv) > 1 else 'narrow'
nrr = int(sys.argv[2]) if len(sys.argv) > 2 else 64
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 12
torch.set_num_threads(32)
man = load_manifest(kind)
P = synth_state_dict(man)
pn = [k for k in man if not (k.endswith('noise_const') or k.endswith('resample_filter') or k.endswith('w_avg') or '.mapping.' in k)]
for k in pn:
    P[k].requires_grad_(True)
G = TriPlaneGenerator(**ffhq512_kwargs(narrow=(kind == 'narrow'), depth_resolution=depth, depth_resolution_importance=depth)).eval()
G.load_state_dict({k: v.detach() for k, v in P.items()})
G = G.to('cuda')
G.neural_rendering_resolution = nrr
g = torch.Generator().manual_seed(int(os.environ.get('SEED', 16)))
ws = torch.randn(1, 14, 512, generator=g)
c = cu.cal_canonical_c(0.4, 0.0)
m = nrr * nrr
xi, u = torch.rand(1, m, depth, 1, generator=g), torch.rand(m, depth, generator=g)
opts = dict(orr.DEFAULT_RENDERING, depth_resolution=depth, depth_resolution_importance=depth)
ref = orr.synthesis(P, ws, c, opts, neural_rendering_resolution=nrr, xi=xi, u=u)
d_img = torch.randn(ref['image'].shape, generator=g)
d_dep = torch.randn(ref['image_depth'].shape, generator=g)
mode = os.environ.get('LOSS', 'both')
def loss_of(o, dev):
    l = 0
    if mode in ('both', 'img'):
        l = l + (o['image'] * d_img.to(dev)).mean()
    if mode in ('both', 'dep'):
        l = l + (o['image_depth'] * d_dep.to(dev)).mean()
    if mode == 'smooth':                          # coherent cotangents: lrelu kink flips (a ~1e-6 fraction of elements) stay at the 1e-6 level
        l = (o['image'] ** 2).mean() + (o['image_depth'] ** 2).mean()
    return l
gref = torch.autograd.grad(loss_of(ref, 'cpu'), [P[k] for k in pn], allow_unused=True)
params = dict(G.named_parameters())
out = G.synthesis(ws.cuda(), c.cuda(), noise_mode='const', render_noise=(xi, u))
for k in ('image', 'image_raw', 'image_depth'):
    print(f'fwd {k}: {((out[k].cpu() - ref[k]).abs().max() / ref[k].abs().max()).item():.2e}')
pl = G._planes(ws.cuda(), noise_mode='const')
print(f'fwd planes: {((pl.cpu() - ref["planes"]).abs().max() / ref["planes"].abs().max()).item():.2e}')
ggpu = torch.autograd.grad(loss_of(out, 'cuda'), [params[k] for k in pn], allow_unused=True)
rows = []
for k, a, b in zip(pn, ggpu, gref):
    if a is None or b is None:
        continue
    e = ((a.cpu().double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()
    rows.append((e, k, tuple(b.shape)))
for e, k, s in sorted(rows, reverse=True):
    print(f'{e:10.2e}  {k}  {s}')
