#!/usr/bin/env python3
"""Diagnostic (round 4): does a W-projector HIP graph stay valid across eager PTI iterations, and does a projector built AFTER them work?

    python tools/debug_graph_after_pti.py

Sequence: projector A (sg, w mode): eager step, capture, 2 replays | 4 PTI iterations (eager) | 2 more replays of A | projector B built now:
eager step, capture, 4 replays | 2 PTI iterations | 2 replays of B.  Prints the feature distance / finiteness of the latent after every step."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spi_amd.configs import global_config, hyperparameters, paths_config
from spi_amd.training.triplane import TriPlaneGenerator, ffhq512_kwargs
from spi_amd.training.projectors.common import Projection
from spi_amd.training.projectors.w_projector import sg_distance
from spi_amd.training.coaches.pti_coach import SingleIDCoach
from spi_amd.data.images_dataset import SyntheticDataset
import tempfile, contextlib

dev = torch.device('cuda:0')
global_config.device = str(dev)
tmp = tempfile.mkdtemp(prefix='spi_dbg_')
for k in ('checkpoints_dir', 'embedding_base_dir', 'experiments_output_dir', 'images_output_dir', 'mirror_images_output_dir'):
    setattr(paths_config, k, f'{tmp}/{k}/')
hyperparameters.first_inv_type, hyperparameters.G_1_type = 'sg', 'pti'
hyperparameters.LPIPS_value_threshold = -1.0
torch.manual_seed(0)
G = TriPlaneGenerator(**ffhq512_kwargs(narrow=bool(os.environ.get('NARROW')), depth_resolution=96, depth_resolution_importance=96)).eval().requires_grad_(False).to(dev)
G.neural_rendering_resolution = 128
d = SyntheticDataset(1)[0]
data = {k: (v[None] if torch.is_tensor(v) else v) for k, v in d.items()}
with contextlib.redirect_stdout(sys.stderr):
    coach = SingleIDCoach(None, False, G=G, synthetic=True)
image = data['img'].to(dev).float()
camera = torch.as_tensor(data['c']).to(dev).float().reshape(-1, 25)


def projector():
    return Projection(coach.G, camera, sg_distance(image, coach._sg_vgg16(), dev), w_mode=os.environ.get('WMODE', 'w'), initial_w=None, num_steps=500, w_avg_samples=600, device=dev)


def steps(tag, proj, first, n):
    for i in range(n):
        o = proj.step(first + i)
        torch.cuda.synchronize()
        print(f'{tag} step {first + i}: dist {float(o["dist"]):.6g} reg {float(o["reg"]):.6g} latent finite {bool(torch.isfinite(proj.w_opt).all())} graph {getattr(proj, "_graph", None) is not None}', flush=True)


A = projector()
w_pivot = A.w_opt.detach().repeat([1, coach.G.backbone.mapping.num_ws, 1]).clone()
target_feats = coach.lpips_loss.features(image)
steps('A', A, 25, 4)


def pti(n):
    part = os.environ.get('PTI_PART', 'all')
    for i in range(n):
        if part == 'all':
            stop, losses = coach.train_step(image, camera, w_pivot, target_feats)
            continue
        from spi_amd.torch_utils import zero_arena
        Gc = coach.G
        m = Gc.neural_rendering_resolution ** 2
        noise = (torch.rand(1, m, 96, 1, device=dev), torch.rand(m, 96, device=dev))
        if part == 'fwd_nograd':
            with torch.no_grad():
                img = Gc.synthesis(w_pivot.detach(), camera, noise_mode='const', render_noise=noise)['image']
            losses = dict(loss=img.mean())
            continue
        if part == 'fwd_nograd_loss':
            with torch.no_grad():
                img = Gc.synthesis(w_pivot.detach(), camera, noise_mode='const', render_noise=noise)['image']
                losses = dict(loss=coach.calc_loss(img, image, target_feats)[0])
            continue
        if part in ('syn_l2', 'syn_lpips', 'syn_sync_loss', 'syn_lpips_clone'):
            from spi_amd.criteria.l2_loss import l2_loss
            with torch.no_grad():
                img = Gc.synthesis(w_pivot.detach(), camera, noise_mode='const', render_noise=noise)['image']
                if part == 'syn_sync_loss':
                    torch.cuda.synchronize()
                    losses = dict(loss=coach.calc_loss(img, image, target_feats)[0])
                elif part == 'syn_l2':
                    losses = dict(loss=l2_loss(img, image))
                elif part == 'syn_lpips':
                    losses = dict(loss=coach.lpips_loss(img, image, y_feats=target_feats).sum())
                else:
                    losses = dict(loss=coach.lpips_loss(img.clone().clamp(-1, 1), image, y_feats=target_feats).sum())
            continue
        if part in ('syn_empty_lpips', 'lpips_syn', 'syn_vgg', 'syn_pool'):
            with torch.no_grad():
                if part == 'lpips_syn':
                    losses = dict(loss=coach.lpips_loss(image * 0.9, image, y_feats=target_feats).sum())
                img = Gc.synthesis(w_pivot.detach(), camera, noise_mode='const', render_noise=noise)['image']
                if part == 'syn_empty_lpips':
                    del noise
                    torch.cuda.synchronize(); torch.cuda.empty_cache()
                    losses = dict(loss=coach.lpips_loss(img, image, y_feats=target_feats).sum())
                elif part == 'syn_vgg':
                    fx = coach.lpips_loss.net(coach.lpips_loss._resize(img.float()))
                    losses = dict(loss=fx[0].mean())
                elif part == 'syn_pool':
                    x_ = coach.lpips_loss._resize(img.float())
                    losses = dict(loss=torch.nn.functional.max_pool2d(x_, 2).mean())
            continue
        if part.startswith('tiny'):
            nl = int(part[4:])
            t_ = torch.zeros(1024, device=dev)
            for _ in range(nl):
                t_.add_(1.0)
            losses = dict(loss=t_.mean())
            continue
        if part.startswith('convs'):
            nl = int(part[5:])
            from spi_amd.torch_utils.ops import conv2d_mfma
            x_ = torch.randn(1, 64, 64, 64, device=dev); w_ = torch.randn(64, 64, 3, 3, device=dev) * 0.05
            with torch.no_grad():
                for _ in range(nl):
                    y_ = conv2d_mfma.conv2d(x_, w_, padding=1)
            losses = dict(loss=y_.mean())
            continue
        if part == 'noise_loss':
            with torch.no_grad():
                losses = dict(loss=coach.calc_loss(torch.randn_like(image) * 2, image, target_feats)[0])
            continue
        if part == 'loss_only':
            with torch.no_grad():
                losses = dict(loss=coach.calc_loss(image * 0.9, image, target_feats)[0])
            continue
        zero_arena.begin(dev, key='pti')
        img = Gc.synthesis(w_pivot.detach(), camera, noise_mode='const', render_noise=noise)['image']
        loss = img.square().mean() if part.startswith('simple') else coach.calc_loss(img, image, target_feats)[0]
        losses = dict(loss=loss.detach())
        if part in ('fwd', 'simple_fwd'):
            continue
        coach.optimizer.zero_grad()
        loss.backward()
        if part in ('bwd', 'simple_bwd'):
            continue
        coach.optimizer.step()
    torch.cuda.synchronize()
    print(f'  {n} PTI iterations ({part}), loss {float(losses["loss"]):.5g}', flush=True)


pti(4)
steps('A', A, 29, 2)
B = projector()
steps('B', B, 25, 6)
pti(2)
steps('B', B, 31, 2)
