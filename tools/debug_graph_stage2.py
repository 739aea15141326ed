#!/usr/bin/env python3
"""Capture pieces of the stage-2 iteration in a HIP graph one at a time (which one breaks hipStreamEndCapture?)."""
import os, sys, tempfile
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from spi_amd.configs import hyperparameters, paths_config, global_config
from spi_amd.training.coaches.rot_bbox_cx_coach import RotBboxCoach
from spi_amd.training.triplane import TriPlaneGenerator, ffhq512_kwargs
from spi_amd.data.images_dataset import SyntheticDataset
from spi_amd.utils.rng import DeviceRNG
from spi_amd.criteria.l2_loss import l2_loss

what = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
dev = 'cuda'
tmp = tempfile.mkdtemp()
for k in ('checkpoints_dir', 'embedding_base_dir', 'experiments_output_dir', 'images_output_dir', 'mirror_images_output_dir'):
    setattr(paths_config, k, f'{tmp}/{k}/')
hyperparameters.first_inv_type, hyperparameters.G_1_type = 'mir', 'RotBbox'
hyperparameters.pt_rot_lambda, hyperparameters.pt_mirror_rot_lambda, hyperparameters.pt_depth_lambda = [float(v) for v in os.environ.get('LAMBDAS', '0.1,0.05,1.0').split(',')]
global_config.stage2_hip_graph = False
if os.environ.get('THR'):
    hyperparameters.LPIPS_value_threshold = float(os.environ['THR'])
FOREACH = os.environ.get('NO_FOREACH')
if FOREACH:
    def _loop_add(a, b):
        for x, y in zip(a, b):
            x.add_(y)
    torch._foreach_add_ = _loop_add
torch.manual_seed(0)
FULL = bool(os.environ.get('FULL'))
WIDE = bool(int(os.environ.get('WIDE', '1' if FULL else '0')))
DEPTH = int(os.environ.get('DEPTH', '96' if FULL else '12'))
NRR = int(os.environ.get('NRR', '128' if FULL else '64'))
global_config.exploit_sparsity = os.environ.get('SPARSE', '1') != '0'
G = TriPlaneGenerator(**ffhq512_kwargs(narrow=not WIDE, depth_resolution=DEPTH, depth_resolution_importance=DEPTH)).eval().requires_grad_(False).to(dev)
G.neural_rendering_resolution = NRR
coach = RotBboxCoach(None, False, G=G, synthetic=True)
data = SyntheticDataset(1)[0]
data = {k: (v[None] if torch.is_tensor(v) else v) for k, v in data.items()}
ctx = coach.prepare_image(data)
w = torch.randn(1, 14, 512, device=dev)
rng = DeviceRNG(dev)
for i in range(int(os.environ.get('N_EAGER', '2'))):
    coach.train_step(i, ctx, w, rng=rng)
torch.cuda.synchronize()
print('eager ok', flush=True)
Gt = coach.G
params = [p for p in Gt.parameters() if p.requires_grad]


FLAG = torch.zeros(1, device=dev, dtype=torch.uint8)
gen_ = torch.Generator().manual_seed(3)
with torch.no_grad():
    DEPTH_MAIN = coach._synth(coach.G, w, ctx['camera'], DeviceRNG(dev))['image_depth'].detach()
IMG4 = (torch.rand(4, 3, 512, 512, generator=gen_) * 2 - 1).to(dev)
TGT4 = (torch.rand(4, 3, 512, 512, generator=gen_) * 2 - 1).to(dev)
MASK4 = (torch.rand(4, 1, 512, 512, generator=gen_) > 0.5).float().to(dev)
KEEP = []


def piece():
    if what == 'fullflag':
        r_ = coach._forward_backward(2, ctx, w, rng, flag_buf=FLAG)
        KEEP.append(r_)
        return r_[1]['l2']
    if what == 'full':
        flag = torch.zeros(1, device=dev, dtype=torch.uint8)
        return coach._forward_backward(2, ctx, w, rng, flag_buf=flag)[1]['l2']
    if what == 'zero':
        coach.optimizer.zero_grad(); return torch.zeros(1, device=dev)
    if what == 'planes':
        planes = Gt._planes(w, noise_mode='const'); leaf = planes.detach().requires_grad_(True); Gt._last_planes = leaf
        out = coach._synth(Gt, w, ctx['camera'], rng, use_cached_backbone=True); Gt._last_planes = None
        g = torch.autograd.grad(l2_loss(out['image'], ctx['image']), [leaf] + [p for k, p in Gt.named_parameters() if p.requires_grad and not k.startswith('backbone.')], allow_unused=True)
        g2 = torch.autograd.grad(planes, [p for k, p in Gt.named_parameters() if p.requires_grad and k.startswith('backbone.')], grad_outputs=g[0], allow_unused=True)
        return g[0].sum()
    if what == 'fwd':
        with torch.no_grad():
            return coach._synth(Gt, w, ctx['camera'], rng)['image'].abs().sum()
    if what in ('mirror_fwd', 'mirror_fb', 'mirror_fb_lpips', 'rot_fb'):
        from spi_amd.utils.camera_utils import sample_surrounding_camera
        from spi_amd.utils.rotate import rotate
        from spi_amd.torch_utils.ops.conv2d_mfma import sparse_gradients
        rot_bs = 4
        mir = what.startswith('mirror')
        cam0 = ctx['camera_m'] if mir else ctx['camera']
        cams_m = sample_surrounding_camera(cam0, batch_size=rot_bs, yaw_range=ctx['yaw_range'], pitch_range=0.1, rand=(rng.rand(rot_bs, 1), rng.rand(rot_bs, 1)))
        warp = {}

        def region(out):
            warp['img'], warp['mask'] = rotate(target_camera=cams_m, target_depth=out['image_depth'], src_image=(ctx['image_m'] if mir else ctx['image']).repeat(rot_bs, 1, 1, 1),
                                               src_camera=cam0.repeat(rot_bs, 1), src_depth=(torch.flip(DEPTH_MAIN, dims=[3]) if mir else DEPTH_MAIN).repeat(rot_bs, 1, 1, 1),
                                               src_mask=(ctx['face_mask_m'] if mir else ctx['face_mask']).repeat(rot_bs, 1, 1, 1), EPS=5e-2,
                                               src_cam2world_inv=ctx['camera_m_inv' if mir else 'camera_inv'])
            return warp['mask']
        if what == 'mirror_fwd':
            with torch.no_grad():
                gm = coach._synth(Gt, w, cams_m, rng, sr_region_fn=region)
                return torch.stack([gm['image'].abs().mean(), warp['mask'].mean()])
        planes = Gt._planes(w, noise_mode='const'); leaf = planes.detach().requires_grad_(True); Gt._last_planes = leaf
        gm = coach._synth(Gt, w, cams_m, rng, sr_region_fn=region, use_cached_backbone=True); Gt._last_planes = None
        if what == 'mirror_fb':
            flip_warp, flip_mask = torch.flip(warp['img'], dims=[3]), torch.flip(warp['mask'], dims=[3])
            l = coach.box_cx_loss(torch.flip(gm['image'], dims=[3]) * flip_mask, flip_warp, ctx['lm'].repeat(rot_bs, 1, 1), plan=ctx.get('box_plan'))
        else:
            l = coach.lpips_loss(gm['image'] * warp['mask'], warp['img'])
        with sparse_gradients(True):
            g = torch.autograd.grad(l, [leaf] + [p for k, p in Gt.named_parameters() if p.requires_grad and not k.startswith('backbone.')], allow_unused=True)
        return torch.stack([l.detach(), g[0].abs().sum(), warp['mask'].mean()])
    if what == 'boxcx':
        x = IMG4.detach().requires_grad_(True)
        l = coach.box_cx_loss(torch.flip(x, dims=[3]) * MASK4, TGT4, ctx['lm'].repeat(4, 1, 1), plan=ctx.get('box_plan'))
        gx, = torch.autograd.grad(l, [x])
        return torch.stack([l.detach(), gx.abs().sum()])
    if what == 'fwd_stats':
        with torch.no_grad():
            o = coach._synth(Gt, w, ctx['camera'], rng)
            return torch.stack([o['image_depth'].min(), o['image_depth'].max(), o['image_depth'].mean(), o['image_raw'].abs().mean(), o['image'].abs().mean()])
    if what == 'fwd_depth':
        with torch.no_grad():
            return coach._synth(Gt, w, ctx['camera'], rng)['image_depth'].abs().sum()
    if what == 'fwd_raw':
        with torch.no_grad():
            o = coach._synth(Gt, w, ctx['camera'], rng)
            return o['image_raw'].abs().sum() + o['image_depth'].abs().sum()
    if what == 'fwd_grad':
        return coach._synth(Gt, w, ctx['camera'], rng)['image'].sum()
    out = coach._synth(Gt, w, ctx['camera'], rng)
    if what == 'bwd_l2':
        loss = l2_loss(out['image'], ctx['image'])
    elif what == 'bwd_lpips':
        loss = torch.squeeze(coach.lpips_loss(out['image'], y_feats=ctx['target_feats']))
    elif what == 'bwd_raw':
        loss = out['image_raw'].square().mean()
    elif what == 'bwd_depth':
        loss = out['image_depth'].square().mean()
    else:
        raise SystemExit('unknown piece')
    g = torch.autograd.grad(loss, params, allow_unused=True)
    return loss


if what == 'direct':
    global_config.stage2_hip_graph = True
    coach._g2_key = (id(ctx), w.data_ptr(), id(coach.G), id(coach.optimizer))
    coach._g2 = {'plain': dict(eager=1, graph=None)}
    print(coach._graph_train_step(2, ctx, w, rng)[0], flush=True)
    raise SystemExit(0)
if what == 'direct2':
    flag = torch.zeros(1, device=coach.device, dtype=torch.uint8)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        _, losses = coach._forward_backward(2, ctx, w, rng, flag_buf=flag)
    print('direct2 captured', flush=True)
    raise SystemExit(0)
if what in ('step', 'step_fixed'):
    class FixedDraws(DeviceRNG):
        def __init__(self, device):
            super().__init__(device)
            self.cache, self.gen = {}, torch.Generator().manual_seed(9)

        def rand(self, *shape):
            if shape not in self.cache:
                self.cache[shape] = torch.rand(*shape, generator=self.gen).to(self.device)
            return self.cache[shape]
    if what == 'step_fixed':
        rng = FixedDraws(dev)
        for i in (0, 1):
            coach.train_step(i, ctx, w, rng=rng)
    global_config.stage2_hip_graph = os.environ.get('GRAPH', '1') != '0'
    hyperparameters.LPIPS_value_threshold = float(os.environ.get('THR', '-1'))
    for i in range(int(os.environ.get('I0', '0')), int(os.environ.get('I1', '10'))):
        stop, losses = coach.train_step(i, ctx, w, rng=rng)
        torch.cuda.synchronize()
        if os.environ.get('DUMP'):
            torch.save({k: (p.grad.norm().item(), p.grad.abs().max().item()) for k, p in coach.G.named_parameters() if p.grad is not None}, f"{os.environ['DUMP']}_{i}.pt")
        print(i, stop, {k: round(float(v), 5) for k, v in losses.items()}, list(getattr(coach, '_g2', {}).keys()), flush=True)
        if os.environ.get('DEL'):
            del losses
    raise SystemExit(0)
if os.environ.get('FIXED'):
    class FixedDraws2(DeviceRNG):
        def __init__(self, device):
            super().__init__(device)
            self.cache, self.gen = {}, torch.Generator().manual_seed(9)

        def rand(self, *shape):
            if shape not in self.cache:
                self.cache[shape] = torch.rand(*shape, generator=self.gen).to(self.device)
            return self.cache[shape]
    rng = FixedDraws2(dev)
if what.endswith('_noeager'):
    what = what[:-8]
else:
    for _ in range(2):
        print('piece eager', piece().flatten().tolist(), flush=True)
    torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    r = piece()
print('captured', flush=True)
for rep in range(4):
    g.replay(); torch.cuda.synchronize()
    print('replayed', rep, what, r.flatten().tolist(), flush=True)
