#!/usr/bin/env python3
"""Time the stage-2 super-cycle (4 iterations) with individual every-4th-step branches switched off (debugging aid)."""
import os, sys, time, tempfile
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spi_amd import hip
from spi_amd.configs import hyperparameters as hp, paths_config, global_config
from spi_amd.training.triplane import TriPlaneGenerator, ffhq512_kwargs
from spi_amd.training.coaches.rot_bbox_cx_coach import RotBboxCoach
from spi_amd.data.images_dataset import SyntheticDataset

dev = torch.device('cuda:0')
global_config.device = str(dev)
tmp = tempfile.mkdtemp(prefix='spi_bt_')
for k in ('checkpoints_dir', 'embedding_base_dir', 'experiments_output_dir', 'images_output_dir', 'mirror_images_output_dir'):
    setattr(paths_config, k, f'{tmp}/{k}/')
hp.LPIPS_value_threshold = -1.0
torch.manual_seed(0)
G = TriPlaneGenerator(**ffhq512_kwargs(depth_resolution=96, depth_resolution_importance=96)).eval().requires_grad_(False).to(dev)
G.neural_rendering_resolution = 128
coach = RotBboxCoach(None, False, G=G, synthetic=True)
d = SyntheticDataset(1)[0]
data = {k: (v[None] if torch.is_tensor(v) else v) for k, v in d.items()}
ctx = coach.prepare_image(data)
w_pivot = torch.randn(1, 14, 512, device=dev) * 0.5


def cycle(reps=2):
    for i in range(4):
        coach.train_step(i, ctx, w_pivot)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(reps):
        for i in range(4):
            coach.train_step(i, ctx, w_pivot)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


full = dict(pt_rot_lambda=0.1, pt_mirror_rot_lambda=0.05, pt_depth_lambda=1.0)
for k, v in full.items():
    setattr(hp, k, v)
t_all = cycle()
print(f'all branches: {t_all:.1f} ms per 4-iteration cycle')
for off in full:
    for k, v in full.items():
        setattr(hp, k, 0.0 if k == off else v)
    t = cycle()
    print(f'  without {off:22s}: {t:.1f} ms  (branch costs {t_all - t:.1f} ms)')
for k in full:
    setattr(hp, k, 0.0)
t = cycle()
print(f'  main view only: {t:.1f} ms per cycle = {t / 4:.1f} ms per plain iteration')
