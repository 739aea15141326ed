#!/usr/bin/env python3
"""Stage-1 projector at full size: eager steps vs HIP-graph replays on identical (fixed) draws -- losses and w+ after K steps."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spi_amd.configs import global_config
from spi_amd.training.triplane import TriPlaneGenerator, ffhq512_kwargs
from spi_amd.training.projectors.common import Projection
from spi_amd.training.projectors.mirror_projector import mirror_setup
from spi_amd.criteria.lpips.lpips import LPIPS
from spi_amd.criteria import weights as pretrained
from spi_amd.utils.rng import DeviceRNG
from spi_amd.utils import camera_utils as cu
from spi_amd.data.images_dataset import SyntheticDataset

dev = 'cuda'
K = int(os.environ.get('K', '16'))


class FixedDraws(DeviceRNG):
    def __init__(self, device):
        super().__init__(device)
        self.cache, self.gen = {}, torch.Generator().manual_seed(5)

    def _get(self, kind, shape):
        key = (kind, tuple(shape))
        if key not in self.cache:
            self.cache[key] = (torch.rand if kind == 'u' else torch.randn)(*shape, generator=self.gen).to(self.device)
        return self.cache[key]

    def rand(self, *shape):
        return self._get('u', shape)

    def randn(self, *shape):
        return self._get('n', shape)


d = SyntheticDataset(1)[0]
target = d['img'][None].to(dev).float()
c = torch.as_tensor(d['c']).reshape(1, 25).to(dev).float()
lp = LPIPS(net_type='vgg', weights=pretrained.lpips_vgg16_weights(True)).to(dev)
res = {}
for graph in (False, True):
    global_config.stage1_hip_graph = graph
    torch.manual_seed(0)
    G = TriPlaneGenerator(**ffhq512_kwargs(narrow=False, depth_resolution=96, depth_resolution_importance=96)).eval().requires_grad_(False).to(dev)
    G.neural_rendering_resolution = 128
    cameras, dist_fn = mirror_setup(target, c, lp, torch.device(dev))
    proj = Projection(G, cameras, dist_fn, w_mode='w+', initial_w=None, num_steps=500, w_avg_samples=64, device=torch.device(dev), rng=FixedDraws(dev))
    outs = [proj.step(i) for i in range(K)]           # no host sync between steps: replays run back to back like in bench.py
    torch.cuda.synchronize()
    res[graph] = ([o['loss'].item() for o in outs], proj.w_opt.detach().clone())
    print('graph' if graph else 'eager', ['%.6g' % v for v in res[graph][0]], 'captured:', getattr(proj, '_graph', None) is not None, flush=True)
a, b = res[True], res[False]
print('max rel loss diff', max(abs(x - y) / abs(y) for x, y in zip(a[0], b[0])), ' w rel diff', ((a[1] - b[1]).abs().max() / b[1].abs().max()).item())
