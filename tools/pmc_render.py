#!/usr/bin/env python3
"""One render forward + backward (decoder trainable) at config size for rocprofv3 --pmc passes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spi_amd.training.volumetric_rendering import renderer as R
from spi_amd.training.triplane import OSGDecoder
from spi_amd.utils import camera_utils as cu
from spi_amd.training.volumetric_rendering.ray_sampler import RaySampler
dev = 'cuda'
torch.manual_seed(0)
dec = OSGDecoder(32, {'decoder_lr_mul': 1, 'decoder_output_dim': 32}).to(dev)
planes = (torch.randn(1, 3, 32, 256, 256, device=dev) * 0.5).requires_grad_(True)
c = cu.cal_canonical_c(0.4, 0.0).to(dev)
ro, rd = RaySampler()(c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3), 128)
opts = dict(depth_resolution=96, depth_resolution_importance=96, ray_start=2.25, ray_end=3.3, box_warp=1, white_back=False)
ren = R.ImportanceRenderer()
xi, u = torch.rand(1, 16384, 96, 1, device=dev), torch.rand(16384, 96, device=dev)
for _ in range(3):
    rgb, depth, _w = ren(planes, dec, ro, rd, opts, noise=(xi, u))
    torch.autograd.grad([rgb, depth], [planes] + list(dec.parameters()), [torch.randn_like(rgb), torch.randn_like(depth)])
torch.cuda.synchronize()
