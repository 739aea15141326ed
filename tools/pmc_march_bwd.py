#!/usr/bin/env python3
"""Launch the march backward (S=192, 16 384 dense rays, d_rgb + d_depth -> d_color_scale + d_densities) a few times: PMC target (tools/pmc_march_bwd.sh)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spi_amd import hip
R, S, dev = 16384, 192, 'cuda'
torch.manual_seed(0)
col = torch.rand(R, S, 32, device=dev); den = torch.randn(R, S, device=dev)
dc = torch.sort(torch.rand(R, 96, device=dev) + 2.25, 1)[0].contiguous(); df = torch.sort(torch.rand(R, 96, device=dev) + 2.25, 1)[0].contiguous()
dep = torch.empty(R, S, device=dev); perm = torch.empty(R, S, device=dev, dtype=torch.int32)
hip.call('spi_merge_sort_depths', hip.ptr(dc), hip.ptr(df), R, 96, 96, hip.ptr(dep), hip.ptr(perm), hip.stream())
cl = torch.tensor([2.25, 3.3], device=dev)
d_rgb = torch.randn(R, 32, device=dev); d_dep = torch.randn(R, device=dev)
d_cs = torch.empty(R, S, device=dev); d_sig = torch.empty(R, S, device=dev); act = torch.empty(R, device=dev, dtype=torch.int32)
for _ in range(6):
    hip.call('spi_raymarch_bwd', hip.ptr(col), hip.ptr(den), hip.ptr(dep), hip.ptr(perm), hip.ptr(cl), hip.ptr(d_rgb), hip.ptr(d_dep), None,
             R, S, S, 32, 0, None, hip.ptr(d_cs), hip.ptr(d_sig), hip.ptr(act), hip.stream())
torch.cuda.synchronize()
print('done')
