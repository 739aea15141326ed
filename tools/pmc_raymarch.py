#!/usr/bin/env python3
"""Launch the final ray march (S=192, C=32, 16384 rays, through a sort permutation) a few times: target of the
rocprofv3 --pmc passes that give roofline.traffic (tools/pmc_raymarch.sh)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spi_amd import hip

R, S = 16384, 192
dev = 'cuda'
torch.manual_seed(0)
col = torch.rand(R, S, 32, device=dev); den = torch.randn(R, S, device=dev)
dc = torch.sort(torch.rand(R, 96, device=dev) + 2.25, 1)[0].contiguous(); df = (torch.rand(R, 96, device=dev) + 2.25).contiguous()
dep = torch.empty(R, S, device=dev); perm = torch.empty(R, S, device=dev, dtype=torch.int32)
hip.call('spi_merge_sort_depths', hip.ptr(dc), hip.ptr(df), R, 96, 96, hip.ptr(dep), hip.ptr(perm), hip.stream())
rgb = torch.empty(R, 32, device=dev); d = torch.empty(R, device=dev); w = torch.empty(R, device=dev)
cl = torch.tensor([2.25, 3.3], device=dev)
for use_perm in (1, 0):
    for _ in range(5):
        hip.call('spi_raymarch_fwd', hip.ptr(col), hip.ptr(den), hip.ptr(dep), hip.ptr(perm) if use_perm else None, hip.ptr(cl), R, S, S, 32, 0,
                 hip.ptr(rgb), hip.ptr(d), None, hip.ptr(w), hip.stream())
torch.cuda.synchronize()
print('done')
