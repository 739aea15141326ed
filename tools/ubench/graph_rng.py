#!/usr/bin/env python3
"""torch.rand / torch.randn inside a captured HIP graph: value range and replay-to-replay variation."""
import torch
dev = 'cuda'
torch.manual_seed(0)
for shape in [(1000,), (1, 16384, 96, 1), (16384, 96)]:
    for _ in range(2):
        torch.rand(*shape, device=dev)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        x = torch.rand(*shape, device=dev)
        y = torch.randn(*shape, device=dev)
    for rep in range(4):
        g.replay(); torch.cuda.synchronize()
        print(shape, rep, 'rand min %.4f max %.4f mean %.4f | randn mean %.4f std %.4f first %.5f' % (x.min().item(), x.max().item(), x.mean().item(), y.mean().item(), y.std().item(), x.flatten()[0].item()), flush=True)
