#!/usr/bin/env python3
"""Device-to-device copies of odd sizes as nodes of a captured HIP graph (tensor.copy_ / clone of small contiguous tensors)."""
import torch
dev = 'cuda'
for n in (1, 3, 9, 16, 25, 33, 100, 400, 1000, 4096):
    src = torch.arange(n, device=dev, dtype=torch.float32) + 1
    dst = torch.zeros(n, device=dev)
    def body():
        dst.copy_(src)                 # contiguous same-dtype copy -> hipMemcpyAsync (a memcpy node when captured)
        c = src.clone()
        return (dst * 2 + c).sum()
    body(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s = body()
    vals = []
    for rep in range(5):
        src.add_(1)                    # new contents every replay
        g.replay(); torch.cuda.synchronize()
        vals.append((s.item() - 3 * src.sum().item(), int((dst != src).sum().item())))
    print(n, '(result error, wrong elements) per replay:', vals, flush=True)
