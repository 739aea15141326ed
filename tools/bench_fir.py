#!/usr/bin/env python3
"""Micro-benchmark of spi_upfirdn2d (4x4 low-pass) at the generator's shapes: GB/s of algorithmic traffic."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spi_amd.torch_utils.ops import upfirdn2d as U

dev = torch.device('cuda')
f = U.setup_filter([1, 3, 3, 1]).to(dev)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


print(f'{"case":44s} {"ms":>8s} {"GB/s":>8s}')
for name, N, C, H, pad, up, down in [
    ('fwd up-layer tail 128ch 513->512', 1, 128, 513, [1, 1, 1, 1], 1, 1),
    ('fwd up-layer tail 128ch 513->512 N=4', 4, 128, 513, [1, 1, 1, 1], 1, 1),
    ('fwd up-layer tail 256ch 257->256', 1, 256, 257, [1, 1, 1, 1], 1, 1),
    ('fwd up-layer tail 512ch 65->64', 1, 512, 65, [1, 1, 1, 1], 1, 1),
    ('bwd of tail 128ch 512->513', 1, 128, 512, [2, 2, 2, 2], 1, 1),
    ('bwd of tail 128ch 512->513 N=4', 4, 128, 512, [2, 2, 2, 2], 1, 1),
    ('img upsample 96ch 128->256', 1, 96, 128, [2, 1, 2, 1], 2, 1),
    ('img upsample 3ch 256->512', 1, 3, 256, [2, 1, 2, 1], 2, 1),
    ('bwd img upsample 96ch 256->128', 1, 96, 256, [1, 2, 1, 2], 1, 2),
]:
    x = torch.randn(N, C, H, H, device=dev)
    noise = torch.randn(H - 1, H - 1, device=dev) if (up == 1 and down == 1 and pad[0] == 1) else None
    b = torch.randn(C, device=dev) if noise is not None else None
    if noise is not None:
        fn = lambda: U.upfirdn2d_bias_act(x, f, noise=noise, noise_strength=torch.ones((), device=dev), bias=b, padding=pad, gain=4, act='lrelu', act_gain=1.414, clamp=256)
    else:
        fn = lambda: U.upfirdn2d(x, f, up=up, down=down, padding=pad, gain=up * up)
    with torch.no_grad():
        y = fn()
        ms = timeit(fn)
    gb = (x.numel() + y.numel()) * 4 / 1e9
    print(f'{name:44s} {ms:8.3f} {gb / ms * 1e3:8.0f}')
# calibration: plain device copy / elementwise of the same footprint
for N in (1, 4):
    x = torch.randn(N, 128, 512, 512, device=dev); y = torch.empty_like(x)
    ms = timeit(lambda: y.copy_(x)); print(f'{"torch copy_ 128ch 512^2 N=%d" % N:44s} {ms:8.3f} {2 * x.numel() * 4 / 1e9 / ms * 1e3:8.0f}')
    ms = timeit(lambda: torch.mul(x, 2.0, out=y)); print(f'{"torch mul 128ch 512^2 N=%d" % N:44s} {ms:8.3f} {2 * x.numel() * 4 / 1e9 / ms * 1e3:8.0f}')
