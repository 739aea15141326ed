#!/usr/bin/env python3
"""GB/s of spi_tail_bwd (layer-tail backward: activation gradient + bias / noise sums) at the generator's shapes; 12 B per element."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spi_amd import hip
if os.environ.get('SPI_HIP_LIB'):                       # A/B against another build of the library
    hip.LIB_PATH = os.environ['SPI_HIP_LIB']
from spi_amd.torch_utils.ops import bias_act as ba

dev = 'cuda'


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


print(f'{"case":40s} {"ms":>8s} {"GB/s":>8s}')
for N, C, H in ((1, 128, 512), (4, 128, 512), (1, 256, 256), (2, 256, 256), (1, 128, 256), (1, 512, 64), (1, 512, 16), (2, 64, 256)):
    dy = torch.randn(N, C, H, H, device=dev); y = torch.randn(N, C, H, H, device=dev)
    noise = torch.randn(H, H, device=dev); st = torch.ones(1, device=dev)
    for pix in (False, True):
        fn = lambda: ba.tail_backward(dy, y, noise if pix else None, st if pix else None, 3, 0.2, 1.414, 256.0, False, pix, True)
        ms = timeit(fn)
        print(f'{f"N={N} C={C} {H}^2 pixsum={pix}":40s} {ms:8.3f} {dy.numel() * 12 / 1e6 / ms:8.0f}')
