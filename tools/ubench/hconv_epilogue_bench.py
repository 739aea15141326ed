#!/usr/bin/env python3
"""hconv_kernel forward with and without its fused epilogue (noise, bias, lrelu, gain, clamp) on the SR conv1 shapes: what the element-wise code paths cost."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spi_amd import hip
from spi_amd.torch_utils.ops import conv2d_mfma as cm


def t(fn, r=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(r):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / r * 1e3


for n, c, h in ((1, 128, 512), (2, 128, 512), (1, 256, 256)):
    x = torch.randn(n, c, h, h, device='cuda').half()
    w = torch.randn(n, c, 3, 3, c, device='cuda') * 0.03
    y = torch.empty_like(x)
    b, nz, ng = torch.randn(c, device='cuda'), torch.randn(h, h, device='cuda'), torch.ones(1, device='cuda')
    res = []
    for epi in (False, True):
        d = cm._desc(n, c, c, h, h, 3, 1, False, True, c * c * 9, *((b, nz, ng, 3, 0.2, 1.414, 256.0) if epi else ()), tap_major=1, f16=1, half=True)
        ws = cm._workspace(d, 0, x.device)
        res.append(t(lambda: hip.call('spi_conv2d_fwd', ctypes.byref(d), hip.ptr(x), hip.ptr(w), hip.ptr(y), hip.stream())))
    fl = 2.0 * n * c * c * 9 * h * h
    print(f'{n}x{c}x{h}: plain {res[0]:.1f} us ({fl / res[0] / 1e6:.0f} TF/s)   noise + bias + lrelu + gain + clamp {res[1]:.1f} us ({fl / res[1] / 1e6:.0f} TF/s)')
