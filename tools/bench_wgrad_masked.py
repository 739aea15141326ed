#!/usr/bin/env python3
"""Masked weight gradient: F(3x3,2x2) Winograd kernel with row skipping vs the sparse implicit GEMM, through the C ABI (kernel time only)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spi_amd import hip
from spi_amd.torch_utils.ops import conv2d_mfma as cm


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


dev = 'cuda'
for (n, i, o, h, per) in [(4, 256, 256, 128, False), (4, 128, 128, 256, False), (4, 128, 128, 512, False), (1, 128, 128, 512, True), (1, 256, 256, 256, True)]:
    x = torch.randn(n, i, h, h, device=dev)
    dw = torch.empty(*((n,) if per else ()), o, 3, 3, i, device=dev)
    for frac in (1.0, 0.35, 0.1):
        dy = torch.randn(n, o, h, h, device=dev)
        if frac < 1:
            side = int(h * frac ** 0.5)
            m = torch.zeros(1, 1, h, h, device=dev); m[:, :, h // 4: h // 4 + side, h // 5: h // 5 + side] = 1
            dy = dy * m
        flags = cm.seg_flags(dy) if frac < 1 else None
        res = []
        for wino in (False, True):
            d = cm._desc(n, i, o, h, h, 3, 1, False, True, o * i * 9 if per else 0, tap_major=1, dy_flags=flags)
            ws = cm._workspace(d, 2, dev) if wino else None
            res.append(timeit(lambda: hip.call('spi_conv2d_wgrad', ctypes.byref(d), hip.ptr(x), hip.ptr(dy), hip.ptr(dw), hip.stream())))
        print(f'N={n} {i}->{o} @{h}^2 per_sample={per} nonzero={frac:4.2f}: implicit GEMM {res[0]:.3f} ms, winograd {res[1]:.3f} ms')
