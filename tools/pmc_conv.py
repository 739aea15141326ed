#!/usr/bin/env python3
"""Launch the big 3x3 conv (128->128 @512^2) fwd/dgrad/wgrad a few times for rocprofv3 --pmc passes."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spi_amd import hip
from spi_amd.torch_utils.ops import conv2d_mfma as cm
N, I, O, H, k = 1, 128, 128, 512, 3
x = torch.randn(N, I, H, H, device='cuda'); w = torch.randn(N, O, I, k, k, device='cuda') * 0.05
y = torch.randn(N, O, H, H, device='cuda')
half = os.environ.get('SPI_BENCH_HALF') == '1'                   # fp16 activation tensors (with SPI_BENCH_F16=1)
if half:
    x, y = x.half(), y.half()
dx = torch.empty_like(x); dw = torch.empty_like(w)
d = cm._desc(N, I, O, H, H, k, 1, False, False, O * I * k * k, tap_major=1, f16=int(os.environ.get('SPI_BENCH_F16', '0')), half=half)
ws = cm._workspace(d, 0, x.device) if os.environ.get('SPI_BENCH_WINO', '1') != '0' else None     # same size for forward and dgrad
dwg = cm._desc(N, I, O, H, H, k, 1, False, False, O * I * k * k, tap_major=1, f16=int(os.environ.get('SPI_BENCH_F16', '0')), half=half)
wsw = cm._workspace(dwg, 2, x.device) if os.environ.get('SPI_BENCH_WINO', '1') != '0' else None     # weight gradient: its own opt-in / partial-sum buffer
for _ in range(3):
    hip.call('spi_conv2d_fwd', ctypes.byref(d), hip.ptr(x), hip.ptr(w), hip.ptr(y), hip.stream())
    hip.call('spi_conv2d_dgrad', ctypes.byref(d), hip.ptr(y), hip.ptr(w), hip.ptr(dx), hip.stream())
    hip.call('spi_conv2d_wgrad', ctypes.byref(dwg), hip.ptr(x), hip.ptr(y), hip.ptr(dw), hip.stream())
torch.cuda.synchronize()
