#!/usr/bin/env python3
"""Micro-benchmark of the MFMA conv kernels on the generator's layer shapes (TFLOP/s per direction)."""
import ctypes, os, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spi_amd import hip
if os.environ.get('SPI_HIP_LIB'):                       # A/B against another build of the library
    hip.LIB_PATH = os.environ['SPI_HIP_LIB']
from spi_amd.torch_utils.ops import conv2d_mfma as cm

SHAPES = [  # name, N, I, O, H, k, transposed, per_sample
    ('sr1.conv1 128->128 @512', 1, 128, 128, 512, 3, False, True),
    ('sr0.conv1 256->256 @256', 1, 256, 256, 256, 3, False, True),
    ('sr1.conv0 256->128 up 256->513', 1, 256, 128, 256, 3, True, True),
    ('sr0.conv0 32->256 up 128->257', 1, 32, 256, 128, 3, True, True),
    ('b256.conv0 256->128 up 128->257', 1, 256, 128, 128, 3, True, True),
    ('sr1.conv0 N=4 256->128 up 256->513', 4, 256, 128, 256, 3, True, True),
    ('b256.conv1 128->128 @256', 1, 128, 128, 256, 3, False, True),
    ('b128.conv1 256->256 @128', 1, 256, 256, 128, 3, False, True),
    ('b64.conv1 512->512 @64', 1, 512, 512, 64, 3, False, True),
    ('b32.conv1 512->512 @32', 1, 512, 512, 32, 3, False, True),
    ('b16.conv1 512->512 @16', 1, 512, 512, 16, 3, False, True),
    ('b8.conv1 512->512 @8', 1, 512, 512, 8, 3, False, True),
    ('b4.conv1 512->512 @4', 1, 512, 512, 4, 3, False, True),
    ('b64.conv0 512->512 up 32->65', 1, 512, 512, 32, 3, True, True),
    ('b256.torgb 128->96 @256', 1, 128, 96, 256, 1, False, True),
    ('sr1.torgb 128->3 @512', 1, 128, 3, 512, 1, False, True),
    ('b64.conv1 N=4', 4, 512, 512, 64, 3, False, True),
    ('sr1.conv1 N=4', 4, 128, 128, 512, 3, False, True),
    ('vgg 3->64 @256 N=4', 4, 3, 64, 256, 3, False, False),
    ('vgg 64->64 @256 N=4', 4, 64, 64, 256, 3, False, False),
    ('vgg 512->512 @32 N=4', 4, 512, 512, 32, 3, False, False),
    ('vgg 64->64 @256 N=1', 1, 64, 64, 256, 3, False, False),
    ('vgg 64->128 @128 N=1', 1, 64, 128, 128, 3, False, False),
    ('vgg 128->128 @128 N=1', 1, 128, 128, 128, 3, False, False),
    ('vgg 128->256 @64 N=1', 1, 128, 256, 64, 3, False, False),
    ('vgg 256->256 @64 N=1', 1, 256, 256, 64, 3, False, False),
    ('vgg 256->512 @32 N=1', 1, 256, 512, 32, 3, False, False),
    ('vgg 512->512 @32 N=1', 1, 512, 512, 32, 3, False, False),
    ('vgg 512->512 @16 N=1', 1, 512, 512, 16, 3, False, False),
    ('vgg 64->64 @256 N=2', 2, 64, 64, 256, 3, False, False),
    ('vgg 128->128 @128 N=2', 2, 128, 128, 128, 3, False, False),
    ('vgg 256->256 @64 N=2', 2, 256, 256, 64, 3, False, False),
]


def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = 'cuda'
    print(f'{"layer":34s} {"GF":>7s} | {"fwd ms":>8s} {"TF/s":>6s} | {"dgrad ms":>8s} {"TF/s":>6s} | {"wgrad ms":>8s} {"TF/s":>6s}')
    only = os.environ.get('SPI_BENCH_ONLY')
    for name, N, I, O, H, k, tr, per in SHAPES:
        if only and only not in name:
            continue
        half = os.environ.get('SPI_BENCH_HALF') == '1'          # fp16 activation TENSORS (with SPI_BENCH_F16=1): spi_conv_desc.act_dtype
        if half and (I % 16 or O % 16):
            continue
        x = torch.randn(N, I, H, H, device=dev)
        w = torch.randn(*((N,) if per else ()), O, I, k, k, device=dev) * 0.05
        pad = k // 2 if not tr else 0
        oh = cm.out_size(H, k, pad, tr)
        y = torch.randn(N, O, oh, oh, device=dev)
        if half:
            x, y = x.half(), y.half()
        dx = torch.empty_like(x); dw = torch.empty_like(w)
        wbs = O * I * k * k if per else 0
        d = cm._desc(N, I, O, H, H, k, pad, tr, False, wbs, tap_major=1, f16=int(os.environ.get('SPI_BENCH_F16', '0')), half=half)
        s = hip.stream()
        wino = os.environ.get('SPI_BENCH_WINO', '1') != '0'
        dg = cm._desc(N, I, O, H, H, k, pad, tr, False, wbs, tap_major=1, f16=int(os.environ.get('SPI_BENCH_F16', '0')), half=half)
        wsf = cm._workspace(d, 0, x.device) if wino else None
        wsg = cm._workspace(dg, 1, x.device) if wino else None
        name = name + (' [W]' if wsf is not None else '')
        flops = 2.0 * N * O * I * k * k * (H * H if tr else oh * oh)
        reps = 3 if flops > 2e10 else 10
        f = timeit(lambda: hip.call('spi_conv2d_fwd', ctypes.byref(d), hip.ptr(x), hip.ptr(w), hip.ptr(y), s), reps)
        g = timeit(lambda: hip.call('spi_conv2d_dgrad', ctypes.byref(dg), hip.ptr(y), hip.ptr(w), hip.ptr(dx), s), reps)
        dwg = cm._desc(N, I, O, H, H, k, pad, tr, False, wbs, tap_major=1, f16=int(os.environ.get('SPI_BENCH_F16', '0')), half=half)
        wsw = cm._workspace(dwg, 2, x.device) if (wino and os.environ.get('SPI_BENCH_WINO_WGRAD', '1') != '0') else None
        if wsw is not None:
            name += '[Wg]'
        h = timeit(lambda: hip.call('spi_conv2d_wgrad', ctypes.byref(dwg), hip.ptr(x), hip.ptr(y), hip.ptr(dw), s), reps) if per else float('nan')
        tf = lambda ms: flops / ms / 1e9
        print(f'{name:34s} {flops / 1e9:7.1f} | {f:8.3f} {tf(f):6.1f} | {g:8.3f} {tf(g):6.1f} | {h:8.3f} {tf(h):6.1f}', flush=True)


if __name__ == '__main__':
    main()
