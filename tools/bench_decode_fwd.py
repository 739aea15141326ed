#!/usr/bin/env python3
"""Decoder forward (tri-plane gather + 32-64-33 MLP) at the benchmark size: the split-bf16 matrix-core kernel (decode_fwd_mfma_kernel)
against the vector-ALU kernel (decode_fwd_kernel, `spi_debug_set(256)`), same inputs: time per launch and the difference of the outputs.
  python tools/bench_decode_fwd.py [N images]      (run on the GPU box)"""
import os, sys, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spi_amd import hip
from spi_amd.utils import camera_utils as cu
from spi_amd.training.volumetric_rendering.ray_sampler import RaySampler


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = 'cuda'
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    torch.manual_seed(0)
    lib = hip.lib()
    lib.spi_debug_set.argtypes = [ctypes.c_int]
    planes = torch.randn(N, 3, 256, 256, 32, device=dev) * 0.5            # channels-last planes
    c = cu.cal_canonical_c(0.4, 0.0).repeat(N, 1).to(dev)
    ro, rd = RaySampler()(c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3), 128)
    M = ro.shape[1]
    w1t = torch.randn(32, 64, device=dev) / 32 ** 0.5
    b1 = torch.randn(64, device=dev) * 0.1
    w2 = torch.randn(33, 64, device=dev) / 8
    b2 = torch.randn(33, device=dev) * 0.1
    for S, tag in ((96, 'coarse pass (S = 96 of 192 rows per ray)'), (192, 'S = 192')):
        depths = torch.sort(torch.rand(N, M, S, device=dev) * 1.05 + 2.25, -1)[0].contiguous()
        outs = {}
        for flag, name in ((256, 'vector ALU'), (0, 'split-bf16 MFMA')):
            lib.spi_debug_set(flag)
            for with_rgb in (True, False):
                rgb = torch.zeros(N * M * 192, 32, device=dev) if with_rgb else None
                sigma = torch.zeros(N * M * 192, device=dev)

                def run():
                    hip.call('spi_triplane_decode_fwd', hip.ptr(planes), None, hip.ptr(ro), hip.ptr(rd), hip.ptr(depths), hip.ptr(w1t), hip.ptr(b1),
                             hip.ptr(w2), hip.ptr(b2), N, M * S, S, 256, 256, 1.0, 192, 0, hip.ptr(rgb) if with_rgb else None, hip.ptr(sigma), hip.stream())
                t = timeit(run)
                pts = N * M * S
                print(f'N={N} {tag} [{name}{"" if with_rgb else ", densities only"}]: {t * 1e3:.1f} us = {t * 1e3 / (pts / 3145728):.1f} us per 3.1 M points, '
                      f'{pts * 8320 / t / 1e9:.1f} TFLOP/s, gather {pts * 1536 / t / 1e6:.0f} GB/s')
                outs[(name, with_rgb)] = (rgb, sigma)
        lib.spi_debug_set(0)
        a, b = outs[('vector ALU', True)], outs[('split-bf16 MFMA', True)]
        print(f'   max |rgb diff| {(a[0] - b[0]).abs().max().item():.3e} (values in [-0.001, 1.001]), max |sigma diff| {(a[1] - b[1]).abs().max().item():.3e} '
              f'(max |sigma| {a[1].abs().max().item():.2f}); densities-only sigma diff {(outs[("vector ALU", False)][1] - outs[("split-bf16 MFMA", False)][1]).abs().max().item():.3e}')


if __name__ == '__main__':
    main()
