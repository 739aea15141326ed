#!/usr/bin/env python3
"""Micro-benchmark of the renderer kernels at BASELINE config size (N images, 128^2 rays, 96+96 samples)."""
import os, sys, time, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from spi_amd import hip
if os.environ.get('SPI_HIP_LIB'):                       # A/B against another build of the library
    hip.LIB_PATH = os.environ['SPI_HIP_LIB']
from spi_amd.training.volumetric_rendering import renderer as R
from spi_amd.training.triplane import OSGDecoder
from spi_amd.utils import camera_utils as cu
from spi_amd.training.volumetric_rendering.ray_sampler import RaySampler


def timeit(fn, reps=5):
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
    dec = OSGDecoder(32, {'decoder_lr_mul': 1, 'decoder_output_dim': 32}).to(dev)
    planes = (torch.randn(N, 3, 32, 256, 256, device=dev) * 0.5).requires_grad_(True)
    c = cu.cal_canonical_c(0.4, 0.0).repeat(N, 1).to(dev)
    ro, rd = RaySampler()(c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3), 128)
    opts = dict(depth_resolution=96, depth_resolution_importance=96, ray_start=2.25, ray_end=3.3, box_warp=1, white_back=False)
    ren = R.ImportanceRenderer()
    lib = hip.lib()
    lib.spi_debug_set.argtypes = [ctypes.c_int]
    xi, u = torch.rand(N, 16384, 96, 1, device=dev), torch.rand(N * 16384, 96, device=dev)

    def fwd():
        return ren(planes, dec, ro, rd, opts, noise=(xi, u))
    print(f'N={N}: render fwd {timeit(lambda: fwd()):.3f} ms')
    for frozen in (True, False):
        for p in dec.parameters():
            p.requires_grad_(not frozen)
        for flags, tag in ((0, 'full'), (1, 'no scatter'), (1 + 4, 'no scatter, no MLP'), (1 + 4 + 64, 'no scatter, no MLP, no gather'), (1 + 64, 'no scatter, no gather'), (8 + 16 + 32, 'scatter loop without atomics'), (8, 'no flush'), (32, 'no LDS atomics'), (16, 'no slow path'), (128, 'no global atomics from the slow path')):
            lib.spi_debug_set(flags)
            rgb, depth, _ = fwd()
            g1, g2 = torch.randn_like(rgb), torch.randn_like(depth)
            t = timeit(lambda: torch.autograd.grad([rgb, depth], [planes] + ([] if frozen else list(dec.parameters())), [g1, g2], retain_graph=True))
            print(f'  render bwd decoder_frozen={frozen} [{tag}]: {t:.3f} ms')
    lib.spi_debug_set(0)
    # ray marcher alone at the materialised boundary
    S = 192
    col = torch.rand(N * 16384, S, 32, device=dev); den = torch.randn(N * 16384, S, device=dev); dep = torch.sort(torch.rand(N * 16384, S, device=dev) + 2.25, 1)[0]
    rgb = torch.empty(N * 16384, 32, device=dev); d = torch.empty(N * 16384, device=dev); w = torch.empty(N * 16384, S - 1, device=dev)
    cl = torch.tensor([2.25, 3.3], device=dev)
    t = timeit(lambda: hip.call('spi_raymarch_fwd', hip.ptr(col), hip.ptr(den), hip.ptr(dep), None, hip.ptr(cl), N * 16384, S, S, 32, 0, hip.ptr(rgb), hip.ptr(d), hip.ptr(w), None, hip.stream()), 20)
    by = N * 16384 * (S * 34 * 4 + (33 + S - 1) * 4)
    print(f'  raymarch fwd S=192: {t * 1e3:.1f} us  -> {by / t / 1e6:.0f} GB/s ({by / t / 1e6 / 80:.1f} % of 8 TB/s)')
    # through a sort permutation like the renderer's (two ascending runs of 96 merged), rgb only (what bench.py times)
    dc = torch.sort(torch.rand(N * 16384, 96, device=dev), 1)[0]; df = torch.sort(torch.rand(N * 16384, 96, device=dev), 1)[0]
    dep2, perm = torch.sort(torch.cat([dc, df], 1) + 2.25, 1)
    perm = perm.int().contiguous(); dep2 = dep2.contiguous()
    ws = torch.empty(N * 16384, device=dev)
    t = timeit(lambda: hip.call('spi_raymarch_fwd', hip.ptr(col), hip.ptr(den), hip.ptr(dep2), hip.ptr(perm), hip.ptr(cl), N * 16384, S, S, 32, 0, hip.ptr(rgb), hip.ptr(d), None, hip.ptr(ws), hip.stream()), 20)
    by = N * 16384 * (S * 34 * 4 + 34 * 4)
    print(f'  raymarch fwd S=192 via perm, no weights: {t * 1e3:.1f} us  -> {by / t / 1e6:.0f} GB/s ({by / t / 1e6 / 80:.1f} % of 8 TB/s)')


if __name__ == '__main__':
    main()
