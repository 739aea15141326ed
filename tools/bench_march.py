#!/usr/bin/env python3
"""Ray-march kernels in isolation, through the C ABI, HIP events around each launch.

Cases (S = 96 + 96 = 192 sorted samples behind a merge permutation, C = 32):
  fwd final      N=1 (16 384 rays): composite with colours                         bytes/ray = S*(32+2)*4 + 32*4 + 8
  fwd depth-only N=4 (65 536 rays): rgb = NULL, depth + wsum only                    bytes/ray = S*3*4 + 8      (density, depth, permutation)
  bwd dense      N=1: d_rgb + d_depth -> d_color_scale + d_densities, flags on       bytes/ray = S*34*4 + 33*4 + S*2*4   (bench.py's figure)
  bwd masked     N=4: the same with the gradient zero outside a box (25 % live)      bytes counted for LIVE rays only
  bwd depth-only N=4 masked: d_rgb = NULL                                            bytes/live ray = S*3*4 + 4 + S*4
usage: [SPI_HIP_LIB=/path/to/other/libspi_hip.so] bench_march.py [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spi_amd import hip
if os.environ.get('SPI_HIP_LIB'):                       # A/B against another build of the library
    hip.LIB_PATH = os.environ['SPI_HIP_LIB']

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
S, dev = 192, 'cuda'
torch.manual_seed(0)


PERM = (lambda p: None) if os.environ.get('NOPERM') else hip.ptr


def setup(n):
    R = n * 16384
    col = torch.rand(R, S, 32, device=dev); den = torch.randn(R, S, device=dev)
    dc = torch.sort(torch.rand(R, 96, device=dev) + 2.25, 1)[0].contiguous(); df = torch.sort(torch.rand(R, 96, device=dev) + 2.25, 1)[0].contiguous()
    dep = torch.empty(R, S, device=dev); perm = torch.empty(R, S, device=dev, dtype=torch.int32)
    hip.call('spi_merge_sort_depths', hip.ptr(dc), hip.ptr(df), R, 96, 96, hip.ptr(dep), hip.ptr(perm), hip.stream())
    return R, col, den, dep, perm


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    flush = torch.zeros(128 << 20, dtype=torch.int32, device=dev)
    for a, b in ev:
        # the 256 MB MALL / L2 must not serve the next launch.  DIRTY=1 flushes by writing instead of reading: every line the timed kernel allocates then
        # evicts a dirty one, which costs the kernels with allocating accesses 10-15 % (plain loads of a cold 430 MB: 3.5 instead of ~6 TB/s)
        if os.environ.get('DIRTY'):
            flush.zero_()
        else:
            flush.max()
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2] * 1e3, sum(ts) / len(ts) * 1e3


def report(name, us_med, us_avg, nbytes):
    print(f'{name:34s} median {us_med:8.1f} us  mean {us_avg:8.1f} us   {nbytes / 1e6:8.1f} MB   {nbytes / us_med / 1e3:7.0f} GB/s = {nbytes / us_med / 1e3 / 8000:.3f} of 8 TB/s', flush=True)


cl = torch.tensor([2.25, 3.3], device=dev)
for n in (1, 4):
    R, col, den, dep, perm = setup(n)
    rgb = torch.empty(R, 32, device=dev); d = torch.empty(R, device=dev); w = torch.empty(R, device=dev)
    if n == 1:
        t = timeit(lambda: hip.call('spi_raymarch_fwd', hip.ptr(col), hip.ptr(den), hip.ptr(dep), PERM(perm), hip.ptr(cl), R, S, S, 32, 0,
                                    hip.ptr(rgb), hip.ptr(d), None, hip.ptr(w), hip.stream()))
        report('fwd final N=1', *t, R * (S * 34 * 4 + 32 * 4 + 8))
    t = timeit(lambda: hip.call('spi_raymarch_fwd', None, hip.ptr(den), hip.ptr(dep), PERM(perm), hip.ptr(cl), R, S, S, 32, 0,
                                None, hip.ptr(d), None, hip.ptr(w), hip.stream()))
    report(f'fwd depth-only N={n}', *t, R * (S * 12 + 8))
    d_rgb = torch.randn(R, 32, device=dev); d_dep = torch.randn(R, device=dev)
    if n == 4:                                          # gradient only inside a box of every view: 64 x 64 of 128 x 128 rays
        m = torch.zeros(n, 128, 128, device=dev); m[:, 40:104, 30:94] = 1
        d_rgb *= m.reshape(R, 1); d_dep *= m.reshape(R)
    live = int((d_rgb != 0).any(1).sum())
    d_cs = torch.empty(R, S, device=dev); d_sig = torch.empty(R, S, device=dev); act = torch.empty(R, device=dev, dtype=torch.int32)
    t = timeit(lambda: hip.call('spi_raymarch_bwd', hip.ptr(col), hip.ptr(den), hip.ptr(dep), PERM(perm), hip.ptr(cl), hip.ptr(d_rgb), hip.ptr(d_dep), None,
                                R, S, S, 32, 0, None, hip.ptr(d_cs), hip.ptr(d_sig), hip.ptr(act), hip.stream()))
    torch.cuda.synchronize(); assert int(act.sum()) == live
    report(f'bwd N={n} ({live} of {R} rays live)', *t, live * (S * 34 * 4 + 33 * 4 + S * 8))
    t = timeit(lambda: hip.call('spi_raymarch_bwd', None, hip.ptr(den), hip.ptr(dep), PERM(perm), hip.ptr(cl), None, hip.ptr(d_dep), None,
                                R, S, S, 32, 0, None, None, hip.ptr(d_sig), hip.ptr(act), hip.stream()))
    report(f'bwd depth-only N={n} ({live} live)', *t, live * (S * 12 + 4 + S * 4))
    del col, den, dep, perm
