#!/usr/bin/env python3
"""How fast does this ROCm replay the nodes of a HIP graph?  A chain of n tiny dependent kernels (x += 1 on 1 K floats), eager vs captured, one chain vs two
independent chains on two streams inside one graph; and the same with a 100 us kernel in front (does the runtime submit ahead while the GPU is busy?).
Run on the GPU box:  python tools/ubench/graph_node_rate.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spi_amd  # noqa: F401  (sets DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 before HIP starts, like the loops)

dev = 'cuda'
x = torch.zeros(1024, device=dev); y = torch.zeros(1024, device=dev)
big = torch.zeros(64 * 1024 * 1024, device=dev)


def chain(t, n):
    for _ in range(n):
        t.add_(1.0)


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for n in (100, 400):
    te = timed(lambda: chain(x, n))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chain(x, n)
    tg = timed(g.replay)
    s2 = torch.cuda.Stream()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        cur = torch.cuda.current_stream()
        s2.wait_stream(cur)
        with torch.cuda.stream(s2):
            chain(y, n // 2)
        chain(x, n // 2)
        cur.wait_stream(s2)
    tg2 = timed(g2.replay)
    gb = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gb):
        big.add_(1.0)                      # ~100 us of GPU work first: can the runtime queue the chain behind it?
        chain(x, n)
    tb = timed(gb.replay)
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        big.add_(1.0)
    t1 = timed(g1.replay)
    print(f'n = {n}: eager {te / n * 1e6:.1f} us per launch | graph {tg / n * 1e6:.1f} us per node | two parallel chains of {n // 2} in one graph {tg2 / n * 1e6:.1f} us per node '
          f'| behind a {t1 * 1e6:.0f} us kernel: {(tb - t1) / n * 1e6:.1f} us per node', flush=True)
print('DEBUG_CLR_GRAPH_PACKET_CAPTURE =', os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE'))
