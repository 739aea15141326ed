#!/usr/bin/env python3
"""Does hipGraphLaunch of a graph that is still executing block the host?  (If it does, a loop that replays ONE graph per iteration can never run ahead of the GPU:
every iteration starts with the launch latency and the first nodes' submission exposed.)  Host time of the 2nd launch, same exec vs a second exec of the same work."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spi_amd  # noqa: F401
big = torch.zeros(64 * 1024 * 1024, device='cuda')
x = torch.zeros(1024, device='cuda')


def body():
    for _ in range(40):
        x.add_(1.0)
    for _ in range(20):
        big.add_(1.0)


def capture():
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    return g


ga, gb = capture(), capture()
for name, seq in (('same exec twice', (ga, ga)), ('two execs alternating', (ga, gb))):
    for g in seq:
        g.replay()
    torch.cuda.synchronize()
    host, wall = [], []
    for _ in range(10):
        t0 = time.perf_counter(); seq[0].replay(); t1 = time.perf_counter(); seq[1].replay(); t2 = time.perf_counter()
        torch.cuda.synchronize(); t3 = time.perf_counter()
        host.append((t1 - t0, t2 - t1)); wall.append(t3 - t0)
    print(f'{name}: host time of launch 1 {1e3 * sum(h[0] for h in host) / 10:.3f} ms, of launch 2 {1e3 * sum(h[1] for h in host) / 10:.3f} ms; both done after {1e3 * sum(wall) / 10:.3f} ms')
