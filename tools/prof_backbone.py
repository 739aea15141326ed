"""Device-time breakdown of one backbone (StyleGAN2 synthesis -> tri-planes) forward + backward with trainable weights."""
import sys, time, torch
sys.path.insert(0, '.')
from torch.profiler import profile, ProfilerActivity
from spi_amd.training.triplane import TriPlaneGenerator, ffhq512_kwargs
dev = 'cuda'
torch.manual_seed(0)
G = TriPlaneGenerator(**ffhq512_kwargs(depth_resolution=12, depth_resolution_importance=12)).eval().to(dev)
ws = torch.randn(1, 14, 512, device=dev) * 0.5
bb = [p for k, p in G.named_parameters() if k.startswith('backbone.')]
g = torch.randn(1, 3, 32, 256, 256, device=dev)


def step():
    planes = G._planes(ws, noise_mode='const')
    return torch.autograd.grad(planes, bb, grad_outputs=g, allow_unused=True)


for _ in range(3):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize(); print(f'backbone fwd+bwd: {(time.perf_counter() - t0) * 100:.2f} ms')
with torch.no_grad():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        G._planes(ws, noise_mode='const')
    torch.cuda.synchronize(); print(f'backbone fwd only (no grad): {(time.perf_counter() - t0) * 100:.2f} ms')
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(5):
        step()
    torch.cuda.synchronize()
ev = [e for e in prof.key_averages() if e.device_time_total > 0 and e.cpu_time_total == 0]
tot = sum(e.device_time_total for e in ev)
print(f'kernel time per step {tot / 5e3:.2f} ms')
for e in sorted(ev, key=lambda e: -e.device_time_total)[:28]:
    print(f'{e.device_time_total / 5e3:8.3f} ms  n={e.count // 5:4d}  avg {e.device_time_total / e.count:8.1f} us  {e.key[:110]}')
