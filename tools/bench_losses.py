"""Forward + backward time of the two pseudo-view losses at the stage-2 batch (N=4, 512^2): LPIPS-VGG16 and BoxCX-VGG19."""
import sys, time, torch
sys.path.insert(0, '.')
from spi_amd.criteria.lpips.lpips import LPIPS
from spi_amd.criteria.bbox_cx_loss import BoxCXLoss
from spi_amd.data.images_dataset import SyntheticDataset
dev = 'cuda'
torch.manual_seed(0)
d = SyntheticDataset(1)[0]
lm = d['lm'].to(dev).float().reshape(1, 68, 2).repeat(4, 1, 1)
lp = LPIPS(net_type='vgg').to(dev).eval()
bx = BoxCXLoss().to(dev).eval()
plan = bx.plan(lm, dev)
x = (torch.rand(4, 3, 512, 512, device=dev) * 2 - 1).requires_grad_(True)
y = torch.rand(4, 3, 512, 512, device=dev) * 2 - 1
m = torch.zeros(4, 1, 512, 512, device=dev); m[:, :, 150:400, 120:380] = 1


def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


print(f'LPIPS N=4 fwd+bwd: {t(lambda: torch.autograd.grad(lp(x * m, y), x)):.2f} ms')
print(f'LPIPS N=1 fwd+bwd: {t(lambda: torch.autograd.grad(lp(x[:1], y[:1]), x)):.2f} ms')
print(f'BoxCX N=4 fwd+bwd: {t(lambda: torch.autograd.grad(bx(x * m, y, lm, plan=plan), x)):.2f} ms')
with torch.no_grad():
    print(f'BoxCX N=4 fwd only: {t(lambda: bx(x * m, y, lm, plan=plan)):.2f} ms')

if len(sys.argv) > 1:
    from torch.profiler import profile, ProfilerActivity
    feats = lp.features(y[:1])
    fn = (lambda: torch.autograd.grad(lp(x[:1], y_feats=feats), x)) if sys.argv[1] == 'lpips1' else (lambda: torch.autograd.grad(bx(x * m, y, lm, plan=plan), x))
    print(f'{sys.argv[1]}: {t(fn):.2f} ms')
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(5): fn()
        torch.cuda.synchronize()
    ev = [e for e in prof.key_averages() if e.device_time_total > 0 and e.cpu_time_total == 0]
    print(f'kernel time per call {sum(e.device_time_total for e in ev) / 5e3:.2f} ms')
    for e in sorted(ev, key=lambda e: -e.device_time_total)[:16]:
        print(f'{e.device_time_total / 5e3:8.3f} ms  n={e.count // 5:4d}  avg {e.device_time_total / e.count:8.1f} us  {e.key[:100]}')
