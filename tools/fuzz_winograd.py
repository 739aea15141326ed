import ctypes, random, sys
import torch
sys.path.insert(0, '.')
from spi_amd import hip
from spi_amd.configs import global_config
from spi_amd.torch_utils.ops import conv2d_mfma as cm
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
dev = 'cuda'
bad = 0
n_cases = 0
while n_cases < 60:
    N = random.choice([1, 1, 2, 3])
    I = random.choice([8, 16, 24, 40, 64, 72, 128])
    O = random.choice([48, 64, 80, 100, 128, 130, 192])
    H, W = random.randint(40, 300), random.randint(40, 300)
    flip, per, epi = random.random() < 0.5, random.random() < 0.6, random.random() < 0.5
    d = cm._desc(N, I, O, H, W, 3, 1, False, flip, O * I * 9 if per else 0, tap_major=1)
    if hip.lib().spi_conv2d_workspace_bytes(ctypes.byref(d), 0) == 0:
        continue
    n_cases += 1
    g = torch.Generator().manual_seed(n_cases)
    x = torch.randn(N, I, H, W, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(*((N,) if per else ()), O, I, 3, 3, generator=g) / (I * 9) ** 0.5).to(dev).requires_grad_(True)
    kw = dict(padding=1, flip=flip)
    if epi:
        kw.update(bias=torch.randn(O, generator=g).to(dev), noise=torch.randn(H, W, generator=g).to(dev), noise_strength=torch.tensor(0.3, device=dev),
                  act=random.choice(['lrelu', 'linear', 'relu']), gain=1.3, clamp=random.choice([None, 2.0]))
    dy = torch.randn(N, O, H, W, generator=g).to(dev)
    outs = []
    for wino in (True, False):
        global_config.conv_winograd = wino
        y = cm.conv2d(x, w, **kw)
        gx, = torch.autograd.grad(y, [x], dy)
        outs.append((y.detach(), gx))
    global_config.conv_winograd = True
    ey = ((outs[0][0] - outs[1][0]).abs().max() / outs[1][0].abs().max()).item()
    diff = (outs[0][1] - outs[1][1]).abs() / outs[1][1].abs().max()
    eg = diff.max().item() if not epi else diff.median().item()
    ok = ey < 1e-5 and eg < 1e-5 and torch.isfinite(outs[0][0]).all() and torch.isfinite(outs[0][1]).all()
    if not ok:
        bad += 1
        print('FAIL', (N, I, O, H, W, flip, per, epi), ey, eg, flush=True)
print('cases', n_cases, 'failures', bad)
