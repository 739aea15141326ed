"""Random weight-gradient problems: the F(3x3, 2x2) Winograd kernel (dense and with a masked gradient) against the implicit GEMM, through the C ABI.
usage: fuzz_winograd_wgrad.py [seed]   -- 80 cases per seed; also channel-split Winograd forward / dgrad shapes (128..255 blocks)."""
import ctypes, random, sys
import torch
sys.path.insert(0, '.')
from spi_amd import hip
from spi_amd.configs import global_config
from spi_amd.torch_utils.ops import conv2d_mfma as cm
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
dev = 'cuda'
bad = n_cases = 0
while n_cases < 80:
    N = random.choice([1, 1, 2, 3, 4])
    I = random.choice([32, 40, 64, 72, 96, 128, 200, 256])
    O = random.choice([32, 48, 64, 80, 100, 128, 130, 256])
    H, W = random.randint(16, 200), random.randint(32, 200)
    flip, per, tap = random.random() < 0.5, random.random() < 0.6, random.choice([0, 1])
    masked = random.random() < 0.5
    d = cm._desc(N, I, O, H, W, 3, 1, False, flip, O * I * 9 if per else 0, tap_major=tap)
    if hip.lib().spi_conv2d_workspace_bytes(ctypes.byref(d), 2) == 0:
        continue
    n_cases += 1
    g = torch.Generator().manual_seed(1000 + n_cases)
    x = torch.randn(N, I, H, W, generator=g).to(dev)
    dy = torch.randn(N, O, H, W, generator=g).to(dev)
    flags = None
    if masked:
        m = torch.zeros(N, 1, H, W, device=dev)
        for _ in range(random.randint(1, 4)):
            y0, x0 = random.randint(0, H - 1), random.randint(0, W - 1)
            m[random.randrange(N), :, y0:y0 + random.randint(1, H), x0:x0 + random.randint(1, W)] = 1
        dy = dy * m
        flags = cm.seg_flags(dy)
    shape = (*((N,) if per else ()), O, 3, 3, I) if tap else (*((N,) if per else ()), O, I, 3, 3)
    res = []
    for wino in (True, False):
        dd = cm._desc(N, I, O, H, W, 3, 1, False, flip, O * I * 9 if per else 0, tap_major=tap, dy_flags=flags)
        ws = None
        if wino:
            nb = hip.lib().spi_conv2d_workspace_bytes(ctypes.byref(dd), 2)
            ws = torch.empty(nb, device=dev, dtype=torch.uint8)
            dd.workspace, dd.workspace_bytes = ws.data_ptr(), nb
        dw = torch.full(shape, 3.0, device=dev)
        hip.call('spi_conv2d_wgrad', ctypes.byref(dd), hip.ptr(x), hip.ptr(dy), hip.ptr(dw), hip.stream())
        res.append(dw)
    scale = res[1].abs().max().clamp_min(1e-20)
    err = ((res[0] - res[1]).abs().max() / scale).item()
    ok = err < 2e-5 and bool(torch.isfinite(res[0]).all())
    if not ok:
        bad += 1
        print('FAIL wgrad', (N, I, O, H, W, flip, per, tap, masked), err, flush=True)
# channel-split forward / dgrad: shapes with 128..255 blocks of 16 x 16 x 64 and >= 256 input channels
n2 = 0
while n2 < 20:
    N = random.choice([1, 2, 4])
    I = random.choice([256, 320, 512])
    O = random.choice([128, 256, 512])
    H, W = random.randint(24, 80), random.randint(24, 80)
    blocks = ((H + 15) // 16) * ((W + 15) // 16) * ((O + 63) // 64) * N
    if not (128 <= blocks < 256):
        continue
    n2 += 1
    flip, per, epi = random.random() < 0.5, random.random() < 0.5, random.random() < 0.5
    g = torch.Generator().manual_seed(5000 + n2)
    x = torch.randn(N, I, H, W, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(*((N,) if per else ()), O, I, 3, 3, generator=g) / (I * 9) ** 0.5).to(dev).requires_grad_(True)
    kw = dict(padding=1, flip=flip)
    if epi:
        kw.update(bias=torch.randn(O, generator=g).to(dev), noise=torch.randn(H, W, generator=g).to(dev), noise_strength=torch.tensor(0.3, device=dev),
                  act='lrelu', gain=1.3, clamp=2.0)
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
    if not (ey < 1e-5 and eg < 1e-5):
        bad += 1
        print('FAIL split fwd/dgrad', (N, I, O, H, W, flip, per, epi), ey, eg, flush=True)
print('wgrad cases', n_cases, 'split cases', n2, 'failures', bad)
