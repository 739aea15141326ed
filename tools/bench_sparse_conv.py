"""Dense vs sparse-gradient backward of the super-resolution convs at the rot-branch batch (N=4), box-masked dy."""
import sys, time, torch
sys.path.insert(0, '.')
from spi_amd.torch_utils.ops import conv2d_mfma as cm
dev = 'cuda'
torch.manual_seed(0)
for (n, i, o, h, k, tr, frac) in [(4, 128, 128, 512, 3, False, 0.35), (4, 128, 128, 512, 3, False, 0.1), (4, 256, 256, 256, 3, False, 0.35),
                                  (4, 256, 128, 256, 3, True, 0.35), (4, 128, 3, 512, 1, False, 0.35), (1, 128, 128, 512, 3, False, 1.0)]:
    x = torch.randn(n, i, h, h, device=dev, requires_grad=True)
    w = (torch.randn(n, o, k, k, i, device=dev) * 0.05).requires_grad_(True)
    y = cm.conv2d(x, w, padding=(0 if tr else k // 2), transposed=tr, flip=not tr, tap_major=True, sparse_grad=True)
    oh = y.shape[2]
    dy = torch.randn_like(y)
    side = int(oh * frac ** 0.5)
    m = torch.zeros(1, 1, oh, oh, device=dev); m[:, :, oh // 4: oh // 4 + side, oh // 5: oh // 5 + side] = 1
    dy = dy * m
    res = {}
    for mode in ('dense', 'sparse'):
        for which, wrt in (('dgrad', [x]), ('wgrad', [w])):
            def run():
                with cm.sparse_gradients(mode == 'sparse'):
                    return torch.autograd.grad(y, wrt, dy, retain_graph=True)
            for _ in range(2): run()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): g = run()
            torch.cuda.synchronize(); res[(mode, which)] = (time.perf_counter() - t0) / 5 * 1e3
    fl = cm.seg_flags(dy)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): cm.seg_flags(dy)
    torch.cuda.synchronize(); tf = (time.perf_counter() - t0) / 5 * 1e3
    print(f'N={n} {i}->{o} {h}^2 k{k} tr={tr} nonzero={fl.float().mean().item():.2f}: dgrad {res[("dense","dgrad")]:.2f} -> {res[("sparse","dgrad")]:.2f} ms, '
          f'wgrad {res[("dense","wgrad")]:.2f} -> {res[("sparse","wgrad")]:.2f} ms (flags pass {tf:.3f} ms, included in both sparse numbers)')
