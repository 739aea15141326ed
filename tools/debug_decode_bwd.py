#!/usr/bin/env python3
"""Compare the tiled MFMA decoder backward with the generic VALU kernel on random inputs (debugging aid)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spi_amd import hip

dev = 'cuda'
torch.manual_seed(0)
N, res, S, H = 1, 16, 8, 64
M = res * res
planes = torch.randn(N, 3, H, H, 32, device=dev) * 0.5
ray_o = torch.randn(N, M, 3, device=dev) * 0.05 + torch.tensor([0., 0., -0.6], device=dev)
ray_d = torch.nn.functional.normalize(torch.randn(N, M, 3, device=dev) * 0.2 + torch.tensor([0., 0., 1.], device=dev), dim=-1)
depths = torch.sort(torch.rand(N, M, S, device=dev) * 1.0 + 0.1, dim=-1)[0].contiguous()
w1t = torch.randn(32, 64, device=dev) * 0.3; b1 = torch.randn(64, device=dev) * 0.1
w2 = torch.randn(33, 64, device=dev) * 0.3; b2 = torch.randn(33, device=dev) * 0.1
d_rgb = torch.randn(N, M, S, 32, device=dev); d_sig = torch.randn(N, M, S, device=dev)
if len(sys.argv) > 1 and sys.argv[1] == 'sigonly':
    d_rgb.zero_()
if len(sys.argv) > 1 and sys.argv[1] == 'rgbonly':
    d_sig.zero_()

# the forward's colour rows (the tiled backward reads the colour layer's sigmoid back from them)
colors = torch.empty(N, M, S, 32, device=dev); sig_fwd = torch.empty(N, M, S, device=dev)
hip.call('spi_triplane_decode_fwd', hip.ptr(planes), None, hip.ptr(ray_o), hip.ptr(ray_d), hip.ptr(depths), hip.ptr(w1t), hip.ptr(b1),
         hip.ptr(w2), hip.ptr(b2), N, M * S, S, H, H, 1.0, 0, 0, hip.ptr(colors), hip.ptr(sig_fwd), hip.stream())
dp_ref = torch.zeros_like(planes)
dump = torch.zeros(193, N * M * S, device=dev)
hip.call('spi_triplane_decode_bwd', hip.ptr(planes), None, hip.ptr(ray_o), hip.ptr(ray_d), hip.ptr(depths), hip.ptr(w1t), hip.ptr(b1),
         hip.ptr(w2), hip.ptr(b2), hip.ptr(d_rgb), hip.ptr(d_sig), N, M * S, S, H, H, 1.0, 0, 0, hip.ptr(dp_ref), hip.ptr(dump), hip.stream())
ref_w = [torch.empty(64, 32, device=dev), torch.empty(64, device=dev), torch.empty(33, 64, device=dev), torch.empty(33, device=dev)]
hip.call('spi_decoder_wgrad', hip.ptr(dump), N * M * S, *[hip.ptr(g) for g in ref_w], hip.stream())

for wgrad in (False, True):
    dp = torch.zeros_like(planes)
    ws = torch.empty(hip.lib().spi_triplane_decode_bwd_sorted_ws(N, M, S, res), device=dev)
    gw = [torch.empty(64, 32, device=dev), torch.empty(64, device=dev), torch.empty(33, 64, device=dev), torch.empty(33, device=dev)]
    hip.call('spi_triplane_decode_bwd_sorted', hip.ptr(planes), hip.ptr(ray_o), hip.ptr(ray_d), hip.ptr(depths), None, hip.ptr(w1t), hip.ptr(b1),
             hip.ptr(w2), hip.ptr(b2), hip.ptr(d_rgb), None, hip.ptr(colors), hip.ptr(d_sig), N, M, S, res, H, H, 1.0, hip.ptr(dp), hip.ptr(ws),
             *([hip.ptr(g) for g in gw] if wgrad else [None] * 4), None, hip.stream())
    torch.cuda.synchronize()
    e = (dp - dp_ref).abs()
    print(f'wgrad={wgrad}: d_planes max|ref| {dp_ref.abs().max():.4e} max err {e.max():.4e} mean err {e.mean():.4e} frac>1e-4*max {(e > 1e-4 * dp_ref.abs().max()).float().mean():.4f}')
    ech = e.amax(dim=(0, 1, 2, 3))
    print('  per-channel max err:', ' '.join(f'{v:.1e}' for v in ech.tolist()))
    if wgrad:
        for nm, a, b in zip(('dw1', 'db1', 'dw2', 'db2'), gw, ref_w):
            print(f'  {nm}: max|ref| {b.abs().max():.4e} max err {(a - b).abs().max():.4e}')
        e2 = (gw[2] - ref_w[2]).abs()
        print('  dw2 row err:', ' '.join(f'{v:.1e}' for v in e2.amax(dim=1).tolist()))
        e1 = (gw[0] - ref_w[0]).abs()
        print('  dw1 row err:', ' '.join(f'{v:.1e}' for v in e1.amax(dim=1).tolist()))
