"""Share of decoder-backward tiles (8x8 rays x 4 sorted samples) whose incoming gradients (d_sigma, d_colour rows) are exactly zero,
for the random-init generator of bench.py -- would sample-level skipping pay off on this workload?"""
import sys, torch
sys.path.insert(0, '.')
from spi_amd import hip
from spi_amd.training.triplane import TriPlaneGenerator, ffhq512_kwargs
from spi_amd.training.volumetric_rendering import renderer as R
from spi_amd.utils import camera_utils as cu
dev = 'cuda'
torch.manual_seed(0)
G = TriPlaneGenerator(**ffhq512_kwargs(depth_resolution=96, depth_resolution_importance=96)).eval().requires_grad_(False).to(dev)
G.neural_rendering_resolution = 128
c = cu.cal_canonical_c(0.2, 0.05).to(dev)
ws = torch.randn(1, 14, 512, device=dev) * 0.5
with torch.no_grad():
    planes = G._planes(ws, noise_mode='const')
    ro, rd = G.ray_sampler(c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3), 128)
    n, m, sc, sf = 1, 16384, 96, 96
    s = sc + sf
    params, gains = R._decoder_params(G.decoder)
    w1, b1, w2, b2 = params
    wg1, bg1, wg2, bg2 = gains
    dec = ((w1 * wg1).t().contiguous(), (b1 * bg1).contiguous(), (w2 * wg2).contiguous(), (b2 * bg2).contiguous())
    pn = R.planes_to_nhwc(planes)
    d_c = torch.empty(n, m, sc, device=dev); xi = torch.rand(n, m, sc, device=dev)
    hip.call('spi_coarse_depths', hip.ptr(xi), m, sc, 2.25, 3.3, hip.ptr(d_c), hip.stream())
    rgb_all = torch.empty(n, m, s, 32, device=dev); sig_all = torch.empty(n, m, s, device=dev)
    R._decode_fwd(pn, dec, rays=(ro.contiguous(), rd.contiguous()), depths=d_c, box_warp=1.0, out=(rgb_all, sig_all), out_S=s, out_off=0)
    w_c = torch.empty(n, m, sc - 1, device=dev)
    hip.call('spi_raymarch_fwd', None, hip.ptr(sig_all), hip.ptr(d_c), None, None, m, sc, s, 32, 0, None, None, hip.ptr(w_c), None, hip.stream())
    d_f = torch.empty(n, m, sf, device=dev); u = torch.rand(n, m, sf, device=dev)
    hip.call('spi_importance_sample', hip.ptr(d_c), hip.ptr(w_c), hip.ptr(u), m, sc, sf, hip.ptr(d_f), 1, hip.stream())
    R._decode_fwd(pn, dec, rays=(ro.contiguous(), rd.contiguous()), depths=d_f, box_warp=1.0, out=(rgb_all, sig_all), out_S=s, out_off=sc)
    d_all = torch.empty(n, m, s, device=dev); perm = torch.empty(n, m, s, device=dev, dtype=torch.int32)
    hip.call('spi_merge_sort_depths', hip.ptr(d_c), hip.ptr(d_f), m, sc, sf, hip.ptr(d_all), hip.ptr(perm), hip.stream())
    cl = R.depth_range(d_all)
    d_rgb = torch.randn(m, 32, device=dev); d_depth = torch.randn(m, device=dev)
    d_col = torch.zeros_like(rgb_all); d_sig = torch.zeros_like(sig_all); act = torch.empty(m, device=dev, dtype=torch.int32)
    hip.call('spi_raymarch_bwd', hip.ptr(rgb_all), hip.ptr(sig_all), hip.ptr(d_all), hip.ptr(perm), hip.ptr(cl), hip.ptr(d_rgb), hip.ptr(d_depth), None,
             m, s, s, 32, 0, hip.ptr(d_col), None, hip.ptr(d_sig), hip.ptr(act), hip.stream())
    # sorted order
    idx = perm.long()
    ds = torch.gather(d_sig, 2, idx)[0]                                    # [m, s]
    dc = torch.gather(d_col[0], 1, idx[0][..., None].expand(-1, -1, 32))   # [m, s, 32]
    zero = (ds == 0) & (dc == 0).all(-1)
    print(f'samples with an exactly-zero gradient: {zero.float().mean().item() * 100:.2f} %')
    tiles = zero.reshape(16, 8, 16, 8, 48, 4).permute(0, 2, 4, 1, 3, 5).reshape(-1, 256).all(1)
    print(f'tiles (8x8 rays x 4 samples) that are all zero: {tiles.float().mean().item() * 100:.2f} %')
    sg = torch.gather(sig_all, 2, idx)[0]
    print(f'sorted densities: mean {sg.mean().item():.3f}, max {sg.max().item():.3f}; softplus(sigma-1) mean {torch.nn.functional.softplus(sg - 1).mean().item():.3f}')
