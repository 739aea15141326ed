"""How many points of a decode_bwd tile (8x8 ray patch x 4 sorted samples) leave the LDS scatter window, per plane, for candidate window
shapes / placements.  Geometry from a real render of the random-init generator (same as bench.py)."""
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
yaw = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
c = cu.cal_canonical_c(yaw, 0.05).to(dev)
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
    d_all = torch.empty(n, m, s, device=dev); perm = torch.empty(n, m, s, device=dev, dtype=torch.int32)
    hip.call('spi_merge_sort_depths', hip.ptr(d_c), hip.ptr(d_f), m, sc, sf, hip.ptr(d_all), hip.ptr(perm), hip.stream())
    pts = (ro[0, :, None, :] + d_all[0, :, :, None] * rd[0, :, None, :]) * 2.0           # [m, s, 3] in [-1, 1]
    # tiles: 8x8 ray patches x 4 consecutive sorted samples
    P = pts.reshape(16, 8, 16, 8, 48, 4, 3).permute(0, 2, 4, 1, 3, 5, 6).reshape(16 * 16 * 48, 256, 3)
    planes_uv = [(0, 1), (0, 2), (2, 0)]
    print(f'yaw {yaw}: {P.shape[0]} tiles')
    P44 = pts.reshape(32, 4, 32, 4, 12, 16, 3).permute(0, 2, 4, 1, 3, 5, 6).reshape(32 * 32 * 12, 256, 3)     # 4x4 rays x 16 samples
    P28 = pts.reshape(64, 2, 64, 2, 3, 64, 3).permute(0, 2, 4, 1, 3, 5, 6).reshape(64 * 64 * 3, 256, 3)        # 2x2 rays x 64 samples
    for tname, PP, shapes in (('4x4 rays x 16 samples', P44, ((16, 16), (6, 42), (42, 6))), ('4x4 rays x 16 samples', P44, ((16, 16), (8, 32), (32, 8))),
                              ('2x2 rays x 64 samples', P28, ((16, 16), (4, 64), (64, 4)))):
        out = []
        for pl, (a, b) in enumerate(planes_uv):
            ix = ((PP[..., a] + 1) * 256 - 1) * 0.5; iy = ((PP[..., b] + 1) * 256 - 1) * 0.5
            x0, y0 = ix.floor(), iy.floor()
            vin = (x0 + 1 >= 0) & (x0 < 256) & (y0 + 1 >= 0) & (y0 < 256)
            ww, wh = shapes[pl]
            cnt = vin.sum(1).clamp_min(1)
            cx = ((x0 * vin).sum(1) / cnt).trunc(); cy = ((y0 * vin).sum(1) / cnt).trunc()
            ox = cx - ww // 2 + 1; oy = cy - wh // 2 + 1
            lx = x0 - ox[:, None]; ly = y0 - oy[:, None]
            fast = vin & (lx >= 0) & (lx + 1 < ww) & (ly >= 0) & (ly + 1 < wh) & (x0 >= 0) & (x0 + 1 < 256) & (y0 >= 0) & (y0 + 1 < 256)
            touched = []
            out.append(((vin & ~fast).sum().item() / vin.sum().item()))
        print(f'  tile {tname}, windows {shapes}: slow share per plane ' + ', '.join(f'{o * 100:5.1f} %' for o in out))
    for name, (wx, wy), anchor in (('16x16 mean', (16, 16), 'mean'), ('12x12 mean', (12, 12), 'mean'), ('16x16 bbox-centre', (16, 16), 'bbox'),
                                   ('12x21 (z long) mean', (0, 0), 'mean'), ('8x32 (z long) mean', (0, 0), 'mean'), ('10x25 (z long) mean', (0, 0), 'mean'), ('16x16 median', (16, 16), 'median')):
        out = []
        for pl, (a, b) in enumerate(planes_uv):
            ix = ((P[..., a] + 1) * 256 - 1) * 0.5; iy = ((P[..., b] + 1) * 256 - 1) * 0.5
            x0, y0 = ix.floor(), iy.floor()
            vin = (x0 + 1 >= 0) & (x0 < 256) & (y0 + 1 >= 0) & (y0 < 256)
            if name.startswith('12x21'):
                ww, wh = (16, 16) if pl == 0 else ((12, 21) if pl == 1 else (21, 12))
            elif name.startswith('8x32'):
                ww, wh = (16, 16) if pl == 0 else ((8, 32) if pl == 1 else (32, 8))
            elif name.startswith('10x25'):
                ww, wh = (16, 16) if pl == 0 else ((10, 25) if pl == 1 else (25, 10))
            else:
                ww, wh = wx, wy
            cnt = vin.sum(1).clamp_min(1)
            if anchor == 'mean':
                cx = ((x0 * vin).sum(1) / cnt).trunc(); cy = ((y0 * vin).sum(1) / cnt).trunc()
            elif anchor == 'median':
                big = 1e9
                cx = torch.where(vin, x0, torch.full_like(x0, float('nan'))).nanmedian(1).values.nan_to_num(0)
                cy = torch.where(vin, y0, torch.full_like(y0, float('nan'))).nanmedian(1).values.nan_to_num(0)
            else:
                xmin = torch.where(vin, x0, torch.full_like(x0, 1e9)).amin(1); xmax = torch.where(vin, x0, torch.full_like(x0, -1e9)).amax(1)
                ymin = torch.where(vin, y0, torch.full_like(y0, 1e9)).amin(1); ymax = torch.where(vin, y0, torch.full_like(y0, -1e9)).amax(1)
                cx = ((xmin + xmax) / 2).floor(); cy = ((ymin + ymax) / 2).floor()
            ox = cx - ww // 2 + 1; oy = cy - wh // 2 + 1
            lx = x0 - ox[:, None]; ly = y0 - oy[:, None]
            fast = vin & (lx >= 0) & (lx + 1 < ww) & (ly >= 0) & (ly + 1 < wh) & (x0 >= 0) & (x0 + 1 < 256) & (y0 >= 0) & (y0 + 1 < 256)
            slow = vin & ~fast
            out.append((slow.sum().item() / vin.sum().item(), 1 - vin.float().mean().item()))
        print(f'  {name:22s}: slow share per plane ' + ', '.join(f'{o[0] * 100:5.1f} %' for o in out) + '   (outside the image: ' + ', '.join(f'{o[1] * 100:4.1f} %' for o in out) + ')')
