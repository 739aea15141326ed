"""Oracle (CPU, plain torch) for the EG3D volumetric renderer and TriPlaneGenerator.synthesis.

TEST INFRASTRUCTURE -- see oracle/__init__.py.

Reference lines followed (relative to /root/reference/eg3d/training):
  rays                 volumetric_rendering/ray_sampler.py:24-63
  coarse depths        volumetric_rendering/renderer.py:169-192 (scalar, per-ray tensor and disparity-space branches)
  'auto' ray limits    volumetric_rendering/renderer.py:91-97, math_utils.py:44-94 (ray / box slab test), :98-118 (linspace)
  density noise        volumetric_rendering/renderer.py:146-147
  tri-plane gather     volumetric_rendering/renderer.py:23-65
  OSG decoder          triplane.py:112-135
  ray march            volumetric_rendering/ray_marcher.py:25-57
  importance sampling  volumetric_rendering/renderer.py:194-253
  merge                volumetric_rendering/renderer.py:157-167
  render               volumetric_rendering/renderer.py:88-140
  synthesis            triplane.py:53-89
Random draws are *injected* (`xi` for the coarse jitter, `u` for the inverse-CDF) so the GPU
path can be compared on identical inputs; None draws them from torch's global generator in
the reference's order (rand_like [N,M,Sc,1] then rand [N*M,Sf]).
"""
import math
import torch
import torch.nn.functional as F

from . import stylegan_ref as sg

DEFAULT_RENDERING = dict(
    depth_resolution=48, depth_resolution_importance=48, ray_start=2.25, ray_end=3.3, box_warp=1.0,
    white_back=False, clamp_mode='softplus', disparity_space_sampling=False, density_noise=0.0,
    superresolution_noise_mode='none',
)


def ray_sampler(cam2world, intrinsics, resolution):
    n = cam2world.shape[0]
    dev = cam2world.device
    fx, fy = intrinsics[:, 0, 0:1], intrinsics[:, 1, 1:2]
    cx, cy, sk = intrinsics[:, 0, 2:3], intrinsics[:, 1, 2:3], intrinsics[:, 0, 1:2]
    t = (torch.arange(resolution, dtype=torch.float32, device=dev) + 0.5) / resolution
    v, u = torch.meshgrid(t, t, indexing='ij')           # row-major pixels: x = column
    u = u.reshape(1, -1).expand(n, -1)
    v = v.reshape(1, -1).expand(n, -1)
    x = (u - cx + cy * sk / fy - sk * v / fy) / fx
    y = (v - cy) / fy
    pts = torch.stack([x, y, torch.ones_like(x), torch.ones_like(x)], dim=-1)   # [N,M,4]
    world = torch.bmm(cam2world, pts.transpose(1, 2)).transpose(1, 2)[:, :, :3]
    origin = cam2world[:, :3, 3]
    dirs = F.normalize(world - origin[:, None, :], dim=2)
    return origin[:, None, :].expand(-1, dirs.shape[1], -1).contiguous(), dirs


def ray_limits_box(ray_o, ray_d, box_side_length):
    """math_utils.get_ray_limits_box: slab test of every ray against the axis-aligned cube of side `box_side_length` centred at the
    origin -> (t_near [N,M,1], t_far [N,M,1]); rays that miss get (-1, -2)."""
    shape = ray_o.shape
    o, d = ray_o.detach().reshape(-1, 3), ray_d.detach().reshape(-1, 3)
    half = box_side_length / 2
    bounds = torch.tensor([[-half] * 3, [half] * 3], dtype=o.dtype, device=o.device)
    valid = torch.ones(o.shape[0], dtype=torch.bool, device=o.device)
    inv = 1 / d
    sign = (inv < 0).long()

    def slab(ax):
        near = (bounds.index_select(0, sign[:, ax])[:, ax] - o[:, ax]) * inv[:, ax]
        far = (bounds.index_select(0, 1 - sign[:, ax])[:, ax] - o[:, ax]) * inv[:, ax]
        return near, far
    tmin, tmax = slab(0)
    for ax in (1, 2):
        near, far = slab(ax)
        valid[torch.logical_or(tmin > far, near > tmax)] = False
        tmin, tmax = torch.max(tmin, near), torch.min(tmax, far)
    tmin[~valid] = -1
    tmax[~valid] = -2
    return tmin.reshape(*shape[:-1], 1), tmax.reshape(*shape[:-1], 1)


def auto_ray_limits(ray_o, ray_d, box_warp):
    """renderer.py:91-97: per-ray limits from the box; rays that miss get [min valid start, max valid START] (sic, :96)."""
    start, end = ray_limits_box(ray_o, ray_d, box_warp)
    ok = end > start
    if torch.any(ok).item():
        start[~ok] = start[ok].min()
        end[~ok] = start[ok].max()
    return start, end


def coarse_depths(n, m, s, ray_start, ray_end, xi=None, device='cpu', disparity=False):
    """sample_stratified (renderer.py:169-192).  ray_start / ray_end: scalars, or per-ray tensors [N,M,1] ('auto')."""
    if disparity:
        base = torch.linspace(0, 1, s, device=device).reshape(1, 1, s, 1).repeat(n, m, 1, 1)
        if xi is None:
            xi = torch.rand_like(base)
        t = base + xi * (1 / (s - 1))
        return 1. / (1. / ray_start * (1. - t) + 1. / ray_end * t)
    if torch.is_tensor(ray_start):
        steps = (torch.arange(s, dtype=torch.float32, device=ray_start.device) / (s - 1)).reshape(s, 1, 1, 1)
        base = (ray_start[None] + steps * (ray_end - ray_start)[None]).permute(1, 2, 0, 3)          # math_utils.linspace -> [N,M,S,1]
        if xi is None:
            xi = torch.rand_like(base)
        return base + xi * ((ray_end - ray_start) / (s - 1))[..., None]
    base = torch.linspace(ray_start, ray_end, s, device=device).reshape(1, 1, s, 1).repeat(n, m, 1, 1)
    if xi is None:
        xi = torch.rand_like(base)
    return base + xi * ((ray_end - ray_start) / (s - 1))


def sample_planes(planes, coords, box_warp=1.0):
    """planes [N,3,C,H,W]; coords [N,P,3] -> [N,3,P,C].  Plane k is sampled at (x,y),(x,z),(z,x)."""
    n, _, c, h, w = planes.shape
    q = coords * (2.0 / box_warp)
    gx = torch.stack([q[..., 0], q[..., 0], q[..., 2]], dim=1)      # [N,3,P]
    gy = torch.stack([q[..., 1], q[..., 2], q[..., 0]], dim=1)
    grid = torch.stack([gx, gy], dim=-1).reshape(n * 3, 1, -1, 2).float()
    out = F.grid_sample(planes.reshape(n * 3, c, h, w), grid, mode='bilinear', padding_mode='zeros',
                        align_corners=False)
    return out.permute(0, 3, 2, 1).reshape(n, 3, -1, c)


def osg_decoder(P, feats, prefix='decoder.', lr_mul=1.0):
    """feats [N,3,P,C] -> rgb [N,P,32], sigma [N,P,1]."""
    x = feats.mean(1)
    n, p, c = x.shape
    x = x.reshape(n * p, c)
    x = sg.fully_connected(x, P[prefix + 'net.0.weight'], P[prefix + 'net.0.bias'], lr_mul=lr_mul)
    x = F.softplus(x)
    x = sg.fully_connected(x, P[prefix + 'net.2.weight'], P[prefix + 'net.2.bias'], lr_mul=lr_mul)
    x = x.reshape(n, p, -1)
    return torch.sigmoid(x[..., 1:]) * 1.002 - 0.001, x[..., 0:1]


def ray_march(colors, densities, depths, white_back=False):
    """[N,M,S,C],[N,M,S,1],[N,M,S,1] -> rgb [N,M,C], depth [N,M,1], weights [N,M,S-1,1]."""
    deltas = depths[:, :, 1:] - depths[:, :, :-1]
    c_mid = (colors[:, :, :-1] + colors[:, :, 1:]) / 2
    s_mid = F.softplus((densities[:, :, :-1] + densities[:, :, 1:]) / 2 - 1)
    d_mid = (depths[:, :, :-1] + depths[:, :, 1:]) / 2
    alpha = 1 - torch.exp(-s_mid * deltas)
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :, :1]), 1 - alpha + 1e-10], -2), -2)[:, :, :-1]
    w = alpha * trans
    rgb = (w * c_mid).sum(-2)
    wsum = w.sum(2)
    depth = (w * d_mid).sum(-2) / wsum
    depth = torch.nan_to_num(depth, float('inf'))
    depth = torch.clamp(depth, torch.min(depths), torch.max(depths))
    if white_back:
        rgb = rgb + 1 - wsum
    return rgb * 2 - 1, depth, w


def importance_depths(depths, weights, n_importance, u=None):
    """depths [N,M,S,1], weights [N,M,S-1,1] -> fine depths [N,M,Sf,1] (unsorted); no grad."""
    with torch.no_grad():
        n, m, s, _ = depths.shape
        z = depths.reshape(n * m, s)
        w = weights.reshape(n * m, s - 1)
        w = F.max_pool1d(w.unsqueeze(1).float(), 2, 1, padding=1)
        w = F.avg_pool1d(w, 2, 1).squeeze(1) + 0.01
        bins = 0.5 * (z[:, :-1] + z[:, 1:])
        pw = w[:, 1:-1] + 1e-5
        pdf = pw / pw.sum(-1, keepdim=True)
        cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)
        if u is None:
            u = torch.rand(n * m, n_importance, device=z.device)
        u = u.contiguous()
        idx = torch.searchsorted(cdf, u, right=True)
        lo = (idx - 1).clamp_min(0)
        hi = idx.clamp_max(pw.shape[1])
        c_lo, c_hi = torch.gather(cdf, 1, lo), torch.gather(cdf, 1, hi)
        b_lo, b_hi = torch.gather(bins, 1, lo), torch.gather(bins, 1, hi)
        den = c_hi - c_lo
        den = torch.where(den < 1e-5, torch.ones_like(den), den)
        t = b_lo + (u - c_lo) / den * (b_hi - b_lo)
        return t.reshape(n, m, n_importance, 1)


def merge_samples(d1, c1, s1, d2, c2, s2):
    d = torch.cat([d1, d2], -2)
    c = torch.cat([c1, c2], -2)
    s = torch.cat([s1, s2], -2)
    _, idx = torch.sort(d, dim=-2)
    return (torch.gather(d, -2, idx), torch.gather(c, -2, idx.expand(-1, -1, -1, c.shape[-1])),
            torch.gather(s, -2, idx))


def run_model(P, planes, coords, opts, eps=None, decoder_fn=None, dirs=None):
    """eps: the density-noise draw (randn_like(sigma), renderer.py:146-147) when opts['density_noise'] > 0; None draws it.
    decoder_fn(feats [N,3,P,C], dirs [N,P,3]) -> (rgb, sigma): any decoder callable in place of the OSG MLP (renderer.py:142-145)."""
    feats = sample_planes(planes, coords, box_warp=opts['box_warp'])
    rgb, sigma = osg_decoder(P, feats) if decoder_fn is None else decoder_fn(feats, dirs)
    if opts.get('density_noise', 0) > 0:
        sigma = sigma + (torch.randn_like(sigma) if eps is None else eps.reshape(sigma.shape)) * opts['density_noise']
    return rgb, sigma


def render(P, planes, ray_o, ray_d, opts, xi=None, u=None, return_aux=False, eps=(None, None), decoder_fn=None):
    """ImportanceRenderer.forward -> (rgb [N,M,32], depth [N,M,1], weight_sum [N,M,1]).  eps = the two density-noise draws
    (coarse pass, fine pass) in the reference's order: rand_like (xi), randn_like (eps[0]), rand (u), randn_like (eps[1])."""
    n, m, _ = ray_o.shape
    sc, sf = opts['depth_resolution'], opts['depth_resolution_importance']
    start, end = opts['ray_start'], opts['ray_end']
    if isinstance(start, str) and start == end == 'auto':
        start, end = auto_ray_limits(ray_o, ray_d, opts['box_warp'])
    d_c = coarse_depths(n, m, sc, start, end, xi=xi, device=ray_o.device, disparity=bool(opts.get('disparity_space_sampling', False)))
    pts = (ray_o.unsqueeze(-2) + d_c * ray_d.unsqueeze(-2)).reshape(n, -1, 3)
    dirs = lambda k: ray_d.unsqueeze(-2).expand(-1, -1, k, -1).reshape(n, -1, 3)
    c_c, s_c = run_model(P, planes, pts, opts, eps[0], decoder_fn, dirs(sc))
    c_c = c_c.reshape(n, m, sc, -1)
    s_c = s_c.reshape(n, m, sc, 1)
    if sf > 0:
        _, _, w_c = ray_march(c_c, s_c, d_c, opts.get('white_back', False))
        d_f = importance_depths(d_c, w_c, sf, u=u)
        pts = (ray_o.unsqueeze(-2) + d_f * ray_d.unsqueeze(-2)).reshape(n, -1, 3)
        c_f, s_f = run_model(P, planes, pts, opts, eps[1], decoder_fn, dirs(sf))
        c_f = c_f.reshape(n, m, sf, -1)
        s_f = s_f.reshape(n, m, sf, 1)
        d_all, c_all, s_all = merge_samples(d_c, c_c, s_c, d_f, c_f, s_f)
        rgb, depth, w = ray_march(c_all, s_all, d_all, opts.get('white_back', False))
    else:
        d_f = None
        rgb, depth, w = ray_march(c_c, s_c, d_c, opts.get('white_back', False))
    if return_aux:
        return rgb, depth, w.sum(2), dict(depths_coarse=d_c, depths_fine=d_f)
    return rgb, depth, w.sum(2)


def synthesis(P, ws, c, opts, neural_rendering_resolution=128, noise_mode='const', xi=None, u=None,
              backbone_resolutions=(4, 8, 16, 32, 64, 128, 256), skip_sr=False):
    """TriPlaneGenerator.synthesis -> {'image','image_raw','image_depth'} (+ 'planes')."""
    n = c.shape[0]
    cam2world = c[:, :16].reshape(n, 4, 4)
    intr = c[:, 16:25].reshape(n, 3, 3)
    ray_o, ray_d = ray_sampler(cam2world, intr, neural_rendering_resolution)
    planes = sg.backbone_synthesis(P, ws, resolutions=backbone_resolutions, noise_mode=noise_mode)
    planes = planes.reshape(n, 3, 32, planes.shape[-2], planes.shape[-1])
    feat, depth, _ = render(P, planes, ray_o, ray_d, opts, xi=xi, u=u)
    r = neural_rendering_resolution
    feat_img = feat.permute(0, 2, 1).reshape(n, feat.shape[-1], r, r).contiguous()
    depth_img = depth.permute(0, 2, 1).reshape(n, 1, r, r)
    rgb = feat_img[:, :3]
    out = {'image_raw': rgb, 'image_depth': depth_img, 'planes': planes, 'feature_image': feat_img}
    if not skip_sr:
        out['image'] = sg.superresolution_8xdc(P, rgb, feat_img, ws,
                                               noise_mode=opts.get('superresolution_noise_mode', 'none'),
                                               fp16_operands=bool(opts.get('sr_fp16_operands', False)),
                                               fp16_storage=bool(opts.get('sr_fp16_storage', False)))
    return out
