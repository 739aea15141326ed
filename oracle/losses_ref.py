"""Oracle (CPU, plain torch) for the SPI losses and the depth-guided warp.

TEST INFRASTRUCTURE -- see oracle/__init__.py.

Reference lines followed (relative to /root/reference/spi):
  l2_loss             criteria/l2_loss.py:3-8
  LPIPS-VGG16         criteria/lpips/lpips.py:32-71, networks.py:36-96, utils.py:6-8
  BoxCX               criteria/bbox_cx_loss.py:20-61 (boxes), :93-129 (contextual), :141-182 (forward)
  rotate              utils/rotate.py:5-116
  noise regulariser   training/projectors/mirror_projector.py:107-115
Third-party arithmetic NOT under /root/reference, restated from its published definition
("parity unpinned" at these edges): torchvision vgg16/vgg19 topology (weights are supplied
by the caller; no pretrained weights exist in this environment), torchvision.ops.roi_align
(aligned=False, spatial_scale=1, sampling_ratio=-1 -> ceil(roi/bin) samples per bin).
"""
import math
import torch
import torch.nn.functional as F

VGG16_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512]
VGG16_TAPS = (1, 3, 6, 9, 12)          # conv index after whose relu a feature is tapped (relu1_2 ... relu5_3)
LPIPS_CH = (64, 128, 256, 512, 512)
LPIPS_MEAN = (-.030, -.088, -.188)
LPIPS_STD = (.458, .448, .450)


def make_vgg16_weights(seed=0, device='cpu'):
    """Seeded stand-in for torchvision's pretrained VGG16 features + LPIPS lin layers."""
    g = torch.Generator().manual_seed(seed)
    convs, cin = [], 3
    for v in VGG16_CFG:
        if v == 'M':
            continue
        w = torch.randn(v, cin, 3, 3, generator=g) * math.sqrt(2.0 / (cin * 9))
        b = torch.randn(v, generator=g) * 0.05
        convs.append((w.to(device), b.to(device)))
        cin = v
    lins = [(torch.rand(1, c, 1, 1, generator=g) / c).to(device) for c in LPIPS_CH]
    return {'convs': convs, 'lins': lins}


def vgg16_features(W, x):
    x = (x - x.new_tensor(LPIPS_MEAN).view(1, 3, 1, 1)) / x.new_tensor(LPIPS_STD).view(1, 3, 1, 1)
    feats, ci = [], 0
    for v in VGG16_CFG:
        if v == 'M':
            x = F.max_pool2d(x, 2, 2)
            continue
        w, b = W['convs'][ci]
        x = F.relu(F.conv2d(x, w, b, padding=1))
        if ci in VGG16_TAPS:
            norm = torch.sqrt(torch.sum(x ** 2, dim=1, keepdim=True))
            feats.append(x / (norm + 1e-10))
        ci += 1
    return feats


def lpips(W, x, y):
    n = x.shape[0]
    if x.shape[-1] > 256:
        x = F.interpolate(x, size=(256, 256), mode='bilinear', align_corners=False)
        y = F.interpolate(y, size=(256, 256), mode='bilinear', align_corners=False)
    fx, fy = vgg16_features(W, x), vgg16_features(W, y)
    loss = 0.0
    for a, b, lin in zip(fx, fy, W['lins']):
        loss = loss + F.conv2d((a - b) ** 2, lin).mean((2, 3), True).sum()
    return loss / n


def sg_vgg_features(W, img, resize_images=False, return_lpips=True):
    """Stand-in for NVIDIA's TorchScript `vgg16.pt` (spi/configs/paths_config.py:5, loaded by load_utils.py:47-50 and called
    as `vgg16(img_0..255, resize_images=False, return_lpips=True)` by w_projector.py:51,86).  The blob is NOT under
    /root/reference and cannot be fetched ("parity unpinned" at this edge); restated from its published contract: images
    in [0, 255] -> [-1, 1] -> LPIPS z-score -> the five VGG16 taps, unit-normalised over channels, scaled by
    sqrt(lin / (H W)) and flattened, so that the squared L2 distance of two feature vectors IS the LPIPS distance."""
    assert not resize_images and return_lpips
    feats = vgg16_features(W, img / 127.5 - 1)
    out = []
    for f, lin in zip(feats, W['lins']):
        hw = f.shape[2] * f.shape[3]
        out.append((f * torch.sqrt(lin / hw)).flatten(1))
    return torch.cat(out, dim=1)


def l2_loss(a, b):
    return F.mse_loss(a, b)


def noise_regulariser(noise_maps):
    reg = 0.0
    for v in noise_maps:
        nz = v[None, None]
        while True:
            reg = reg + (nz * torch.roll(nz, 1, 3)).mean() ** 2 + (nz * torch.roll(nz, 1, 2)).mean() ** 2
            if nz.shape[2] <= 8:
                break
            nz = F.avg_pool2d(nz, 2)
    return reg


# ---- BoxCX ---------------------------------------------------------------------------------

def make_vgg19_head_weights(seed=1, device='cpu'):
    g = torch.Generator().manual_seed(seed)
    out = []
    for cin, cout in ((3, 64), (64, 64), (64, 128)):
        w = torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (cin * 9))
        b = torch.randn(cout, generator=g) * 0.05
        out.append((w.to(device), b.to(device)))
    return out


def vgg19_head(W, x):
    """torchvision vgg19.features[:6] = conv-relu-conv-relu-maxpool-conv (no relu after the last)."""
    x = F.relu(F.conv2d(x, W[0][0], W[0][1], padding=1))
    x = F.relu(F.conv2d(x, W[1][0], W[1][1], padding=1))
    x = F.max_pool2d(x, 2, 2)
    return F.conv2d(x, W[2][0], W[2][1], padding=1)


def landmark_boxes(lm):
    """[mouth, l_eye, r_eye] boxes (x1,y1,x2,y2); pad 8 for the mouth then 15 (the p=15 assignment
    persists, bbox_cx_loss.py:32-33)."""
    boxes, p = [], 8
    for i, (a, b) in enumerate(((48, 68), (36, 42), (42, 48))):
        pts = lm[:, a:b]
        ly, ry = pts[:, :, 0].min(1)[0], pts[:, :, 0].max(1)[0]
        lx, rx = pts[:, :, 1].min(1)[0], pts[:, :, 1].max(1)[0]
        lx, rx, ly, ry = lx.long(), rx.long(), ly.long(), ry.long()
        if i in (1, 2):
            p = 15
        boxes.append(torch.stack([ly - p, lx - p, ry + p, rx + p], dim=1).float())
    return boxes


def roi_align(x, boxes, out=80):
    """x [N,C,H,W]; boxes [N,4] (x1,y1,x2,y2), one per batch element; aligned=False, ratio=-1."""
    n, c, h, w = x.shape
    res = []
    for i in range(n):
        x1, y1, x2, y2 = [float(v) for v in boxes[i]]
        rw, rh = max(x2 - x1, 1.0), max(y2 - y1, 1.0)
        bw, bh = rw / out, rh / out
        gw, gh = int(math.ceil(rw / out)), int(math.ceil(rh / out))
        ix = torch.arange(out, dtype=torch.float32).view(-1, 1) * bw + (torch.arange(gw, dtype=torch.float32).view(1, -1) + 0.5) * bw / gw + x1
        iy = torch.arange(out, dtype=torch.float32).view(-1, 1) * bh + (torch.arange(gh, dtype=torch.float32).view(1, -1) + 0.5) * bh / gh + y1
        xs, ys = ix.reshape(-1).to(x.device), iy.reshape(-1).to(x.device)

        def axis(t, size):
            valid = (t >= -1.0) & (t <= size)
            t = t.clamp(min=0)
            lo = t.floor().long()
            hi_edge = lo >= size - 1
            lo = torch.where(hi_edge, torch.full_like(lo, size - 1), lo)
            hi = torch.where(hi_edge, lo, lo + 1)
            t = torch.where(hi_edge, lo.float(), t)
            frac = t - lo.float()
            return lo, hi, frac, valid.float()
        xl, xh, xf, xv = axis(xs, w)
        yl, yh, yf, yv = axis(ys, h)
        img = x[i]
        rows_l, rows_h = img[:, yl, :], img[:, yh, :]
        val = ((rows_l[:, :, xl] * (1 - xf) + rows_l[:, :, xh] * xf) * (1 - yf).view(1, -1, 1)
               + (rows_h[:, :, xl] * (1 - xf) + rows_h[:, :, xh] * xf) * yf.view(1, -1, 1))
        val = val * yv.view(1, -1, 1) * xv.view(1, 1, -1)
        res.append(val.reshape(c, out, gh, out, gw).mean(dim=(2, 4)))
    return torch.stack(res)


def contextual_cx(sim, band_width=0.5):
    """[B,P1,P2] cosine matrix -> mean_j max_i CX [B]: compute_relative_distance, compute_cx and the max / mean of compute_cx_loss
    (spi/criteria/bbox_cx_loss.py:111-129)."""
    dist = 1 - sim
    dmin = dist.min(dim=2, keepdim=True)[0]
    dt = torch.clamp(dist / (dmin + 1e-5), max=10., min=-10)
    wgt = torch.exp((1 - dt) / band_width)
    cx = wgt / wgt.sum(dim=2, keepdim=True)
    return cx.max(dim=1)[0].mean(dim=1)


def contextual_loss(fx, fy, band_width=0.5):
    mu = fy.mean(dim=(0, 2, 3), keepdim=True)
    xn = F.normalize(fx - mu, p=2, dim=1).flatten(2)
    yn = F.normalize(fy - mu, p=2, dim=1).flatten(2)
    cx = contextual_cx(torch.bmm(xn.transpose(1, 2), yn), band_width)
    return torch.mean(-torch.log(cx + 1e-5))


def box_cx_loss(W19, x, y, lm):
    if x.shape[-1] > 256:
        x = F.interpolate(x, (256, 256), mode='bilinear', align_corners=False)
    if y.shape[-1] > 256:
        y = F.interpolate(y, (256, 256), mode='bilinear', align_corners=False)
    mean = x.new_tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = x.new_tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    x, y = (x - mean) / std, (y - mean) / std
    loss = 0.0
    for box in landmark_boxes(lm):
        loss = loss + contextual_loss(vgg19_head(W19, roi_align(x, box)), vgg19_head(W19, roi_align(y, box)))
    return loss * 0.1


# ---- depth-guided warp -----------------------------------------------------------------------

def _pixel_grid(n, res, device):
    t = (torch.arange(res, dtype=torch.float32, device=device) + 0.5) / res
    v, u = torch.meshgrid(t, t, indexing='ij')
    return u.reshape(1, -1).expand(n, -1), v.reshape(1, -1).expand(n, -1)


def rotate(target_camera, target_depth, src_image, src_camera, src_depth, src_mask=None, EPS=5e-2):
    n = src_image.shape[0]
    res = src_image.shape[-1]
    tex, tin = target_camera[:, :16].reshape(n, 4, 4), target_camera[:, 16:].reshape(n, 3, 3)
    gex, gin = src_camera[:, :16].reshape(n, 4, 4), src_camera[:, 16:].reshape(n, 3, 3)

    def up(d):
        dr = d.shape[-1]                       # 128 in the reference (rotate.py:102,108 hard-code it)
        d = d.reshape(n, 1, dr, dr)
        if res != dr:
            d = F.interpolate(d, (res, res), mode='bilinear', align_corners=False)
        return d.reshape(n, res, res)
    td, gd = up(target_depth), up(src_depth)
    u, v = _pixel_grid(n, res, src_image.device)
    fx, fy, cx, cy, sk = tin[:, 0, 0:1], tin[:, 1, 1:2], tin[:, 0, 2:3], tin[:, 1, 2:3], tin[:, 0, 1:2]
    z = td.reshape(n, -1)
    xl = (u - cx + cy * sk / fy - sk * v / fy) / fx * z
    yl = (v - cy) / fy * z
    world = torch.bmm(tex, torch.stack([xl, yl, z, torch.ones_like(z)], -1).transpose(1, 2))
    cam = torch.bmm(torch.inverse(gex), world).transpose(1, 2)
    fx, fy, cx, cy, sk = gin[:, 0, 0:1], gin[:, 1, 1:2], gin[:, 0, 2:3], gin[:, 1, 2:3], gin[:, 0, 1:2]
    zc = cam[:, :, 2]
    yc = cam[:, :, 1] / zc * fy + cy
    xc = cam[:, :, 0] / zc * fx + sk * yc / fy - cy * sk / fy + cx
    grid = (2 * torch.stack([xc, yc], -1) - 1).reshape(n, res, res, 2)
    inside = 1 - ((grid[..., 0] < -1) | (grid[..., 0] > 1) | (grid[..., 1] < -1) | (grid[..., 1] > 1)).float()
    d_src = F.grid_sample(gd.reshape(n, 1, res, res), grid, align_corners=False).reshape(n, res, res)
    mask = ((d_src - zc.reshape(n, res, res)).abs() < EPS) * inside
    mask = mask.unsqueeze(1)
    rgb = F.grid_sample(src_image, grid, align_corners=False) * mask
    if src_mask is not None:
        m2 = F.grid_sample(src_mask.reshape(n, 1, res, res), grid, align_corners=False)
        rgb = rgb * m2
        mask = mask * m2
    return rgb, mask
