"""Box contextual loss on eye / mouth crops (drop-in surface of spi/criteria/bbox_cx_loss.py:141-182).

``BoxCXLoss()(x, y, lm)``: reduce to 256^2, ImageNet-normalise, RoI-align three landmark boxes to
80x80, VGG19 ``features[:6]`` (MFMA conv kernel), contextual loss between the [n,128,40,40] maps.
``roi_align`` is torchvision's op in the reference (not installed here): it is restated from its
published definition (aligned=False, spatial_scale=1, sampling_ratio=-1) as batched gathers, one box
per batch element as ``get_bbox`` builds them (:41-61).  The 1600x1600 cosine matrices are plain
batched GEMMs (``torch.bmm``).
"""
import math
import torch
import torch.nn.functional as F

from .lpips.networks import VGG19Head


def get_landmark_bbox(lm, scale=1):
    """[mouth, l_eye, r_eye, nose] boxes (x1, y1, x2, y2) per batch element.  The pad is 8 for the mouth and 15
    from the first eye on (the ``p = 15`` assignment persists in the reference, :32-33)."""
    boxes, p = [], 8
    for i, (a, b) in enumerate(((48, 68), (36, 42), (42, 48), (27, 36))):
        pts = lm[:, a:b]
        ly, ry = pts[:, :, 0].min(1)[0], pts[:, :, 0].max(1)[0]
        lx, rx = pts[:, :, 1].min(1)[0], pts[:, :, 1].max(1)[0]
        lx, rx, ly, ry = (lx * scale).long(), (rx * scale).long(), (ly * scale).long(), (ry * scale).long()
        if i in (1, 2):
            p = 15
        boxes.append(torch.stack([ly - p, lx - p, ry + p, rx + p], dim=1))
    return boxes


def _axis(t, size):
    valid = ((t >= -1.0) & (t <= size)).float()
    t = t.clamp(min=0)
    lo = t.floor().long()
    edge = lo >= size - 1
    lo = torch.where(edge, torch.full_like(lo, size - 1), lo)
    hi = torch.where(edge, lo, lo + 1)
    t = torch.where(edge, lo.float(), t)
    return lo, hi, t - lo.float(), valid


def roi_plan(boxes, h, w, device, output_size=80):
    """Sampling geometry of ``roi_align`` for boxes [N,4] (x1,y1,x2,y2), one box per image: a list of per-image index / weight
    tensors on ``device``.  This is the only host-side part (the sampling grid size depends on the box size); for SPI the
    landmarks are fixed per image, so the coach builds the plan once per image and the loss itself never synchronises."""
    out = output_size
    bx = boxes.detach().float().cpu()
    plans = []
    for i in range(bx.shape[0]):
        x1, y1, x2, y2 = [float(v) for v in bx[i]]
        rw, rh = max(x2 - x1, 1.0), max(y2 - y1, 1.0)
        gw, gh = int(math.ceil(rw / out)), int(math.ceil(rh / out))
        bw, bh = rw / out, rh / out
        ar = torch.arange(out, dtype=torch.float32).view(-1, 1)
        xs = (ar * bw + (torch.arange(gw, dtype=torch.float32).view(1, -1) + 0.5) * bw / gw + x1).reshape(-1)
        ys = (ar * bh + (torch.arange(gh, dtype=torch.float32).view(1, -1) + 0.5) * bh / gh + y1).reshape(-1)
        key = (x1, y1, x2, y2)
        if plans and plans[-1]['key'] == key:            # SPI repeats one landmark set over the batch: share the tensors
            plans.append(plans[-1])
            continue
        xl, xh, xf, xv = _axis(xs, w)
        yl, yh, yf, yv = _axis(ys, h)
        plans.append(dict(key=key, gw=gw, gh=gh, xl=xl.to(device), xh=xh.to(device), xf=xf.to(device), xv=xv.to(device),
                          yl=yl.to(device), yh=yh.to(device), yf=yf.to(device), yv=yv.to(device)))
    return plans


def roi_align(x, boxes, output_size=80, plan=None):
    """x [N,C,H,W]; boxes [N,4] float (x1,y1,x2,y2), box i applies to image i.  Differentiable wrt x.
    ``plan`` (from ``roi_plan``) skips the host-side geometry; images that share a plan entry are gathered together."""
    n, c, h, w = x.shape
    out = output_size
    if plan is None:
        plan = roi_plan(boxes, h, w, x.device, output_size)
    res, i = [], 0
    while i < n:
        pl = plan[i]
        j = i
        while j + 1 < n and plan[j + 1] is pl:
            j += 1
        img = x[i:j + 1]
        xf, yf = pl['xf'], pl['yf']
        # index_select, not img[:, :, idx]: the backward of advanced indexing is a sort-based index_put_, which faulted when the iteration
        # was replayed from a captured HIP graph; index_select's backward is an atomic index_add_
        top, bot = img.index_select(2, pl['yl']), img.index_select(2, pl['yh'])
        val = ((top.index_select(3, pl['xl']) * (1 - xf) + top.index_select(3, pl['xh']) * xf) * (1 - yf).view(1, 1, -1, 1)
               + (bot.index_select(3, pl['xl']) * (1 - xf) + bot.index_select(3, pl['xh']) * xf) * yf.view(1, 1, -1, 1))
        val = val * pl['yv'].view(1, 1, -1, 1) * pl['xv'].view(1, 1, 1, -1)
        res.append(val.reshape(j + 1 - i, c, out, pl['gh'], out, pl['gw']).mean(dim=(3, 5)))
        i = j + 1
    return torch.cat(res)


def compute_cosine_distance(x, y):
    y_mu = y.mean(dim=(0, 2, 3), keepdim=True)
    xn = F.normalize(x - y_mu, p=2, dim=1).flatten(2)
    yn = F.normalize(y - y_mu, p=2, dim=1).flatten(2)
    return 1 - torch.bmm(xn.transpose(1, 2), yn)


def compute_relative_distance(dist_raw):
    # amin, not min(dim)[0] as the reference writes it (bbox_cx_loss.py:113): the same value; the backward is a mask multiply instead of an
    # index scatter through the saved argmin -- which is what faulted when the iteration was replayed from a captured HIP graph.
    # (Exact ties share the gradient instead of giving it to the first index: measure zero on float features.)
    dist_min = torch.amin(dist_raw, dim=2, keepdim=True)
    return torch.clamp(dist_raw / (dist_min + 1e-5), max=10., min=-10)


def compute_cx(dist_tilde, band_width):
    w = torch.exp((1 - dist_tilde) / band_width)
    return w / torch.sum(w, dim=2, keepdim=True)


class _ContextualCX(torch.autograd.Function):
    """sim [B,P1,P2] -> mean_j max_i cx  [B]: `1 - sim`, compute_relative_distance, compute_cx and the max / mean of the reference's
    compute_cx_loss (bbox_cx_loss.py:93-129) in two launches forward, one backward (csrc/losses.hip) instead of ~25 ATen passes."""

    @staticmethod
    def forward(ctx, sim, band_width):
        from .. import hip
        sim = sim.contiguous()
        b, p1, p2 = sim.shape
        dev = sim.device
        out = torch.empty(b, device=dev)
        stats = torch.empty(2, b, p1, device=dev)
        row_argmin = torch.empty(b, p1, dtype=torch.int32, device=dev)
        col_argmax = torch.empty(b, p2, dtype=torch.int32, device=dev)
        ws = torch.empty(hip.lib().spi_contextual_workspace_bytes(b, p1, p2), dtype=torch.uint8, device=dev)
        hip.call('spi_contextual_fwd', hip.ptr(sim), b, p1, p2, float(band_width), hip.ptr(out), hip.ptr(stats[0]), hip.ptr(stats[1]),
                 row_argmin.data_ptr(), col_argmax.data_ptr(), ws.data_ptr(), hip.stream())
        ctx.save_for_backward(sim, stats, row_argmin, col_argmax)
        ctx.band_width = float(band_width)
        return out

    @staticmethod
    def backward(ctx, d_out):
        from .. import hip
        sim, stats, row_argmin, col_argmax = ctx.saved_tensors
        b, p1, p2 = sim.shape
        d_sim = torch.empty_like(sim)
        hip.call('spi_contextual_bwd', hip.ptr(sim), hip.ptr(d_out.contiguous().float()), b, p1, p2, ctx.band_width, hip.ptr(stats[0]), hip.ptr(stats[1]),
                 row_argmin.data_ptr(), col_argmax.data_ptr(), hip.ptr(d_sim), hip.stream())
        return d_sim, None


def contextual_cx(sim, band_width):
    """mean_j max_i CX of a cosine matrix sim [B,P1,P2] -> [B] as one fused HIP op (GPU tensors only, like every kernel of this package).
    ``compute_relative_distance`` / ``compute_cx`` above keep the reference's step-by-step functions."""
    return _ContextualCX.apply(sim, band_width)


class BoxCXLoss(torch.nn.Module):
    def __init__(self, band_width=0.5, weights=None, seed=1):
        super().__init__()
        self.band_width = band_width
        self.vgg_model = VGG19Head(weights=weights, seed=seed)
        self.register_buffer('vgg_mean', torch.tensor([[[0.485]], [[0.456]], [[0.406]]]))
        self.register_buffer('vgg_std', torch.tensor([[[0.229]], [[0.224]], [[0.225]]]))

    def plan(self, lm, device):
        """Host-side box geometry for ``forward(..., plan=)``: build once per image (the landmarks do not change)."""
        return [roi_plan(box.float(), 256, 256, device) for box in get_landmark_bbox(lm)[:3]]

    def forward(self, x, y, lm, plan=None):
        if x.shape[-1] > 256:
            x = F.interpolate(x, (256, 256), mode='bilinear', align_corners=False)
        if y.shape[-1] > 256:
            y = F.interpolate(y, (256, 256), mode='bilinear', align_corners=False)
        x = (x - self.vgg_mean) / self.vgg_std
        y = (y - self.vgg_mean) / self.vgg_std
        loss = 0
        if plan is not None and tuple(x.shape[-2:]) != (256, 256):
            plan = None                                      # the plan is built for the 256^2 working resolution
        boxes = get_landmark_bbox(lm)[:3] if plan is None else [None] * 3
        # The three boxes (eyes, nose, mouth) go through the VGG head and the contextual loss as ONE batch [3N, ...] instead of three
        # passes of a few hundred tiny launches each; every reduction of the reference (feature mean over the batch of a box, min /
        # sum / max over positions, mean over the batch) keeps its own group.
        n = x.shape[0]
        cx_in, cy_in = [], []
        for bi, box in enumerate(boxes):
            pl = plan[bi] if plan is not None else roi_plan(box.float(), x.shape[-2], x.shape[-1], x.device)
            cx_in.append(roi_align(x, box, plan=pl))
            cy_in.append(roi_align(y, box, plan=pl))
        nb = len(cx_in)
        if y.requires_grad:
            f = self.vgg_model(torch.cat(cx_in + cy_in))
            fx, fy = f[:nb * n], f[nb * n:]
        else:
            fx = self.vgg_model(torch.cat(cx_in))
            with torch.no_grad():
                fy = self.vgg_model(torch.cat(cy_in))
        c, fh, fw = fx.shape[1:]
        fx5, fy5 = fx.reshape(nb, n, c, fh, fw), fy.reshape(nb, n, c, fh, fw)
        y_mu = fy5.mean(dim=(1, 3, 4), keepdim=True)                               # per box: mean over its batch and positions (:93)
        xn = F.normalize(fx5 - y_mu, p=2, dim=2).reshape(nb * n, c, fh * fw)
        yn = F.normalize(fy5 - y_mu, p=2, dim=2).reshape(nb * n, c, fh * fw)
        cx = contextual_cx(torch.bmm(xn.transpose(1, 2), yn), self.band_width)      # [nb * n]
        loss = (-torch.log(cx + 1e-5)).reshape(nb, n).mean(dim=1).sum()
        return loss * 0.1
