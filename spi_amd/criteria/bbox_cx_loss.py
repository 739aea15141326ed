"""Box contextual loss on eye / mouth crops (drop-in surface of spi/criteria/bbox_cx_loss.py:141-182).

``BoxCXLoss()(x, y, lm)``: reduce to 256^2, ImageNet-normalise, RoI-align three landmark boxes to
80x80, VGG19 ``features[:6]`` (MFMA conv kernel), contextual loss between the [n,128,40,40] maps.
``roi_align`` is torchvision's op in the reference (not installed here): it is restated from its
published definition (aligned=False, spatial_scale=1, sampling_ratio=-1) as a HIP kernel pair, one box
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


class _RoiAlign(torch.autograd.Function):
    """x [N,C,H,W], boxes [N,4] float (x1,y1,x2,y2; box i applies to image i) -> [N,C,O,O]: ``torchvision.ops.roi_align`` with spatial_scale = 1,
    sampling_ratio = -1, aligned = False as one HIP launch each way (csrc/losses.hip: roi_align_kernel; until round 3 ~25 gather / multiply /
    mean launches per box and direction).  The boxes stay in device memory: no host-side geometry, nothing to synchronise on."""

    @staticmethod
    def forward(ctx, x, boxes, out_size):
        from .. import hip
        x = x.contiguous().float()
        boxes = boxes.to(x.device).contiguous().float()          # landmarks may live on the host (the reference moves them, bbox_cx_loss.py:46)
        n, c, h, w = x.shape
        assert boxes.shape == (n, 4), f'one box per batch element: boxes {tuple(boxes.shape)} for a batch of {n}'
        out = torch.empty(n, c, out_size, out_size, device=x.device)
        hip.call('spi_roi_align_fwd', hip.ptr(x), hip.ptr(boxes), hip.ptr(out), n, c, h, w, int(out_size), hip.stream())
        ctx.save_for_backward(boxes)
        ctx.shape = (n, c, h, w, int(out_size))
        return out

    @staticmethod
    def backward(ctx, dy):
        from .. import hip
        boxes, = ctx.saved_tensors
        n, c, h, w, o = ctx.shape
        dx = torch.empty(n, c, h, w, device=dy.device)
        dy = dy.contiguous().float()                           # bound to a name: a temporary would be freed before the launch is enqueued
        hip.call('spi_roi_align_bwd', hip.ptr(boxes), hip.ptr(dy), hip.ptr(dx), n, c, h, w, o, hip.stream())
        return dx, None, None


def roi_align(x, boxes, output_size=80):
    """x [N,C,H,W]; boxes [N,4] float (x1,y1,x2,y2), box i applies to image i (host or device tensor).  Differentiable wrt x."""
    return _RoiAlign.apply(x, boxes, int(output_size))


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
        d_out = d_out.contiguous().float()
        hip.call('spi_contextual_bwd', hip.ptr(sim), hip.ptr(d_out), b, p1, p2, ctx.band_width, hip.ptr(stats[0]), hip.ptr(stats[1]),
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
        """The three landmark boxes as float tensors on ``device`` for ``forward(..., plan=)``: built once per image (the landmarks do not change)."""
        return [box.float().to(device) for box in get_landmark_bbox(lm)[:3]]

    def forward(self, x, y, lm, plan=None):
        if x.shape[-1] > 256:
            x = F.interpolate(x, (256, 256), mode='bilinear', align_corners=False)
        if y.shape[-1] > 256:
            y = F.interpolate(y, (256, 256), mode='bilinear', align_corners=False)
        x = (x - self.vgg_mean) / self.vgg_std
        y = (y - self.vgg_mean) / self.vgg_std
        loss = 0
        boxes = plan if plan is not None else [b.float() for b in get_landmark_bbox(lm)[:3]]
        # The three boxes (eyes, nose, mouth) go through the VGG head and the contextual loss as ONE batch [3N, ...] instead of three
        # passes of a few hundred tiny launches each; every reduction of the reference (feature mean over the batch of a box, min /
        # sum / max over positions, mean over the batch) keeps its own group.
        n = x.shape[0]
        cx_in = [roi_align(x, box) for box in boxes]
        cy_in = [roi_align(y, box) for box in boxes]
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
