"""S3FD face detector (the default ``face_detector='sfd'`` of the ``face_alignment`` package the reference uses, extract_landmark.py:10),
inference on the MI355X matrix cores.  Third-party code that is not under /root/reference ("parity unpinned"); restated from the published
detector (Zhang et al., "S3FD: Single Shot Scale-invariant Face Detector", ICCV 2017; `face_alignment/detection/sfd/{net_s3fd,detect,bbox}.py`
of the 1.x releases): a VGG16 trunk with fc6 / fc7 as convolutions and two extra stages, L2-normalised conv3_3 / conv4_3 / conv5_3, six
(confidence, box) head pairs at strides 4 ... 128 with max-out background on the first, softmax, anchors of 4 x stride, variances (0.1, 0.2),
score threshold 0.05, NMS at IoU 0.3, final threshold 0.5.  Parameter names follow the package, so its ``s3fd`` state dict loads unchanged.
All convolutions run on ``spi_conv2d_fwd`` with bias + ReLU fused (stride-2 3x3 convs evaluated at stride 1 and decimated); pooling,
normalisation, softmax and the box decoding are library launches; NMS runs on the host over the few hundred surviving boxes.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...torch_utils.ops import conv2d_mfma


class L2Norm(nn.Module):
    def __init__(self, n_channels, scale=1.0):
        super().__init__()
        self.n_channels, self.scale, self.eps = n_channels, scale, 1e-10
        self.weight = nn.Parameter(torch.full((n_channels,), float(scale)))


_VGG = (('conv1_1', 3, 64), ('conv1_2', 64, 64), 'P', ('conv2_1', 64, 128), ('conv2_2', 128, 128), 'P', ('conv3_1', 128, 256), ('conv3_2', 256, 256),
        ('conv3_3', 256, 256), 'P', ('conv4_1', 256, 512), ('conv4_2', 512, 512), ('conv4_3', 512, 512), 'P', ('conv5_1', 512, 512),
        ('conv5_2', 512, 512), ('conv5_3', 512, 512), 'P')
_HEADS = (('conv3_3_norm', 256, 4), ('conv4_3_norm', 512, 2), ('conv5_3_norm', 512, 2), ('fc7', 1024, 2), ('conv6_2', 512, 2), ('conv7_2', 256, 2))


class s3fd(nn.Module):
    def __init__(self):
        super().__init__()
        for item in _VGG:
            if item != 'P':
                setattr(self, item[0], nn.Conv2d(item[1], item[2], kernel_size=3, stride=1, padding=1))
        self.fc6 = nn.Conv2d(512, 1024, kernel_size=3, stride=1, padding=3)
        self.fc7 = nn.Conv2d(1024, 1024, kernel_size=1, stride=1, padding=0)
        self.conv6_1 = nn.Conv2d(1024, 256, kernel_size=1, stride=1, padding=0)
        self.conv6_2 = nn.Conv2d(256, 512, kernel_size=3, stride=2, padding=1)
        self.conv7_1 = nn.Conv2d(512, 128, kernel_size=1, stride=1, padding=0)
        self.conv7_2 = nn.Conv2d(128, 256, kernel_size=3, stride=2, padding=1)
        self.conv3_3_norm, self.conv4_3_norm, self.conv5_3_norm = L2Norm(256, scale=10), L2Norm(512, scale=8), L2Norm(512, scale=5)
        for name, cin, ncls in _HEADS:
            setattr(self, name + '_mbox_conf', nn.Conv2d(cin, ncls, kernel_size=3, stride=1, padding=1))
            setattr(self, name + '_mbox_loc', nn.Conv2d(cin, 4, kernel_size=3, stride=1, padding=1))
        self._memo = {}
        self.eval()

    def _apply(self, fn, *a, **k):
        self._memo = {}
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._memo = {}
        return super().load_state_dict(*a, **k)

    def _conv(self, name, x, relu=True, extra_pad=0, stride=1):
        m = getattr(self, name)
        if name not in self._memo:
            self._memo[name] = (conv2d_mfma.to_tap_major(m.weight.detach().float()).contiguous(), m.bias.detach().float().contiguous())
        w, b = self._memo[name]
        if extra_pad:
            x = F.pad(x, (extra_pad,) * 4)
        y = conv2d_mfma.conv2d(x, w, bias=b, padding=m.kernel_size[0] // 2, act='relu' if relu else None, gain=1.0 if relu else None, tap_major=True)
        return y[:, :, ::stride, ::stride] if stride > 1 else y      # a stride-2 / pad-1 3x3 conv = the stride-1 result at the even positions

    @staticmethod
    def _l2norm(m, x):
        return x / (x.pow(2).sum(dim=1, keepdim=True).sqrt() + m.eps) * m.weight.view(1, -1, 1, 1)

    @torch.no_grad()
    def forward(self, x):
        """x [N,3,H,W] (BGR, mean-subtracted 0..255) -> [cls1, reg1, ..., cls6, reg6] raw head outputs (cls1 after the background max-out)"""
        h = x.float()
        feats = {}
        for item in _VGG:
            if item == 'P':
                h = F.max_pool2d(h, 2, 2)
            else:
                h = self._conv(item[0], h)
                feats[item[0]] = h
        h = self._conv('fc6', h, extra_pad=2)                      # 3x3 with padding 3 (the package's port keeps the dilated layer's padding): +2 of explicit zero padding
        h = self._conv('fc7', h)
        ffc7 = h
        h = self._conv('conv6_1', h)
        f6_2 = h = self._conv('conv6_2', h, stride=2)
        h = self._conv('conv7_1', h)
        f7_2 = self._conv('conv7_2', h, stride=2)
        srcs = (self._l2norm(self.conv3_3_norm, feats['conv3_3']), self._l2norm(self.conv4_3_norm, feats['conv4_3']),
                self._l2norm(self.conv5_3_norm, feats['conv5_3']), ffc7, f6_2, f7_2)
        out = []
        for (name, _, _), f in zip(_HEADS, srcs):
            cls = self._conv(name + '_mbox_conf', f, relu=False)
            reg = self._conv(name + '_mbox_loc', f, relu=False)
            if name == 'conv3_3_norm':                             # max-out background label: max of the first three maps vs the face map
                chunk = torch.chunk(cls, 4, 1)
                cls = torch.cat([torch.max(torch.max(chunk[0], chunk[1]), chunk[2]), chunk[3]], dim=1)
            out += [cls, reg]
        return out


def nms(dets, thresh):
    """greedy NMS over [K,5] (x1, y1, x2, y2, score) with the +1 pixel convention of the package's bbox.py -> kept indices, best first"""
    if len(dets) == 0:
        return []
    x1, y1, x2, y2, scores = dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3], dets[:, 4]
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = scores.argsort()[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(int(i))
        xx1, yy1 = np.maximum(x1[i], x1[order[1:]]), np.maximum(y1[i], y1[order[1:]])
        xx2, yy2 = np.minimum(x2[i], x2[order[1:]]), np.minimum(y2[i], y2[order[1:]])
        w, h = np.maximum(0.0, xx2 - xx1 + 1), np.maximum(0.0, yy2 - yy1 + 1)
        ovr = w * h / (areas[i] + areas[order[1:]] - w * h)
        order = order[np.where(ovr <= thresh)[0] + 1]
    return keep


@torch.no_grad()
def detect(net, image_rgb, device):
    """image_rgb: uint8 / float array [H,W,3] (RGB, 0..255) -> float array [K,5] of (x1, y1, x2, y2, score), NMS-ed, score > 0.5, best first
    (sfd_detector.detect_from_image + detect.py of the package)."""
    img = np.asarray(image_rgb)[..., ::-1].astype(np.float32) - np.array([104.0, 117.0, 123.0], dtype=np.float32)       # BGR, mean-subtracted
    x = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))).unsqueeze(0).to(device)
    olist = net(x)
    boxes = []
    for i in range(len(olist) // 2):
        ocls, oreg = F.softmax(olist[i * 2], dim=1), olist[i * 2 + 1]
        stride = 2 ** (i + 2)
        score = ocls[0, 1]
        hi, wi = torch.nonzero(score > 0.05, as_tuple=True)
        if hi.numel() == 0:
            continue
        axc, ayc = stride / 2 + wi.float() * stride, stride / 2 + hi.float() * stride
        loc = oreg[0, :, hi, wi].t()                               # [K,4]
        pw = torch.full_like(axc, stride * 4.0)
        cxy = torch.stack([axc, ayc], 1) + loc[:, :2] * 0.1 * torch.stack([pw, pw], 1)
        wh = torch.stack([pw, pw], 1) * torch.exp(loc[:, 2:] * 0.2)
        x1y1 = cxy - wh / 2
        boxes.append(torch.cat([x1y1, x1y1 + wh, score[hi, wi].unsqueeze(1)], 1))
    if not boxes:
        return np.zeros((0, 5), dtype=np.float32)
    dets = torch.cat(boxes).cpu().numpy()
    dets = dets[nms(dets, 0.3)]
    return dets[dets[:, 4] > 0.5]
