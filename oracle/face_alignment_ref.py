"""Oracle (CPU, plain torch) for the landmark producer of the dataset preparation: S3FD face detector + 2D-FAN-4.

TEST INFRASTRUCTURE -- see oracle/__init__.py.

PARITY UNPINNED: the reference calls the third-party pip package `face_alignment` (preprocess/extract_landmark.py:3,10,19:
`face_alignment.FaceAlignment(LandmarksType._2D).get_landmarks_from_image(np.array(image))`); the package is not under /root/reference, is not
installed and cannot be fetched (no version is pinned either: README.md:19 "Follow EG3D to install the environment"; the `_2D` enum name exists
in its 1.x releases up to 1.3.5).  This file restates the published algorithms -- Zhang et al., S3FD (ICCV 2017) and Bulat & Tzimiropoulos,
2D-FAN (ICCV 2017) as the package's 1.x `detection/sfd/{net_s3fd,detect,bbox}.py`, `models.py`, `utils.py`, `api.py` implement them --
functionally over a state dict with plain torch.nn.functional calls, independently of spi_amd/third_part/face_alignment (which holds modules and
runs the HIP convs).  It checks the HIP path against an independent CPU implementation of the same published definition; it cannot check
either against the package.
"""
import numpy as np
import torch
import torch.nn.functional as F


# ---- 2D-FAN-4 ------------------------------------------------------------------------------------------------------------------------
def _bn(P, pfx, x):
    return F.batch_norm(x, P[pfx + 'running_mean'], P[pfx + 'running_var'], P[pfx + 'weight'], P[pfx + 'bias'], False, 0.0, 1e-5)


def _conv_block(P, pfx, x):
    o1 = F.conv2d(F.relu(_bn(P, pfx + 'bn1.', x)), P[pfx + 'conv1.weight'], None, 1, 1)
    o2 = F.conv2d(F.relu(_bn(P, pfx + 'bn2.', o1)), P[pfx + 'conv2.weight'], None, 1, 1)
    o3 = F.conv2d(F.relu(_bn(P, pfx + 'bn3.', o2)), P[pfx + 'conv3.weight'], None, 1, 1)
    res = x
    if pfx + 'downsample.2.weight' in P:
        res = F.conv2d(F.relu(_bn(P, pfx + 'downsample.0.', x)), P[pfx + 'downsample.2.weight'])
    return torch.cat([o1, o2, o3], dim=1) + res


def _hourglass(P, pfx, level, x):
    up1 = _conv_block(P, f'{pfx}b1_{level}.', x)
    low1 = _conv_block(P, f'{pfx}b2_{level}.', F.avg_pool2d(x, 2, stride=2))
    low2 = _hourglass(P, pfx, level - 1, low1) if level > 1 else _conv_block(P, f'{pfx}b2_plus_{level}.', low1)
    low3 = _conv_block(P, f'{pfx}b3_{level}.', low2)
    return up1 + F.interpolate(low3, scale_factor=2, mode='nearest')


def fan_forward(P, x, num_modules=4):
    """x [N,3,256,256] in [0,1] -> list of heat maps [N,68,64,64]"""
    x = F.relu(_bn(P, 'bn1.', F.conv2d(x, P['conv1.weight'], P['conv1.bias'], 2, 3)))
    x = F.avg_pool2d(_conv_block(P, 'conv2.', x), 2, stride=2)
    x = _conv_block(P, 'conv4.', _conv_block(P, 'conv3.', x))
    previous, outs = x, []
    for i in range(num_modules):
        ll = _conv_block(P, f'top_m_{i}.', _hourglass(P, f'm{i}.', 4, previous))
        ll = F.relu(_bn(P, f'bn_end{i}.', F.conv2d(ll, P[f'conv_last{i}.weight'], P[f'conv_last{i}.bias'])))
        tmp = F.conv2d(ll, P[f'l{i}.weight'], P[f'l{i}.bias'])
        outs.append(tmp)
        if i < num_modules - 1:
            previous = previous + F.conv2d(ll, P[f'bl{i}.weight'], P[f'bl{i}.bias']) + F.conv2d(tmp, P[f'al{i}.weight'], P[f'al{i}.bias'])
    return outs


# ---- S3FD ----------------------------------------------------------------------------------------------------------------------------
def s3fd_forward(P, x):
    def c(name, h, stride=1, pad=1):
        return F.conv2d(h, P[name + '.weight'], P[name + '.bias'], stride, pad)
    h = x
    taps = {}
    for block, n in ((1, 2), (2, 2), (3, 3), (4, 3), (5, 3)):
        for j in range(1, n + 1):
            h = F.relu(c(f'conv{block}_{j}', h))
        taps[block] = h
        h = F.max_pool2d(h, 2, 2)
    h = F.relu(c('fc6', h, pad=3))
    ffc7 = h = F.relu(c('fc7', h, pad=0))
    h = F.relu(c('conv6_1', h, pad=0))
    f6 = h = F.relu(c('conv6_2', h, stride=2))
    h = F.relu(c('conv7_1', h, pad=0))
    f7 = F.relu(c('conv7_2', h, stride=2))

    def l2n(name, t):
        return t / (t.pow(2).sum(dim=1, keepdim=True).sqrt() + 1e-10) * P[name + '.weight'].view(1, -1, 1, 1)
    srcs = (('conv3_3_norm', l2n('conv3_3_norm', taps[3])), ('conv4_3_norm', l2n('conv4_3_norm', taps[4])), ('conv5_3_norm', l2n('conv5_3_norm', taps[5])),
            ('fc7', ffc7), ('conv6_2', f6), ('conv7_2', f7))
    out = []
    for name, f in srcs:
        cls, reg = c(name + '_mbox_conf', f), c(name + '_mbox_loc', f)
        if name == 'conv3_3_norm':
            ch = torch.chunk(cls, 4, 1)
            cls = torch.cat([torch.max(torch.max(ch[0], ch[1]), ch[2]), ch[3]], dim=1)
        out += [cls, reg]
    return out


def nms(dets, thresh):
    order = dets[:, 4].argsort()[::-1]
    area = (dets[:, 2] - dets[:, 0] + 1) * (dets[:, 3] - dets[:, 1] + 1)
    keep = []
    while order.size:
        i, rest = order[0], order[1:]
        keep.append(int(i))
        w = np.maximum(0.0, np.minimum(dets[i, 2], dets[rest, 2]) - np.maximum(dets[i, 0], dets[rest, 0]) + 1)
        h = np.maximum(0.0, np.minimum(dets[i, 3], dets[rest, 3]) - np.maximum(dets[i, 1], dets[rest, 1]) + 1)
        iou = w * h / (area[i] + area[rest] - w * h)
        order = rest[iou <= thresh]
    return keep


def detect(P, image_rgb):
    """per position, like the package's loop (detect.py): anchor centre (stride/2 + index * stride), anchor size 4 * stride, variances 0.1 / 0.2"""
    img = np.asarray(image_rgb)[..., ::-1].astype(np.float32) - np.array([104.0, 117.0, 123.0], dtype=np.float32)
    with torch.no_grad():
        olist = s3fd_forward(P, torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1)))[None])
    out = []
    for i in range(len(olist) // 2):
        ocls, oreg = F.softmax(olist[2 * i], dim=1), olist[2 * i + 1]
        stride = 2 ** (i + 2)
        for hi, wi in zip(*np.where(ocls[0, 1].numpy() > 0.05)):
            axc, ayc, s = stride / 2 + wi * stride, stride / 2 + hi * stride, stride * 4.0
            loc = oreg[0, :, hi, wi].numpy()
            cx, cy = axc + loc[0] * 0.1 * s, ayc + loc[1] * 0.1 * s
            w, h = s * np.exp(loc[2] * 0.2), s * np.exp(loc[3] * 0.2)
            out.append([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2, float(ocls[0, 1, hi, wi])])
    if not out:
        return np.zeros((0, 5), dtype=np.float32)
    dets = np.array(out, dtype=np.float32)
    dets = dets[nms(dets, 0.3)]
    return dets[dets[:, 4] > 0.5]


# ---- crop / heat map decoding -----------------------------------------------------------------------------------------------------------
def _t(center, scale, res):
    h = 200.0 * scale
    return np.array([[res / h, 0, res * (-center[0] / h + 0.5)], [0, res / h, res * (-center[1] / h + 0.5)], [0, 0, 1]])


def landmarks_from_face(Pfan, image, box):
    """one detected box (x1, y1, x2, y2, ...) -> [68,2] landmarks in image coordinates"""
    image = np.asarray(image)[..., :3]
    cx, cy = box[2] - (box[2] - box[0]) / 2.0, box[3] - (box[3] - box[1]) / 2.0 - (box[3] - box[1]) * 0.12
    scale = (box[2] - box[0] + box[3] - box[1]) / 195.0
    inv = np.linalg.inv(_t((cx, cy), scale, 256))
    ul = (inv @ np.array([1, 1, 1.0]))[:2].astype(np.int64)
    br = (inv @ np.array([256, 256, 1.0]))[:2].astype(np.int64)
    H, W = image.shape[:2]
    patch = np.zeros((br[1] - ul[1], br[0] - ul[0], 3), dtype=np.float32)
    for yy in range(patch.shape[0]):                               # pixel (ul + 1 .. br) of the image, 1-based as in the package's crop
        sy = ul[1] + yy
        if 0 <= sy < H:
            x0, x1 = max(0, ul[0]), min(W, br[0])
            if x1 > x0:
                patch[yy, x0 - ul[0]:x1 - ul[0]] = image[sy, x0:x1]
    inp = F.interpolate(torch.from_numpy(patch).permute(2, 0, 1)[None], size=(256, 256), mode='bilinear', align_corners=False)
    inp = torch.floor(inp + 0.5).clamp(0, 255) / 255.0              # the package resizes the uint8 crop with cv2 (rounds to uint8) before scaling
    with torch.no_grad():
        hm = fan_forward(Pfan, inp)[-1][0]
    inv64 = np.linalg.inv(_t((cx, cy), scale, 64))
    pts = []
    for k in range(hm.shape[0]):
        idx = int(hm[k].reshape(-1).argmax())
        px, py = idx % 64, idx // 64
        x, y = float(px + 1), float(py + 1)
        if 0 < px < 63 and 0 < py < 63:
            x += 0.25 * float(torch.sign(hm[k, py, px + 1] - hm[k, py, px - 1]))
            y += 0.25 * float(torch.sign(hm[k, py + 1, px] - hm[k, py - 1, px]))
        pts.append((inv64 @ np.array([x - 0.5, y - 0.5, 1.0]))[:2])
    return np.array(pts, dtype=np.float32)
