"""Oracle (CPU, plain torch) for the BiSeNet face-parsing network of the mask producer.

TEST INFRASTRUCTURE -- see oracle/__init__.py.

Functional restatement over a ``state_dict`` with the reference's key names, following (relative to /root/reference/third_part/bisenet):
  ResNet-18 trunk        resnet.py:18-95   (BasicBlock: conv-bn-relu-conv-bn + (1x1 stride conv, bn) shortcut, relu)
  ConvBNReLU / heads     bisenet.py:15-52
  attention refinement   bisenet.py:68-94
  context path           bisenet.py:97-131
  feature fusion         bisenet.py:176-213
  BiSeNet.forward        bisenet.py:231-256 (the spatial path is replaced by the res3b1 feature; three bilinear align_corners=True outputs)
Pinned bit-exact against the imported reference module by tests/golden/make_golden.py (section `bisenet`).
"""
import torch
import torch.nn.functional as F


def synthetic_state_dict(manifest, seed=0):
    """Seeded weights keyed by name (no bisenet.pth / resnet18 download exists offline): He-style conv weights, BatchNorm statistics
    away from the identity so that folding errors would show."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shape in manifest.items():
        leaf = k.split('.')[-1]
        if leaf == 'num_batches_tracked':
            sd[k] = torch.zeros((), dtype=torch.int64)
        elif leaf == 'running_var':
            sd[k] = 0.5 + torch.rand(shape, generator=g)
        elif leaf == 'running_mean':
            sd[k] = 0.2 * torch.randn(shape, generator=g)
        elif leaf == 'bias':
            sd[k] = 0.1 * torch.randn(shape, generator=g)
        elif leaf == 'weight' and len(shape) == 1:
            sd[k] = 1.0 + 0.2 * torch.randn(shape, generator=g)
        else:
            fan_in = shape[1] * shape[2] * shape[3]
            sd[k] = torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5
    return sd


def _bn(P, pfx, x):
    return F.batch_norm(x, P[pfx + 'running_mean'], P[pfx + 'running_var'], P[pfx + 'weight'], P[pfx + 'bias'], False, 0.0, 1e-5)


def _cbr(P, pfx, x, stride=1, pad=1):
    return F.relu(_bn(P, pfx + 'bn.', F.conv2d(x, P[pfx + 'conv.weight'], None, stride, pad)))


def _block(P, pfx, x, stride):
    r = F.relu(_bn(P, pfx + 'bn1.', F.conv2d(x, P[pfx + 'conv1.weight'], None, stride, 1)))
    r = _bn(P, pfx + 'bn2.', F.conv2d(r, P[pfx + 'conv2.weight'], None, 1, 1))
    sc = x
    if pfx + 'downsample.0.weight' in P:
        sc = _bn(P, pfx + 'downsample.1.', F.conv2d(x, P[pfx + 'downsample.0.weight'], None, stride, 0))
    return F.relu(sc + r)


def resnet18(P, x, pfx='cp.resnet.'):
    x = F.relu(_bn(P, pfx + 'bn1.', F.conv2d(x, P[pfx + 'conv1.weight'], None, 2, 3)))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for li, stride in ((1, 1), (2, 2), (3, 2), (4, 2)):
        x = _block(P, f'{pfx}layer{li}.0.', x, stride)
        x = _block(P, f'{pfx}layer{li}.1.', x, 1)
        feats.append(x)
    return feats[1], feats[2], feats[3]


def _arm(P, pfx, x):
    feat = _cbr(P, pfx + 'conv.', x)
    att = F.avg_pool2d(feat, feat.shape[2:])
    att = torch.sigmoid(_bn(P, pfx + 'bn_atten.', F.conv2d(att, P[pfx + 'conv_atten.weight'])))
    return feat * att


def _head(P, pfx, x, size):
    x = F.conv2d(_cbr(P, pfx + 'conv.', x), P[pfx + 'conv_out.weight'])
    return F.interpolate(x, size, mode='bilinear', align_corners=True)


def bisenet_forward(P, x):
    H, W = x.shape[2:]
    feat8, feat16, feat32 = resnet18(P, x)
    avg = _cbr(P, 'cp.conv_avg.', F.avg_pool2d(feat32, feat32.shape[2:]), pad=0)
    avg_up = F.interpolate(avg, feat32.shape[2:], mode='nearest')
    f32 = _arm(P, 'cp.arm32.', feat32) + avg_up
    f32_up = _cbr(P, 'cp.conv_head32.', F.interpolate(f32, feat16.shape[2:], mode='nearest'))
    f16 = _arm(P, 'cp.arm16.', feat16) + f32_up
    f16_up = _cbr(P, 'cp.conv_head16.', F.interpolate(f16, feat8.shape[2:], mode='nearest'))
    feat = _cbr(P, 'ffm.convblk.', torch.cat([feat8, f16_up], 1), pad=0)
    att = F.avg_pool2d(feat, feat.shape[2:])
    att = torch.sigmoid(F.conv2d(F.relu(F.conv2d(att, P['ffm.conv1.weight'])), P['ffm.conv2.weight']))
    fuse = feat * att + feat
    return _head(P, 'conv_out.', fuse, (H, W)), _head(P, 'conv_out16.', f16_up, (H, W)), _head(P, 'conv_out32.', f32_up, (H, W))


def cal_mask(P, image):
    return torch.argmax(bisenet_forward(P, image)[0], dim=1, keepdim=True)
