"""TEST INFRASTRUCTURE (oracle) -- never imported by the product path.

CPU restatement of the identity metric of the reference:
  * ``IDLoss.extract_feats / calculate_similarity``  spi/criteria/id_loss/id_loss.py:17-29
  * ``Backbone(input_size=112, num_layers=50, mode='ir_se')`` forward in eval mode
        spi/criteria/id_loss/model_irse.py:11-52, helpers.py:13-121 (get_blocks, SEModule, bottleneck_IR_SE, l2_norm)
written as one functional pass over a state dict with the reference's key names, so a real ``model_ir_se50.pth`` loads as is.

Pinned: tests/golden/idloss.npz holds features / similarities computed by the REFERENCE's own ``Backbone`` / ``IDLoss``
code (imported from /root/reference by tests/golden/make_idloss_golden.py) on ``synthetic_state_dict(seed)`` weights and
seeded inputs; tests/test_oracle_cpu.py checks this file against them.  The trained ArcFace checkpoint itself is not
available offline: parity on *trained* weights is unpinned, the arithmetic is not.
"""
import torch
import torch.nn.functional as F

# (in_channel, depth, stride) of the 24 units of IR-50 (helpers.py:27-40: get_block(64,64,3), (64,128,4), (128,256,14), (256,512,3))
UNITS = []
for _in, _d, _n in ((64, 64, 3), (64, 128, 4), (128, 256, 14), (256, 512, 3)):
    UNITS += [(_in, _d, 2)] + [(_d, _d, 1)] * (_n - 1)
BN_EPS = 1e-5


def state_dict_spec():
    """Ordered (key, shape, kind) list of the reference module's state dict (num_batches_tracked omitted)."""
    spec = []

    def bn(prefix, c):
        spec.extend([(prefix + '.weight', (c,), 'bn_w'), (prefix + '.bias', (c,), 'bn_b'), (prefix + '.running_mean', (c,), 'bn_m'),
                     (prefix + '.running_var', (c,), 'bn_v')])
    spec.append(('input_layer.0.weight', (64, 3, 3, 3), 'conv'))
    bn('input_layer.1', 64)
    spec.append(('input_layer.2.weight', (64,), 'prelu'))
    for i, (cin, d, s) in enumerate(UNITS):
        p = f'body.{i}.'
        if cin != d:
            spec.append((p + 'shortcut_layer.0.weight', (d, cin, 1, 1), 'conv'))
            bn(p + 'shortcut_layer.1', d)
        bn(p + 'res_layer.0', cin)
        spec.append((p + 'res_layer.1.weight', (d, cin, 3, 3), 'conv'))
        spec.append((p + 'res_layer.2.weight', (d,), 'prelu'))
        spec.append((p + 'res_layer.3.weight', (d, d, 3, 3), 'conv'))
        bn(p + 'res_layer.4', d)
        spec.append((p + 'res_layer.5.fc1.weight', (d // 16, d, 1, 1), 'conv'))
        spec.append((p + 'res_layer.5.fc2.weight', (d, d // 16, 1, 1), 'conv'))
    bn('output_layer.0', 512)
    spec.append(('output_layer.3.weight', (512, 512 * 7 * 7), 'linear'))
    spec.append(('output_layer.3.bias', (512,), 'bn_b'))
    bn('output_layer.4', 512)
    return spec


def synthetic_state_dict(seed=0):
    """Deterministic stand-in weights (the trained checkpoint does not exist offline): one seeded generator per tensor."""
    sd = {}
    for idx, (key, shape, kind) in enumerate(state_dict_spec()):
        g = torch.Generator().manual_seed(seed * 100003 + idx)
        r = torch.randn(shape, generator=g)
        if kind == 'conv':
            v = r * (1.6 / (shape[1] * shape[2] * shape[3])) ** 0.5
        elif kind == 'linear':
            v = r * (1.0 / shape[1]) ** 0.5
        elif kind == 'bn_w':
            v = 1.0 + 0.1 * r
        elif kind in ('bn_b', 'bn_m'):
            v = 0.1 * r
        elif kind == 'bn_v':
            v = 1.0 + 0.3 * torch.rand(shape, generator=g)
        elif kind == 'prelu':
            v = 0.25 + 0.05 * r
        sd[key] = v.float()
    return sd


def _bn(x, sd, p):
    """eval-mode BatchNorm (running statistics)."""
    shape = (1, -1) + (1,) * (x.ndim - 2)
    scale = sd[p + '.weight'] / torch.sqrt(sd[p + '.running_var'] + BN_EPS)
    return (x - sd[p + '.running_mean'].reshape(shape)) * scale.reshape(shape) + sd[p + '.bias'].reshape(shape)


def backbone_forward(sd, x):
    """model_irse.py:48-52: input_layer -> body -> output_layer -> l2_norm; x [N,3,112,112] -> [N,512] unit vectors."""
    sd = {k: v.to(x.device, torch.float32) for k, v in sd.items() if v.is_floating_point()}
    x = F.conv2d(x, sd['input_layer.0.weight'], padding=1)
    x = F.prelu(_bn(x, sd, 'input_layer.1'), sd['input_layer.2.weight'])
    for i, (cin, d, s) in enumerate(UNITS):
        p = f'body.{i}.'
        if cin == d:
            shortcut = x[:, :, ::s, ::s]                                             # MaxPool2d(1, stride) (helpers.py:101)
        else:
            shortcut = _bn(F.conv2d(x, sd[p + 'shortcut_layer.0.weight'], stride=s), sd, p + 'shortcut_layer.1')
        r = _bn(x, sd, p + 'res_layer.0')
        r = F.prelu(F.conv2d(r, sd[p + 'res_layer.1.weight'], padding=1), sd[p + 'res_layer.2.weight'])
        r = _bn(F.conv2d(r, sd[p + 'res_layer.3.weight'], stride=s, padding=1), sd, p + 'res_layer.4')
        se = r.mean((2, 3), keepdim=True)                                            # SEModule (helpers.py:56-73)
        se = torch.sigmoid(F.conv2d(F.relu(F.conv2d(se, sd[p + 'res_layer.5.fc1.weight'])), sd[p + 'res_layer.5.fc2.weight']))
        x = r * se + shortcut
    x = _bn(x, sd, 'output_layer.0')                                                 # Dropout is the identity in eval mode
    x = F.linear(x.flatten(1), sd['output_layer.3.weight'], sd['output_layer.3.bias'])
    x = _bn(x, sd, 'output_layer.4')
    return x / torch.norm(x, 2, 1, True)


def extract_feats(sd, img):
    """id_loss.py:17-21: fixed crop [35:223, 32:220], adaptive average pool to 112^2, backbone."""
    x = img[:, :, 35:223, 32:220]
    x = F.adaptive_avg_pool2d(x, (112, 112))
    return backbone_forward(sd, x)


def calculate_similarity(sd, x, y):
    """id_loss.py:23-28 (batch of one)."""
    assert x.shape[0] == 1
    return extract_feats(sd, x)[0].dot(extract_feats(sd, y)[0])
