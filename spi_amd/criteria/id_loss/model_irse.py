"""ArcFace IR / IR-SE backbone for the identity metric, inference on the MI355X matrix cores.

Mirror of the reference's ``Backbone`` (spi/criteria/id_loss/model_irse.py:11-52) and its building blocks
(helpers.py:27-121: ``get_blocks``, ``SEModule``, ``bottleneck_IR``, ``bottleneck_IR_SE``, ``l2_norm``): same constructor, same
module tree and parameter names, so ``load_state_dict(torch.load('model_ir_se50.pth'))`` works unchanged.  The modules only HOLD
the parameters; the forward is one functional pass in which
  * every convolution runs on ``spi_conv2d_fwd`` (fp32 MFMA implicit GEMM) -- the eval-mode BatchNorm that follows a conv is
    folded into its weights and the kernel's fused bias epilogue; stride-2 3x3 convs are evaluated at stride 1 and decimated
    (exactly the same numbers: the identity metric is evaluated a handful of times per image, the extra FLOPs are irrelevant);
  * the BatchNorm in front of a conv (zero padding forbids folding it), PReLU, the squeeze-excite gate and the final
    ``Linear(25088, 512)`` are a few elementwise / library-GEMM launches.
The metric is inference only (``Metric.run``, spi/utils/metric_utils.py:14-17): the module stays in eval mode.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...torch_utils.ops import conv2d_mfma

_UNITS_PER_STAGE = {50: (3, 4, 14, 3), 100: (3, 13, 30, 3), 152: (3, 8, 36, 3)}     # helpers.py:31-52
_STAGE_DEPTH = (64, 128, 256, 512)


def get_blocks(num_layers):
    """[(in_channel, depth, stride)] per residual unit: every stage opens with a stride-2 unit (helpers.py:27-28)."""
    if num_layers not in _UNITS_PER_STAGE:
        raise ValueError('Invalid number of layers: {}. Must be one of [50, 100, 152]'.format(num_layers))
    units, cin = [], 64
    for depth, n in zip(_STAGE_DEPTH, _UNITS_PER_STAGE[num_layers]):
        units += [(cin, depth, 2)] + [(depth, depth, 1)] * (n - 1)
        cin = depth
    return units


def l2_norm(x, axis=1):
    return x / torch.norm(x, 2, axis, True)


class Flatten(nn.Module):
    def forward(self, x):
        return x.reshape(x.shape[0], -1)


class SEModule(nn.Module):
    """parameter holder: fc1 / fc2 are bias-free 1x1 convs on the pooled vector (helpers.py:56-73)"""
    def __init__(self, channels, reduction):
        super().__init__()
        self.fc1 = nn.Conv2d(channels, channels // reduction, kernel_size=1, bias=False)
        self.fc2 = nn.Conv2d(channels // reduction, channels, kernel_size=1, bias=False)


class _Unit(nn.Module):
    """bottleneck_IR / bottleneck_IR_SE parameter layout (helpers.py:76-121): shortcut_layer = [conv1x1(stride), BN] when the
    channel count changes (else MaxPool2d(1, stride), no parameters); res_layer = [BN, conv3x3, PReLU, conv3x3(stride), BN(, SE)]."""
    def __init__(self, in_channel, depth, stride, se):
        super().__init__()
        self.in_channel, self.depth, self.stride = in_channel, depth, stride
        if in_channel == depth:
            self.shortcut_layer = nn.MaxPool2d(1, stride)
        else:
            self.shortcut_layer = nn.Sequential(nn.Conv2d(in_channel, depth, 1, stride, bias=False), nn.BatchNorm2d(depth))
        layers = [nn.BatchNorm2d(in_channel), nn.Conv2d(in_channel, depth, 3, 1, 1, bias=False), nn.PReLU(depth),
                  nn.Conv2d(depth, depth, 3, stride, 1, bias=False), nn.BatchNorm2d(depth)]
        if se:
            layers.append(SEModule(depth, 16))
        self.res_layer = nn.Sequential(*layers)


def bottleneck_IR(in_channel, depth, stride):
    return _Unit(in_channel, depth, stride, se=False)


def bottleneck_IR_SE(in_channel, depth, stride):
    return _Unit(in_channel, depth, stride, se=True)


def _bn_affine(bn):
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps) if bn.affine else 1.0 / torch.sqrt(bn.running_var + bn.eps)
    shift = (bn.bias if bn.affine else 0.0) - bn.running_mean * scale
    return scale, shift


def _fold(conv, bn):
    """conv followed by eval-mode BN -> (tap-major weights, bias) of one conv with a bias epilogue."""
    scale, shift = _bn_affine(bn)
    return conv2d_mfma.to_tap_major(conv.weight * scale.reshape(-1, 1, 1, 1)), shift.contiguous()


class Backbone(nn.Module):
    def __init__(self, input_size, num_layers, mode='ir', drop_ratio=0.4, affine=True):
        super().__init__()
        assert input_size in [112, 224], 'input_size should be 112 or 224'
        assert num_layers in [50, 100, 152], 'num_layers should be 50, 100 or 152'
        assert mode in ['ir', 'ir_se'], 'mode should be ir or ir_se'
        self.input_layer = nn.Sequential(nn.Conv2d(3, 64, 3, 1, 1, bias=False), nn.BatchNorm2d(64), nn.PReLU(64))
        side = input_size // 16
        self.output_layer = nn.Sequential(nn.BatchNorm2d(512), nn.Dropout(drop_ratio), Flatten(), nn.Linear(512 * side * side, 512),
                                          nn.BatchNorm1d(512, affine=affine))
        self.body = nn.Sequential(*[_Unit(cin, depth, stride, se=(mode == 'ir_se')) for cin, depth, stride in get_blocks(num_layers)])
        self._folded = None
        self.eval()

    def train(self, mode=True):
        if mode:
            raise RuntimeError('the identity backbone is inference-only (eval-mode BatchNorm is folded into the convolutions)')
        return super().train(False)

    def load_state_dict(self, *args, **kw):
        self._folded = None
        return super().load_state_dict(*args, **kw)

    def _apply(self, fn, *a, **kw):
        self._folded = None
        return super()._apply(fn, *a, **kw)

    @torch.no_grad()
    def _prepare(self):
        f = {'input': _fold(self.input_layer[0], self.input_layer[1]), 'units': []}
        for u in self.body:
            r = u.res_layer
            d = {'pre': _bn_affine(r[0]), 'conv1': conv2d_mfma.to_tap_major(r[1].weight), 'conv2': _fold(r[3], r[4])}
            if not isinstance(u.shortcut_layer, nn.MaxPool2d):
                d['short'] = _fold(u.shortcut_layer[0], u.shortcut_layer[1])
            f['units'].append(d)
        f['out_bn'] = _bn_affine(self.output_layer[0])
        f['out_bn1d'] = _bn_affine(self.output_layer[4])
        self._folded = f
        return f

    @torch.no_grad()
    def forward(self, x):
        f = self._folded or self._prepare()
        conv = conv2d_mfma.conv2d
        x = x.contiguous().float()
        w, b = f['input']
        x = F.prelu(conv(x, w, bias=b, padding=1, tap_major=True), self.input_layer[2].weight)
        for u, d in zip(self.body, f['units']):
            s = u.stride
            sub = x[:, :, ::s, ::s] if s > 1 else x
            if 'short' in d:
                shortcut = conv(sub.contiguous(), d['short'][0], bias=d['short'][1], padding=0, tap_major=True)
            else:
                shortcut = sub
            scale, shift = d['pre']
            r = torch.addcmul(shift.reshape(1, -1, 1, 1), x, scale.reshape(1, -1, 1, 1))
            r = F.prelu(conv(r, d['conv1'], padding=1, tap_major=True), u.res_layer[2].weight)
            r = conv(r, d['conv2'][0], bias=d['conv2'][1], padding=1, tap_major=True)
            if s > 1:
                r = r[:, :, ::s, ::s]
            if len(u.res_layer) == 6:
                se = u.res_layer[5]
                g = r.mean((2, 3))
                g = torch.sigmoid(F.linear(F.relu(F.linear(g, se.fc1.weight.flatten(1))), se.fc2.weight.flatten(1)))
                r = r * g[:, :, None, None]
            x = r + shortcut
        scale, shift = f['out_bn']
        x = torch.addcmul(shift.reshape(1, -1, 1, 1), x, scale.reshape(1, -1, 1, 1))
        lin = self.output_layer[3]
        x = F.linear(x.flatten(1), lin.weight, lin.bias)
        scale, shift = f['out_bn1d']
        return l2_norm(x * scale + shift)


def IR_50(input_size):
    return Backbone(input_size, num_layers=50, mode='ir', drop_ratio=0.4, affine=False)


def IR_101(input_size):
    return Backbone(input_size, num_layers=100, mode='ir', drop_ratio=0.4, affine=False)


def IR_152(input_size):
    return Backbone(input_size, num_layers=152, mode='ir', drop_ratio=0.4, affine=False)


def IR_SE_50(input_size):
    return Backbone(input_size, num_layers=50, mode='ir_se', drop_ratio=0.4, affine=False)


def IR_SE_101(input_size):
    return Backbone(input_size, num_layers=100, mode='ir_se', drop_ratio=0.4, affine=False)


def IR_SE_152(input_size):
    return Backbone(input_size, num_layers=152, mode='ir_se', drop_ratio=0.4, affine=False)
