"""2D-FAN-4 landmark network (68 heat maps at 64^2 from a 256^2 face crop), inference on the MI355X matrix cores.

The reference gets its landmarks from the third-party pip package ``face_alignment`` (preprocess/extract_landmark.py:3,10:
``face_alignment.FaceAlignment(LandmarksType._2D)``), which is NOT under /root/reference and cannot be installed offline -- "parity
unpinned" at this edge.  This module restates the package's published network (Bulat & Tzimiropoulos, "How far are we from solving the 2D &
3D face alignment problem?", ICCV 2017; `face_alignment/models.py` of the 1.x releases): a 7x7 / stride-2 stem, three pre-activation
``ConvBlock``s (BN-ReLU-conv3x3 x3 with a concatenated 1/2 + 1/4 + 1/4 output and a BN-ReLU-1x1 shortcut when the widths differ), and four
stacked depth-4 hourglasses with intermediate supervision heads.  Module tree and parameter names follow the package (``conv1 bn1 conv2 conv3
conv4 m{i}.b{1,2,3}_{level} m{i}.b2_plus_1 top_m_{i} conv_last{i} bn_end{i} l{i} bl{i} al{i}``), so its ``2DFAN4`` state dict loads unchanged.
The modules only HOLD parameters; ``forward`` is one functional inference pass: every convolution runs on ``spi_conv2d_fwd`` (fp32 MFMA);
the pre-activation BatchNorm + ReLU in front of each 3x3 is a per-channel affine + ReLU (library elementwise launches), the stem's BN is
folded into its weights with the ReLU fused into the conv epilogue; pooling / nearest up-sampling are library launches.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...torch_utils.ops import conv2d_mfma


def conv3x3(in_planes, out_planes):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=1, padding=1, bias=False)


class ConvBlock(nn.Module):
    def __init__(self, in_planes, out_planes):
        super().__init__()
        self.bn1 = nn.BatchNorm2d(in_planes)
        self.conv1 = conv3x3(in_planes, out_planes // 2)
        self.bn2 = nn.BatchNorm2d(out_planes // 2)
        self.conv2 = conv3x3(out_planes // 2, out_planes // 4)
        self.bn3 = nn.BatchNorm2d(out_planes // 4)
        self.conv3 = conv3x3(out_planes // 4, out_planes // 4)
        self.downsample = None
        if in_planes != out_planes:
            self.downsample = nn.Sequential(nn.BatchNorm2d(in_planes), nn.ReLU(True), nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=1, bias=False))


class HourGlass(nn.Module):
    def __init__(self, num_modules, depth, num_features):
        super().__init__()
        self.num_modules, self.depth, self.features = num_modules, depth, num_features
        self._generate_network(depth)

    def _generate_network(self, level):
        self.add_module('b1_' + str(level), ConvBlock(self.features, self.features))
        self.add_module('b2_' + str(level), ConvBlock(self.features, self.features))
        if level > 1:
            self._generate_network(level - 1)
        else:
            self.add_module('b2_plus_' + str(level), ConvBlock(self.features, self.features))
        self.add_module('b3_' + str(level), ConvBlock(self.features, self.features))


_MEMO = {}          # id(module) -> prepared tensors (inference only: parameters are fixed between calls; FAN.load_state_dict clears it)


def _bn_affine(bn):
    key = ('bn', id(bn))
    if key not in _MEMO:
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        _MEMO[key] = (scale.view(1, -1, 1, 1), (bn.bias - bn.running_mean * scale).view(1, -1, 1, 1))
    return _MEMO[key]


def _tm(conv):
    """the conv's weight in the kernels' tap-major layout [O, kh, kw, I]"""
    key = ('w', id(conv))
    if key not in _MEMO:
        _MEMO[key] = conv2d_mfma.to_tap_major(conv.weight.detach().float()).contiguous()
    return _MEMO[key]


def _pre(bn, x):
    s, b = _bn_affine(bn)
    return torch.relu(x * s + b)


def _conv_block(m, x):
    o1 = conv2d_mfma.conv2d(_pre(m.bn1, x), _tm(m.conv1), padding=1, tap_major=True)
    o2 = conv2d_mfma.conv2d(_pre(m.bn2, o1), _tm(m.conv2), padding=1, tap_major=True)
    o3 = conv2d_mfma.conv2d(_pre(m.bn3, o2), _tm(m.conv3), padding=1, tap_major=True)
    res = x if m.downsample is None else conv2d_mfma.conv2d(_pre(m.downsample[0], x), _tm(m.downsample[2]), padding=0, tap_major=True)
    return torch.cat((o1, o2, o3), 1) + res


def _hourglass(hg, level, x):
    up1 = _conv_block(getattr(hg, 'b1_' + str(level)), x)
    low1 = _conv_block(getattr(hg, 'b2_' + str(level)), F.avg_pool2d(x, 2, stride=2))
    low2 = _hourglass(hg, level - 1, low1) if level > 1 else _conv_block(getattr(hg, 'b2_plus_' + str(level)), low1)
    low3 = _conv_block(getattr(hg, 'b3_' + str(level)), low2)
    return up1 + F.interpolate(low3, scale_factor=2, mode='nearest')


class FAN(nn.Module):
    def __init__(self, num_modules=4):
        super().__init__()
        self.num_modules = num_modules
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3)
        self.bn1 = nn.BatchNorm2d(64)
        self.conv2 = ConvBlock(64, 128)
        self.conv3 = ConvBlock(128, 128)
        self.conv4 = ConvBlock(128, 256)
        for i in range(num_modules):
            self.add_module('m' + str(i), HourGlass(1, 4, 256))
            self.add_module('top_m_' + str(i), ConvBlock(256, 256))
            self.add_module('conv_last' + str(i), nn.Conv2d(256, 256, kernel_size=1, stride=1, padding=0))
            self.add_module('bn_end' + str(i), nn.BatchNorm2d(256))
            self.add_module('l' + str(i), nn.Conv2d(256, 68, kernel_size=1, stride=1, padding=0))
            if i < num_modules - 1:
                self.add_module('bl' + str(i), nn.Conv2d(256, 256, kernel_size=1, stride=1, padding=0))
                self.add_module('al' + str(i), nn.Conv2d(68, 256, kernel_size=1, stride=1, padding=0))
        self.eval()

    def train(self, mode=True):
        if mode:
            raise NotImplementedError('inference only (eval-mode BatchNorm is folded into the pass)')
        return super().train(False)

    def _apply(self, fn, *a, **k):                               # .to() / .float(): prepared tensors are rebuilt on the next call
        _MEMO.clear()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        _MEMO.clear()
        return super().load_state_dict(*a, **k)

    @torch.no_grad()
    def forward(self, x):
        """x [N,3,256,256] in [0,1] -> list of num_modules heat-map tensors [N,68,64,64] (the last one is the prediction)"""
        n, _, h, w = x.shape
        # stem: 7x7 / stride 2 / pad 3 as im2col + a 1x1 MFMA conv with 147 input channels; BN folded, ReLU in the epilogue
        s, b = _bn_affine(self.bn1)
        wt = (self.conv1.weight * s.view(-1, 1, 1, 1)).reshape(64, 147, 1, 1)
        bias = self.conv1.bias * s.view(-1) + b.view(-1)
        cols = F.unfold(x.float(), kernel_size=7, padding=3, stride=2).reshape(n, 147, h // 2, w // 2)
        x = conv2d_mfma.conv2d(cols, wt, bias=bias, padding=0, act='relu', gain=1.0)
        x = F.avg_pool2d(_conv_block(self.conv2, x), 2, stride=2)
        x = _conv_block(self.conv3, x)
        x = _conv_block(self.conv4, x)
        previous, outputs = x, []
        for i in range(self.num_modules):
            ll = _conv_block(getattr(self, 'top_m_' + str(i)), _hourglass(getattr(self, 'm' + str(i)), 4, previous))
            cl, be = getattr(self, 'conv_last' + str(i)), getattr(self, 'bn_end' + str(i))
            s, b = _bn_affine(be)
            ll = conv2d_mfma.conv2d(ll, cl.weight * s.view(-1, 1, 1, 1), bias=cl.bias * s.view(-1) + b.view(-1), padding=0, act='relu', gain=1.0)
            li = getattr(self, 'l' + str(i))
            tmp_out = conv2d_mfma.conv2d(ll, li.weight, bias=li.bias, padding=0)
            outputs.append(tmp_out)
            if i < self.num_modules - 1:
                bl, al = getattr(self, 'bl' + str(i)), getattr(self, 'al' + str(i))
                previous = previous + conv2d_mfma.conv2d(ll, bl.weight, bias=bl.bias, padding=0) + conv2d_mfma.conv2d(tmp_out, al.weight, bias=al.bias, padding=0)
        return outputs
