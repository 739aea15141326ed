"""3DMM coefficient regressor of the crop + camera producer, inference on the MI355X matrix cores.

``ReconNetWrapper('resnet50', use_last_fc=False)`` of third_part/Deep3DFaceRecon_pytorch/models/networks.py:69-104: a ResNet-50 v1.5 trunk
(:205-370, the stride sits on the 3x3 conv of a bottleneck), global average pooling and seven 1x1 heads (id 80, exp 64, tex 80, angle 3,
gamma 27, tx/ty 2, tz 1) -> 257 coefficients per 224^2 image.  Same module tree and parameter names, so the ``net_recon`` entry of the
reference's ``epoch_20.pth`` loads unchanged (``backbone.conv1 / bn1 / layer{1..4}.{i}.conv{1,2,3} / bn{1,2,3} / downsample.{0,1}``,
``final_layers.{0..6}``).  The modules only HOLD the parameters; the forward is one functional inference pass in which
  * all 53 convolutions run on ``spi_conv2d_fwd`` (fp32 MFMA) with the eval-mode BatchNorm that follows folded into the weights and the ReLU
    fused into the kernel's epilogue; stride-2 3x3 convs are evaluated at stride 1 and decimated, stride-2 1x1 shortcuts decimate their input
    first (the same numbers); the 7x7 / stride-2 stem is an im2col (``F.unfold``) + a 1x1 MFMA conv with 147 input channels;
  * max-pool, the global average pool and the seven heads on the pooled 2048-vector (ONE 2048 x 257 library GEMM) are library launches.
Only resnet50 with use_last_fc=False is built -- what Extract3dmm constructs through TestOptions (facerecon_model.py:24,89-91).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ....torch_utils.ops import conv2d_mfma

LAYERS = (3, 4, 6, 3)
HEADS = (80, 64, 80, 3, 27, 2, 1)


def conv1x1(in_planes, out_planes, stride=1, bias=False):
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=bias)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv1x1(inplanes, planes)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = conv1x1(planes, planes * self.expansion)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.downsample = downsample
        self.stride = stride


class ResNet50(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        inplanes = 64
        for li, (planes, n) in enumerate(zip((64, 128, 256, 512), LAYERS), start=1):
            stride = 1 if li == 1 else 2
            blocks = []
            for bi in range(n):
                ds = None
                if bi == 0 and (stride != 1 or inplanes != planes * 4):
                    ds = nn.Sequential(conv1x1(inplanes, planes * 4, stride), nn.BatchNorm2d(planes * 4))
                blocks.append(Bottleneck(inplanes, planes, stride if bi == 0 else 1, ds))
                inplanes = planes * 4
            setattr(self, f'layer{li}', nn.Sequential(*blocks))


def _bn_affine(bn):
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    return scale, bn.bias - bn.running_mean * scale


class ReconNetWrapper(nn.Module):
    fc_dim = 257

    def __init__(self, net_recon='resnet50', use_last_fc=False, init_path=None):
        super().__init__()
        if net_recon != 'resnet50' or use_last_fc:
            raise NotImplementedError('only net_recon="resnet50", use_last_fc=False (what preprocess/extract_3dmm.py builds) is implemented')
        self.use_last_fc = use_last_fc
        self.backbone = ResNet50()
        self.final_layers = nn.ModuleList([conv1x1(2048, d, bias=True) for d in HEADS])
        self._folded = None
        self.eval()

    def train(self, mode=True):
        if mode:
            raise RuntimeError('ReconNetWrapper here is inference-only (eval-mode BatchNorm is folded into the convolutions)')
        return super().train(False)

    def load_state_dict(self, *args, **kw):
        self._folded = None
        return super().load_state_dict(*args, **kw)

    def _apply(self, fn, *a, **kw):
        self._folded = None
        return super()._apply(fn, *a, **kw)

    @torch.no_grad()
    def _prepare(self):
        def fold(conv, bn):
            scale, shift = _bn_affine(bn)
            w = conv.weight * scale.reshape(-1, 1, 1, 1)
            if conv.kernel_size == (7, 7):                               # stem: im2col columns (c, ky, kx) -> a 1x1 conv over 147 channels
                w = w.reshape(w.shape[0], -1, 1, 1)
            return conv2d_mfma.to_tap_major(w.float()), shift.float().contiguous()
        f = {'stem': fold(self.backbone.conv1, self.backbone.bn1)}
        for li in (1, 2, 3, 4):
            for bi, blk in enumerate(getattr(self.backbone, f'layer{li}')):
                key = f'l{li}.{bi}'
                f[key + '.c1'], f[key + '.c2'], f[key + '.c3'] = fold(blk.conv1, blk.bn1), fold(blk.conv2, blk.bn2), fold(blk.conv3, blk.bn3)
                if blk.downsample is not None:
                    f[key + '.ds'] = fold(blk.downsample[0], blk.downsample[1])
        f['heads'] = (torch.cat([m.weight.reshape(m.weight.shape[0], -1) for m in self.final_layers]).float().t().contiguous(),
                      torch.cat([m.bias for m in self.final_layers]).float())
        self._folded = f

    @staticmethod
    def _conv(x, wb, pad, relu, stride=1):
        w, b = wb
        y = conv2d_mfma.conv2d(x, w, bias=b, padding=pad, act='relu' if relu else None, gain=1.0 if relu else None, tap_major=True)
        return y[:, :, ::stride, ::stride].contiguous() if stride > 1 else y

    @torch.no_grad()
    def forward(self, x):
        """x [N,3,224,224] in [0,1] (extract_3dmm.py:133: np.array(im) / 255) -> [N,257]."""
        if self._folded is None:
            self._prepare()
        f = self._folded
        x = x.float()
        n, _, H, W = x.shape
        oh, ow = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        cols = F.unfold(x, kernel_size=7, padding=3, stride=2).reshape(n, 147, oh, ow)
        y = self._conv(cols, f['stem'], 0, True)
        y = F.max_pool2d(y, kernel_size=3, stride=2, padding=1)
        for li in (1, 2, 3, 4):
            for bi, blk in enumerate(getattr(self.backbone, f'layer{li}')):
                key = f'l{li}.{bi}'
                out = self._conv(y, f[key + '.c1'], 0, True)
                out = self._conv(out, f[key + '.c2'], 1, True, stride=blk.stride)
                out = self._conv(out, f[key + '.c3'], 0, False)
                sc = y
                if blk.downsample is not None:
                    sc = self._conv(y[:, :, ::blk.stride, ::blk.stride].contiguous() if blk.stride > 1 else y, f[key + '.ds'], 0, False)
                y = F.relu(out + sc)
        pooled = y.mean(dim=(2, 3))
        wh, bh = f['heads']
        return torch.addmm(bh, pooled, wh)


def define_net_recon(net_recon, use_last_fc=False, init_path=None):
    return ReconNetWrapper(net_recon, use_last_fc=use_last_fc, init_path=init_path)
