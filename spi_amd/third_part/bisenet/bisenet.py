"""BiSeNet face parsing (19 classes), inference on the MI355X matrix cores -- SURVEY.md 8f-4, the input producer of the parsing masks
SPI reads (`preprocess/extract_mask.py:35-62` -> `third_part/bisenet/bisenet.py:231-256`, `resnet.py:62-95`).

Same constructor, module tree and parameter names as the reference's ``BiSeNet(n_classes)`` (so ``load_state_dict(torch.load('bisenet.pth'))``
works unchanged: cp.resnet.*, cp.arm16/arm32, cp.conv_head16/32, cp.conv_avg, ffm.*, conv_out/conv_out16/conv_out32), same forward contract
``(feat_out, feat_out16, feat_out32)`` = three ``[N, n_classes, H, W]`` logit maps.  The modules only HOLD the parameters; the forward is one
functional inference pass in which
  * every convolution runs on ``spi_conv2d_fwd`` (fp32 MFMA implicit GEMM) with the eval-mode BatchNorm that follows folded into its
    weights and the ReLU fused into the kernel's epilogue; stride-2 3x3 convs are evaluated at stride 1 and decimated, stride-2 1x1
    convs decimate their input first (the same numbers); the 7x7 / stride-2 stem is an im2col (``F.unfold``) + a 1x1 MFMA conv with
    147 input channels;
  * max-pool, the global average pools, the 1x1 attention convs on pooled vectors, the sigmoid gates and the final
    ``align_corners=True`` bilinear upsampling are a handful of library launches.
Unlike the reference, constructing the module never downloads ResNet-18 weights (resnet.py:79-85): they arrive with the checkpoint.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...torch_utils.ops import conv2d_mfma


class ConvBNReLU(nn.Module):
    def __init__(self, in_chan, out_chan, ks=3, stride=1, padding=1, *args, **kwargs):
        super().__init__()
        self.conv = nn.Conv2d(in_chan, out_chan, kernel_size=ks, stride=stride, padding=padding, bias=False)
        self.bn = nn.BatchNorm2d(out_chan)


class BiSeNetOutput(nn.Module):
    def __init__(self, in_chan, mid_chan, n_classes, *args, **kwargs):
        super().__init__()
        self.conv = ConvBNReLU(in_chan, mid_chan, ks=3, stride=1, padding=1)
        self.conv_out = nn.Conv2d(mid_chan, n_classes, kernel_size=1, bias=False)


class AttentionRefinementModule(nn.Module):
    def __init__(self, in_chan, out_chan, *args, **kwargs):
        super().__init__()
        self.conv = ConvBNReLU(in_chan, out_chan, ks=3, stride=1, padding=1)
        self.conv_atten = nn.Conv2d(out_chan, out_chan, kernel_size=1, bias=False)
        self.bn_atten = nn.BatchNorm2d(out_chan)


class BasicBlock(nn.Module):
    def __init__(self, in_chan, out_chan, stride=1):
        super().__init__()
        self.stride = stride
        self.conv1 = nn.Conv2d(in_chan, out_chan, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(out_chan)
        self.conv2 = nn.Conv2d(out_chan, out_chan, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(out_chan)
        self.downsample = None
        if in_chan != out_chan or stride != 1:
            self.downsample = nn.Sequential(nn.Conv2d(in_chan, out_chan, kernel_size=1, stride=stride, bias=False), nn.BatchNorm2d(out_chan))


def create_layer_basic(in_chan, out_chan, bnum, stride=1):
    return nn.Sequential(BasicBlock(in_chan, out_chan, stride=stride), *[BasicBlock(out_chan, out_chan, stride=1) for _ in range(bnum - 1)])


class Resnet18(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.layer1 = create_layer_basic(64, 64, bnum=2, stride=1)
        self.layer2 = create_layer_basic(64, 128, bnum=2, stride=2)
        self.layer3 = create_layer_basic(128, 256, bnum=2, stride=2)
        self.layer4 = create_layer_basic(256, 512, bnum=2, stride=2)


class ContextPath(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        self.resnet = Resnet18()
        self.arm16 = AttentionRefinementModule(256, 128)
        self.arm32 = AttentionRefinementModule(512, 128)
        self.conv_head32 = ConvBNReLU(128, 128, ks=3, stride=1, padding=1)
        self.conv_head16 = ConvBNReLU(128, 128, ks=3, stride=1, padding=1)
        self.conv_avg = ConvBNReLU(512, 128, ks=1, stride=1, padding=0)


class FeatureFusionModule(nn.Module):
    def __init__(self, in_chan, out_chan, *args, **kwargs):
        super().__init__()
        self.convblk = ConvBNReLU(in_chan, out_chan, ks=1, stride=1, padding=0)
        self.conv1 = nn.Conv2d(out_chan, out_chan // 4, kernel_size=1, stride=1, padding=0, bias=False)
        self.conv2 = nn.Conv2d(out_chan // 4, out_chan, kernel_size=1, stride=1, padding=0, bias=False)


def _bn_affine(bn):
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    return scale, bn.bias - bn.running_mean * scale


class BiSeNet(nn.Module):
    def __init__(self, n_classes, *args, **kwargs):
        super().__init__()
        self.n_classes = n_classes
        self.cp = ContextPath()
        self.ffm = FeatureFusionModule(256, 256)
        self.conv_out = BiSeNetOutput(256, 256, n_classes)
        self.conv_out16 = BiSeNetOutput(128, 64, n_classes)
        self.conv_out32 = BiSeNetOutput(128, 64, n_classes)
        self._folded = None
        self.eval()

    def train(self, mode=True):
        if mode:
            raise RuntimeError('BiSeNet here is inference-only (eval-mode BatchNorm is folded into the convolutions)')
        return super().train(False)

    def load_state_dict(self, *args, **kw):
        self._folded = None
        return super().load_state_dict(*args, **kw)

    def _apply(self, fn, *a, **kw):
        self._folded = None
        return super()._apply(fn, *a, **kw)

    # ---- folded parameters ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _prepare(self):
        def fold(conv, bn):
            scale, shift = _bn_affine(bn)
            w = conv.weight * scale.reshape(-1, 1, 1, 1)
            if conv.kernel_size == (7, 7):                               # stem: im2col columns (c, ky, kx) -> a 1x1 conv over 147 channels
                w = w.reshape(w.shape[0], -1, 1, 1)
            return conv2d_mfma.to_tap_major(w.float()), shift.float().contiguous()
        f = {}
        r = self.cp.resnet
        f['stem'] = fold(r.conv1, r.bn1)
        for li in (1, 2, 3, 4):
            for bi, blk in enumerate(getattr(r, f'layer{li}')):
                f[f'l{li}.{bi}.c1'] = fold(blk.conv1, blk.bn1)
                f[f'l{li}.{bi}.c2'] = fold(blk.conv2, blk.bn2)
                if blk.downsample is not None:
                    f[f'l{li}.{bi}.ds'] = fold(blk.downsample[0], blk.downsample[1])
        for name, m in (('arm16', self.cp.arm16), ('arm32', self.cp.arm32)):
            f[name] = fold(m.conv.conv, m.conv.bn)
            scale, shift = _bn_affine(m.bn_atten)
            f[name + '.att'] = ((m.conv_atten.weight.reshape(m.conv_atten.weight.shape[0], -1) * scale.reshape(-1, 1)).float(), shift.float())
        for name, m in (('head32', self.cp.conv_head32), ('head16', self.cp.conv_head16), ('ffm', self.ffm.convblk),
                        ('out', self.conv_out.conv), ('out16', self.conv_out16.conv), ('out32', self.conv_out32.conv)):
            f[name] = fold(m.conv, m.bn)
        scale, shift = _bn_affine(self.cp.conv_avg.bn)
        f['avg'] = ((self.cp.conv_avg.conv.weight.reshape(128, 512) * scale.reshape(-1, 1)).float(), shift.float())
        for name, m in (('out', self.conv_out), ('out16', self.conv_out16), ('out32', self.conv_out32)):
            f[name + '.cls'] = conv2d_mfma.to_tap_major(m.conv_out.weight.float())
        f['ffm.a1'] = self.ffm.conv1.weight.reshape(self.ffm.conv1.weight.shape[0], -1).float()
        f['ffm.a2'] = self.ffm.conv2.weight.reshape(self.ffm.conv2.weight.shape[0], -1).float()
        self._folded = f

    # ---- forward ------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _conv(x, wb, pad, relu=True, stride=1):
        w, b = wb
        y = conv2d_mfma.conv2d(x, w, bias=b, padding=pad, act='relu' if relu else None, gain=1.0 if relu else None, tap_major=True)
        return y[:, :, ::stride, ::stride] if stride > 1 else y

    def _block(self, x, key, stride, has_ds):
        f = self._folded
        res = self._conv(x, f[key + '.c1'], 1, relu=True, stride=stride)
        res = self._conv(res, f[key + '.c2'], 1, relu=False)
        sc = x
        if has_ds:
            sc = self._conv(x[:, :, ::stride, ::stride].contiguous(), f[key + '.ds'], 0, relu=False)
        return F.relu(sc + res)

    @torch.no_grad()
    def forward(self, x, aux=True):
        """x [N,3,H,W] in [-1,1] (extract_mask.py:58-59 feeds (image / 127.5 - 1)) -> (feat_out, feat_out16, feat_out32); aux=False
        skips the two auxiliary heads (only the first output is read by cal_mask / cal_face_mask, extract_mask.py:17,39)."""
        if self._folded is None:
            self._prepare()
        f = self._folded
        x = x.float()
        n, _, H, W = x.shape
        # 7x7 / stride 2 / pad 3 stem as im2col + 1x1 conv
        oh, ow = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        cols = F.unfold(x, kernel_size=7, padding=3, stride=2).reshape(n, 147, oh, ow)
        y = self._conv(cols, f['stem'], 0, relu=True)
        y = F.max_pool2d(y, kernel_size=3, stride=2, padding=1)
        r = self.cp.resnet
        feats = {}
        for li in (1, 2, 3, 4):
            for bi, blk in enumerate(getattr(r, f'layer{li}')):
                y = self._block(y, f'l{li}.{bi}', blk.stride, blk.downsample is not None)
            feats[li] = y
        feat8, feat16, feat32 = feats[2], feats[3], feats[4]

        def arm(feat, key):
            v = self._conv(feat, f[key], 1, relu=True)
            wa, ba = f[key + '.att']
            att = torch.sigmoid(v.mean(dim=(2, 3)) @ wa.t() + ba)
            return v * att[:, :, None, None]
        wavg, bavg = f['avg']
        avg = F.relu(feat32.mean(dim=(2, 3)) @ wavg.t() + bavg)[:, :, None, None]              # conv_avg on the pooled vector; 'nearest' upsampling = broadcast
        feat32_sum = arm(feat32, 'arm32') + avg
        feat32_up = self._conv(F.interpolate(feat32_sum, feat16.shape[2:], mode='nearest'), f['head32'], 1)
        feat16_sum = arm(feat16, 'arm16') + feat32_up
        feat16_up = self._conv(F.interpolate(feat16_sum, feat8.shape[2:], mode='nearest'), f['head16'], 1)
        # feature fusion (the spatial path is replaced by the res3b1 feature, bisenet.py:245-246)
        feat = self._conv(torch.cat([feat8, feat16_up], dim=1), f['ffm'], 0)
        att = torch.sigmoid(F.relu(feat.mean(dim=(2, 3)) @ f['ffm.a1'].t()) @ f['ffm.a2'].t())
        fuse = feat * att[:, :, None, None] + feat

        def head(v, key):
            v = self._conv(v, f[key], 1)
            v = conv2d_mfma.conv2d(v, f[key + '.cls'], padding=0, tap_major=True)
            return F.interpolate(v, (H, W), mode='bilinear', align_corners=True)
        out = head(fuse, 'out')
        if not aux:
            return out, None, None
        return out, head(feat16_up, 'out16'), head(feat32_up, 'out32')
