"""VGG feature extractors for the perceptual losses, on the MFMA conv kernel.

Mirrors spi/criteria/lpips/networks.py:36-96 (BaseNet z-score, VGG16 taps after relu1_2, 2_2, 3_3,
4_3, 5_3) and the VGG19 ``features[:6]`` head of spi/criteria/bbox_cx_loss.py:76-90.  The reference
pulls the weights from torchvision / a download (neither exists offline): here they come from a
torchvision-format ``state_dict`` file if one is given, else from a seeded He-style init
("synthetic", parity unpinned at that edge -- SURVEY.md 8c).  All parameters are frozen.
"""
import math
import torch

from ...torch_utils.ops import conv2d_mfma

VGG16_CFG = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512)
VGG16_TAPS = (1, 3, 6, 9, 12)
N_CHANNELS = (64, 128, 256, 512, 512)


def normalize_activation(x, eps=1e-10):
    return x / (torch.sqrt(torch.sum(x ** 2, dim=1, keepdim=True)) + eps)


def _seeded_convs(cfg, cin, seed):
    g = torch.Generator().manual_seed(seed)
    out = []
    for v in cfg:
        if v == 'M':
            continue
        out.append((torch.randn(v, cin, 3, 3, generator=g) * math.sqrt(2.0 / (cin * 9)), torch.randn(v, generator=g) * 0.05))
        cin = v
    return out, g


class ConvStack(torch.nn.Module):
    """conv3x3 + bias + ReLU layers (fused epilogue of the MFMA kernel) with 2x2 max-pools in between."""
    def __init__(self, cfg, convs, last_relu=True):
        super().__init__()
        self.cfg = tuple(cfg)
        self.last_relu = last_relu
        self.n_conv = len(convs)
        for i, (w, b) in enumerate(convs):
            self.register_buffer(f'w{i}', conv2d_mfma.to_tap_major(w.clone().float()))      # [O,3,3,I]
            self.register_buffer(f'b{i}', b.clone().float())

    def run(self, x, taps=None):
        feats, ci = [], 0
        for v in self.cfg:
            if v == 'M':
                x = torch.nn.functional.max_pool2d(x, 2, 2)
                continue
            relu = self.last_relu or ci < self.n_conv - 1
            x = conv2d_mfma.conv2d(x, getattr(self, f'w{ci}'), bias=getattr(self, f'b{ci}'), padding=1, act='relu' if relu else None,
                                   gain=1.0 if relu else None, tap_major=True)
            if taps is not None and ci in taps:
                feats.append(x)
            ci += 1
        return feats if taps is not None else x


class VGG16(torch.nn.Module):
    def __init__(self, weights=None, seed=0):
        super().__init__()
        convs = weights['convs'] if weights is not None else _seeded_convs(VGG16_CFG, 3, seed)[0]
        self.layers = ConvStack(VGG16_CFG, convs)
        self.n_channels_list = list(N_CHANNELS)
        self.register_buffer('mean', torch.tensor([-.030, -.088, -.188]).view(1, 3, 1, 1))
        self.register_buffer('std', torch.tensor([.458, .448, .450]).view(1, 3, 1, 1))

    def forward(self, x):
        """-> list of the five RAW tap activations (the unit-normalisation lives in the fused LPIPS tail kernel)."""
        return self.layers.run((x - self.mean) / self.std, taps=VGG16_TAPS)


class VGG19Head(torch.nn.Module):
    """torchvision vgg19.features[:6]: conv-relu-conv-relu-maxpool-conv."""
    def __init__(self, weights=None, seed=1):
        super().__init__()
        convs = weights if weights is not None else _seeded_convs((64, 64, 'M', 128), 3, seed)[0]
        self.slice1 = ConvStack((64, 64, 'M', 128), convs, last_relu=False)

    def forward(self, x):
        return self.slice1.run(x)


def load_torchvision_state(path, n_convs):
    """[(weight, bias)] from a torchvision ``vgg*.features`` state_dict file."""
    sd = torch.load(path, map_location='cpu')
    keys = sorted({int(k.split('.')[-2]) for k in sd if k.endswith('.weight') and sd[k].ndim == 4})
    pre = 'features.' if any(k.startswith('features.') for k in sd) else ''
    return [(sd[f'{pre}{i}.weight'], sd[f'{pre}{i}.bias']) for i in keys[:n_convs]]
