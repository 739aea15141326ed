"""Feature extractor of the W projector (``first_inv_type='sg'``): the callable the reference loads as NVIDIA's TorchScript
``vgg16.pt`` (spi/utils/load_utils.py:47-50, paths_config.VGG_PATH) and calls as
``vgg16(img_0..255, resize_images=False, return_lpips=True)`` (spi/training/projectors/w_projector.py:51,86).

That blob is not part of the reference tree and cannot be fetched offline, so its published contract is restated here on the
MI355X conv kernels ("parity unpinned" at this edge): images in [0, 255] -> [-1, 1] -> LPIPS z-score -> the
five VGG16 taps (relu1_2 ... relu5_3), unit-normalised over channels, scaled by sqrt(lin / (H W)) and flattened, so that the
squared L2 distance between two feature vectors IS the LPIPS-VGG distance.  Weights: the same container LPIPS uses
(``{'convs': [(w, b)] * 13, 'lins': [c] * 5}``) -- converted torchvision / LPIPS weights when the user has them, seeded ones
with ``--synthetic``.
"""
import torch

from .lpips.networks import VGG16, N_CHANNELS


class SgVgg16(torch.nn.Module):
    def __init__(self, weights=None, seed=0):
        super().__init__()
        self.net = VGG16(weights=weights, seed=seed)
        if weights is not None:
            lins = [l.reshape(-1) for l in weights['lins']]
        else:
            g = torch.Generator().manual_seed(seed + 1000)
            lins = [torch.rand(c, generator=g) / c for c in N_CHANNELS]
        for i, l in enumerate(lins):
            self.register_buffer(f'lin{i}', l.clone().float().contiguous())

    def forward(self, img, resize_images=False, return_lpips=True):
        if resize_images or not return_lpips:
            raise NotImplementedError('the W projector calls vgg16(img, resize_images=False, return_lpips=True) only (w_projector.py:51,86)')
        feats = self.net(img.float() / 127.5 - 1)
        out = []
        for i, f in enumerate(feats):
            hw = f.shape[2] * f.shape[3]
            f = f / (torch.sqrt(torch.sum(f * f, dim=1, keepdim=True)) + 1e-10)
            out.append((f * torch.sqrt(getattr(self, f'lin{i}') / hw).view(1, -1, 1, 1)).flatten(1))
        return torch.cat(out, dim=1)
