"""Pretrained weights of the perceptual losses.

The reference gets them implicitly: ``torchvision.models.vgg16(True)`` / ``vgg19(pretrained=True)`` pull the ImageNet weights
(spi/criteria/lpips/networks.py:92, spi/criteria/bbox_cx_loss.py:79) and ``get_state_dict`` downloads the LPIPS v0.1 linear
layers (spi/criteria/lpips/utils.py:11-30).  There is no torchvision and no network here, so the same files are read from
``paths_config`` (torchvision ``state_dict`` files and richzhang's ``vgg.pth``), and a missing file is an ERROR: silently
optimising a random-feature perceptual loss would diverge from the reference without a trace.  Seeded stand-ins are used only
when the caller asks for them (``--synthetic`` / ``global_config.synthetic_weights``), i.e. the offline benchmark and tests.
"""
import os
import re
import torch

from ..configs import paths_config, global_config
from .lpips.networks import load_torchvision_state, N_CHANNELS


def _need(path, what):
    if not path or not os.path.isfile(path):
        raise FileNotFoundError(
            f'{what}: {path!r} not found.  The reference downloads these weights; offline they must be placed there '
            f'(see spi_amd/configs/paths_config.py).  Pass --synthetic only for benchmarking / tests with seeded stand-ins.')
    return path


def want_synthetic(synthetic=None):
    return bool(global_config.synthetic_weights if synthetic is None else synthetic)


def load_lpips_lins(path):
    """LPIPS v0.1 linear layers from richzhang's ``vgg.pth`` (keys ``lin<i>.model.1.weight``) or the reference's renamed form
    (``<i>.1.weight``, lpips/utils.py:24-28).  -> five tensors [C_i]"""
    sd = torch.load(path, map_location='cpu', weights_only=True)
    found = {}
    for k, v in sd.items():
        m = re.match(r'^(?:lin)?(\d)\.(?:model\.)?1\.weight$', k)
        if m:
            found[int(m.group(1))] = v.reshape(-1).float()
    if sorted(found) != [0, 1, 2, 3, 4] or [found[i].numel() for i in range(5)] != list(N_CHANNELS):
        raise ValueError(f'{path}: not an LPIPS-VGG v0.1 lin state_dict (keys {list(sd)[:6]}...)')
    return [found[i] for i in range(5)]


def lpips_vgg16_weights(synthetic=None):
    """-> ``{'convs': [(w, b)] * 13, 'lins': [c] * 5}`` for LPIPS / SgVgg16, or None = seeded stand-ins (synthetic mode only)."""
    if want_synthetic(synthetic):
        return None
    convs = load_torchvision_state(_need(paths_config.VGG16_PATH, 'torchvision VGG16 (LPIPS backbone)'), 13)
    if len(convs) != 13:
        raise ValueError(f'{paths_config.VGG16_PATH}: expected the 13 conv layers of vgg16.features, found {len(convs)}')
    return {'convs': convs, 'lins': load_lpips_lins(_need(paths_config.LPIPS_PATH, 'LPIPS v0.1 VGG linear layers'))}


def vgg19_head_weights(synthetic=None):
    """-> [(w, b)] * 3 = torchvision vgg19.features[:6] for BoxCXLoss, or None = seeded stand-ins (synthetic mode only)."""
    if want_synthetic(synthetic):
        return None
    convs = load_torchvision_state(_need(paths_config.VGG19_PATH, 'torchvision VGG19 (BoxCX head)'), 3)
    if len(convs) != 3 or convs[2][0].shape[:2] != (128, 64):
        raise ValueError(f'{paths_config.VGG19_PATH}: not a vgg19 features state_dict')
    return convs
