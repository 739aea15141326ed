"""Parsing-mask producer of the SPI dataset (mirror of preprocess/extract_mask.py:14-62): BiSeNet logits -> 19-class parsing map
(``cal_mask``, what ``extract_mask`` stores as ``<name>.pt`` and ``PTIDataset`` reads back) or the binary face mask at 256^2
(``cal_face_mask``).  The network is built on first use from ``paths_config.BISENET_PATH`` (spi/utils/load_utils.py:36-44)."""
import glob
import os

import numpy as np
import torch
import torch.nn.functional as F

from ..configs import global_config

FACE_ATTS = (1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 13)     # skin, brows, eyes, glasses, ears, nose, mouth, lips (extract_mask.py:19-21)


def cal_mask(bisenet, image):
    """image [N,3,512,512] in [-1,1] -> int64 parsing labels [N,1,512,512]"""
    out = bisenet(image.clone(), aux=False)[0] if _takes_aux(bisenet) else bisenet(image.clone())[0]
    return torch.argmax(out, dim=1, keepdim=True)


def cal_face_mask(bisenet, image):
    parsing = cal_mask(bisenet, image)
    mask = torch.zeros(image.shape[0], 1, *parsing.shape[-2:], device=image.device)
    for att in FACE_ATTS:
        mask += (parsing == att)
    return F.interpolate(mask, size=(256, 256), mode='nearest')


def _takes_aux(net):
    from ..third_part.bisenet import BiSeNet
    return isinstance(net, BiSeNet)


_net = [None]


def get_bisenet():
    if _net[0] is None:
        from ..utils.load_utils import load_bisenet
        _net[0] = load_bisenet()
    return _net[0]


def extract_mask(input_dir, output_dir, mode='png', bisenet=None, device=None):
    from PIL import Image
    bisenet = bisenet if bisenet is not None else get_bisenet()
    device = device or global_config.device
    os.makedirs(output_dir, exist_ok=True)
    for image_path in sorted(glob.glob(f'{input_dir}/*.{mode}')):
        image_name = os.path.basename(image_path).split('.')[0]
        image = Image.open(image_path).resize((512, 512))
        image = torch.from_numpy(np.array(image)).unsqueeze(0).permute(0, 3, 1, 2)
        image = image.to(device).to(torch.float32) / 127.5 - 1
        torch.save(cal_mask(bisenet, image).cpu(), os.path.join(output_dir, image_name + '.pt'))
