"""Evaluation metrics of an inversion (mirror of spi/utils/metric_utils.py:6-28): L2, LPIPS and identity similarity between
the input photo and the re-synthesised view (and their mirrored counterparts, base_coach.py:141-152).

The identity term is ArcFace cosine similarity from an IR-SE50 checkpoint (``paths_config.IDLOSS_PATH``): when that file
exists, ``Metric`` builds ``IDLoss`` on it (criteria/id_loss, convolutions on the HIP kernels) exactly like the reference; the
checkpoint is not available offline, so without it the identity column is reported as NaN (never from stand-in weights).  An
``id_fn(gt, fake) -> 0-dim tensor`` can be injected instead.  L2 and LPIPS run on the HIP LPIPS path."""
import math
import os

import torch

from ..criteria.l2_loss import l2_loss
from ..criteria.lpips.lpips import LPIPS
from ..configs import paths_config


class Metric:
    def __init__(self, lpips_loss=None, id_fn=None, device='cuda'):
        if lpips_loss is None:
            from ..criteria import weights as pretrained
            lpips_loss = LPIPS(net_type='vgg', weights=pretrained.lpips_vgg16_weights()).to(device).eval()
        self.lpips_loss = lpips_loss
        if id_fn is None and os.path.isfile(paths_config.IDLOSS_PATH):
            from ..criteria.id_loss import IDLoss
            id_fn = IDLoss(paths_config.IDLOSS_PATH).to(device).eval().calculate_similarity
        self.id_fn = id_fn

    @torch.no_grad()
    def run(self, gt, fake):
        l2 = l2_loss(gt, fake)
        lp = self.lpips_loss(gt, fake)
        ids = self.id_fn(gt, fake) if self.id_fn is not None else None
        return float(l2), float(lp), (float(ids) if ids is not None else math.nan)


@torch.no_grad()
def metric(gt, w, c, G, lpips_func):
    fake = G.synthesis(w, c, noise_mode='const')['image']
    return l2_loss(gt, fake), lpips_func(gt, fake)
