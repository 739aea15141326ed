"""Evaluation metrics of an inversion (mirror of spi/utils/metric_utils.py:6-28): L2, LPIPS and identity similarity between
the input photo and the re-synthesised view (and their mirrored counterparts, base_coach.py:141-152).

The identity term is ArcFace cosine similarity from an IR-SE50 checkpoint (``paths_config.IDLOSS_PATH``) in the reference;
neither that checkpoint nor its backbone definition's weights exist offline, so ``Metric`` takes the identity function as an
injectable callable ``id_fn(gt, fake) -> 0-dim tensor`` and reports NaN for it when none is given.  L2 and LPIPS run on
the HIP LPIPS path."""
import math

import torch

from ..criteria.l2_loss import l2_loss
from ..criteria.lpips.lpips import LPIPS


class Metric:
    def __init__(self, lpips_loss=None, id_fn=None, device='cuda'):
        self.lpips_loss = lpips_loss if lpips_loss is not None else LPIPS(net_type='vgg').to(device).eval()
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
