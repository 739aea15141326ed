"""Identity similarity (ArcFace IR-SE50 cosine), mirror of the reference's ``IDLoss`` (spi/criteria/id_loss/id_loss.py:7-75).

Used by ``Metric.run`` only (spi/utils/metric_utils.py:9-16): crop [35:223, 32:220], adaptive average pool to 112^2, backbone,
dot product of the unit feature vectors.  Needs the trained checkpoint ``paths_config.IDLOSS_PATH``; a missing file raises (no
silent stand-in weights)."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .model_irse import Backbone


class IDLoss(nn.Module):
    def __init__(self, path_ir_se50, num_scales=1, state_dict=None):
        super().__init__()
        self.facenet = Backbone(input_size=112, num_layers=50, drop_ratio=0.6, mode='ir_se')
        if state_dict is None:
            if not os.path.isfile(path_ir_se50):
                raise FileNotFoundError(f'IDLoss: ArcFace checkpoint not found: {path_ir_se50}')
            state_dict = torch.load(path_ir_se50, map_location='cpu', weights_only=True)
        self.facenet.load_state_dict(state_dict)
        self.face_pool = nn.AdaptiveAvgPool2d((112, 112))
        self.facenet.eval()
        self.num_scales = num_scales

    def train(self, mode=True):
        return super().train(False)

    def extract_feats(self, x):
        x = x[:, :, 35:223, 32:220]                   # the reference's fixed crop (:18)
        return self.facenet(self.face_pool(x))

    def calculate_similarity(self, x, y):
        assert x.shape[0] == 1
        return self.extract_feats(x)[0].dot(self.extract_feats(y)[0])

    def calculate_batch_similarity(self, x, y):
        return (self.extract_feats(x) * self.extract_feats(y)).sum(-1).mean()

    def forward(self, x, y):
        n, loss = x.shape[0], 0.0
        for scale in range(self.num_scales):
            loss = loss + (1 - (self.extract_feats(y) * self.extract_feats(x)).sum(-1)).sum()
            if scale != self.num_scales - 1:
                x = F.interpolate(x, mode='bilinear', scale_factor=0.5, align_corners=False, recompute_scale_factor=True)
                y = F.interpolate(y, mode='bilinear', scale_factor=0.5, align_corners=False, recompute_scale_factor=True)
        return loss / n

    def psp_forward(self, y_hat, y, x):
        """(:52-75) loss = mean(1 - <f(y_hat), f(y)>), improvement over <f(y), f(x)>, per-sample log."""
        fx, fy, fh = self.extract_feats(x), self.extract_feats(y).detach(), self.extract_feats(y_hat)
        d_target, d_input, d_views = (fh * fy).sum(-1), (fh * fx).sum(-1), (fy * fx).sum(-1)
        logs = [{'diff_target': float(a), 'diff_input': float(b), 'diff_views': float(c)} for a, b, c in zip(d_target, d_input, d_views)]
        n = x.shape[0]
        return (1 - d_target).sum() / n, sum(l['diff_target'] - l['diff_views'] for l in logs) / n, logs
