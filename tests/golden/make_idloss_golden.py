#!/usr/bin/env python3
"""Golden vectors of the identity metric from the REFERENCE's own code: imports
spi/criteria/id_loss/{model_irse,helpers}.py from /root/reference (build container only), loads
``oracle.irse_ref.synthetic_state_dict(seed)`` into its ``Backbone(112, 50, 'ir_se')`` and records features / similarities
for seeded inputs.  The committed idloss.npz is data (inputs are regenerated from the seeds it stores).

    python tests/golden/make_idloss_golden.py
"""
import os
import sys

os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path[:0] = [ROOT, REF, REF + '/eg3d', REF + '/spi']

import numpy as np
import torch

from oracle import irse_ref  # noqa: E402
from spi.criteria.id_loss.model_irse import Backbone  # noqa: E402  (the reference)

torch.set_num_threads(8)
SEED = 3


def inputs():
    g = torch.Generator().manual_seed(77)
    faces = torch.rand(2, 3, 112, 112, generator=g) * 2 - 1
    img_a = torch.rand(1, 3, 512, 512, generator=g) * 2 - 1
    img_b = (img_a + 0.3 * torch.randn(1, 3, 512, 512, generator=g)).clamp(-1, 1)
    return faces, img_a, img_b


@torch.no_grad()
def main():
    sd = irse_ref.synthetic_state_dict(SEED)
    net = Backbone(input_size=112, num_layers=50, drop_ratio=0.6, mode='ir_se').eval()
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith('num_batches_tracked') for k in missing), (missing, unexpected)
    ref_shapes = {k: tuple(v.shape) for k, v in net.state_dict().items() if not k.endswith('num_batches_tracked')}
    assert ref_shapes == {k: tuple(s) for k, s, _ in irse_ref.state_dict_spec()}
    faces, img_a, img_b = inputs()
    feats = net(faces)
    pool = torch.nn.AdaptiveAvgPool2d((112, 112))                       # id_loss.py:13,17-21 restated with the reference's modules

    def extract(x):
        return net(pool(x[:, :, 35:223, 32:220]))
    fa, fb = extract(img_a), extract(img_b)
    sim = fa[0].dot(fb[0])
    o_feats = irse_ref.backbone_forward(sd, faces)
    o_sim = irse_ref.calculate_similarity(sd, img_a, img_b)
    print('pin backbone features  max|ref-oracle| =', (feats - o_feats).abs().max().item())
    print('pin similarity         |ref-oracle|    =', (sim - o_sim).abs().item(), ' value', sim.item())
    np.savez_compressed(os.path.join(HERE, 'idloss.npz'), seed=SEED, feats=feats.numpy(), feat_a=fa.numpy(), feat_b=fb.numpy(),
                        similarity=sim.numpy(), faces_checksum=faces.double().sum().numpy(), img_checksum=(img_a.double().sum() + img_b.double().sum()).numpy())
    print('wrote idloss.npz')


if __name__ == '__main__':
    main()
