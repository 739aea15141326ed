"""GPU parity of the input producer: BiSeNet face parsing on the MFMA conv kernels against the logits of the reference's
own network (tests/golden/bisenet.npz, third_part/bisenet on seeded weights) and the oracle, plus the extract_mask file contract."""
import json
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import ROOT, assert_close, rel_err
from oracle import bisenet_ref as obr

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _net(seed):
    from spi_amd.third_part.bisenet import BiSeNet
    man = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, 'tests', 'golden', 'manifest_bisenet.json'))).items()}
    sd = obr.synthetic_state_dict(man, seed=seed)
    net = BiSeNet(19)
    net.load_state_dict(sd)
    return net.to(DEV), sd


def _photo(seed):
    g = torch.Generator().manual_seed(seed)
    return F.interpolate(torch.rand(1, 3, 24, 24, generator=g), size=(512, 512), mode='bicubic', align_corners=False).clamp(0, 1) * 2 - 1


def test_bisenet_vs_reference_golden(golden):
    g = golden('bisenet')
    net, sd = _net(int(g['seed'][0]))
    img = _photo(int(g['img_seed'][0]))
    out, out16, out32 = net(img.to(DEV))
    assert out.shape == (1, 19, 512, 512)
    scale = g['out_absmax'].item()
    for a, b, nm, st in ((out, g['out_sub'], 'feat_out', 8), (out16, g['out16_sub'], 'feat_out16', 16), (out32, g['out32_sub'], 'feat_out32', 16)):
        assert (a[:, :, ::st, ::st].cpu() - b).abs().max().item() <= 1e-4 * scale, nm        # 20 convolutions deep, eval-BN folded
    assert_close(out.mean(dim=(2, 3)), g['out_mean'], 1e-4, 'mean logits')
    # the parsing map (what extract_mask stores): identical wherever the reference's top-2 margin is not within rounding
    from spi_amd.preprocess.extract_mask import cal_mask, cal_face_mask
    parsing = cal_mask(net, img.to(DEV)).cpu()
    ref = g['parsing'].long()
    assert parsing.dtype == torch.int64 and parsing.shape == (1, 1, 512, 512)
    assert (parsing != ref).float().mean().item() < 2e-4
    top2 = out.topk(2, dim=1).values
    safe = ((top2[:, 0] - top2[:, 1]) > 1e-3 * scale).cpu()
    assert torch.equal(parsing[:, 0][safe], ref[:, 0][safe])
    fm = cal_face_mask(net, img.to(DEV))
    assert fm.shape == (1, 1, 256, 256) and set(fm.unique().tolist()) <= {0.0, 1.0}
    # second input: batch of 2, non-square, vs the reference golden and the oracle
    img2 = torch.rand(2, 3, 96, 160, generator=torch.Generator().manual_seed(62)) * 2 - 1
    o2 = net(img2.to(DEV), aux=False)[0]
    assert (o2[:, :, ::4, ::4].cpu() - g['out2_sub']).abs().max().item() <= 1e-4 * g['out2_sub'].abs().max().item()
    with torch.no_grad():
        assert_close(o2, obr.bisenet_forward(sd, img2)[0], 1e-4, 'batch-2 logits vs oracle')


def test_extract_mask_file_contract(tmp_path):
    """extract_mask(input_dir, output_dir, mode) (preprocess/extract_mask.py:50-62): one <name>.pt per image holding int64 labels
    [1,1,512,512] -- what PTIDataset loads as data['mask'] -- and load_bisenet reading paths_config.BISENET_PATH."""
    from PIL import Image
    import numpy as np
    from spi_amd.preprocess import extract_mask as em
    from spi_amd.utils import load_utils
    from spi_amd.configs import paths_config
    net, sd = _net(3)
    ck = str(tmp_path / 'bisenet.pth')
    torch.save(sd, ck)
    old = paths_config.BISENET_PATH
    paths_config.BISENET_PATH = ck
    try:
        loaded = load_utils.load_bisenet(device=DEV)
    finally:
        paths_config.BISENET_PATH = old
    src, dst = tmp_path / 'crop', tmp_path / 'mask'
    src.mkdir()
    arr = ((_photo(7)[0].permute(1, 2, 0).numpy() + 1) * 127.5).astype(np.uint8)[:300, :400]
    Image.fromarray(arr).save(src / 'target.png')
    em.extract_mask(str(src), str(dst), mode='png', bisenet=loaded, device=DEV)
    m = torch.load(dst / 'target.pt')
    assert m.dtype == torch.int64 and m.shape == (1, 1, 512, 512) and 0 <= int(m.min()) and int(m.max()) < 19
    img = torch.from_numpy(np.asarray(Image.open(src / 'target.png').resize((512, 512)))).unsqueeze(0).permute(0, 3, 1, 2).float() / 127.5 - 1
    with torch.no_grad():
        ref = obr.cal_mask(sd, img)
    assert (m != ref).float().mean().item() < 1e-3
    with pytest.raises(FileNotFoundError):
        load_utils.load_bisenet(device=DEV, path=str(tmp_path / 'missing.pth'))
