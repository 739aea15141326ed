"""GPU parity of the input producer: BiSeNet face parsing on the MFMA conv kernels against the logits of the reference's
own network (tests/golden/bisenet.npz, third_part/bisenet on seeded weights) and the oracle, plus the extract_mask file contract."""
import json
import os

import numpy as np
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


# ---- crop + camera producer (Deep3DFaceRecon regressor + camera arithmetic) ---------------------------------------------------------------

def _recon(seed):
    from oracle import recon_ref as orr2
    from spi_amd.third_part.Deep3DFaceRecon_pytorch.models.networks import ReconNetWrapper
    man = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, 'tests', 'golden', 'manifest_recon.json'))).items()}
    sd = orr2.synthetic_state_dict(man, seed=seed)
    net = ReconNetWrapper('resnet50', use_last_fc=False)
    net.load_state_dict(sd)
    return net.to(DEV), sd


def test_recon_net_vs_reference_golden(golden):
    """The 3DMM regressor (ResNet-50 v1.5 + seven heads, 53 convolutions on spi_conv2d_fwd with folded BatchNorm) against the coefficients of
    the reference's own ReconNetWrapper on the same seeded weights (golden/recon.npz)."""
    g = golden('recon')
    net, _ = _recon(int(g['seed'][0]))
    gen = torch.Generator().manual_seed(int(g['img_seed'][0]))
    img = F.interpolate(torch.rand(2, 3, 28, 28, generator=gen), size=(224, 224), mode='bicubic', align_corners=False).clamp(0, 1)
    out = net(img.to(DEV))
    assert out.shape == (2, 257)
    assert (out.cpu() - g['coeffs']).abs().max().item() <= 1e-4 * g['coeffs'].abs().max().item()
    with pytest.raises(RuntimeError):
        net.train()


def test_camera_extractor_end_to_end_vs_oracle(tmp_path):
    """CameraExtractor.extract (preprocess/extract_camera.py:140-158) on a synthetic photo with an injected landmark detector: the 512^2 crop and
    the 25-float camera label are written, the label equals the oracle's chain (align_img -> CPU ResNet-50 -> cal_camera -> process_camera) on the
    same weights, the camera sits at radius 2.7 with EG3D's normalised intrinsics, and the mirror pair is consistent."""
    import numpy as np
    from PIL import Image
    from oracle import recon_ref as orr2
    from spi_amd.preprocess.extract_camera import CameraExtractor
    net, sd = _recon(0)
    # tame heads: the golden weights give |angle| ~ 15 rad, where 1e-5 of coefficient noise is amplified by the trigonometry
    sd = {k: (v * 0.02 if k.startswith('final_layers') else v) for k, v in sd.items()}
    rng = np.random.default_rng(11)
    photo = Image.fromarray((rng.random((360, 330, 3)) * 255).astype(np.uint8)).resize((720, 660))
    photo.save(tmp_path / 'face.png')
    lm = np.stack([330 + 120 * np.cos(np.linspace(0, 6.2, 68)), 320 + 140 * np.sin(np.linspace(0, 6.2, 68))], axis=1).astype(np.float32)
    from spi_amd.preprocess import extract_3dmm as e3
    up = lm.astype(np.float64).copy()
    up[:, 1] = photo.size[1] - 1 - up[:, 1]
    lm3d = np.concatenate([(e3.extract_5p(up) - up.mean(0)) / 150.0, np.zeros((5, 1))], axis=1)
    (tmp_path / 'crop').mkdir(); (tmp_path / 'c').mkdir()
    ex = CameraExtractor(str(tmp_path / 'crop'), str(tmp_path / 'c'), 'png', landmark_fn=lambda im: lm.copy(), device=DEV, state_dict=sd, lm3d_std=lm3d)
    cam = ex.extract(str(tmp_path / 'face.png'))
    crop = Image.open(tmp_path / 'crop' / 'face.png')
    assert crop.size == (512, 512) and np.array_equal(np.load(tmp_path / 'c' / 'face.npy'), cam) and cam.shape == (25,)
    # oracle chain
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    _, im224, _, im1024 = orr2.align_img(photo.convert('RGB'), up.copy(), lm3d)
    x = torch.tensor(np.array(im224) / 255., dtype=torch.float32).permute(2, 0, 1)[None]
    with torch.no_grad():
        co = orr2.split_coeff(orr2.recon_net(sd, x))
    cc = orr2.cal_camera(co['angle'], co['trans'][0].clone())
    ref = orr2.process_camera(cc['pose'], cc['intrinsics'])
    assert np.abs(cam - ref).max() <= 1e-4, np.abs(cam - ref).max()
    pose = cam[:16].reshape(4, 4)
    assert abs(np.linalg.norm(pose[:3, 3]) - 2.7) < 1e-6 and np.allclose(pose[:3, :3] @ pose[:3, :3].T, np.eye(3), atol=1e-5) and np.allclose(pose[3], [0, 0, 0, 1])
    assert np.allclose(cam[16:], [2985.29 / 700, 0, .5, 0, 2985.29 / 700, .5, 0, 0, 1])
    # the crop is the oracle's: align at rescale 300, centre 700^2 window, LANCZOS to 512
    _, _, _, hi = orr2.align_img(photo.convert('RGB'), up.copy(), lm3d, rescale_factor=300)
    want = hi.crop((162, 162, 862, 862)).resize((512, 512), resample=Image.LANCZOS)
    assert np.array_equal(np.array(crop), np.array(want))
    ex.cal_mirror_c(str(tmp_path / 'face.png'))
    cm_ = np.load(tmp_path / 'c' / 'face_m.npy')
    assert np.allclose(cm_, orr2.mirror_camera(cam)) and np.array_equal(np.array(Image.open(tmp_path / 'crop' / 'face_m.png')), np.array(crop)[:, ::-1])


# ---- landmark producer (round 4): S3FD + 2D-FAN-4 on the HIP convs, the one call the reference makes into the `face_alignment` package ----------
def _fa_nets():
    from spi_amd.third_part import face_alignment as own
    from spi_amd.third_part.face_alignment.api import synthetic_state_dict
    from spi_amd.third_part.face_alignment.fan import FAN
    from spi_amd.third_part.face_alignment.sfd import s3fd
    fan_sd, sfd_sd = synthetic_state_dict(FAN(4), 11), synthetic_state_dict(s3fd(), 12)
    fa = own.FaceAlignment(own.LandmarksType._2D, device=DEV, fan_state_dict=fan_sd, sfd_state_dict=sfd_sd)
    return fa, fan_sd, sfd_sd


@pytest.mark.timeout(900)
def test_fan_and_s3fd_networks_vs_oracle():
    """The two networks of the landmark producer on seeded weights, HIP convs against the CPU oracle's functional restatement
    (oracle/face_alignment_ref.py): all four FAN heat-map stacks and the twelve S3FD head outputs."""
    from oracle import face_alignment_ref as ofr
    fa, fan_sd, sfd_sd = _fa_nets()
    g = torch.Generator().manual_seed(5)
    x = F.interpolate(torch.rand(2, 3, 32, 32, generator=g), size=(256, 256), mode='bicubic', align_corners=False).clamp(0, 1)
    hip_out = fa.face_alignment_net(x.to(DEV))
    ref_out = ofr.fan_forward(fan_sd, x)
    assert len(hip_out) == 4 and hip_out[-1].shape == (2, 68, 64, 64)
    for i, (a, b) in enumerate(zip(hip_out, ref_out)):
        assert_close(a, b, 2e-4, f'FAN stack {i} heat maps')
    img = (torch.rand(1, 3, 160, 128, generator=g) * 255 - 115)
    hip_o = fa.face_detector(img.to(DEV))
    ref_o = ofr.s3fd_forward(sfd_sd, img)
    assert len(hip_o) == 12
    for i, (a, b) in enumerate(zip(hip_o, ref_o)):
        assert a.shape == b.shape
        assert_close(a, b, 2e-4, f'S3FD head output {i}')


@pytest.mark.timeout(900)
def test_landmark_pipeline_vs_oracle():
    """get_landmarks_from_image end to end on seeded weights: detection boxes (softmax, anchors, decoding, NMS, thresholds) and, for a given
    face box, the crop -> FAN -> heat-map decoding -> image coordinates chain, HIP against the oracle; then the preprocess entry point
    (extract_landmark.get_landmark) with the built-in detector injected."""
    from oracle import face_alignment_ref as ofr
    from spi_amd.third_part.face_alignment.sfd import detect
    from spi_amd.preprocess import extract_landmark as el
    fa, fan_sd, sfd_sd = _fa_nets()
    g = torch.Generator().manual_seed(6)
    photo = (F.interpolate(torch.rand(1, 3, 20, 16, generator=g), size=(160, 128), mode='bicubic', align_corners=False).clamp(0, 1)[0].permute(1, 2, 0) * 255).to(torch.uint8).numpy()
    d_hip, d_ref = detect(fa.face_detector, photo, DEV), ofr.detect(sfd_sd, photo)
    # random weights: scores sit near 0.5, so a box can fall on either side of a threshold; the bulk must coincide
    assert abs(len(d_hip) - len(d_ref)) <= max(2, len(d_ref) // 20), (len(d_hip), len(d_ref))
    matched = 0
    for b in d_ref:
        tol = 1e-3 * np.maximum(1.0, np.abs(b))                   # (seeded weights: exp(loc) makes some boxes 1e4 pixels wide -- relative tolerance)
        j = (np.abs(d_hip[:, :4] - b[:4]) / tol[:4]).max(axis=1).argmin() if len(d_hip) else None
        matched += int(j is not None and bool((np.abs(d_hip[j] - b) <= tol).all()))
    assert len(d_ref) == 0 or matched >= 0.9 * len(d_ref), (matched, len(d_ref))
    box = np.array([30.0, 40.0, 100.0, 130.0, 0.9])
    lm_hip = fa.get_landmarks_from_image(photo, detected_faces=[box])[0]
    lm_ref = ofr.landmarks_from_face(fan_sd, photo, box)
    assert lm_hip.shape == (68, 2)
    # an arg-max over a random-weight heat map can tie within round-off: the landmarks must coincide except for isolated flips
    same = np.abs(lm_hip - lm_ref).max(axis=1) < 1e-3
    assert same.mean() >= 0.95, (same.mean(), np.abs(lm_hip - lm_ref).max())
    from PIL import Image
    lm = el.get_landmark(Image.fromarray(photo), landmark_fn=lambda im: fa.get_landmarks_from_image(np.array(im), detected_faces=[box])[0])
    assert lm.shape == (68, 2) and lm.dtype == np.float32 and np.array_equal(lm, lm_hip)
