"""``FaceAlignment(LandmarksType._2D).get_landmarks_from_image(image)`` -- the one call the reference makes into the third-party
``face_alignment`` package (preprocess/extract_landmark.py:10,19) -- on this package's own S3FD + 2D-FAN-4 networks (sfd.py, fan.py).

Restated from the package's published 1.x API ("parity unpinned": it is not under /root/reference and cannot be installed offline):
  detect faces -> per face: centre = box centre moved up by 12 % of the box height, scale = (w + h) / 195; crop a (200 * scale)-pixel
  square around the centre, resized to 256^2 (bilinear); FAN -> the last stack's 68 heat maps at 64^2; landmark = arg-max location, moved a
  quarter pixel towards the larger neighbour, minus half a pixel, mapped back through the crop transform.
The networks' trained weights (``s3fd-*.pth``, ``2DFAN4-*.pth(.tar)`` -- downloaded by the package) are read from
``paths_config.SFD_PATH`` / ``paths_config.FAN_PATH`` or handed in; a missing file raises (seeded stand-ins only with ``synthetic=True``, for tests).
"""
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

from .fan import FAN
from .sfd import s3fd, detect


class LandmarksType:
    _2D = 1
    TWO_D = 1
    _2halfD = 2
    _3D = 3


def transform(point, center, scale, resolution, invert=False):
    """a point through the crop transform (or its inverse): the square of side 200 * scale around `center` maps onto [0, resolution]"""
    h = 200.0 * scale
    t = np.eye(3)
    t[0, 0] = t[1, 1] = resolution / h
    t[0, 2] = resolution * (-center[0] / h + 0.5)
    t[1, 2] = resolution * (-center[1] / h + 0.5)
    if invert:
        t = np.linalg.inv(t)
    return (t @ np.array([point[0], point[1], 1.0]))[:2]


def crop(image, center, scale, resolution=256):
    """image uint8 [H,W,3] -> the crop, bilinearly resized to resolution^2 (zeros outside the image)"""
    ul = transform([1, 1], center, scale, resolution, True).astype(np.int64)
    br = transform([resolution, resolution], center, scale, resolution, True).astype(np.int64)
    ht, wd = image.shape[:2]
    new = np.zeros([br[1] - ul[1], br[0] - ul[0], image.shape[2]], dtype=np.float32)
    nx = (max(1, -ul[0] + 1), min(br[0], wd) - ul[0])
    ny = (max(1, -ul[1] + 1), min(br[1], ht) - ul[1])
    ox = (max(1, ul[0] + 1), min(br[0], wd))
    oy = (max(1, ul[1] + 1), min(br[1], ht))
    new[ny[0] - 1:ny[1], nx[0] - 1:nx[1]] = image[oy[0] - 1:oy[1], ox[0] - 1:ox[1], :]
    t = torch.from_numpy(new).permute(2, 0, 1).unsqueeze(0)
    # cv2.resize(..., INTER_LINEAR) in the package: half-pixel-centre bilinear without antialiasing == align_corners=False
    # ... applied to a uint8 image: the package's crop stays uint8 (cv2 rounds to nearest, saturating) before it is scaled by 1 / 255
    r = F.interpolate(t, size=(resolution, resolution), mode='bilinear', align_corners=False)[0]
    return torch.floor(r + 0.5).clamp_(0, 255)


def get_preds_fromhm(hm, center, scale):
    """hm [68,64,64] (host tensor) -> landmarks [68,2] in image coordinates"""
    n, res = hm.shape[0], hm.shape[-1]
    idx = hm.reshape(n, -1).argmax(dim=1)
    preds = torch.stack([(idx % res).float() + 1, (idx // res).float() + 1], dim=1)
    for i in range(n):
        px, py = int(preds[i, 0]) - 1, int(preds[i, 1]) - 1
        if 0 < px < res - 1 and 0 < py < res - 1:
            diff = torch.tensor([hm[i, py, px + 1] - hm[i, py, px - 1], hm[i, py + 1, px] - hm[i, py - 1, px]])
            preds[i] += diff.sign() * 0.25
    preds -= 0.5
    return np.stack([transform(preds[i].numpy(), center, scale, res, True) for i in range(n)]).astype(np.float32)


def synthetic_state_dict(module, seed):
    """seeded stand-in weights (He-scaled, BatchNorm statistics near identity): tests only"""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in module.state_dict().items():
        leaf = k.rsplit('.', 1)[-1]
        if leaf == 'num_batches_tracked':
            sd[k] = torch.zeros_like(v)
        elif leaf == 'running_var':
            sd[k] = 0.8 + 0.4 * torch.rand(v.shape, generator=g)
        elif leaf == 'running_mean' or leaf == 'bias':
            sd[k] = 0.05 * torch.randn(v.shape, generator=g)
        elif v.ndim == 1:
            sd[k] = v.clone() if 'norm' in k else 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        else:
            sd[k] = torch.randn(v.shape, generator=g) * (1.4 / (v.shape[1] * v.shape[2] * v.shape[3]) ** 0.5)
    return sd


class FaceAlignment:
    def __init__(self, landmarks_type=LandmarksType._2D, device='cuda', flip_input=False, face_detector='sfd', fan_state_dict=None,
                 sfd_state_dict=None, synthetic=False):
        if landmarks_type != LandmarksType._2D or face_detector != 'sfd' or flip_input:
            raise NotImplementedError('only FaceAlignment(LandmarksType._2D) with the sfd detector (what preprocess/extract_landmark.py:10 builds)')
        from ...configs import paths_config
        self.device = device
        self.face_alignment_net, self.face_detector = FAN(4), s3fd()
        for net, sd, attr, seed in ((self.face_alignment_net, fan_state_dict, 'FAN_PATH', 11), (self.face_detector, sfd_state_dict, 'SFD_PATH', 12)):
            if sd is None:
                path = getattr(paths_config, attr, '')
                if os.path.isfile(path):
                    sd = torch.load(path, map_location='cpu', weights_only=True)
                    sd = sd.get('state_dict', sd) if isinstance(sd, dict) else sd
                elif synthetic:
                    sd = synthetic_state_dict(net, seed)
                else:
                    raise FileNotFoundError(f'{path!r} (paths_config.{attr}): the {attr[:3]} weights of the third-party face_alignment package are needed for '
                                            'landmark extraction; they are downloaded by that package and are not part of this one')
            net.load_state_dict(sd)
            net.to(device)

    def get_landmarks_from_image(self, image, detected_faces=None):
        """image: uint8 array [H,W,3] (RGB) -> list of [68,2] float32 arrays, one per detected face (best first), or None"""
        image = np.asarray(image)
        if image.ndim == 2:
            image = np.stack([image] * 3, axis=2)
        image = image[..., :3]
        if detected_faces is None:
            detected_faces = detect(self.face_detector, image, self.device)
        if len(detected_faces) == 0:
            return None
        landmarks = []
        for d in detected_faces:
            center = [d[2] - (d[2] - d[0]) / 2.0, d[3] - (d[3] - d[1]) / 2.0]
            center[1] = center[1] - (d[3] - d[1]) * 0.12
            scale = (d[2] - d[0] + d[3] - d[1]) / 195.0
            inp = (crop(image, center, scale) / 255.0).unsqueeze(0).to(self.device)
            hm = self.face_alignment_net(inp)[-1][0].float().cpu()
            landmarks.append(get_preds_fromhm(hm, center, scale))
        return landmarks
