"""Face alignment and 3DMM coefficients for the crop + camera producer (mirror of preprocess/extract_3dmm.py:15-224).

``align_img`` fits a similarity transform between five facial landmarks and the five standard 3D landmarks of the BFM
(``similarity_Lm3D_all.mat``, util/load_mats.py:105-116) by least squares, rescales the photo so that the face has the standard size and
crops a 1024^2 window around it (PIL LANCZOS, like the reference); ``Extract3dmm.get_3dmm`` runs the 224^2 aligned image through the
ResNet-50 regressor (third_part/Deep3DFaceRecon_pytorch/models/networks.py, here on the MI355X conv kernels) and splits the 257
coefficients (models/bfm.py:252-273).  The reference builds the network through Deep3DFaceRecon's option parser and its checkpoint loader
(``epoch_20.pth``, entry ``net_recon``); here the two files are read directly -- or a state dict / landmark array is handed in, which is how
the tests run without the trained files (they are not redistributable and not available offline).
"""
import os

import numpy as np
import torch


def POS(xp, x):
    """least squares: 2D points xp [2,n] ~ s * R[:2] x + t for 3D points x [3,n] -> (t [2,1], s)"""
    n = xp.shape[1]
    A = np.zeros([2 * n, 8])
    A[0::2, :3], A[0::2, 3] = x.T, 1                     # x-rows: [X Y Z 1 | 0 0 0 0]
    A[1::2, 4:7], A[1::2, 7] = x.T, 1                    # y-rows: [0 0 0 0 | X Y Z 1]
    k = np.linalg.lstsq(A, xp.T.reshape(2 * n, 1), rcond=None)[0]
    s = (np.linalg.norm(k[0:3]) + np.linalg.norm(k[4:7])) / 2
    return np.stack([k[3], k[7]], axis=0), s


def extract_5p(lm):
    """68 landmarks -> (left eye, right eye, nose, left mouth corner, right mouth corner)"""
    idx = np.array([31, 37, 40, 43, 46, 49, 55]) - 1
    lm5p = np.stack([lm[idx[0], :], np.mean(lm[idx[[1, 2]], :], 0), np.mean(lm[idx[[3, 4]], :], 0), lm[idx[5], :], lm[idx[6], :]], axis=0)
    return lm5p[[1, 2, 0, 3, 4], :]


def load_lm3d(bfm_folder):
    """the five standard 3D landmarks from the BFM's similarity_Lm3D_all.mat (util/load_mats.py:105-116)"""
    from scipy.io import loadmat
    path = os.path.join(bfm_folder, 'similarity_Lm3D_all.mat')
    if not os.path.isfile(path):
        raise FileNotFoundError(f'{path}: the Basel Face Model files of Deep3DFaceRecon are needed for the crop / camera step '
                                '(reference README, "checkpoints/BFM"); they cannot be redistributed with this package')
    return extract_5p(loadmat(path)['lm'])


def resize_n_crop_img(img, lm, t, s, target_size=1024., mask=None):
    from PIL import Image
    w0, h0 = img.size
    t = [float(np.ravel(t)[0]), float(np.ravel(t)[1])]
    w, h = (w0 * s).astype(np.int32), (h0 * s).astype(np.int32)
    left = (w / 2 - target_size / 2 + float((t[0] - w0 / 2) * s)).astype(np.int32)
    up = (h / 2 - target_size / 2 + float((h0 / 2 - t[1]) * s)).astype(np.int32)
    box = (left, up, left + target_size, up + target_size)
    img = img.resize((w, h), resample=Image.LANCZOS).crop(box)
    if mask is not None:
        mask = mask.resize((w, h), resample=Image.LANCZOS).crop(box)
    lm = np.stack([lm[:, 0] - t[0] + w0 / 2, lm[:, 1] - t[1] + h0 / 2], axis=1) * s
    lm = lm - np.reshape(np.array([(w / 2 - target_size / 2), (h / 2 - target_size / 2)]), [1, 2])
    return img, lm, mask


def align_img(img, lm, lm3D, mask=None, target_size=1024., rescale_factor=466.285):
    """img PIL, lm [68,2] (or [5,2]) with y pointing UP, lm3D [5,3] -> (trans_params [w0, h0, s, tx, ty], 224^2 image, landmarks at the 224
    scale, cropped mask, 1024^2 image)   (extract_3dmm.py:67-102)"""
    from PIL import Image
    w0, h0 = img.size
    lm5p = extract_5p(lm) if lm.shape[0] != 5 else lm
    t, s = POS(lm5p.transpose(), lm3D.transpose())
    s = rescale_factor / s
    img_new, lm_new, mask_new = resize_n_crop_img(img, lm, t, s, target_size=target_size, mask=mask)
    # (the reference builds np.array([w0, h0, s, t[0], t[1]]) with t[i] of shape (1,): a ragged array, an error since numpy 1.24; the five numbers are meant)
    trans_params = np.array([w0, h0, float(s), float(np.ravel(t)[0]), float(np.ravel(t)[1])])
    lm_new = lm_new * (224 / 1024.0)
    return trans_params, img_new.resize((224, 224), resample=Image.LANCZOS), lm_new, mask_new, img_new


def split_coeff(coeffs):
    return {'id': coeffs[:, :80], 'exp': coeffs[:, 80:144], 'tex': coeffs[:, 144:224], 'angle': coeffs[:, 224:227], 'gamma': coeffs[:, 227:254],
            'trans': coeffs[:, 254:]}


class Extract3dmm:
    def __init__(self, PRETRAINED_MODELS_PATH=None, device='cuda', state_dict=None, lm3d_std=None):
        """PRETRAINED_MODELS_PATH: {'BFM': folder, '3DMM': '.../epoch_20.pth'} like the reference (extract_3dmm.py:107-124); ``state_dict`` /
        ``lm3d_std`` replace the two files."""
        from ..third_part.Deep3DFaceRecon_pytorch.models.networks import ReconNetWrapper
        paths = PRETRAINED_MODELS_PATH or {}
        if state_dict is None:
            ckpt = paths.get('3DMM', '')
            if not os.path.isfile(ckpt):
                raise FileNotFoundError(f'{ckpt!r}: the Deep3DFaceRecon checkpoint (epoch_20.pth) is needed for the crop / camera step; '
                                        'it is a third-party trained model that is not part of this package')
            state_dict = torch.load(ckpt, map_location='cpu', weights_only=True)['net_recon']
        self.device = device
        self.model = ReconNetWrapper('resnet50', use_last_fc=False)
        self.model.load_state_dict(state_dict)
        self.model.to(device)
        self.lm3d_std = np.asarray(lm3d_std) if lm3d_std is not None else load_lm3d(paths.get('BFM', ''))

    def image_transform(self, images, lm):
        """PIL image + [68,2] landmarks (image coordinates, y down) -> (float image [3,224,224] in [0,1], landmarks at 224)   (:126-138)"""
        _, H = images.size
        lm = np.array(lm, dtype=np.float64).reshape(-1, 2)               # (a copy: the reference flips the caller's array in place)
        lm[:, -1] = H - 1 - lm[:, -1]
        _, im_pil, lm, _, _ = align_img(images, lm, self.lm3d_std, rescale_factor=466.285)
        img = torch.tensor(np.array(im_pil) / 255., dtype=torch.float32).permute(2, 0, 1)
        return img, torch.tensor(lm)

    def get_3dmm(self, images_pil, lms_np):
        """-> dict of CPU tensors id [B,80], exp [B,64], tex [B,80], angle [B,3], gamma [B,27], trans [B,3]   (:140-224; batches of 20)"""
        images = torch.stack([self.image_transform(img, lm)[0] for img, lm in zip(images_pil, lms_np)])
        out = []
        for i in range(0, images.shape[0], 20):
            out.append(self.model(images[i:i + 20].to(self.device)).cpu())
        return split_coeff(torch.cat(out))
