"""Crop + EG3D camera label of a photo (mirror of preprocess/extract_camera.py:14-186).

``CameraExtractor.extract(path)``: 68 landmarks -> 3DMM coefficients (Extract3dmm: alignment + ResNet-50 on the MI355X conv kernels) ->
``cal_camera`` (head rotation / translation -> cam2world pose in EG3D's convention) -> ``process_camera`` (radius 2.7, normalised
intrinsics) -> ``<c_outdir>/<name>.npy`` (25 floats); the 512^2 crop (align at rescale 300, centre 700^2 window, LANCZOS) goes to
``<crop_outdir>/<name>.<mode>``.  ``cal_mirror_c`` writes the flipped crop and its mirrored camera.
The landmark detector is injected (extract_landmark.py); the regressor's trained weights and the BFM landmarks come from the files the
reference names (checkpoints/BFM, checkpoints/model_name/epoch_20.pth) or are handed in.
"""
import copy
import os

import numpy as np
import torch

from .extract_3dmm import Extract3dmm, align_img
from .extract_landmark import get_landmark
from .process_camera import process_camera


def compute_rotation(angles):
    """angles [B,3] (radians, x / y / z) -> rotation of the FIRST sample as `pts @ rot` (i.e. (R_z R_y R_x)^T), [3,3]   (:14-47)"""
    b = angles.shape[0]
    ones, zeros = torch.ones([b, 1]), torch.zeros([b, 1])
    x, y, z = angles[:, :1], angles[:, 1:2], angles[:, 2:]
    rot_x = torch.cat([ones, zeros, zeros, zeros, torch.cos(x), -torch.sin(x), zeros, torch.sin(x), torch.cos(x)], dim=1).reshape([b, 3, 3])
    rot_y = torch.cat([torch.cos(y), zeros, torch.sin(y), zeros, ones, zeros, -torch.sin(y), zeros, torch.cos(y)], dim=1).reshape([b, 3, 3])
    rot_z = torch.cat([torch.cos(z), -torch.sin(z), zeros, torch.sin(z), torch.cos(z), zeros, zeros, zeros, ones], dim=1).reshape([b, 3, 3])
    return (rot_z @ rot_y @ rot_x).permute(0, 2, 1)[0]


class CameraExtractor:
    def __init__(self, crop_outdir, c_outdir, mode, model_paths=None, landmark_fn=None, device='cuda', state_dict=None, lm3d_std=None):
        self.crop_outdir, self.c_outdir, self.mode = crop_outdir, c_outdir, mode
        self.landmark_fn = landmark_fn
        self.model_3dmm = Extract3dmm(model_paths or {'BFM': 'checkpoints/BFM/', '3DMM': 'checkpoints/model_name/epoch_20.pth'},   # (:57-60)
                                      device=device, state_dict=state_dict, lm3d_std=lm3d_std)
        self.lm3d_std = self.model_3dmm.lm3d_std

    def set_path(self, crop_outdir, c_outdir, mode):
        self.crop_outdir, self.c_outdir, self.mode = crop_outdir, c_outdir, mode

    def crop(self, img_pil, lm, image_name):
        """the 512^2 training crop (:72-85).  lm: y pointing up (Extract3dmm.image_transform has flipped it in place in the reference; here
        the flip is explicit in `_extract`)."""
        from PIL import Image
        rescale_factor, center_crop_size, output_size = 300, 700, 512
        _, _, _, _, im_high = align_img(img_pil, lm, self.lm3d_std, rescale_factor=rescale_factor)
        left = int(im_high.size[0] / 2 - center_crop_size / 2)
        upper = int(im_high.size[1] / 2 - center_crop_size / 2)
        im = im_high.crop((left, upper, left + center_crop_size, upper + center_crop_size)).resize((output_size, output_size), resample=Image.LANCZOS)
        im.save(os.path.join(self.crop_outdir, f'{image_name}.{self.mode}'), compress_level=0)
        return im

    def cal_camera(self, coeff_3dmm):
        """3DMM pose -> {'intrinsics' 3x3, 'pose' 4x4 cam2world, 'angle'}   (:87-138)"""
        angle = coeff_3dmm['angle']
        trans = coeff_3dmm['trans'][0].clone()
        R = compute_rotation(angle).numpy()
        trans[2] += -10
        c = -np.dot(R, trans.numpy())
        pose = np.eye(4)
        pose[:3, :3] = R
        c *= 0.27                      # factor to match EG3D's ("tripleganger") scale
        c[1] += 0.006                  # offsets to align with it
        c[2] += 0.161
        pose[:3, 3] = c
        focal, w, h = 2985.29, 1024, 1024          # = 1015 * 1024 / 224 * (300 / 466.285)
        K = np.eye(3)
        K[0][0] = K[1][1] = focal
        K[0][2], K[1][2] = w / 2.0, h / 2.0
        pose[:3, :3] = np.dot(pose[:3, :3], np.diag([1.0, -1.0, -1.0]))
        return {'intrinsics': K.tolist(), 'pose': pose.tolist(), 'angle': (angle * torch.tensor([1, -1, 1])).flatten().tolist()}

    def _extract(self, image_pil, image_name):
        lm_np = get_landmark(image_pil, self.landmark_fn)
        coeff_3dmm = self.model_3dmm.get_3dmm([image_pil], [lm_np])
        lm_up = np.array(lm_np, dtype=np.float64)
        lm_up[:, -1] = image_pil.size[1] - 1 - lm_up[:, -1]             # the crop sees the flipped landmarks (the reference's in-place side effect, extract_3dmm.py:134)
        self.crop(image_pil, lm_up, image_name)
        cam = self.cal_camera(coeff_3dmm)
        camera = process_camera(pose=cam['pose'], intrinsics=cam['intrinsics'])
        np.save(os.path.join(self.c_outdir, f'{image_name}.npy'), camera)
        return camera

    def extract(self, image_path):
        from PIL import Image
        image_pil = Image.open(image_path).convert('RGB')                # (png with alpha)
        return self._extract(image_pil, os.path.basename(image_path).split('.')[0])

    @staticmethod
    def flip_yaw(pose_matrix):
        flipped = copy.deepcopy(pose_matrix)
        for i, j in ((0, 1), (0, 2), (1, 0), (2, 0), (0, 3)):
            flipped[i, j] *= -1
        return flipped

    def _cal_mirror_c(self, c):
        pose, intrinsics = c[:16].reshape(4, 4), c[16:].reshape(3, 3)
        mirror_c = np.concatenate([self.flip_yaw(pose).reshape(-1), intrinsics.reshape(-1)])
        assert mirror_c.shape == c.shape
        return mirror_c

    def cal_mirror_c(self, image_path):
        from PIL import Image
        image_name = os.path.basename(image_path).split('.')[0]
        crop = Image.open(os.path.join(self.crop_outdir, f'{image_name}.{self.mode}')).convert('RGB')
        crop.transpose(Image.FLIP_LEFT_RIGHT).save(os.path.join(self.crop_outdir, f'{image_name}_m.{self.mode}'))
        camera = np.load(os.path.join(self.c_outdir, f'{image_name}.npy'))
        np.save(os.path.join(self.c_outdir, f'{image_name}_m.npy'), self._cal_mirror_c(camera))
