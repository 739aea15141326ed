"""Inputs of the inversion: the on-disk layout of the reference's ``PTIDataset`` and a synthetic stand-in.

``PTIDataset`` mirrors spi/data/images_dataset.py:102-198: ``crop/<name>/target.<ext>`` (RGB, resized to 512^2,
scaled to [-1,1]), ``c/<name>/target.npy`` (25 fp32), ``mask/<name>/target.pt`` (int64 parsing [1,1,512,512]),
``lm/<name>/target.npy`` (68x2 at 256 scale); ``dataset_block='i/N'`` keeps the reference's contiguous
blocks (block = total // N + 1).  ``SyntheticDataset`` produces the inputs of SURVEY.md 8d without any
file: seeded image, ``cal_canonical_c(yaw=0.4)``, box-shaped parsing mask, fixed landmark template.
"""
import glob
import os
import numpy as np
import torch

from ..utils.camera_utils import cal_canonical_c


def shard_block(paths, dataset_block):
    """The reference's static sharding (images_dataset.py:149-158)."""
    index, total = (int(v) for v in dataset_block.split('/'))
    block = len(paths) // total + 1
    return paths[(index - 1) * block: index * block]


class PTIDataset(torch.utils.data.Dataset):
    def __init__(self, source_root, c_root=None, w_root=None, mask_root=None, lm_root=None, target_name='target', mode='jpg',
                 dataset_block=None, output_root=None, select_range=None, filter_index=None):
        self.source_root, self.c_root, self.w_root = source_root, c_root, w_root
        self.mask_root, self.lm_root, self.mode, self.target_name = mask_root, lm_root, mode, target_name
        self.source_paths = sorted(glob.glob(f'{source_root}/*/'))
        if select_range is not None:
            self.source_paths = self.source_paths[:select_range]
        if output_root is not None:
            done = set(sorted(glob.glob(f'{output_root}/*.jpg')))
            self.source_paths = [p for p in self.source_paths if os.path.join(output_root, p.split('/')[-2] + '.jpg') not in done]
        if dataset_block is not None:
            self.source_paths = shard_block(self.source_paths, dataset_block)
        if filter_index is not None:
            self.source_paths = [os.path.join(source_root, f'{ff}/') for ff in filter_index]

    def __len__(self):
        return len(self.source_paths)

    def __getitem__(self, index):
        from PIL import Image
        path = self.source_paths[index]
        name = os.path.dirname(path).split('/')[-1]
        img = Image.open(os.path.join(path, f'{self.target_name}.{self.mode}')).convert('RGB').resize((512, 512))
        img = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255.0)
        img = (img - 0.5) / 0.5
        data = {'img': img, 'fname': self.target_name, 'name': name,
                'c': np.load(os.path.join(self.c_root, name, self.target_name + '.npy')).astype(np.float32)}
        if self.w_root is not None:
            data['w'] = torch.load(os.path.join(self.w_root, name, self.target_name + '.pt'))
        if self.mask_root is not None:
            data['mask'] = torch.load(os.path.join(self.mask_root, name, self.target_name + '.pt'))
        if self.lm_root is not None:
            data['lm'] = torch.from_numpy(np.load(os.path.join(self.lm_root, name, self.target_name + '.npy'))).float()
        return data


def synthetic_landmarks():
    """Fixed 68x2 template inside [60,196] at 256 scale: jaw arc, brows, nose, two eyes, mouth ring."""
    t = np.linspace(0, 1, 17)
    jaw = np.stack([60 + 136 * t, 110 + 80 * np.sin(np.pi * t)], 1)
    brow_l = np.stack([np.linspace(78, 112, 5), np.full(5, 92.0)], 1)
    brow_r = np.stack([np.linspace(144, 178, 5), np.full(5, 92.0)], 1)
    nose = np.stack([np.r_[np.full(4, 128.0), np.linspace(116, 140, 5)], np.r_[np.linspace(104, 134, 4), np.full(5, 142.0)]], 1)
    ang = np.linspace(0, 2 * np.pi, 6, endpoint=False)
    eye_l = np.stack([95 + 12 * np.cos(ang), 108 + 6 * np.sin(ang)], 1)
    eye_r = np.stack([161 + 12 * np.cos(ang), 108 + 6 * np.sin(ang)], 1)
    ang = np.linspace(0, 2 * np.pi, 20, endpoint=False)
    mouth = np.stack([128 + 24 * np.cos(ang), 168 + 10 * np.sin(ang)], 1)
    lm = np.concatenate([jaw, brow_l, brow_r, nose, eye_l, eye_r, mouth], 0).astype(np.float32)
    assert lm.shape == (68, 2)
    return torch.from_numpy(lm)


def synthetic_parsing_mask():
    m = torch.zeros(1, 1, 512, 512, dtype=torch.int64)
    m[:, :, 86:426, 106:406] = 17            # "hair" ring around ...
    m[:, :, 106:406, 126:386] = 1             # ... a 300 x 260 "skin" box
    return m


class SyntheticDataset(torch.utils.data.Dataset):
    def __init__(self, n_images=1, seed=1, yaw=0.4):
        self.n, self.seed, self.yaw = n_images, seed, yaw

    def __len__(self):
        return self.n

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(self.seed + index)
        img = torch.rand(3, 512, 512, generator=g) * 2 - 1
        return {'img': img, 'c': cal_canonical_c(self.yaw, 0.0)[0].numpy(), 'fname': 'target', 'name': f'synthetic_{index:05d}',
                'mask': synthetic_parsing_mask()[0], 'lm': synthetic_landmarks()}
