"""Deterministic, name-keyed synthetic weights shared by the golden generator and the tests.

No checkpoint exists in this environment, so every generator tensor is a seeded draw keyed by
its state_dict name.  The same function fills the reference module (make_golden.py), the
oracle's parameter dict and the product module, so fixtures only need to hold outputs.
"""
import hashlib
import json
import os
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _gen(name, seed):
    h = int.from_bytes(hashlib.sha256(f'{seed}:{name}'.encode()).digest()[:8], 'little') % (2 ** 62)
    return torch.Generator().manual_seed(h)


def synth_tensor(name, shape, seed=0):
    leaf = name.split('.')[-1]
    if leaf == 'resample_filter':
        f = torch.tensor([1.0, 3.0, 3.0, 1.0])
        f = torch.outer(f, f)
        return f / f.sum()
    t = torch.randn(tuple(shape), generator=_gen(name, seed))
    if leaf == 'bias':
        return 1.0 + 0.1 * t if '.affine.' in name else 0.1 * t
    if leaf == 'noise_strength':
        return 0.1 * t
    if leaf == 'w_avg':
        return 0.1 * t
    if '.mapping.fc' in name and leaf == 'weight':
        return 100.0 * t
    return t


def load_manifest(kind):
    with open(os.path.join(GOLDEN_DIR, f'manifest_{kind}.json')) as f:
        return {k: tuple(v) for k, v in json.load(f).items()}


def synth_state_dict(manifest, seed=0):
    return {k: synth_tensor(k, s, seed) for k, s in manifest.items()}
