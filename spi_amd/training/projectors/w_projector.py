"""W projection ("sg", the PTI baseline's stage 1), spi/training/projectors/w_projector.py:9-113.

A single ``w [1,1,512]`` is broadcast to all 14 layers; the distance is the squared difference of the
feature vectors of an injected extractor ``vgg16(img_0..255, resize_images=False, return_lpips=True)``
on 256^2 area-downsampled images (:48-51,81-87).  NVIDIA's TorchScript ``vgg16.pt`` is not available
offline, so the extractor is a constructor argument ("parity unpinned" at that edge).
"""
import torch
import torch.nn.functional as F
from ...configs import global_config
from .common import run_projection


def sg_distance(target, vgg16, device):
    """-> dist_fn(images) of the W projector: squared difference of the extractor's feature vectors on 256^2 area-downsampled 0..255 images
    (:48-51,81-87); the fixed target's features are computed once."""
    def prep(img):
        img = (img + 1) * (255 / 2)
        if img.shape[2] > 256:
            img = F.interpolate(img, size=(256, 256), mode='area')
        return img

    with torch.no_grad():
        target_features = vgg16(prep(target.to(device).float()), resize_images=False, return_lpips=True)

    def dist_fn(images):
        return (target_features - vgg16(prep(images), resize_images=False, return_lpips=True)).square().sum()
    if torch.is_tensor(target_features):
        dist_fn.state = [target_features]          # the per-image tensor of the objective (run_projection re-uses the projector across images)
    return dist_fn


def project(G, target, camera, vgg16, *, num_steps=1000, w_avg_samples=10000, initial_learning_rate=0.01,
            initial_noise_factor=0.05, lr_rampdown_length=0.25, lr_rampup_length=0.05, noise_ramp_length=0.75,
            regularize_noise_weight=1e5, verbose=False, device, use_wandb=False, initial_w=None, image_log_step=global_config.log_snapshot,
            w_name='', rng=None, log=None):
    """Signature of w_projector.py:9-29 (``vgg16`` positional; ``use_wandb`` / ``image_log_step`` accepted, they only gate logging
    there) plus the two test hooks ``rng`` (draw source) and ``log`` (per-step losses)."""
    assert target.shape[1:] == (G.img_channels, G.img_resolution, G.img_resolution)

    dist_fn = sg_distance(target, vgg16, device)

    sched = dict(initial_learning_rate=initial_learning_rate, initial_noise_factor=initial_noise_factor,
                 lr_rampdown_length=lr_rampdown_length, lr_rampup_length=lr_rampup_length, noise_ramp_length=noise_ramp_length)
    w = run_projection(G, camera.to(device), dist_fn, w_mode='w', initial_w=initial_w, num_steps=num_steps,
                       w_avg_samples=w_avg_samples, device=device, rng=rng, log=log,
                       regularize_noise_weight=regularize_noise_weight, schedule_kwargs=sched)
    return w.repeat([1, G.backbone.mapping.num_ws, 1])
