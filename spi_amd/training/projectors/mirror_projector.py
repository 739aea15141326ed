"""SPI stage 1: W+ projection with the mirrored pseudo-view ("mir").

Same surface as spi/training/projectors/mirror_projector.py:12-140 ``project(G, target, c, lpips_func, fg_mask, *,
initial_w, num_steps, w_avg_samples, ..., device, w_name) -> w_opt [1,14,512]``: each step renders the
view and its mirror camera as one batch of 2 and minimises
``LPIPS(view, target) + LPIPS(mirror view, flip(target)) * weight_m + 1e5 * noise_reg``.
"""
import torch

from ...utils.camera_utils import cal_mirror_c, cal_camera_weight
from .common import run_projection, Projection


def mirror_setup(target, c, lpips_func, device):
    """-> (cameras [2,25], dist_fn) of the view + mirrored-view objective (:66-72,99-104)."""
    target = target.to(device).float()
    target_m = torch.flip(target, dims=[3])
    camera_m = cal_mirror_c(camera=c)
    cameras = torch.cat([c, camera_m], dim=0).to(device)
    weight_m = cal_camera_weight(camera_m)[0]
    feats, feats_m = (lpips_func.features(target), lpips_func.features(target_m)) if hasattr(lpips_func, 'features') else (None, None)
    both = sw = None
    from ...criteria.lpips.lpips import LPIPS
    if feats is not None and isinstance(lpips_func, LPIPS):
        both = [torch.cat([a, b], dim=0) for a, b in zip(feats, feats_m)]         # the two fixed targets as one batch
        sw = torch.stack([torch.ones((), device=device), torch.as_tensor(weight_m, device=device, dtype=torch.float32).reshape(())])

    def dist_fn(images):
        if both is not None:                                 # same value: lpips(view) * 1 + lpips(mirror view) * weight_m, one VGG pass
            return lpips_func(images, y_feats=both, sample_weights=sw)
        if feats is not None:
            return lpips_func(images[:1], y_feats=feats) + lpips_func(images[1:], y_feats=feats_m) * weight_m
        return lpips_func(images[:1], target) + lpips_func(images[1:], target_m) * weight_m
    if both is not None:
        dist_fn.state = list(both) + [sw]          # every per-image tensor the objective reads: lets the projector be re-used for the next image
    return cameras, dist_fn


def project(G, target, c, lpips_func, fg_mask=None, *, initial_w=None, num_steps=1000, w_avg_samples=10000,
            initial_learning_rate=0.01, initial_noise_factor=0.05, lr_rampdown_length=0.25, lr_rampup_length=0.05,
            noise_ramp_length=0.75, regularize_noise_weight=1e5, verbose=False, device, image_log_step=500, w_name='', rng=None,
            log=None):
    assert target.shape[1:] == (G.img_channels, G.img_resolution, G.img_resolution)
    cameras, dist_fn = mirror_setup(target, c, lpips_func, device)

    sched = dict(initial_learning_rate=initial_learning_rate, initial_noise_factor=initial_noise_factor,
                 lr_rampdown_length=lr_rampdown_length, lr_rampup_length=lr_rampup_length, noise_ramp_length=noise_ramp_length)
    return run_projection(G, cameras, dist_fn, w_mode='w+', initial_w=initial_w, num_steps=num_steps, w_avg_samples=w_avg_samples,
                          device=device, rng=rng, log=log, regularize_noise_weight=regularize_noise_weight, schedule_kwargs=sched)
