"""W+ projection of a single view ("sgw+"), spi/training/projectors/w_plus_projector.py:10-113."""
import torch
from .common import run_projection


def project(G, target, c, lpips_func, *, initial_w=None, num_steps=1000, w_avg_samples=10000, initial_learning_rate=0.01,
            initial_noise_factor=0.05, lr_rampdown_length=0.25, lr_rampup_length=0.05, noise_ramp_length=0.75,
            regularize_noise_weight=1e5, verbose=False, device, image_log_step=500, w_name='', rng=None, log=None):
    assert target.shape[1:] == (G.img_channels, G.img_resolution, G.img_resolution)
    target = target.to(device).float()
    feats = lpips_func.features(target) if hasattr(lpips_func, 'features') else None

    def dist_fn(images):
        return lpips_func(images, y_feats=feats) if feats is not None else lpips_func(images, target)
    if feats is not None:
        dist_fn.state = list(feats)                # the per-image tensors of the objective (run_projection re-uses the projector across images)

    sched = dict(initial_learning_rate=initial_learning_rate, initial_noise_factor=initial_noise_factor,
                 lr_rampdown_length=lr_rampdown_length, lr_rampup_length=lr_rampup_length, noise_ramp_length=noise_ramp_length)
    return run_projection(G, c.to(device), dist_fn, w_mode='w+', initial_w=initial_w, num_steps=num_steps,
                          w_avg_samples=w_avg_samples, device=device, rng=rng, log=log,
                          regularize_noise_weight=regularize_noise_weight, schedule_kwargs=sched)
