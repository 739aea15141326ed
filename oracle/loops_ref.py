"""Oracle (CPU, plain torch) for the two SPI optimisation loops.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Also the `cpu_baseline` leg of bench.py.

Reference lines followed (relative to /root/reference/spi):
  w statistics / schedules / stage-1 loop   training/projectors/mirror_projector.py:34-140
  W+ projector                              training/projectors/w_plus_projector.py:10-113
  W projector ("sg")                        training/projectors/w_projector.py:9-113
  stage-2 SPI loop                          training/coaches/rot_bbox_cx_coach.py:52-157
  PTI loop                                  training/coaches/pti_coach.py:62-82
  optimiser / seeds                         training/coaches/base_coach.py:28-33,53-60,133-135
Random draws go through a `Draws` object in the reference's order so a GPU run can replay them.
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

from . import renderer_ref as rr
from . import losses_ref as lo
from . import stylegan_ref as sg
from . import camera_ref as cu                     # independent restatement, pinned against the reference's outputs (golden/geometry.npz)


class Draws:
    """CPU draw source; records every draw (shape-ordered) for replay on the device path."""
    def __init__(self):
        self.log = []

    def rand(self, *shape):
        t = torch.rand(*shape)
        self.log.append(t.clone())
        return t

    def randn(self, *shape):
        t = torch.randn(*shape)
        self.log.append(t.clone())
        return t


def noise_keys(P):
    return [k for k in P if k.startswith('backbone.synthesis.') and k.endswith('noise_const')]


def w_stats(P, c, w_avg_samples=600, z_dim=512):
    z = np.random.RandomState(123).randn(w_avg_samples, z_dim)
    w = sg.mapping(P, torch.from_numpy(z), c.repeat(w_avg_samples, 1))[:, :1, :].numpy().astype(np.float32)
    w_avg = np.mean(w, axis=0, keepdims=True)
    w_std = (np.sum((w - w_avg) ** 2) / w_avg_samples) ** 0.5
    return w_avg, float(w_std)


def stage1_schedule(step, num_steps, w_std, lr0=0.01, noise0=0.05, rampdown=0.25, rampup=0.05, noise_ramp=0.75):
    t = step / num_steps
    w_noise_scale = w_std * noise0 * max(0.0, 1.0 - t / noise_ramp) ** 2
    ramp = min(1.0, (1.0 - t) / rampdown)
    ramp = 0.5 - 0.5 * np.cos(ramp * np.pi)
    ramp = ramp * min(1.0, t / rampup)
    return lr0 * ramp, w_noise_scale


def _synth(P, ws, c, opts, nrr, draws, noise_mode='const', **kw):
    n = c.shape[0]
    m = nrr * nrr
    xi = draws.rand(n, m, opts['depth_resolution'], 1)
    u = draws.rand(n * m, opts['depth_resolution_importance'])
    return rr.synthesis(P, ws, c, opts, neural_rendering_resolution=nrr, noise_mode=noise_mode, xi=xi, u=u, **kw)


def project_w_plus(P, target, c, lpips_fn, opts, *, mirror, num_steps, w_avg_samples=600, nrr=128, draws=None,
                   first_inv_lr=5e-3, log=None):
    """mirror=True -> mirror_projector.project; False -> w_plus_projector.project.  Returns w_opt [1,14,512]."""
    draws = draws or Draws()
    P = {k: v.detach().clone().float() for k, v in P.items()}
    w_avg, w_std = w_stats(P, c, w_avg_samples)
    nk = noise_keys(P)
    w_opt = torch.tensor(np.repeat(w_avg, 14, axis=1), dtype=torch.float32, requires_grad=True)
    for k in nk:
        P[k] = draws.randn(*P[k].shape).requires_grad_(True)
    opt = torch.optim.Adam([w_opt] + [P[k] for k in nk], betas=(0.9, 0.999), lr=first_inv_lr)
    if mirror:
        target_m = torch.flip(target, dims=[3])
        cam = torch.cat([c, cu.cal_mirror_c(c)], dim=0)
        weight_m = cu.cal_camera_weight(cam[1:])[0]
    else:
        cam = c
    for step in range(num_steps):
        lr, w_noise_scale = stage1_schedule(step, num_steps, w_std)
        for gp in opt.param_groups:
            gp['lr'] = lr
        ws = (w_opt + draws.randn(*w_opt.shape) * w_noise_scale).repeat(cam.shape[0], 1, 1)
        img = _synth(P, ws, cam, opts, nrr, draws)['image']
        if mirror:
            dist = lpips_fn(img[:1], target) + lpips_fn(img[1:], target_m) * weight_m
        else:
            dist = lpips_fn(img, target)
        reg = lo.noise_regulariser([P[k] for k in nk])
        loss = dist + reg * 1e5
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        with torch.no_grad():
            for k in nk:
                P[k] -= P[k].mean()
                P[k] *= P[k].square().mean().rsqrt()
        if log is not None:
            log.append(dict(dist=float(dist.detach()), reg=float(reg.detach()), loss=float(loss.detach()), w=w_opt.detach().clone(),
                            grad_w=w_opt.grad.detach().clone()))
    return w_opt


def project_w(P, target, c, vgg16_fn, opts, *, num_steps, w_avg_samples=600, nrr=128, draws=None, first_inv_lr=5e-3,
              log=None):
    """w_projector.project (`first_inv_type='sg'`): ONE w [1,1,512] broadcast to the 14 layers (:77), distance = squared
    difference of the feature vectors of an injected extractor on 256^2 area-downsampled 0..255 images (:48-51,81-87).
    Returns w_opt.repeat([1,14,1])."""
    draws = draws or Draws()
    P = {k: v.detach().clone().float() for k, v in P.items()}
    w_avg, w_std = w_stats(P, c, w_avg_samples)
    nk = noise_keys(P)

    def prep(img):
        img = (img + 1) * (255 / 2)
        if img.shape[2] > 256:
            img = F.interpolate(img, size=(256, 256), mode='area')
        return img
    target_features = vgg16_fn(prep(target), resize_images=False, return_lpips=True)
    w_opt = torch.tensor(w_avg, dtype=torch.float32, requires_grad=True)
    opt = torch.optim.Adam([w_opt] + [P[k] for k in nk], betas=(0.9, 0.999), lr=first_inv_lr)
    for k in nk:
        P[k] = draws.randn(*P[k].shape).requires_grad_(True)
    opt.param_groups[0]['params'][1:] = [P[k] for k in nk]
    for step in range(num_steps):
        lr, w_noise_scale = stage1_schedule(step, num_steps, w_std)
        for gp in opt.param_groups:
            gp['lr'] = lr
        ws = (w_opt + draws.randn(*w_opt.shape) * w_noise_scale).repeat([1, 14, 1])
        img = _synth(P, ws, c, opts, nrr, draws)['image']
        dist = (target_features - vgg16_fn(prep(img), resize_images=False, return_lpips=True)).square().sum()
        reg = lo.noise_regulariser([P[k] for k in nk])
        loss = dist + reg * 1e5
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        with torch.no_grad():
            for k in nk:
                P[k] -= P[k].mean()
                P[k] *= P[k].square().mean().rsqrt()
        if log is not None:
            log.append(dict(dist=float(dist.detach()), reg=float(reg.detach()), loss=float(loss.detach()), w=w_opt.detach().clone(),
                            grad_w=w_opt.grad.detach().clone()))
    return w_opt.detach().repeat([1, 14, 1])


# ---- stage 2 -----------------------------------------------------------------------------------

HP = dict(pt_l2_lambda=1.0, pt_lpips_lambda=1.0, pt_rot_lambda=0.1, pt_mirror_rot_lambda=0.05, pt_depth_lambda=1.0,
          LPIPS_value_threshold=0.05, pti_learning_rate=3e-4)


def face_mask_from_parsing(parsing):
    m = torch.zeros_like(parsing)
    for a in (1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 13):
        m += (parsing == a)
    return m


class Stage2State:
    """G (trainable copy), original_G (frozen copy), Adam over every G parameter (buffers excluded)."""
    def __init__(self, P, param_names, lr=3e-4):
        self.P = {k: v.detach().clone().float() for k, v in P.items()}
        self.P0 = {k: v.detach().clone().float() for k, v in P.items()}
        self.param_names = list(param_names)
        for k in self.param_names:
            self.P[k].requires_grad_(True)
        self.opt = torch.optim.Adam([self.P[k] for k in self.param_names], lr=lr)


def stage2_iteration(st, i, data, w_pivot, opts, lpips_fn, boxcx_fn, hp=HP, nrr=128, draws=None, rot_bs=4,
                     adapt_yaw_range=0.2, pti_only=False):
    """One iteration of rot_bbox_cx_coach.py:68-157 (pti_only -> pti_coach.py:62-82).  Returns a dict of
    loss values; performs the optimiser step unless the early-stop condition fires."""
    draws = draws or Draws()
    image, camera = data['img'], data['c']
    st.opt.zero_grad()
    out = {}
    gen = _synth(st.P, w_pivot, camera, opts, nrr, draws)
    l2 = lo.l2_loss(gen['image'], image)
    lp = torch.squeeze(lpips_fn(gen['image'], image))
    (l2 * hp['pt_l2_lambda'] + lp * hp['pt_lpips_lambda']).backward()
    out.update(l2=float(l2), lpips=float(lp))
    if not pti_only and i % rot_bs == 0:
        face_mask = data['face_mask']
        depth_main = gen['image_depth'].detach()
        if hp['pt_rot_lambda'] > 0:
            cams = cu.sample_surrounding_camera(camera, rot_bs, adapt_yaw_range, 0.1,
                                                rand=(draws.rand(rot_bs, 1), draws.rand(rot_bs, 1)))
            gs = _synth(st.P, w_pivot.repeat(rot_bs, 1, 1), cams, opts, nrr, draws)
            with torch.no_grad():
                warp, wmask = lo.rotate(cams, gs['image_depth'], image.repeat(rot_bs, 1, 1, 1), camera.repeat(rot_bs, 1),
                                        depth_main.repeat(rot_bs, 1, 1, 1), face_mask.repeat(rot_bs, 1, 1, 1), EPS=5e-2)
            l_rot = lpips_fn(gs['image'] * wmask, warp) * hp['pt_rot_lambda'] * rot_bs
            l_rot.backward()
            out['rot'] = float(l_rot)
        weight_m = cu.cal_camera_weight(camera)
        if hp['pt_mirror_rot_lambda'] > 0 and bool(weight_m > 0):
            camera_m = cu.cal_mirror_c(camera)
            cams_m = cu.sample_surrounding_camera(camera_m, rot_bs, adapt_yaw_range, 0.1,
                                                  rand=(draws.rand(rot_bs, 1), draws.rand(rot_bs, 1)))
            gm = _synth(st.P, w_pivot.repeat(rot_bs, 1, 1), cams_m, opts, nrr, draws)
            with torch.no_grad():
                warp_m, wmask_m = lo.rotate(cams_m, gm['image_depth'], torch.flip(image, [3]).repeat(rot_bs, 1, 1, 1),
                                            camera_m.repeat(rot_bs, 1), torch.flip(depth_main, [3]).repeat(rot_bs, 1, 1, 1),
                                            torch.flip(face_mask, [3]).repeat(rot_bs, 1, 1, 1), EPS=5e-2)
                fw, fm = torch.flip(warp_m, [3]), torch.flip(wmask_m, [3])
            l_m = boxcx_fn(torch.flip(gm['image'], [3]) * fm, fw, data['lm'].repeat(rot_bs, 1, 1))
            l_m = l_m * hp['pt_mirror_rot_lambda'] * rot_bs
            l_m.backward()
            out['mirror_rot'] = float(l_m)
        if hp['pt_depth_lambda'] > 0:
            cams_d = cu.sample_camera(4, 0.7, 0.4, rand=(draws.rand(4, 1), draws.rand(4, 1)))
            ws4 = w_pivot.repeat(4, 1, 1)
            d_new = _synth(st.P, ws4, cams_d, opts, nrr, draws)['image_depth']
            with torch.no_grad():
                d_old = _synth(st.P0, ws4, cams_d, opts, nrr, draws)['image_depth']
            l_d = lo.l2_loss(d_old, d_new) * hp['pt_depth_lambda']
            l_d.backward()
            out['depth'] = float(l_d)
    if float(lp) <= hp['LPIPS_value_threshold']:
        out['stopped'] = True
        return out
    st.opt.step()
    return out
