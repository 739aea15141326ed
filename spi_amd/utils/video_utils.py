"""Novel-view orbit and shape export after the inversion (drop-in surface of spi/utils/video_utils.py:14-230).

``gen_interp_video(G, {'w': w}, mp4=path)`` is what ``BaseCoach.log_video`` calls (base_coach.py:236-237): 120 frames on a
yaw +-0.7 / pitch +-0.4 orbit around the look-at point (0, 0, 0.2), radius 2.7, FFHQ intrinsics.  The reference evaluates
``G.synthesis`` once per frame, i.e. runs the StyleGAN2 backbone 120 times on the same ``w``; here the frames go through the
generator in batches with ONE ``w`` (``TriPlaneGenerator.synthesis`` shares the backbone and the weight modulation across the
views), under ``no_grad``.  Several latents (``G_kwargs['w']`` of shape [K * grid_w * grid_h, 14, 512]) are keyframes: every grid cell walks
its ``num_keyframes`` latents with the reference's wrapped cubic interpolation (scipy ``interp1d`` over the keyframes tiled ``2 * wraps + 1``
times, :118-134) while the camera completes ONE orbit over the ``num_keyframes * w_frames`` frames (:155-160), and the cells are laid out
by ``layout_grid`` (:31-44).
Output: frames are written as ``<mp4 stem>_frames/%04d.jpg`` (PIL).  The ``.mp4`` container itself needs ``imageio`` + ffmpeg/libx264
(the reference's writer, video_utils.py:141), which are NOT dependencies of this package: with imageio importable the file is encoded,
otherwise the run prints where the frames are and the ffmpeg command that encodes them (stated dependency, no silent skip).  ``gen_shapes`` exports the density grid of frame 0 like video_utils.py:198-218: the
iso-surface at level 10 as ``interpolation_shape/0000_shape.ply`` (``output_ply=True`` there; ``shape_format='mrc'`` for the
other branch), through utils/shape_utils.py, plus the raw grid as ``.npy`` and the camera path as ``_trajectory.npy``.
"""
import math
import os

import numpy as np
import torch

from . import camera_utils as cu

_warned_no_imageio = False


def orbit_cameras(num_frames, yaw_range=0.7, pitch_range=0.4, lookat=(0.0, 0.0, 0.2), radius=2.7, device='cpu'):
    """[F,25] cameras of the reference's orbit (video_utils.py:155-160).  It writes 3.14, not pi: kept, the path is part of
    what a user sees."""
    t = np.arange(num_frames, dtype=np.float64)                                   # the reference evaluates the angles in float64 (np.sin)
    h = torch.tensor(3.14 / 2 + yaw_range * np.sin(2 * 3.14 * t / num_frames), dtype=torch.float32, device=device).view(-1, 1)
    v = torch.tensor(3.14 / 2 - 0.05 + pitch_range * np.cos(2 * 3.14 * t / num_frames), dtype=torch.float32, device=device).view(-1, 1)
    ext = cu.look_at_pose(h, v, torch.tensor(lookat, device=device), radius)
    return torch.cat([ext.reshape(-1, 16), cu._intrinsics(num_frames, device)], dim=1)


def create_samples(N=256, voxel_origin=(0, 0, 0), cube_length=2.0):
    """N^3 query points, x fastest (video_utils.py:41-70).  -> ([1, N^3, 3], origin, voxel size).
    The reference divides the running index as a FLOAT (`(overall_index.float() / N) % N`, :57-58), so the second and third
    coordinates are fractional grid positions (each row / slab is sheared by up to one voxel); kept, since the exported
    density grid is defined by these points (pinned by tests/golden/orbit.npz)."""
    origin = np.array(voxel_origin) - cube_length / 2
    voxel = cube_length / (N - 1)
    idx = torch.arange(0, N ** 3, 1, dtype=torch.int64)
    s = torch.zeros(N ** 3, 3)
    s[:, 2] = idx % N
    s[:, 1] = (idx.float() / N) % N
    s[:, 0] = ((idx.float() / N) / N) % N
    s[:, 0] = s[:, 0] * voxel + origin[2]
    s[:, 1] = s[:, 1] * voxel + origin[1]
    s[:, 2] = s[:, 2] * voxel + origin[0]
    return s.unsqueeze(0), origin, voxel


def to_uint8(img, image_mode='image'):
    """[N,C,H,W] in [-1,1] -> uint8 [N,H,W,3] like layout_grid (video_utils.py:26-38)."""
    if image_mode == 'image_depth':
        img = -img
        lo, hi = img.amin(dim=(1, 2, 3), keepdim=True), img.amax(dim=(1, 2, 3), keepdim=True)
        img = (img - lo) / (hi - lo) * 2 - 1
    if img.shape[1] == 1:
        img = img.repeat(1, 3, 1, 1)
    return (img * 127.5 + 128).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1)


@torch.no_grad()
def sigma_grid(G, w, resolution=128, max_batch=1 << 22):
    """Density on a resolution^3 grid over the tri-plane box through ``G.sample_mixed`` (video_utils.py:177-196), with the
    reference's border clean-up (:198-207).  -> float32 [R,R,R] numpy."""
    dev = w.device
    samples, _, _ = create_samples(N=resolution, voxel_origin=[0, 0, 0], cube_length=G.rendering_kwargs['box_warp'])
    samples = samples.to(dev)
    sig = torch.zeros(1, samples.shape[1], 1, device=dev)
    dirs = torch.zeros(1, min(max_batch, samples.shape[1]), 3, device=dev)
    dirs[..., -1] = -1
    head = 0
    while head < samples.shape[1]:
        chunk = samples[:, head:head + max_batch]
        sig[:, head:head + max_batch] = G.sample_mixed(chunk, dirs[:, :chunk.shape[1]], w, noise_mode='const')['sigma']
        head += max_batch
    s = np.flip(sig.reshape(resolution, resolution, resolution).cpu().numpy(), 0).copy()
    pad, pad_top = int(30 * resolution / 256), int(38 * resolution / 256)
    if pad:
        s[:pad] = 0; s[-pad:] = 0; s[:, :pad] = 0; s[:, :, :pad] = 0; s[:, :, -pad:] = 0
    if pad_top:
        s[:, -pad_top:] = 0
    return s


def layout_grid(img, grid_w=None, grid_h=1):
    """uint8 [B,H,W,3] cells -> one [grid_h * H, grid_w * W, 3] frame, row-major cells (video_utils.py:31-44)."""
    b, h, w, ch = img.shape
    grid_w = b // grid_h if grid_w is None else grid_w
    assert b == grid_w * grid_h
    return img.reshape(grid_h, grid_w, h, w, ch).permute(0, 2, 1, 3, 4).reshape(grid_h * h, grid_w * w, ch)


def keyframe_interpolators(ws, grid_dims=(1, 1), num_keyframes=None, wraps=2, kind='cubic'):
    """ws [grid_h * grid_w * K, 14, 512] -> (K, [[f(frame_idx / w_frames) -> [14,512] float64 numpy] per cell]) exactly as video_utils.py:101-134
    builds them: cell (yi, xi) owns K consecutive latents, tiled 2 * wraps + 1 times over x = -K * wraps .. K * (wraps + 1) - 1, so the
    interpolant is periodic over the video and cubic through every keyframe."""
    import scipy.interpolate
    grid_w, grid_h = grid_dims
    if num_keyframes is None:
        if len(ws) % (grid_w * grid_h) != 0:
            raise ValueError('Number of input seeds must be divisible by grid W*H')
        num_keyframes = len(ws) // (grid_w * grid_h)
    wk = ws.detach().cpu().numpy().reshape(grid_h, grid_w, num_keyframes, *ws.shape[1:])
    x = np.arange(-num_keyframes * wraps, num_keyframes * (wraps + 1))
    return num_keyframes, [[scipy.interpolate.interp1d(x, np.tile(wk[yi][xi], [wraps * 2 + 1, 1, 1]), kind=kind, axis=0) for xi in range(grid_w)]
                           for yi in range(grid_h)]


@torch.no_grad()
def gen_interp_video(G, G_kwargs, mp4, w_frames=30 * 4, image_mode='image', gen_shapes=False, batch=4, device=None,
                     voxel_resolution=128, save_frames=True, shape_format='ply', shape_level=10, render_noise=None, return_float=False,
                     grid_dims=(1, 1), num_keyframes=None, wraps=2, kind='cubic', **_unused):
    """Orbit video.  Returns the frames as uint8 [F, grid_h * H, grid_w * W, 3] (numpy); F = num_keyframes * w_frames.
    One latent (what BaseCoach.log_video passes): 120 frames of that latent.  Several: keyframe interpolation, see the module docstring.
    render_noise (extension, tests): callable frame index -> (xi [1,M,Sc,1], u [M,Sf]) or None -- the renderer's two draws of that frame
    (the reference draws them with torch.rand inside G.synthesis, once per frame); return_float: also the float frames before uint8."""
    w = G_kwargs['w']
    if w.ndim == 2:
        w = w.unsqueeze(0)
    device = device or w.device
    grid_w, grid_h = grid_dims
    cells = grid_w * grid_h
    single = w.shape[0] == 1 and cells == 1
    if single:
        num_keyframes, interp = 1, None
    else:
        num_keyframes, interp = keyframe_interpolators(w, grid_dims, num_keyframes, wraps, kind)
    total = num_keyframes * w_frames
    cams = orbit_cameras(total, device=device)                   # ONE orbit over the whole video (:155-160 divide by num_keyframes * w_frames)
    frames, floats = [], []
    m, rk = G.neural_rendering_resolution ** 2, G.rendering_kwargs
    if not single and (render_noise is not None or return_float):
        raise NotImplementedError('render_noise / return_float are test hooks of the single-latent path')
    for i in range(0, total, batch):
        c = cams[i:i + batch]
        if not single:
            # frame f, cell (yi, xi): latent interp[yi][xi](f / w_frames), camera of frame f; cells of a frame are consecutive (row-major), frames of a
            # batch follow each other -- distinct latents, so the generator runs its per-sample path
            wb = np.stack([interp[yi][xi]((i + f) / w_frames) for f in range(len(c)) for yi in range(grid_h) for xi in range(grid_w)])
            out = G.synthesis(torch.from_numpy(wb).float().to(device), c.repeat_interleave(cells, dim=0), noise_mode='const')[image_mode]
            u8 = to_uint8(out, image_mode).cpu()
            frames.append(torch.stack([layout_grid(u8[f * cells:(f + 1) * cells], grid_w, grid_h) for f in range(len(c))]))
            continue
        noise = None
        if render_noise is not None:
            per = [render_noise(k) for k in range(i, i + len(c))]
            per = [p if p is not None else (torch.rand(1, m, int(rk['depth_resolution']), 1), torch.rand(m, max(int(rk['depth_resolution_importance']), 1))) for p in per]
            noise = (torch.cat([p[0].reshape(1, m, -1, 1) for p in per]).to(device), torch.cat([p[1].reshape(m, -1) for p in per]).to(device))
        out = G.synthesis(w.to(device), c, noise_mode='const', render_noise=noise)[image_mode]      # one w, len(c) cameras: backbone shared
        frames.append(to_uint8(out, image_mode).cpu())
        if return_float:
            floats.append(out.float().cpu())
    frames = torch.cat(frames).numpy()
    stem = os.path.splitext(mp4)[0]
    if save_frames:
        from PIL import Image
        os.makedirs(stem + '_frames', exist_ok=True)
        for i, f in enumerate(frames):
            Image.fromarray(f).save(os.path.join(stem + '_frames', f'{i:04d}.jpg'))
        # The .mp4 container needs imageio + an ffmpeg build with libx264 (what the reference uses, video_utils.py:141); neither is a
        # dependency of this package.  With imageio installed the same file is written; without it the frames above ARE the output, and
        # the run says so once instead of silently dropping the video.
        try:
            import imageio
        except ImportError:
            imageio = None
            global _warned_no_imageio
            if not _warned_no_imageio:
                _warned_no_imageio = True
                print(f'[spi_amd] imageio is not installed: {os.path.basename(mp4)} is not encoded, its {len(frames)} frames are in {stem}_frames/ '
                      '(ffmpeg -framerate 60 -i %04d.jpg -c:v libx264 makes the same file)', flush=True)
        if imageio is not None:
            with imageio.get_writer(mp4, mode='I', fps=60, codec='libx264') as vw:
                for f in frames:
                    vw.append_data(f)
    if gen_shapes:
        from . import shape_utils
        outdir = os.path.join(os.path.dirname(mp4) or '.', 'interpolation_shape')
        os.makedirs(outdir, exist_ok=True)
        sigmas = sigma_grid(G, w[:1].to(device), voxel_resolution)           # frame 0 = the first keyframe of cell (0, 0)
        np.save(os.path.join(outdir, '0000_sigma.npy'), sigmas)
        if shape_format == 'ply':                                                    # video_utils.py:209-214
            shape_utils.convert_sdf_samples_to_ply(np.transpose(sigmas, (2, 1, 0)), [0, 0, 0], 1, os.path.join(outdir, '0000_shape.ply'), level=shape_level)
        else:                                                                        # :215-217
            shape_utils.write_mrc(os.path.join(outdir, '0000_shape.mrc'), sigmas)
        np.save(stem + '_trajectory.npy', cams[:, :16].reshape(-1, 4, 4).cpu().numpy())
    return (frames, torch.cat(floats)) if return_float else frames
