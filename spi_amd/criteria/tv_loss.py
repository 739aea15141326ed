"""Density total-variation regulariser (mirror of spi/criteria/tv_loss.py:9-19): L1 between the density at
1000 random points and at copies perturbed by N(0, 0.004^2), evaluated with ``G.sample_mixed`` (the fused
gather+decode kernel at explicit coordinates).  lambda = 0 in the README commands, so this is off by default."""
import torch

density_reg_p_dist = 0.004
box_warp = 1


def cal_tv_loss(ws, G, rng=None):
    """The backbone pass inside ``sample_mixed`` runs with the layers' default ``noise_mode='random'`` exactly as in the reference
    (tv_loss.py:14 passes no noise_mode).  ``rng`` (extension): draw source for parity tests; it is asked for the reference's draws
    in the reference's order -- coordinates, perturbation, the (unused) directions, then one noise map per synthesis layer."""
    dev = ws.device
    init = (torch.rand((ws.shape[0], 1000, 3), device=dev) * 2 - 1) if rng is None else rng.rand(ws.shape[0], 1000, 3) * 2 - 1
    pert = init + (torch.randn_like(init) if rng is None else rng.randn(*init.shape)) * density_reg_p_dist
    coords = torch.cat([init, pert], dim=1)
    kw = {}
    if rng is not None:
        rng.randn(*coords.shape)                       # the decoder ignores directions (triplane.py:124); the draw keeps the stream aligned
        kw['noise_rng'] = rng
    sigma = G.sample_mixed(coords, None, ws, update_emas=False, **kw)['sigma']
    half = sigma.shape[1] // 2
    return torch.nn.functional.l1_loss(sigma[:, :half], sigma[:, half:])
