"""GPU parity of the HIP renderer kernels (through the C ABI) against the oracle and the golden vectors."""
import json
import math
import pytest
import torch

from conftest import assert_close, rel_err
from oracle import renderer_ref as orr

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _decoder(P):
    from spi_amd.training.triplane import OSGDecoder
    dec = OSGDecoder(32, {'decoder_lr_mul': 1, 'decoder_output_dim': 32}).to(DEV)
    dec.load_state_dict({k[len('decoder.'):]: v for k, v in P.items()})
    return dec


def _P(g):
    return {k[2:]: g[k] for k in g.keys() if k.startswith('P_decoder.')}


def test_ray_sampler_golden(golden):
    from spi_amd.training.volumetric_rendering.ray_sampler import RaySampler
    g = golden('renderer')
    c = g['cam'].to(DEV)
    ro, rd = RaySampler()(c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3), 8)
    assert_close(ro, g['ray_o'], 1e-6, 'ray origins')
    assert_close(rd, g['ray_d'], 1e-6, 'ray dirs')


def test_ray_sampler_full_res():
    from spi_amd.training.volumetric_rendering.ray_sampler import RaySampler
    from spi_amd.utils import camera_utils as cu
    c = torch.cat([cu.cal_canonical_c(0.4, 0.0), cu.cal_canonical_c(-0.2, 0.1)], 0)
    ro, rd = RaySampler()(c[:, :16].view(-1, 4, 4).to(DEV), c[:, 16:25].view(-1, 3, 3).to(DEV), 128)
    oo, od = orr.ray_sampler(c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3), 128)
    assert_close(ro, oo, 1e-6, 'ray origins 128')
    assert_close(rd, od, 2e-6, 'ray dirs 128')


def test_raymarch_golden_fwd_bwd(golden):
    from spi_amd.training.volumetric_rendering.ray_marcher import MipRayMarcher2
    g = golden('renderer')
    for wb in (0, 1):
        t = f'rm{wb}_'
        col = g['rm_col'].to(DEV).requires_grad_(True)
        den = g['rm_den'].to(DEV).requires_grad_(True)
        dep = g['rm_dep'].to(DEV)
        rgb, depth, w = MipRayMarcher2()(col, den, dep, {'clamp_mode': 'softplus', 'white_back': bool(wb)})
        assert_close(rgb, g[t + 'rgb'], 2e-6, 'march rgb')
        assert_close(depth, g[t + 'depth'], 2e-6, 'march depth')
        assert_close(w, g[t + 'w'], 2e-6, 'march weights')
        gc, gd = torch.autograd.grad([rgb, depth, w], [col, den], [g[t + 'drgb'].to(DEV), g[t + 'ddepth'].to(DEV), g[t + 'dw'].to(DEV)])
        ref_c, ref_d = g[t + 'gcol'], g[t + 'gden']
        ok = ~torch.isnan(ref_d).flatten(2).any(2)        # rays where the reference's own backward is finite
        assert ok.sum() == ok.numel() - 1                  # exactly the one empty ray of the fixture
        assert_close(gc.cpu()[ok], ref_c[ok], 1e-5, 'march grad colors')
        assert_close(gd.cpu()[ok], ref_d[ok], 1e-5, 'march grad densities')
        assert torch.isfinite(gc).all() and torch.isfinite(gd).all()     # documented deviation: 0 instead of NaN


@pytest.mark.parametrize('S', [2, 3, 64, 65, 96, 192, 256])
def test_raymarch_sizes_vs_oracle(S):
    from spi_amd.training.volumetric_rendering.ray_marcher import MipRayMarcher2
    gen = torch.Generator().manual_seed(S)
    n, m = 1, 37
    col = torch.rand(n, m, S, 32, generator=gen)
    den = torch.randn(n, m, S, 1, generator=gen) * 3 + 1
    dep = torch.sort(torch.rand(n, m, S, 1, generator=gen) * 1.05 + 2.25, dim=2)[0]
    d1, d2 = torch.randn(n, m, 32, generator=gen), torch.randn(n, m, 1, generator=gen)
    cr, dr = col.clone().requires_grad_(True), den.clone().requires_grad_(True)
    a, b, c = orr.ray_march(cr, dr, dep)
    gca, gda = torch.autograd.grad([a, b], [cr, dr], [d1, d2])
    cg, dg = col.to(DEV).requires_grad_(True), den.to(DEV).requires_grad_(True)
    x, y, z = MipRayMarcher2()(cg, dg, dep.to(DEV), {'clamp_mode': 'softplus'})
    gcb, gdb = torch.autograd.grad([x, y], [cg, dg], [d1.to(DEV), d2.to(DEV)])
    assert_close(x, a, 3e-6, 'rgb'); assert_close(y, b, 3e-6, 'depth'); assert_close(z, c, 3e-6, 'weights')
    assert_close(gcb, gca, 2e-5, 'grad colors'); assert_close(gdb, gda, 2e-5, 'grad densities')


@pytest.mark.parametrize('R,S,depth_only', [(1003, 192, False), (1003, 192, True), (61, 65, False), (5, 7, False)])
def test_raymarch_bwd_flags_and_live_rays_through_the_abi(R, S, depth_only):
    """One wave of the backward owns 8 consecutive rays and walks the live ones: flags, untouched rows of dead rays and the live rays'
    gradients are those of a launch without flags (every ray marched), for ragged ray counts, dead / live runs and lone live rays."""
    from spi_amd import hip
    gen = torch.Generator().manual_seed(R * 1000 + S)
    sc = S // 2
    col = torch.rand(R, S, 32, generator=gen).to(DEV); den = (torch.randn(R, S, generator=gen) * 3 + 1).to(DEV)
    dc = torch.sort(torch.rand(R, sc, generator=gen) + 2.25, 1)[0].to(DEV).contiguous()
    df = torch.sort(torch.rand(R, S - sc, generator=gen) + 2.25, 1)[0].to(DEV).contiguous()
    dep = torch.empty(R, S, device=DEV); perm = torch.empty(R, S, device=DEV, dtype=torch.int32)
    hip.call('spi_merge_sort_depths', hip.ptr(dc), hip.ptr(df), R, sc, S - sc, hip.ptr(dep), hip.ptr(perm), hip.stream())
    cl = torch.tensor([2.25, 3.3], device=DEV)
    live = torch.rand(R, generator=gen) < 0.3
    live[R // 3: R // 3 + 40] = True                         # a run of live rays across several waves
    live[R // 2: R // 2 + 100] = False                       # ... and of dead ones
    live[-1] = True                                          # the last ray of the ragged tail
    d_rgb = torch.randn(R, 32, generator=gen) * live[:, None]
    d_rgb[live.nonzero()[0, 0]] = 0.; d_rgb[live.nonzero()[0, 0], 31] = 1e-30      # live through one tiny component of the last channel quad
    d_dep = torch.randn(R, generator=gen) * live
    if not depth_only:
        only_depth = (~live).nonzero()[:3, 0]                # rays that are live through their depth gradient alone
        d_dep[only_depth] = 0.5; live[only_depth] = True
    d_rgb, d_dep = d_rgb.to(DEV), d_dep.to(DEV)
    out = {}
    for flags in (False, True):
        d_cs = torch.full((R, S), 7.5, device=DEV); d_sig = torch.full((R, S), 7.5, device=DEV)
        act = torch.full((R,), -3, device=DEV, dtype=torch.int32)
        hip.call('spi_raymarch_bwd', None if depth_only else hip.ptr(col), hip.ptr(den), hip.ptr(dep), hip.ptr(perm), hip.ptr(cl),
                 None if depth_only else hip.ptr(d_rgb), hip.ptr(d_dep), None, R, S, S, 32, 1, None, None if depth_only else hip.ptr(d_cs), hip.ptr(d_sig),
                 hip.ptr(act) if flags else None, hip.stream())
        out[flags] = (d_cs, d_sig, act)
    torch.cuda.synchronize()
    (cs0, sig0, _), (cs1, sig1, act) = out[False], out[True]
    lv = live.to(DEV)
    assert torch.equal(act, lv.int()), 'flags'
    assert (sig1[~lv] == 7.5).all() and (cs1[~lv] == 7.5).all(), 'rows of dead rays must stay unwritten'
    assert torch.equal(sig1[lv], sig0[lv]) and (depth_only or torch.equal(cs1[lv], cs0[lv])), 'live rays: same arithmetic with and without flags'
    assert (sig0[~lv] == 0).all(), 'a dead ray marched anyway has an exactly-zero density gradient'
    # and against the oracle's autograd on the live rays
    k = lv.nonzero()[:64, 0]
    idx = perm[k].long()
    cr = torch.gather(col[k], 1, idx[:, :, None].expand(-1, -1, 32)).cpu()[None].requires_grad_(True)
    dr = torch.gather(den[k], 1, idx).cpu()[None, :, :, None].requires_grad_(True)
    a, b, _ = orr.ray_march(cr, dr, dep[k].cpu()[None, :, :, None], white_back=True)
    heads = [b] if depth_only else [a, b]
    gs = [d_dep[k].cpu()[None, :, None]] if depth_only else [d_rgb[k].cpu()[None], d_dep[k].cpu()[None, :, None]]
    gden = torch.autograd.grad(heads, [dr], gs)[0][0, :, :, 0]
    finite = torch.isfinite(gden).all(1)
    got = torch.gather(sig1[k], 1, idx).cpu()
    assert_close(got[finite], gden[finite], 3e-5, 'density gradient of live rays vs the oracle')


def test_raymarch_subset_of_stored_rows_through_perm():
    """S of S_store stored rows per ray are marched (perm picks them, sorted by depth): same results as on the compacted arrays; the rows that
    are not picked keep their contents in the gradient outputs."""
    from spi_amd import hip
    gen = torch.Generator().manual_seed(11)
    R, S, SS = 77, 100, 160
    col = torch.rand(R, SS, 32, generator=gen).to(DEV); den = (torch.randn(R, SS, generator=gen) * 3 + 1).to(DEV)
    pick = torch.stack([torch.randperm(SS, generator=gen)[:S] for _ in range(R)]).to(DEV)                 # rows marched, in depth order
    dep = torch.sort(torch.rand(R, S, generator=gen) + 2.25, 1)[0].to(DEV).contiguous()
    perm = pick.int().contiguous()
    cl = torch.tensor([2.25, 3.3], device=DEV)
    colc = torch.gather(col, 1, pick[:, :, None].expand(-1, -1, 32)).contiguous(); denc = torch.gather(den, 1, pick).contiguous()
    d_rgb = torch.randn(R, 32, generator=gen).to(DEV); d_dep = torch.randn(R, generator=gen).to(DEV)
    res = []
    for c, d, pm, ss in ((col, den, perm, SS), (colc, denc, None, S)):
        rgb = torch.empty(R, 32, device=DEV); depth = torch.empty(R, device=DEV); w = torch.empty(R, S - 1, device=DEV)
        hip.call('spi_raymarch_fwd', hip.ptr(c), hip.ptr(d), hip.ptr(dep), hip.ptr(pm), hip.ptr(cl), R, S, ss, 32, 1, hip.ptr(rgb), hip.ptr(depth), hip.ptr(w), None, hip.stream())
        d_cs = torch.full((R, ss), 7.5, device=DEV); d_sig = torch.full((R, ss), 7.5, device=DEV)
        hip.call('spi_raymarch_bwd', hip.ptr(c), hip.ptr(d), hip.ptr(dep), hip.ptr(pm), hip.ptr(cl), hip.ptr(d_rgb), hip.ptr(d_dep), None, R, S, ss, 32, 1,
                 None, hip.ptr(d_cs), hip.ptr(d_sig), None, hip.stream())
        res.append((rgb, depth, w, d_cs, d_sig))
    (rgb0, dp0, w0, cs0, sg0), (rgb1, dp1, w1, cs1, sg1) = res
    assert torch.equal(rgb0, rgb1) and torch.equal(dp0, dp1) and torch.equal(w0, w1)
    assert torch.equal(torch.gather(cs0, 1, pick), cs1) and torch.equal(torch.gather(sg0, 1, pick), sg1)
    rest = torch.ones(R, SS, dtype=torch.bool, device=DEV).scatter_(1, pick, False)
    assert (cs0[rest] == 7.5).all() and (sg0[rest] == 7.5).all()
    with pytest.raises(RuntimeError):
        hip.call('spi_raymarch_fwd', hip.ptr(col), hip.ptr(den), hip.ptr(dep), hip.ptr(perm), hip.ptr(cl), R, S, 300, 32, 1, hip.ptr(rgb0), None, None, None, hip.stream())


def test_gather_decode_golden_fwd_bwd(golden):
    from spi_amd.training.volumetric_rendering.renderer import ImportanceRenderer
    g = golden('renderer')
    P = _P(g)
    dec = _decoder(P)
    planes = g['planes'].to(DEV).requires_grad_(True)
    out = ImportanceRenderer().run_model(planes, dec, g['coords'].to(DEV), None, {'box_warp': 1})
    assert_close(out['rgb'], g['gd_rgb'], 2e-6, 'decoder rgb')
    assert_close(out['sigma'], g['gd_sigma'], 2e-6, 'decoder sigma')
    params = list(dec.parameters())
    grads = torch.autograd.grad([out['rgb'], out['sigma']], [planes] + params, [g['gd_drgb'].to(DEV), g['gd_dsigma'].to(DEV)])
    assert_close(grads[0], g['gd_gplanes'], 1e-5, 'grad planes')
    for (k, _), gv in zip(dec.named_parameters(), grads[1:]):
        assert_close(gv, g['gd_g_' + k], 2e-5, 'grad ' + k)


def test_importance_golden(golden):
    from spi_amd import hip
    g = golden('renderer')
    dep, w, u = g['rm_dep'].to(DEV)[..., 0].contiguous(), g['is_w'].to(DEV)[..., 0].contiguous(), g['is_u'].to(DEV).contiguous()
    n, m, s = dep.shape
    fine = torch.empty(n, m, 20, device=DEV)
    hip.call('spi_importance_sample', hip.ptr(dep), hip.ptr(w), hip.ptr(u), n * m, s, 20, hip.ptr(fine), 0, hip.stream())
    assert (fine.cpu() - g['is_fine'][..., 0]).abs().max() < 3e-6                 # draw order, as the reference returns them (the inverse CDF amplifies the rounding of the cumulative sums: parallel scan here, sequential cumsum there)
    hip.call('spi_importance_sample', hip.ptr(dep), hip.ptr(w), hip.ptr(u), n * m, s, 20, hip.ptr(fine), 1, hip.stream())
    assert (fine.cpu() - torch.sort(g['is_fine'][..., 0], dim=-1)[0]).abs().max() < 3e-6     # ascending variant: same multiset


def test_merge_sort_matches_torch_sort():
    from spi_amd import hip
    gen = torch.Generator().manual_seed(3)
    r, sc, sf = 300, 96, 96
    dc = torch.sort(torch.rand(r, sc, generator=gen), dim=1)[0].to(DEV)
    df = torch.rand(r, sf, generator=gen).to(DEV)
    out = torch.empty(r, sc + sf, device=DEV)
    perm = torch.empty(r, sc + sf, device=DEV, dtype=torch.int32)
    hip.call('spi_merge_sort_depths', hip.ptr(dc), hip.ptr(df), r, sc, sf, hip.ptr(out), hip.ptr(perm), hip.stream())
    ref, idx = torch.sort(torch.cat([dc, df], 1), dim=1)
    assert torch.equal(out, ref)
    assert torch.equal(perm.long(), idx)
    # the renderer's case: two ascending runs (binary-search ranks), with ties inside and across the runs, ragged sizes;
    # a stable sort of the concatenation puts equal coarse samples first
    for sc2, sf2 in ((96, 96), (48, 20), (7, 129)):
        dc = torch.sort((torch.rand(r, sc2, generator=gen) * 40).round() / 40, dim=1)[0].to(DEV)
        df = torch.sort((torch.rand(r, sf2, generator=gen) * 40).round() / 40, dim=1)[0].to(DEV)
        df[5] = dc[5, :1]                                                     # a whole run of ties
        out = torch.empty(r, sc2 + sf2, device=DEV)
        perm = torch.empty(r, sc2 + sf2, device=DEV, dtype=torch.int32)
        hip.call('spi_merge_sort_depths', hip.ptr(dc), hip.ptr(df), r, sc2, sf2, hip.ptr(out), hip.ptr(perm), hip.stream())
        ref, idx = torch.sort(torch.cat([dc, df], 1), dim=1, stable=True)
        assert torch.equal(out, ref)
        assert torch.equal(perm.long(), idx)


def test_rank_sorts_stay_permutations_with_nans():
    """A diverged ray (NaN depths / weights) must show up as NaN in the loss, like in the reference's torch.sort (NaNs last), not as an
    out-of-range read: the rank sorts of spi_merge_sort_depths / spi_importance_sample order NaNs last by position, so every output slot is
    written and `perm` is a permutation.  (Until round 4 every NaN got rank 0: the slots at the row's end kept what the buffer held before,
    and the decoder backward indexed with that -- an illegal address once the buffer's previous tenant was not a permutation.)"""
    from spi_amd import hip
    gen = torch.Generator().manual_seed(5)
    r, sc, sf = 64, 96, 96
    dc = torch.sort(torch.rand(r, sc, generator=gen), dim=1)[0]
    df = torch.sort(torch.rand(r, sf, generator=gen), dim=1)[0]
    df[3, 10] = float('nan'); df[7, :] = float('nan'); dc[9, 0] = float('nan'); df[11, 95] = float('nan'); df[11, 0] = float('nan')
    dc, df = dc.to(DEV), df.to(DEV)
    out = torch.full((r, sc + sf), -7.0, device=DEV)                       # dirty buffers: an unwritten slot keeps the sentinel
    perm = torch.full((r, sc + sf), 10 ** 9, device=DEV, dtype=torch.int32)
    hip.call('spi_merge_sort_depths', hip.ptr(dc), hip.ptr(df), r, sc, sf, hip.ptr(out), hip.ptr(perm), hip.stream())
    ref, idx = torch.sort(torch.cat([dc, df], 1), dim=1, stable=True)      # torch: NaNs last, stable
    assert torch.equal(torch.sort(perm.long(), dim=1)[0], torch.arange(sc + sf, device=DEV).expand(r, -1)), 'perm is not a permutation'
    assert torch.equal(perm.long(), idx)
    assert torch.equal(torch.isnan(out), torch.isnan(ref)) and torch.equal(torch.nan_to_num(out, nan=9.0), torch.nan_to_num(ref, nan=9.0))
    # importance sampling with NaN weights on some rays: all Sf slots of every ray are written (NaN where the ray diverged), no stale values
    dep = torch.sort(torch.rand(r, sc, generator=gen) + 2.0, dim=1)[0].to(DEV)
    w = torch.rand(r, sc - 1, generator=gen)
    w[5, 40] = float('nan'); w[6, :] = float('nan')
    u = torch.rand(r, sf, generator=gen).to(DEV)
    fine = torch.full((r, sf), -7.0, device=DEV)
    wd = w.to(DEV)             # (kept alive across the launch: `hip.ptr(w.to(DEV))` frees the temporary BEFORE the kernel runs -- harmless with the
    #                            caching allocator, a use-after-free under the guard-page allocator of tools/efence, which is how it was found)
    hip.call('spi_importance_sample', hip.ptr(dep), hip.ptr(wd), hip.ptr(u), r, sc, sf, hip.ptr(fine), 1, hip.stream())
    assert not bool((fine == -7.0).any()), 'unwritten slots'
    ok_rows = [i for i in range(r) if i not in (5, 6)]
    assert bool(torch.isfinite(fine[ok_rows]).all()) and bool((fine[ok_rows][:, 1:] >= fine[ok_rows][:, :-1]).all())
    assert bool(torch.isnan(fine[6]).all())


def test_full_render_golden_fwd_bwd(golden):
    from spi_amd.training.volumetric_rendering.renderer import ImportanceRenderer
    g = golden('renderer')
    P = _P(g)
    dec = _decoder(P)
    opts = dict(orr.DEFAULT_RENDERING, depth_resolution=12, depth_resolution_importance=12)
    c = g['cam']
    ro, rd = orr.ray_sampler(c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3), 8)
    planes = g['planes'].to(DEV).requires_grad_(True)
    rgb, depth, wsum = ImportanceRenderer()(planes, dec, ro.to(DEV), rd.to(DEV), opts, noise=(g['fr_xi'], g['fr_u']))
    assert_close(rgb, g['fr_rgb'], 1e-5, 'render rgb')
    assert_close(depth, g['fr_depth'], 1e-5, 'render depth')
    assert_close(wsum, g['fr_wsum'], 1e-5, 'render weight sum')
    grads = torch.autograd.grad([rgb, depth], [planes] + list(dec.parameters()), [g['fr_drgb'].to(DEV), g['fr_ddepth'].to(DEV)])
    assert_close(grads[0], g['fr_gplanes'], 5e-5, 'render grad planes')
    for (k, _), gv in zip(dec.named_parameters(), grads[1:]):
        assert_close(gv, g['fr_g_' + k], 5e-5, 'render grad ' + k)


@pytest.mark.parametrize('tag', ['auto', 'disparity', 'dnoise', 'tiny', 'tiny_auto_dnoise'])
def test_render_options_vs_reference_golden(golden, tag):
    """The ImportanceRenderer surface off the SPI path (VERDICT r02 missing #3): 'auto' ray limits (renderer.py:91-97), disparity-space
    sampling (:175-182), density noise (:146-147) on the fused kernels, and a decoder callable that is not the OSG MLP (:88,142-148)
    through sample_from_planes (its own HIP kernels) + the callable + MipRayMarcher2 -- against the REFERENCE's outputs and plane
    gradients with its recorded draws replayed."""
    from spi_amd.training.volumetric_rendering.renderer import ImportanceRenderer
    from render_variants import RENDER_VARIANTS, tiny_decoder
    g = golden('renderer_options')
    opts = dict(orr.DEFAULT_RENDERING, depth_resolution=12, depth_resolution_importance=12, **RENDER_VARIANTS[tag])

    class Tiny(torch.nn.Module):
        def forward(self, sampled_features, ray_directions):
            rgb, sigma = tiny_decoder(g, sampled_features, ray_directions)
            return {'rgb': rgb, 'sigma': sigma}
    dec = Tiny() if tag.startswith('tiny') else _decoder(_P(g))
    planes = g['planes'].to(DEV).requires_grad_(True)
    noise = [g[tag + '_xi'], g[tag + '_u']] + ([g[tag + '_eps0'], g[tag + '_eps1']] if 'dnoise' in tag else [])
    rgb, depth, wsum = ImportanceRenderer()(planes, dec, g['ray_o'].to(DEV), g['ray_d'].to(DEV), opts, noise=noise)
    assert rgb.shape == g[tag + '_rgb'].shape
    assert_close(rgb, g[tag + '_rgb'], 2e-5, tag + ' rgb')
    assert_close(depth, g[tag + '_depth'], 2e-5, tag + ' depth')
    assert_close(wsum, g[tag + '_wsum'], 2e-5, tag + ' weight sum')
    gp, = torch.autograd.grad([rgb, depth], [planes], [g[tag + '_drgb'].to(DEV), g[tag + '_ddepth'].to(DEV)])
    assert_close(gp, g[tag + '_gplanes'], 1e-4, tag + ' grad planes')


def test_sample_from_planes_golden_fwd_bwd(golden):
    """sample_from_planes as its own operator (renderer.py:55-65) against the reference's per-plane features and their plane gradient."""
    from spi_amd.training.volumetric_rendering.renderer import sample_from_planes, generate_planes
    g = golden('renderer')
    planes = g['planes'].to(DEV).requires_grad_(True)
    feats = sample_from_planes(generate_planes(), planes, g['coords'].to(DEV), padding_mode='zeros', box_warp=1)
    assert_close(feats, g['gd_feats'], 1e-6, 'sample_from_planes')
    gen = torch.Generator().manual_seed(3)
    d = torch.randn(feats.shape, generator=gen)
    ref_planes = g['planes'].clone().requires_grad_(True)
    gref, = torch.autograd.grad(orr.sample_planes(ref_planes, g['coords']), ref_planes, d)
    ggpu, = torch.autograd.grad(feats, planes, d.to(DEV))
    assert_close(ggpu, gref, 1e-5, 'sample_from_planes plane gradient')


@pytest.mark.parametrize('depth', [96, 128])
def test_full_render_config_size_vs_oracle(depth):
    """BASELINE config 2 (96+96) and config 5 (128+128 = the kernels' 256-sample limit) on a ray subset: 256^2 planes,
    forward and gradients wrt planes and decoder (oracle finishes in seconds on 1024 rays)."""
    from spi_amd.training.volumetric_rendering.renderer import ImportanceRenderer
    from spi_amd.utils import camera_utils as cu
    from synth_weights import synth_tensor
    gen = torch.Generator().manual_seed(17)
    P = {f'decoder.net.{i}.{k}': synth_tensor(f'decoder.net.{i}.{k}', s) for i, k, s in
         ((0, 'weight', (64, 32)), (0, 'bias', (64,)), (2, 'weight', (33, 64)), (2, 'bias', (33,)))}
    dec = _decoder(P)
    planes = torch.randn(1, 3, 32, 256, 256, generator=gen) * 0.7
    c = cu.cal_canonical_c(0.4, 0.0)
    ro, rd = orr.ray_sampler(c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3), 128)
    sel = torch.randperm(128 * 128, generator=gen)[:1024]
    ro, rd = ro[:, sel].contiguous(), rd[:, sel].contiguous()
    opts = dict(orr.DEFAULT_RENDERING, depth_resolution=depth, depth_resolution_importance=depth)
    xi, u = torch.rand(1, 1024, depth, 1, generator=gen), torch.rand(1024, depth, generator=gen)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    pl_ref = planes.clone().requires_grad_(True)
    a, b, cc = orr.render(Pg, pl_ref, ro, rd, opts, xi=xi, u=u)
    pl = planes.to(DEV).requires_grad_(True)
    x, y, z = ImportanceRenderer()(pl, dec, ro.to(DEV), rd.to(DEV), opts, noise=(xi, u))
    # north_star tolerance: 1e-3 relative on rendered RGB / depth
    assert_close(x, a, 1e-3, 'rgb'); assert_close(y, b, 1e-3, 'depth'); assert_close(z, cc, 1e-3, 'weight sum')
    assert rel_err(x, a) < 2e-4 and rel_err(y, b) < 2e-5      # what the kernels actually achieve
    d1, d2 = torch.randn(a.shape, generator=gen), torch.randn(b.shape, generator=gen)
    gref = torch.autograd.grad([a, b], [pl_ref] + [Pg[k] for k in sorted(Pg)], [d1, d2])
    names = {f'decoder.net.{i}.{k}': getattr(dec.net[i], k) for i in (0, 2) for k in ('weight', 'bias')}
    ggpu = torch.autograd.grad([x, y], [pl] + [names[k] for k in sorted(Pg)], [d1.to(DEV), d2.to(DEV)])
    for gg, gr, nm in zip(ggpu, gref, ['planes'] + sorted(Pg)):
        assert_close(gg, gr, 1e-3, f'config-size grad {nm}')


@pytest.mark.parametrize('frozen', [False, True])
def test_depth_only_and_rgb_only_backward(golden, frozen):
    """A loss that uses only one renderer output takes the short paths (no colour gradient buffers for a depth-only loss,
    `d_rgb = NULL` kernels): same gradients as the full path fed with an explicit zero gradient for the unused output."""
    from spi_amd.training.volumetric_rendering.renderer import ImportanceRenderer
    g = golden('renderer')
    dec = _decoder(_P(g))
    for p in dec.parameters():
        p.requires_grad_(not frozen)
    opts = dict(orr.DEFAULT_RENDERING, depth_resolution=12, depth_resolution_importance=12)
    c = g['cam']
    ro, rd = orr.ray_sampler(c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3), 8)
    planes = g['planes'].to(DEV).requires_grad_(True)
    wrt = [planes] + ([] if frozen else list(dec.parameters()))
    rgb, depth, _ = ImportanceRenderer()(planes, dec, ro.to(DEV), rd.to(DEV), opts, noise=(g['fr_xi'], g['fr_u']))
    drgb, ddep = g['fr_drgb'].to(DEV), g['fr_ddepth'].to(DEV)
    full_d = torch.autograd.grad([rgb, depth], wrt, [torch.zeros_like(drgb), ddep], retain_graph=True)
    only_d = torch.autograd.grad([depth], wrt, [ddep], retain_graph=True, allow_unused=True)
    full_c = torch.autograd.grad([rgb, depth], wrt, [drgb, torch.zeros_like(ddep)], retain_graph=True)
    only_c = torch.autograd.grad([rgb], wrt, [drgb], retain_graph=True)
    names = ['planes'] + ([] if frozen else [k for k, _ in dec.named_parameters()])
    for a, b, nm in zip(only_d, full_d, names):
        assert_close(a if a is not None else torch.zeros_like(b), b, 2e-5, 'depth-only grad ' + nm)
    for a, b, nm in zip(only_c, full_c, names):
        assert_close(a, b, 2e-5, 'rgb-only grad ' + nm)


@pytest.mark.parametrize('frozen', [False, True])
def test_render_backward_skips_zero_gradient_rays(frozen):
    """Masked losses (SPI's rot / mirror-rot branches) leave most rays with an exactly-zero gradient: those rays are flagged by
    the march backward and skipped by the decoder backward (whole 8x8 patches at no cost).  Gradients equal the oracle's
    autograd on the same masked gradient; fully and partially masked patches are both present."""
    from spi_amd.training.volumetric_rendering.renderer import ImportanceRenderer
    from spi_amd.utils import camera_utils as cu
    from synth_weights import synth_tensor
    gen = torch.Generator().manual_seed(23)
    P = {f'decoder.net.{i}.{k}': synth_tensor(f'decoder.net.{i}.{k}', s) for i, k, s in
         ((0, 'weight', (64, 32)), (0, 'bias', (64,)), (2, 'weight', (33, 64)), (2, 'bias', (33,)))}
    dec = _decoder(P)
    for p in dec.parameters():
        p.requires_grad_(not frozen)
    planes = torch.randn(1, 3, 32, 64, 64, generator=gen) * 0.7
    c = cu.cal_canonical_c(0.3, 0.1)
    ro, rd = orr.ray_sampler(c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3), 32)           # 32 x 32 rays = 16 patches of 8 x 8
    opts = dict(orr.DEFAULT_RENDERING, depth_resolution=12, depth_resolution_importance=12)
    xi, u = torch.rand(1, 1024, 12, 1, generator=gen), torch.rand(1024, 12, generator=gen)
    keep = torch.zeros(32, 32)
    keep[3:13, 5:11] = 1                          # touches 4 patches partially, leaves 12 patches untouched
    keep = keep.reshape(1, 1024, 1)
    Pg = {k: v.clone().requires_grad_(not frozen) for k, v in P.items()}
    pl_ref = planes.clone().requires_grad_(True)
    a, b, _ = orr.render(Pg, pl_ref, ro, rd, opts, xi=xi, u=u)
    d1, d2 = torch.randn(a.shape, generator=gen) * keep, torch.randn(b.shape, generator=gen) * keep
    wrt_ref = [pl_ref] + ([] if frozen else [Pg[k] for k in sorted(Pg)])
    gref = torch.autograd.grad([a, b], wrt_ref, [d1, d2])
    pl = planes.to(DEV).requires_grad_(True)
    x, y, _ = ImportanceRenderer()(pl, dec, ro.to(DEV), rd.to(DEV), opts, noise=(xi, u))
    names = {f'decoder.net.{i}.{k}': getattr(dec.net[i], k) for i in (0, 2) for k in ('weight', 'bias')}
    wrt = [pl] + ([] if frozen else [names[k] for k in sorted(Pg)])
    ggpu = torch.autograd.grad([x, y], wrt, [d1.to(DEV), d2.to(DEV)], retain_graph=True)
    for gg, gr, nm in zip(ggpu, gref, ['planes'] + sorted(Pg)):
        assert torch.isfinite(gg).all()
        assert_close(gg, gr, 1e-4, f'masked-gradient render grad {nm}')
    # everything masked: exact zeros, no NaN from unwritten rows
    g0 = torch.autograd.grad([x, y], wrt, [torch.zeros_like(x), torch.zeros_like(y)])
    assert all(float(g.abs().max()) == 0.0 for g in g0)


def test_depth_only_rendering_equals_full_depth():
    """renderer(..., depth_only=True): same depth map and the same plane / decoder gradients of a depth loss as the full render
    (whose colour half is then simply unused)."""
    from spi_amd.training.volumetric_rendering.renderer import ImportanceRenderer
    from spi_amd.training.volumetric_rendering.ray_sampler import RaySampler
    from spi_amd.training.triplane import OSGDecoder
    from spi_amd.utils import camera_utils as cu
    torch.manual_seed(11)
    n = 2
    dec = OSGDecoder(32, {'decoder_lr_mul': 1, 'decoder_output_dim': 32}).to(DEV)
    planes = (torch.randn(n, 3, 32, 64, 64, device=DEV) * 0.5).requires_grad_(True)
    c = torch.cat([cu.cal_canonical_c(0.3, 0.1), cu.cal_canonical_c(-0.2, 0.0)]).to(DEV)
    ro, rd = RaySampler()(c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3), 32)
    opts = dict(depth_resolution=24, depth_resolution_importance=24, ray_start=2.25, ray_end=3.3, box_warp=1, white_back=False)
    noise = (torch.rand(n, 1024, 24, 1, device=DEV), torch.rand(n * 1024, 24, device=DEV))
    ren = ImportanceRenderer()
    rgb, depth, _ = ren(planes, dec, ro, rd, opts, noise=noise)
    none, depth_o, _ = ren(planes, dec, ro, rd, opts, noise=noise, depth_only=True)
    assert none is None and torch.equal(depth_o, depth)
    g = torch.randn_like(depth)
    wrt = [planes] + list(dec.parameters())
    full = torch.autograd.grad(depth, wrt, g, allow_unused=True)
    only = torch.autograd.grad(depth_o, wrt, g, allow_unused=True)
    for a, b in zip(only, full):
        assert (a is None) == (b is None)
        if a is not None:
            assert_close(a, b, 1e-5, 'depth-only gradient')
