"""ImportanceRenderer on HIP kernels: stratified coarse pass, tri-plane gather + OSG decoder,
hierarchical importance resampling, merge-sort and alpha-composite march.

Same surface as the reference's ``ImportanceRenderer`` (eg3d/training/volumetric_rendering/renderer.py:88-148):
  ``forward(planes [N,3,32,H,W], decoder, ray_origins [N,M,3], ray_directions [N,M,3], rendering_options)
      -> (rgb [N,M,32], depth [N,M,1], weight_sum [N,M,1])``
  ``run_model(planes, decoder, sample_coordinates [N,P,3], sample_directions, options) -> {'rgb','sigma'}``

MI355X-first layout (not the reference's op-by-op graph):
  * planes are re-laid out channels-last once per call, so each bilinear corner is one 128-B line;
  * gather + plane mean + 32-64-33 MLP + sigmoid are ONE kernel per pass; nothing of size
    [N,3,M*S,32] or [M*S,64] is ever materialised;
  * coarse and fine passes write into one [R, Sc+Sf, .] buffer; the sort produces a permutation that
    the march reads through, so the reference's cat + 3 gathers (renderer.py:157-167) never happen;
  * the backward recomputes gather + hidden layer instead of saving them.
The two random draws (``rand_like`` [N,M,Sc,1] then ``rand`` [N*M,Sf], renderer.py:190,237) can be
injected with ``noise=(xi, u)``; by default they come from the device generator in that order.

The whole ``ImportanceRenderer`` surface is served (round 3):
  * scalar ``ray_start`` / ``ray_end`` (renderer.py:188-190): the FFHQ configuration SPI runs -- everything in the fused kernels;
  * ``ray_start = ray_end = 'auto'`` (:91-97, box limits per ray) and ``disparity_space_sampling`` (:175-182): the coarse depths are
    computed by tensor ops (math_utils.py) and handed to the same fused kernels; the coarse / fine merge then uses a stable sort
    instead of the two-run merge kernel (a ray that misses the box may carry descending depths, as in the reference);
  * ``density_noise`` (:146-147): ``sigma += randn_like(sigma) * density_noise`` after each decoder pass; the draws can be injected as
    ``noise=(xi, u, eps_coarse, eps_fine)`` (the reference's order: rand_like, randn_like, rand, randn_like);
  * any decoder callable ``decoder(sampled_features [N,3,P,C], sample_directions) -> {'rgb', 'sigma'}`` (:88,142-148): an
    ``OSGDecoder``-shaped module takes the fused gather + MLP kernels; anything else goes through ``sample_from_planes`` (its own HIP
    kernel pair) + the callable under autograd + ``MipRayMarcher2`` -- the reference's op-by-op composition.
"""
import math
import os
import torch

from ... import hip
from ...torch_utils import zero_arena
from .ray_marcher import MipRayMarcher2, depth_range

DEC_DUMP_ROWS = 193
MARCH_EVENTS = None      # bench.py: list collecting (start, end, rays) HIP events around every final-march launch
MARCH_BWD_EVENTS = None  # bench.py: the same around every march-backward launch with a colour gradient: (start, end, rays, active-ray flags)
DECODE_FWD_EVENTS = None # bench.py: (start, end, points) around every tri-plane gather + decoder forward launch that evaluates the colour rows
DECODE_BWD_EVENTS = None # bench.py: (start, end, rays, samples per ray, active-ray flags, wgrad?, rgb?) around every tiled decoder-backward call


def _timed(events):
    return events is not None and not torch.cuda.is_current_stream_capturing()      # (a captured step cannot hold timing events)


def _scaled_decoder(w1, b1, w2, b2, gains, transpose_w1):
    """(w1 * g, b1 * g, w2 * g, b2 * g) in one launch; w1 comes back transposed ([32,64], the kernels' operand layout) if asked."""
    w1, b1, w2, b2 = (t.detach().contiguous().float() for t in (w1, b1, w2, b2))
    flat = torch.empty(64 * 32 + 64 + 33 * 64 + 33, device=w1.device, dtype=torch.float32)
    o1, ob1, o2, ob2 = flat[:2048].view(32, 64) if transpose_w1 else flat[:2048].view(64, 32), flat[2048:2112], flat[2112:4224].view(33, 64), flat[4224:]
    hip.call('spi_decoder_gains', hip.ptr(w1), hip.ptr(b1), hip.ptr(w2), hip.ptr(b2), *[float(g) for g in gains], hip.ptr(o1), hip.ptr(ob1), hip.ptr(o2), hip.ptr(ob2),
             int(transpose_w1), hip.stream())
    return o1, ob1, o2, ob2


def decoder_tensors(decoder):
    """(w1t [32,64], b1 [64], w2 [33,64], b2 [33]) with the FullyConnectedLayer gains folded in."""
    l0, l2 = decoder.net[0], decoder.net[2]
    w1t = (l0.weight * l0.weight_gain).t().contiguous()
    b1 = (l0.bias * l0.bias_gain).contiguous()
    w2 = (l2.weight * l2.weight_gain).contiguous()
    b2 = (l2.bias * l2.bias_gain).contiguous()
    return w1t, b1, w2, b2


def planes_to_nhwc(planes):
    n, p, c, h, w = planes.shape
    src = planes.contiguous().float()
    dst = torch.empty(n, p, h, w, c, device=planes.device, dtype=torch.float32)
    hip.call('spi_nchw_to_nhwc', hip.ptr(src), hip.ptr(dst), n * p, c, h, w, hip.stream())
    return dst


def planes_to_nchw(planes_nhwc):
    n, p, h, w, c = planes_nhwc.shape
    dst = torch.empty(n, p, c, h, w, device=planes_nhwc.device, dtype=torch.float32)
    hip.call('spi_nhwc_to_nchw', hip.ptr(planes_nhwc), hip.ptr(dst), n * p, c, h, w, hip.stream())
    return dst


def _decode_fwd(planes_nhwc, dec, *, coords=None, rays=None, depths=None, box_warp, out=None, out_S=0, out_off=0):
    n, _, h, w, _ = planes_nhwc.shape
    w1t, b1, w2, b2 = dec
    if coords is not None:
        p, s = coords.shape[1], 1
        rgb = torch.empty(n, p, 32, device=planes_nhwc.device, dtype=torch.float32)
        sigma = torch.empty(n, p, device=planes_nhwc.device, dtype=torch.float32)
        ro = rd = dp = None
    else:
        ray_o, ray_d = rays
        s = depths.shape[-1]
        p = ray_o.shape[1] * s
        ro, rd, dp = hip.ptr(ray_o), hip.ptr(ray_d), hip.ptr(depths)
        rgb, sigma = out
    timed = _timed(DECODE_FWD_EVENTS) and rgb is not None
    if timed:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    hip.call('spi_triplane_decode_fwd', hip.ptr(planes_nhwc), hip.ptr(coords) if coords is not None else None, ro, rd, dp,
             hip.ptr(w1t), hip.ptr(b1), hip.ptr(w2), hip.ptr(b2), n, p, s, h, w, float(box_warp), out_S, out_off,
             hip.ptr(rgb), hip.ptr(sigma), hip.stream())
    if timed:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        DECODE_FWD_EVENTS.append((e0, e1, n * p))
    return rgb, sigma


def _decode_bwd(planes_nhwc, dec, d_rgb, d_sigma, d_planes, *, coords=None, rays=None, depths=None, box_warp, out_S=0, out_off=0,
                want_wgrad=False):
    """Accumulates into d_planes; returns (dW1 [64,32], db1, dW2 [33,64], db2) wrt the GAINED weights or None."""
    n, _, h, w, _ = planes_nhwc.shape
    w1t, b1, w2, b2 = dec
    if coords is not None:
        p, s = coords.shape[1], 1
        ro = rd = dp = None
    else:
        ray_o, ray_d = rays
        s = depths.shape[-1]
        p = ray_o.shape[1] * s
        ro, rd, dp = hip.ptr(ray_o), hip.ptr(ray_d), hip.ptr(depths)
    dump = torch.empty(DEC_DUMP_ROWS, n * p, device=planes_nhwc.device, dtype=torch.float32) if want_wgrad else None
    hip.call('spi_triplane_decode_bwd', hip.ptr(planes_nhwc), hip.ptr(coords) if coords is not None else None, ro, rd, dp,
             hip.ptr(w1t), hip.ptr(b1), hip.ptr(w2), hip.ptr(b2), hip.ptr(d_rgb), hip.ptr(d_sigma), n, p, s, h, w, float(box_warp),
             out_S, out_off, hip.ptr(d_planes), hip.ptr(dump), hip.stream())
    return _decoder_wgrad(dump) if want_wgrad else None


def _decoder_grad_buffers(dev):
    """dW1 [64,32] | db1 [64] | dW2 [33,64] | db2 [33] carved from ONE buffer in this order: the library then clears them with one fill
    launch instead of four (spi_triplane_decode_bwd_sorted / spi_decoder_wgrad test for exactly this adjacency)."""
    buf = torch.empty(64 * 32 + 64 + 33 * 64 + 33, device=dev, dtype=torch.float32)
    return buf[:2048].view(64, 32), buf[2048:2112], buf[2112:4224].view(33, 64), buf[4224:4257]


def _decoder_wgrad(dump):
    """(dW1 [64,32], db1 [64], dW2 [33,64], db2 [33]) from an activation dump [193, cols]: one streaming MFMA kernel."""
    dev = dump.device
    cols = dump.shape[1]
    if cols % 4 != 0:                      # explicit-coordinate path with an odd point count: pad the columns
        pad = 4 - cols % 4
        dump = torch.cat([dump, dump.new_zeros(dump.shape[0], pad)], dim=1).contiguous()
        cols += pad
    gw1, gb1, gw2, gb2 = _decoder_grad_buffers(dev)
    hip.call('spi_decoder_wgrad', hip.ptr(dump), cols, hip.ptr(gw1), hip.ptr(gb1), hip.ptr(gw2), hip.ptr(gb2), hip.stream())
    return gw1, gb1, gw2, gb2


class _Render(torch.autograd.Function):
    @staticmethod
    def forward(ctx, planes, w1, b1, w2, b2, gains, ray_o, ray_d, xi, u, opts):
        n, _, c, h, w = planes.shape
        m = ray_o.shape[1]
        sc, sf = int(opts['depth_resolution']), int(opts['depth_resolution_importance'])
        s = sc + sf
        r = n * m
        dev = planes.device
        dec = _scaled_decoder(w1, b1, w2, b2, gains, True)
        planes_nhwc = planes_to_nhwc(planes.detach())
        ray_o = ray_o.detach().contiguous().float()
        ray_d = ray_d.detach().contiguous().float()
        box_warp = float(opts['box_warp'])
        white_back = int(bool(opts.get('white_back', False)))
        # coarse pass
        general = opts.get('coarse_depths') is not None          # 'auto' limits / disparity sampling: depths precomputed by ImportanceRenderer
        if general:
            d_c = opts['coarse_depths'].reshape(n, m, sc).contiguous().float()
        else:
            d_c = torch.empty(n, m, sc, device=dev, dtype=torch.float32)
            xi = xi.reshape(n, m, sc).contiguous().float()
            hip.call('spi_coarse_depths', hip.ptr(xi), r, sc, float(opts['ray_start']), float(opts['ray_end']), hip.ptr(d_c), hip.stream())
        dnoise = float(opts.get('density_noise', 0) or 0)
        eps_c, eps_f = opts.get('density_eps', (None, None))
        # depth_only (SPI's depth-regularisation branch reads nothing but image_depth): no colour rows are decoded, stored or composited
        depth_only = bool(opts.get('depth_only', False))
        rgb_all = torch.empty(n, m, s, 32, device=dev, dtype=torch.float32) if not depth_only else None
        sig_all = torch.empty(n, m, s, device=dev, dtype=torch.float32)
        _decode_fwd(planes_nhwc, dec, rays=(ray_o, ray_d), depths=d_c, box_warp=box_warp, out=(rgb_all, sig_all), out_S=s, out_off=0)
        if dnoise > 0:                                           # renderer.py:146-147 (a constant wrt every gradient)
            e = torch.randn(n, m, sc, device=dev) if eps_c is None else eps_c.to(dev).reshape(n, m, sc).float()
            sig_all[:, :, :sc] += e * dnoise
        if sf > 0:
            w_c = torch.empty(n, m, sc - 1, device=dev, dtype=torch.float32)
            hip.call('spi_raymarch_fwd', None, hip.ptr(sig_all), hip.ptr(d_c), None, None, r, sc, s, 32, white_back, None, None,
                     hip.ptr(w_c), None, hip.stream())
            d_f = torch.empty(n, m, sf, device=dev, dtype=torch.float32)
            u = u.reshape(n, m, sf).contiguous().float()
            hip.call('spi_importance_sample', hip.ptr(d_c), hip.ptr(w_c), hip.ptr(u), r, sc, sf, hip.ptr(d_f), 1, hip.stream())
            _decode_fwd(planes_nhwc, dec, rays=(ray_o, ray_d), depths=d_f, box_warp=box_warp, out=(rgb_all, sig_all), out_S=s, out_off=sc)
            if dnoise > 0:
                if eps_f is None:
                    e = torch.randn(n, m, sf, device=dev)
                else:       # an injected draw is indexed like the reference's UNSORTED fine samples (draw order); ours are emitted ascending:
                    raw = torch.empty_like(d_f)                  # the same samples in draw order -> the kernel's stable rank order
                    hip.call('spi_importance_sample', hip.ptr(d_c), hip.ptr(w_c), hip.ptr(u), r, sc, sf, hip.ptr(raw), 0, hip.stream())
                    e = torch.gather(eps_f.to(dev).reshape(n, m, sf).float(), -1, torch.argsort(raw, dim=-1, stable=True))
                sig_all[:, :, sc:] += e * dnoise
            if general:
                # per-ray limits: the coarse run of a ray that misses the box may be descending (reference behaviour, renderer.py:95-97),
                # which the two-ascending-runs merge kernel does not accept -> the reference's own cat + sort (:157-167), kept as a permutation
                d_all, perm = torch.sort(torch.cat([d_c, d_f], dim=-1), dim=-1, stable=True)
                d_all, perm = d_all.contiguous(), perm.to(torch.int32).contiguous()
            else:
                d_all = torch.empty(n, m, s, device=dev, dtype=torch.float32)
                perm = torch.empty(n, m, s, device=dev, dtype=torch.int32)
                hip.call('spi_merge_sort_depths', hip.ptr(d_c), hip.ptr(d_f), r, sc, sf, hip.ptr(d_all), hip.ptr(perm), hip.stream())
        else:
            d_f, d_all, perm = None, d_c, None
        clamp2 = depth_range(d_all)
        rgb = torch.empty(n, m, 32, device=dev, dtype=torch.float32) if not depth_only else None
        depth = torch.empty(n, m, 1, device=dev, dtype=torch.float32)
        wsum = torch.empty(n, m, 1, device=dev, dtype=torch.float32)
        timed_fwd = MARCH_EVENTS is not None and not depth_only and not torch.cuda.is_current_stream_capturing()   # (a captured step cannot hold timing events)
        if timed_fwd:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        hip.call('spi_raymarch_fwd', hip.ptr(rgb_all), hip.ptr(sig_all), hip.ptr(d_all), hip.ptr(perm), hip.ptr(clamp2), r, s, s, 32,
                 white_back, hip.ptr(rgb), hip.ptr(depth), None, hip.ptr(wsum), hip.stream())
        if timed_fwd:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            MARCH_EVENTS.append((e0, e1, r))
        ctx.save_for_backward(planes_nhwc, *dec, ray_o, ray_d, d_c, d_f, rgb_all, sig_all, d_all, perm, clamp2)
        ctx.meta = (n, m, sc, sf, box_warp, white_back, gains)
        ctx.aux = dict(depths_coarse=d_c, depths_fine=d_f, depths_sorted=d_all, perm=perm)
        ctx.mark_non_differentiable(wsum)
        ctx.set_materialize_grads(False)          # an unused output arrives as None in backward (depth-only / image-only losses)
        return rgb, depth, wsum

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_rgb, d_depth, _d_wsum):
        planes_nhwc, w1t, b1, w2, b2, ray_o, ray_d, d_c, d_f, rgb_all, sig_all, d_all, perm, clamp2 = ctx.saved_tensors
        n, m, sc, sf, box_warp, white_back, gains = ctx.meta
        dec = (w1t, b1, w2, b2)
        s = sc + sf
        r = n * m
        dev = planes_nhwc.device
        if d_rgb is None and d_depth is None:
            return (None,) * 11
        if os.environ.get('SPI_DEBUG_SPARSITY'):                  # debugging aid: how many rays carry a gradient at all
            z = torch.ones(n, m, dtype=torch.bool, device=dev)
            if d_rgb is not None:
                z &= (d_rgb.reshape(n, m, -1) == 0).all(-1)
            if d_depth is not None:
                z &= (d_depth.reshape(n, m) == 0)
            print(f'[sparsity] render backward N={n}: {float(z.float().mean()) * 100:.1f} % of rays have an all-zero gradient', flush=True)
        # d_rgb None: only the depth map feeds the loss (SPI's depth branch) -> no colour gradient buffers or traffic at all
        d_rgb = d_rgb.contiguous().float() if d_rgb is not None else None
        dd = d_depth.contiguous().float() if d_depth is not None else None
        # the colour-row gradient is d_rgb[ray] * (w_{k-1} + w_k): the marcher writes the per-sample scalar only and the decoder
        # backward rebuilds the rows from the per-ray d_rgb -- the [N,M,S,32] gradient tensor (403 MB per image) never exists
        d_cs = torch.empty_like(sig_all) if d_rgb is not None else None
        d_sig = torch.empty_like(sig_all)
        # rays with an exactly-zero incoming gradient (SPI's masked pseudo-view losses: 65-90 % of those views) are flagged by
        # the march backward and skipped by the decoder backward; their rows of d_col / d_sig stay unwritten
        from ...configs import global_config
        active = torch.empty(r, device=dev, dtype=torch.int32) if global_config.exploit_sparsity else None
        timed = MARCH_BWD_EVENTS is not None and d_rgb is not None and not torch.cuda.is_current_stream_capturing()
        if timed:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        hip.call('spi_raymarch_bwd', hip.ptr(rgb_all), hip.ptr(sig_all), hip.ptr(d_all), hip.ptr(perm), hip.ptr(clamp2), hip.ptr(d_rgb),
                 hip.ptr(dd), None, r, s, s, 32, white_back, None, hip.ptr(d_cs), hip.ptr(d_sig), hip.ptr(active), hip.stream())
        if timed:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            MARCH_BWD_EVENTS.append((e0, e1, r, active))
        want_w = any(ctx.needs_input_grad[1:5])
        d_planes = zero_arena.zeros_like(planes_nhwc)
        # one pass over all Sc+Sf samples in sorted order, 8x8 ray patches (LDS-aggregated scatter)
        _, _, hh, ww, _ = planes_nhwc.shape
        res = int(round(math.sqrt(m)))
        ray_w = res if res * res == m else m
        ws = torch.empty(hip.lib().spi_triplane_decode_bwd_sorted_ws(n, m, s, ray_w), device=dev, dtype=torch.float32)
        gw = None
        if want_w:                               # decoder weight gradients come out of the same kernel (no activation dump)
            gw = _decoder_grad_buffers(dev)
        timed_dec = _timed(DECODE_BWD_EVENTS)
        if timed_dec:
            f0 = torch.cuda.Event(enable_timing=True)
            f0.record()
        hip.call('spi_triplane_decode_bwd_sorted', hip.ptr(planes_nhwc), hip.ptr(ray_o), hip.ptr(ray_d), hip.ptr(d_all), hip.ptr(perm),
                 hip.ptr(w1t), hip.ptr(b1), hip.ptr(w2), hip.ptr(b2), hip.ptr(d_rgb), hip.ptr(d_cs), hip.ptr(rgb_all if d_rgb is not None else None), hip.ptr(d_sig), n, m, s, ray_w, hh, ww, box_warp,
                 hip.ptr(d_planes), hip.ptr(ws), *([hip.ptr(g) for g in gw] if want_w else [None] * 4), hip.ptr(active), hip.stream())
        if timed_dec:
            f1 = torch.cuda.Event(enable_timing=True)
            f1.record()
            DECODE_BWD_EVENTS.append((f0, f1, r, s, active, bool(want_w), d_rgb is not None))
        g_planes = planes_to_nchw(d_planes) if ctx.needs_input_grad[0] else None
        gw1 = gb1 = gw2 = gb2 = None
        if want_w:
            gw1, gb1, gw2, gb2 = _scaled_decoder(*gw, gains, False)
        return g_planes, gw1, gb1, gw2, gb2, None, None, None, None, None, None


class _RunModel(torch.autograd.Function):
    """sample_from_planes + decoder at explicit coordinates (ImportanceRenderer.run_model)."""
    @staticmethod
    def forward(ctx, planes, w1, b1, w2, b2, gains, coords, box_warp):
        dec = _scaled_decoder(w1, b1, w2, b2, gains, True)
        planes_nhwc = planes_to_nhwc(planes.detach())
        coords = coords.detach().contiguous().float()
        rgb, sigma = _decode_fwd(planes_nhwc, dec, coords=coords, box_warp=box_warp)
        ctx.save_for_backward(planes_nhwc, *dec, coords)
        ctx.meta = (box_warp, gains)
        return rgb, sigma.unsqueeze(-1)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_rgb, d_sigma):
        planes_nhwc, w1t, b1, w2, b2, coords = ctx.saved_tensors
        box_warp, gains = ctx.meta
        n, p = coords.shape[:2]
        d_rgb = d_rgb.contiguous().float() if d_rgb is not None else torch.zeros(n, p, 32, device=coords.device)
        d_sigma = d_sigma.reshape(n, p).contiguous().float() if d_sigma is not None else torch.zeros(n, p, device=coords.device)
        want_w = any(ctx.needs_input_grad[1:5])
        d_planes = zero_arena.zeros_like(planes_nhwc)
        gw = _decode_bwd(planes_nhwc, (w1t, b1, w2, b2), d_rgb, d_sigma, d_planes, coords=coords, box_warp=box_warp, want_wgrad=want_w)
        g_planes = planes_to_nchw(d_planes) if ctx.needs_input_grad[0] else None
        gw1 = gb1 = gw2 = gb2 = None
        if want_w:
            gw1, gb1, gw2, gb2 = _scaled_decoder(*gw, gains, False)
        return g_planes, gw1, gb1, gw2, gb2, None, None, None


def _decoder_params(decoder):
    l0, l2 = decoder.net[0], decoder.net[2]
    gains = (float(l0.weight_gain), float(l0.bias_gain), float(l2.weight_gain), float(l2.bias_gain))
    return (l0.weight, l0.bias, l2.weight, l2.bias), gains


def _is_osg_decoder(decoder):
    """True for a module with the OSG decoder's exact arithmetic (triplane.py:112-135): net = [FC 32->64, Softplus, FC 64->33] and the
    class's own forward -- the only thing the fused gather + MLP kernels compute.  Anything else takes the generic path."""
    from ..triplane import OSGDecoder
    net = getattr(decoder, 'net', None)
    if not isinstance(decoder, OSGDecoder) or type(decoder).forward is not OSGDecoder.forward or net is None or len(net) != 3:
        return False
    l0, l2 = net[0], net[2]
    return (isinstance(net[1], torch.nn.Softplus) and all(hasattr(l, a) for l in (l0, l2) for a in ('weight', 'bias', 'weight_gain', 'bias_gain'))
            and tuple(l0.weight.shape) == (64, 32) and tuple(l2.weight.shape) == (33, 64) and l0.bias is not None and l2.bias is not None
            and getattr(l0, 'activation', 'linear') == 'linear' and getattr(l2, 'activation', 'linear') == 'linear')


def generate_planes():
    """The three plane-axis triples (renderer.py:23-37).  The kernels sample plane 0 at (x, y), plane 1 at (x, z), plane 2 at (z, x) --
    what these axes (through project_onto_planes' inverse, :39-53) amount to."""
    return torch.tensor([[[1, 0, 0], [0, 1, 0], [0, 0, 1]],
                         [[1, 0, 0], [0, 0, 1], [0, 1, 0]],
                         [[0, 0, 1], [1, 0, 0], [0, 1, 0]]], dtype=torch.float32)


class _SamplePlanes(torch.autograd.Function):
    @staticmethod
    def forward(ctx, planes, coords, box_warp):
        n, _, c, h, w = planes.shape
        planes_nhwc = planes_to_nhwc(planes.detach())
        coords = coords.detach().contiguous().float()
        p = coords.shape[1]
        out = torch.empty(n, 3, p, c, device=planes.device, dtype=torch.float32)
        hip.call('spi_sample_from_planes_fwd', hip.ptr(planes_nhwc), hip.ptr(coords), n, p, h, w, float(box_warp), hip.ptr(out), hip.stream())
        ctx.save_for_backward(coords)
        ctx.meta = (n, p, c, h, w, float(box_warp))
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        coords, = ctx.saved_tensors
        n, p, c, h, w, box_warp = ctx.meta
        d_planes = zero_arena.zeros((n, 3, h, w, c), coords.device)
        d_out = d_out.contiguous().float()                     # bound to a name: a temporary would be freed before the launch is enqueued
        hip.call('spi_sample_from_planes_bwd', hip.ptr(d_out), hip.ptr(coords), n, p, h, w, box_warp, hip.ptr(d_planes), hip.stream())
        return planes_to_nchw(d_planes), None, None


def sample_from_planes(plane_axes, plane_features, coordinates, mode='bilinear', padding_mode='zeros', box_warp=None):
    """renderer.py:55-65: [N,3,C,H,W] planes sampled at [N,P,3] coordinates -> [N,3,P,C] (bilinear, zeros padding, align_corners False).
    One HIP kernel per direction instead of project_onto_planes + three grid_samples; C = 32 and the default plane axes only."""
    assert padding_mode == 'zeros' and mode == 'bilinear'
    if plane_axes is not None and not torch.equal(plane_axes.detach().cpu().float(), generate_planes()):
        raise NotImplementedError('sample_from_planes: only the EG3D plane axes of generate_planes() are built into the kernels')
    if plane_features.shape[1] != 3 or plane_features.shape[2] != 32:
        raise NotImplementedError(f'sample_from_planes: planes must be [N,3,32,H,W], got {tuple(plane_features.shape)}')
    return _SamplePlanes.apply(plane_features, coordinates, float(box_warp))


class ImportanceRenderer(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.ray_marcher = MipRayMarcher2()
        self.plane_axes = generate_planes()
        self.last_aux = None

    @staticmethod
    def _check(opts):
        if opts.get('clamp_mode', 'softplus') != 'softplus':
            raise AssertionError('MipRayMarcher only supports `clamp_mode`=`softplus`!')

    # ---- sampling (renderer.py:169-253) ------------------------------------------------------------------------------------------
    def sample_stratified(self, ray_origins, ray_start, ray_end, depth_resolution, disparity_space_sampling=False, xi=None):
        """renderer.py:169-192 -> [N,M,S,1]: scalar limits, per-ray tensor limits, or uniform in disparity.  xi: the rand_like draw."""
        n, m, _ = ray_origins.shape
        dev = ray_origins.device
        s = int(depth_resolution)
        if disparity_space_sampling:
            t = torch.linspace(0, 1, s, device=dev).reshape(1, 1, s, 1).repeat(n, m, 1, 1)
            t = t + (torch.rand_like(t) if xi is None else xi.to(dev).reshape(n, m, s, 1)) * (1 / (s - 1))
            return 1. / (1. / ray_start * (1. - t) + 1. / ray_end * t)
        if torch.is_tensor(ray_start):
            from . import math_utils
            d = math_utils.linspace(ray_start, ray_end, s).permute(1, 2, 0, 3)
            delta = (ray_end - ray_start) / (s - 1)
            return d + (torch.rand_like(d) if xi is None else xi.to(dev).reshape(n, m, s, 1)) * delta[..., None]
        d = torch.linspace(ray_start, ray_end, s, device=dev).reshape(1, 1, s, 1).repeat(n, m, 1, 1)
        return d + (torch.rand_like(d) if xi is None else xi.to(dev).reshape(n, m, s, 1)) * ((ray_end - ray_start) / (s - 1))

    def sample_importance(self, z_vals, weights, N_importance, u=None, return_order=False):
        """renderer.py:194-215 on the HIP kernel: [N,M,S,1] depths + [N,M,S-1,1] weights -> [N,M,N_importance,1] fine depths (no grad;
        emitted ascending per ray -- the same multiset as the reference's unsorted draws).  u: the torch.rand [N*M, N_importance] draw.
        return_order: also the index [N,M,N_importance] of each emitted sample in the reference's draw order."""
        with torch.no_grad():
            n, m, s, _ = z_vals.shape
            dev = z_vals.device
            d_c = z_vals.detach().reshape(n, m, s).contiguous().float()
            w_c = weights.detach().reshape(n, m, s - 1).contiguous().float()
            u = (torch.rand(n * m, N_importance, device=dev) if u is None else u.to(dev)).reshape(n, m, N_importance).contiguous().float()
            d_f = torch.empty(n, m, N_importance, device=dev, dtype=torch.float32)
            hip.call('spi_importance_sample', hip.ptr(d_c), hip.ptr(w_c), hip.ptr(u), n * m, s, N_importance, hip.ptr(d_f), 1, hip.stream())
            if return_order:
                raw = torch.empty_like(d_f)
                hip.call('spi_importance_sample', hip.ptr(d_c), hip.ptr(w_c), hip.ptr(u), n * m, s, N_importance, hip.ptr(raw), 0, hip.stream())
                return d_f.unsqueeze(-1), torch.argsort(raw, dim=-1, stable=True)
        return d_f.unsqueeze(-1)

    def unify_samples(self, depths1, colors1, densities1, depths2, colors2, densities2):
        """renderer.py:157-167 (generic-decoder path only; the fused path folds this into a permutation the march reads through)."""
        all_depths = torch.cat([depths1, depths2], dim=-2)
        all_colors = torch.cat([colors1, colors2], dim=-2)
        all_densities = torch.cat([densities1, densities2], dim=-2)
        _, idx = torch.sort(all_depths, dim=-2)
        return (torch.gather(all_depths, -2, idx), torch.gather(all_colors, -2, idx.expand(-1, -1, -1, all_colors.shape[-1])),
                torch.gather(all_densities, -2, idx))

    # ---- forward ------------------------------------------------------------------------------------------------------------------
    def forward(self, planes, decoder, ray_origins, ray_directions, rendering_options, noise=None, depth_only=False):
        """depth_only (extension): returns (None, depth, weight sums) -- the colour half of the decoder and of the composite is skipped.
        noise (extension): (xi, u) or (xi, u, eps_coarse, eps_fine) -- the reference's draws in its order, injected."""
        self._check(rendering_options)
        opts = rendering_options
        n, m, _ = ray_origins.shape
        dev = planes.device
        sc, sf = int(opts['depth_resolution']), int(opts['depth_resolution_importance'])
        noise = tuple(noise) if noise is not None else ()
        xi = noise[0].to(dev) if len(noise) > 0 and noise[0] is not None else None
        u = noise[1].to(dev) if len(noise) > 1 and noise[1] is not None else None
        eps = (noise[2] if len(noise) > 2 else None, noise[3] if len(noise) > 3 else None)
        dnoise = float(opts.get('density_noise', 0) or 0)
        auto = isinstance(opts['ray_start'], str) or isinstance(opts['ray_end'], str)
        if auto and not (opts['ray_start'] == opts['ray_end'] == 'auto'):
            raise ValueError("ray_start / ray_end: numbers, or both 'auto' (renderer.py:91)")
        disparity = bool(opts.get('disparity_space_sampling', False))
        fused = _is_osg_decoder(decoder)
        coarse = None
        if auto or disparity or not fused:
            # draws in the reference's order: the coarse jitter first
            if xi is None:
                xi = torch.rand(n, m, sc, 1, device=dev)
            if auto:
                from . import math_utils
                ray_start, ray_end = math_utils.get_ray_limits_box(ray_origins, ray_directions, box_side_length=opts['box_warp'])
                ok = ray_end > ray_start
                if torch.any(ok).item():
                    ray_start = torch.where(ok, ray_start, ray_start[ok].min())
                    ray_end = torch.where(ok, ray_end, ray_start[ok].max())      # (sic, renderer.py:96: the largest valid START)
                coarse = self.sample_stratified(ray_origins, ray_start, ray_end, sc, disparity, xi=xi)
            else:
                coarse = self.sample_stratified(ray_origins, opts['ray_start'], opts['ray_end'], sc, disparity, xi=xi)
        if not fused:
            if depth_only:
                raise NotImplementedError('depth_only rendering needs the OSG decoder (fused kernels)')
            return self._forward_generic(planes, decoder, ray_origins, ray_directions, opts, coarse, u, eps)
        if xi is None:
            xi = torch.rand(n, m, sc, 1, device=dev)
        if dnoise > 0 and eps[0] is None:                        # keep the reference's draw order: rand_like, randn_like, rand, randn_like
            eps = (torch.randn(n, m * sc, 1, device=dev), None)
        if u is None:
            u = torch.rand(n * m, max(sf, 1), device=dev)
        if dnoise > 0 and eps[1] is None and sf > 0:
            eps = (eps[0], torch.randn(n, m * sf, 1, device=dev))
        params, gains = _decoder_params(decoder)
        ro = dict(opts, depth_only=bool(depth_only), coarse_depths=coarse, density_eps=eps)
        if auto:
            ro['ray_start'] = ro['ray_end'] = 0.0                # unused beside coarse_depths
        rgb, depth, wsum = _Render.apply(planes, *params, gains, ray_origins, ray_directions, xi, u, ro)
        return rgb, depth, wsum

    def _forward_generic(self, planes, decoder, ray_origins, ray_directions, opts, depths_coarse, u, eps):
        """The reference's composition (renderer.py:103-140) for an arbitrary decoder callable: sample_from_planes (HIP) -> decoder (autograd)
        -> MipRayMarcher2 (HIP) -> sample_importance (HIP) -> second pass -> unify_samples -> MipRayMarcher2."""
        n, m, sc, _ = depths_coarse.shape
        coords = (ray_origins.unsqueeze(-2) + depths_coarse * ray_directions.unsqueeze(-2)).reshape(n, -1, 3)
        dirs = ray_directions.unsqueeze(-2).expand(-1, -1, sc, -1).reshape(n, -1, 3)
        out = self.run_model(planes, decoder, coords, dirs, opts, _eps=eps[0])
        colors_coarse = out['rgb'].reshape(n, m, sc, out['rgb'].shape[-1])
        dens_coarse = out['sigma'].reshape(n, m, sc, 1)
        sf = int(opts['depth_resolution_importance'])
        if sf > 0:
            _, _, weights = self.ray_marcher(colors_coarse, dens_coarse, depths_coarse, opts)
            depths_fine, order = self.sample_importance(depths_coarse, weights, sf, u=u, return_order=True)
            dirs = ray_directions.unsqueeze(-2).expand(-1, -1, sf, -1).reshape(n, -1, 3)
            coords = (ray_origins.unsqueeze(-2) + depths_fine * ray_directions.unsqueeze(-2)).reshape(n, -1, 3)
            e1 = eps[1]
            if e1 is not None:                                   # injected draw: indexed like the reference's unsorted fine samples (see _Render.forward)
                e1 = torch.gather(e1.to(planes.device).reshape(n, m, sf), -1, order)
            out = self.run_model(planes, decoder, coords, dirs, opts, _eps=e1)
            colors_fine = out['rgb'].reshape(n, m, sf, out['rgb'].shape[-1])
            dens_fine = out['sigma'].reshape(n, m, sf, 1)
            all_d, all_c, all_s = self.unify_samples(depths_coarse, colors_coarse, dens_coarse, depths_fine, colors_fine, dens_fine)
            rgb, depth, weights = self.ray_marcher(all_c, all_s, all_d, opts)
        else:
            rgb, depth, weights = self.ray_marcher(colors_coarse, dens_coarse, depths_coarse, opts)
        return rgb, depth, weights.sum(2)

    def run_model(self, planes, decoder, sample_coordinates, sample_directions, options, _eps=None):
        self._check(options)
        if _is_osg_decoder(decoder):
            params, gains = _decoder_params(decoder)
            rgb, sigma = _RunModel.apply(planes, *params, gains, sample_coordinates, float(options['box_warp']))
            out = {'rgb': rgb, 'sigma': sigma}
        else:
            feats = sample_from_planes(self.plane_axes, planes, sample_coordinates, padding_mode='zeros', box_warp=options['box_warp'])
            out = dict(decoder(feats, sample_directions))
        if options.get('density_noise', 0) > 0:                  # renderer.py:146-147
            e = torch.randn_like(out['sigma']) if _eps is None else _eps.to(out['sigma'].device).reshape(out['sigma'].shape)
            out['sigma'] = out['sigma'] + e * options['density_noise']
        return out
