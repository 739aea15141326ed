"""SPI stage 2: generator tuning with rotated / mirrored pseudo-views (mirror of
spi/training/coaches/rot_bbox_cx_coach.py:17-171).

Per iteration (``train_step``), exactly as the reference's loop body :68-157:
  main view     G.synthesis(w, c) -> L2 + LPIPS -> backward                                (:69-85)
  every 4th     rot:        4 surrounding cameras -> warp the input into them (HIP ``rotate``) -> LPIPS * 0.1 * 4   (:87-105)
                mirror-rot: same around the mirror camera with the flipped input -> BoxCX * 0.05 * 4              (:107-131)
                depth:      4 random cameras, depth of G vs depth of the frozen original G -> L2 * 1               (:133-141)
  early stop if LPIPS <= 0.05, else ONE optimiser step over the accumulated gradients               (:148-151)
Result-identical savings: target LPIPS features are cached; the depth branch skips the
super-resolution network (only ``image_depth`` is consumed, :136-138); w_pivot is detached; the
4-view branches pass ONE w with 4 cameras, so the camera-independent StyleGAN2 backbone (46 % of the
generator's conv FLOPs) and every weight modulation run once instead of on 4 identical rows.
"""
import contextlib
import os
import torch

from ...configs import paths_config, hyperparameters, global_config
from ...criteria.l2_loss import l2_loss
from ...criteria.bbox_cx_loss import BoxCXLoss
from ...utils.rotate import rotate
from ...utils.mask_utils import calculate_face_mask
from ...utils.camera_utils import cal_mirror_c, cal_camera_weight, sample_surrounding_camera, sample_camera, cal_camera_gauss_weight
from ...utils.rng import DeviceRNG
from ...torch_utils.ops.conv2d_mfma import sparse_gradients
from ...torch_utils.misc import trace_range, capture_graph
from ...torch_utils import zero_arena
from .base_coach import BaseCoach


class RotBboxCoach(BaseCoach):
    def __init__(self, data_loader, use_wandb, box_cx_loss=None, **kw):
        super().__init__(data_loader, use_wandb, **kw)
        self.coach_name = 'RotBboxCoach'
        self.build_name()
        if box_cx_loss is None:
            from ...criteria import weights as pretrained
            box_cx_loss = BoxCXLoss(weights=pretrained.vgg19_head_weights(self.synthetic))
        self.box_cx_loss = box_cx_loss.to(self.device).eval()
        self.rot_bs = 4

    def prepare_image(self, data):
        """Per-image constants of the loop (:34-43,58-66)."""
        dev = self.device
        image = data['img'].to(dev).float()
        camera = torch.as_tensor(data['c']).to(dev).float().reshape(-1, 25)
        mask = data['mask'].to(dev)[:, 0] if data['mask'].ndim == 5 else data['mask'].to(dev)
        mask = mask.reshape(1, 1, *mask.shape[-2:])
        ctx = dict(image=image, camera=camera, image_m=torch.flip(image, dims=[3]), camera_m=cal_mirror_c(camera=camera),
                   fg_mask=1 - (mask == 0).float(), face_mask=calculate_face_mask(mask).float(), lm=data['lm'].to(dev).float().reshape(1, 68, 2))
        ctx['face_mask_m'] = torch.flip(ctx['face_mask'], dims=[3])
        # inverse source extrinsics of the two warps (rotate): constant per image, and torch.inverse cannot run inside a HIP-graph capture
        for k in ('camera', 'camera_m'):
            ctx[k + '_inv'] = torch.inverse(ctx[k].repeat(self.rot_bs, 1)[:, :16].reshape(-1, 4, 4)).reshape(-1, 16).contiguous()
        ctx['weight_m'] = float(cal_camera_weight(camera)[0])
        ctx['yaw_range'] = float(cal_camera_gauss_weight(camera)[0]) if hyperparameters.use_adapt_yaw_range else 0.2
        ctx['target_feats'] = self.lpips_loss.features(image)
        ctx['box_plan'] = self.box_cx_loss.plan(ctx['lm'].repeat(self.rot_bs, 1, 1), dev)      # host-side RoI geometry, once per image
        return self._adopt_ctx(ctx)

    # ---- persistent per-coach input buffers (round 5) --------------------------------------------------------------------------------
    # A captured iteration bakes in the ADDRESSES of everything it reads.  The per-image constants above are therefore kept in ONE set of
    # tensors per coach: the next image's values are copied into them (same shapes: every image is 512^2 with 68 landmarks), its pivot into
    # the pivot buffer, the frozen generator's tri-planes for that pivot into the tensor the depth branch cached -- and image k >= 2 replays
    # image 1's two stage-2 graphs from its first iteration on (reference: the loop is per image, base_coach.py:53-60; until round 4 every
    # image paid two eager warm-up iterations + two captures of ~1000 / ~3500 nodes).  What is not a tensor but shapes the captured launch
    # sequence (is the mirror branch on: `weight_m > 0`; the yaw range, a host float folded into launch arguments) must match, otherwise
    # the image gets buffers -- and graphs -- of its own.
    def _adopt_ctx(self, new):
        old = getattr(self, '_ctx_persist', None)
        frozen_planes = self.original_G._last_planes             # the tensor the captured depth branch reads (None before the first branch iteration)
        self.original_G._last_planes = None                      # per-image backbone cache of the frozen generator (depth branch)

        def same_layout(a, b):
            if torch.is_tensor(a):
                return torch.is_tensor(b) and a.shape == b.shape and a.dtype == b.dtype and a.device == b.device
            if isinstance(a, (list, tuple)):
                return isinstance(b, (list, tuple)) and len(a) == len(b) and all(same_layout(x, y) for x, y in zip(a, b))
            return True
        reuse = (global_config.reuse_graphs_across_images and old is not None and set(k for k in old if k not in self._CTX_RUNTIME_KEYS) == set(new)
                 and all(same_layout(old[k], new[k]) for k in new)
                 and (old['weight_m'] > 0) == (new['weight_m'] > 0) and old['yaw_range'] == new['yaw_range'])
        if not reuse:
            new['generation'] = self._next_generation()          # identity of this image's constants for the graph cache (ids / pointers get recycled)
            self._ctx_persist = new
            self._frozen_planes_buf = None
            return new
        with torch.no_grad():
            for k, v in new.items():
                if torch.is_tensor(v):
                    old[k].copy_(v)
                elif isinstance(v, (list, tuple)):
                    for a, b in zip(old[k], v):
                        a.copy_(b)
                else:
                    old[k] = v
        old['stable_planes_cached'] = False
        self._frozen_planes_buf = frozen_planes if frozen_planes is not None else getattr(self, '_frozen_planes_buf', None)
        old['images_adopted'] = old.get('images_adopted', 1) + 1
        return old

    _CTX_RUNTIME_KEYS = ('generation', 'pivot_ptr', 'pivot_generation', 'stable_planes_cached', 'images_adopted')

    def _bind_pivot(self, ctx, w_pivot):
        """The pivot latent of this image in the coach's persistent pivot buffer (+ the frozen generator's tri-planes for it, in the tensor a
        captured depth branch reads).  -> the tensor the loop iterates with."""
        if not (global_config.reuse_graphs_across_images and ctx is getattr(self, '_ctx_persist', None)):
            return w_pivot
        buf = getattr(self, '_pivot_buf', None)
        if buf is None or buf.shape != w_pivot.shape or buf.device != w_pivot.device or ctx.get('images_adopted', 1) == 1:
            self._pivot_buf = buf = w_pivot.detach().clone()
        else:
            with torch.no_grad():
                buf.copy_(w_pivot.detach())
        self._refresh_frozen_planes(ctx, buf)
        return buf

    def _refresh_frozen_planes(self, ctx, w_pivot):
        """The frozen generator's tri-planes for `w_pivot`, written into the tensor a captured depth branch reads (only when the ctx says they are
        stale: `_adopt_ctx` clears `stable_planes_cached` for every new image)."""
        planes = getattr(self, '_frozen_planes_buf', None)
        if planes is not None and not ctx.get('stable_planes_cached', False) and hyperparameters.pt_depth_lambda > 0:
            with torch.no_grad():
                fresh = self.original_G._planes(w_pivot, noise_mode='const')
                if fresh.shape == planes.shape:
                    planes.copy_(fresh)
                    self.original_G._last_planes = planes
                    ctx['stable_planes_cached'] = True

    def _next_generation(self):
        self._generation = getattr(self, '_generation', 0) + 1
        return self._generation

    def reset_pipeline(self):
        """Start of an image's loop: no early stop is pending, the sticky stop byte is clear (a stop belongs to ONE image's loop)."""
        self._late_stop = None
        g2 = getattr(self, '_g2', None)
        if g2:
            g2['pending'].clear()
            g2['stop'].zero_()

    def _side_streams(self):
        if getattr(self, '_streams', None) is None:
            self._streams = [torch.cuda.Stream(device=self.device) for _ in range(3)]
        return self._streams

    def _trainable_params(self):
        """(backbone parameters, all other trainable parameters) of G"""
        key = id(self.G)
        if getattr(self, '_params_of', None) != key:
            named = [(k, p) for k, p in self.G.named_parameters() if p.requires_grad]
            self._params_of = key
            self._params = ([p for k, p in named if k.startswith('backbone.')], [p for k, p in named if not k.startswith('backbone.')])
        return self._params

    def _synth(self, G, ws, cams, rng, **kw):
        n, m = cams.shape[0], G.neural_rendering_resolution ** 2
        rk = G.rendering_kwargs
        noise = (rng.rand(n, m, int(rk['depth_resolution']), 1), rng.rand(n * m, max(int(rk['depth_resolution_importance']), 1)))
        return G.synthesis(ws, cams, noise_mode='const', render_noise=noise, **kw)

    # ---- HIP-graph replay (MI355X-first; the eager iteration below is the definition) --------------------------------------------
    # An iteration is ~1000 launches (every 4th: ~3500) whose sequence depends on nothing the GPU computes -- the data-driven skipping
    # happens inside the kernels -- so after one eager iteration of its kind (plain / with the pseudo-view branches) the next one is
    # CAPTURED (forward, losses, every backward, gradient folding) and later ones are a single graph launch: ~25 ms of enqueue work per
    # iteration leave the host, which otherwise cannot keep the GPU fed through the backward passes (the eager loop idles the GPU for
    # ~13 % of a plain iteration, tools/trace_gaps.py).
    # The early-stop test (`loss_lpips <= threshold: break` BEFORE optimizer.step(), :148-151) is the loop's one host decision.  Reading it
    # after every replay would drain the GPU before each launch; instead the graph ORs the comparison into a sticky device byte, the Adam
    # launch that follows is PREDICATED on that byte (spi_adam_multi_pred: once it is set, no step changes anything any more), and the
    # host reads the byte of iteration i - GRAPH_LAG when it launches iteration i.  Parameters, moments and the reported iteration count
    # end exactly where the reference's break leaves them; what differs is up to GRAPH_LAG speculative iterations of discarded GPU work
    # after a stop (and as many extra draws from the renderer's random stream).  Graphs are keyed to the image's tensors, the pivot and
    # the generator / optimiser instances and dropped when any of them changes.
    GRAPH_WARMUP = int(os.environ.get('SPI_GRAPH_WARMUP', '1'))          # eager iterations of a kind before its capture
    captures_total = 0                                                   # stage-2 graph captures of this process (all coaches)
    GRAPH_LAG = max(1, int(os.environ.get('SPI_STAGE2_GRAPH_LAG', '2')))  # iterations the host may run ahead of the early-stop byte it has read

    def _graph_ok(self, rng):
        from ..projectors.common import graph_policy              # the same answer with and without a process group (multi-GPU = measured path)
        return (graph_policy(global_config.stage2_hip_graph) and isinstance(rng, DeviceRNG) and torch.device(self.device).type == 'cuda'
                and not global_config.concurrent_branches and not getattr(self, '_graph_failed', False))

    def _resolve_pending(self, keep):
        """Wait for the early-stop bytes of all but the newest `keep` replays in flight.  -> (iteration, losses) of the FIRST one that had it set."""
        pend = self._g2.get('pending') if getattr(self, '_g2', None) else None
        while pend and len(pend) > keep:
            it, ev, host, losses, steps_before = pend.popleft()
            ev.synchronize()
            if bool(host[0]):
                pend.clear()                                     # everything launched after it was speculative:
                self.optimizer.step_count = steps_before         # their predicated Adam launches changed nothing and do not count as steps
                return it, losses
        return None

    def drain_pipeline(self):
        """After the loop: the early stop that pipelined replays found late (or still have pending).  -> (iteration index, its losses) or None.
        The caller counts its iterations like the reference's loop: the stop belongs to THAT iteration."""
        late = getattr(self, '_late_stop', None)
        self._late_stop = None
        if late is None and getattr(self, '_g2', None):
            late = self._resolve_pending(0)
        return late

    def _graph_train_step(self, i, ctx, w_pivot, rng):
        import collections
        if 'generation' not in ctx:                             # a hand-built ctx (tests): give it an identity once
            ctx['generation'] = self._next_generation()
        if ctx.get('pivot_ptr') != w_pivot.data_ptr():          # a new pivot for the same image constants is a new set of baked-in addresses
            ctx['pivot_ptr'], ctx['pivot_generation'] = w_pivot.data_ptr(), self._next_generation()
        key = (ctx['generation'], ctx['pivot_generation'], id(self.G), id(self.optimizer), float(hyperparameters.LPIPS_value_threshold),    # (the threshold is baked in,
               global_config.conv_precision, global_config.conv_winograd, global_config.conv_winograd_f4, global_config.enable_fp16_blocks, global_config.exploit_sparsity)   # and so is the arithmetic)
        if getattr(self, '_g2_key', None) != key:
            self._g2_key = key
            self._g2 = dict(stop=torch.zeros(1, device=self.device, dtype=torch.uint8), pending=collections.deque(),
                            host=[torch.zeros(1, dtype=torch.uint8).pin_memory() for _ in range(self.GRAPH_LAG + 1)])
            self._late_stop = None
        if self._late_stop is not None:                         # the loop was told to stop and called again: nothing runs, nothing is counted
            return True, self._late_stop[1]
        st = self._g2.setdefault('branch' if i % self.rot_bs == 0 else 'plain', dict(eager=0, graph=None))
        if st['graph'] is None:
            late = self._resolve_pending(0)                      # eager iterations and captures start from a drained pipeline
            if late is not None:
                self._late_stop = late
                return True, late[1]
            if st['eager'] < self.GRAPH_WARMUP:                  # the first iterations of a kind run eagerly (lazy initialisations, allocator
                st['eager'] += 1                                 # warm-up, the frozen generator's tri-plane cache of the depth branch)
                return self._eager_train_step(i, ctx, w_pivot, rng)
            try:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                from ..projectors.common import capture_mode
                with capture_graph(g, capture_error_mode=capture_mode()):
                    _, losses = self._forward_backward(i, ctx, w_pivot, rng, flag_buf=self._g2['stop'])
            except Exception as e:                               # noqa: BLE001  (capture is an optimisation: the eager iteration is always valid)
                import sys
                self._graph_failed = True
                torch.cuda.synchronize()
                print(f'[spi_amd] stage-2 HIP-graph capture failed ({type(e).__name__}: {e}); continuing with eager iterations', file=sys.stderr)
                import traceback
                print(''.join(traceback.format_tb(e.__traceback__)[-6:]), file=sys.stderr)
                return self._eager_train_step(i, ctx, w_pivot, rng)
            st.update(graph=g, losses=losses)
            RotBboxCoach.captures_total += 1                     # (tests / bench: image k >= 2 must not capture again)
        late = self._resolve_pending(self.GRAPH_LAG - 1)         # the byte of iteration i - GRAPH_LAG, before iteration i is launched
        if late is not None:
            self._late_stop = late
            return True, late[1]
        steps_before = self.optimizer.step_count
        # A caller that drives train_step itself after a second prepare_image (bench.py, tools) never went through optimise_image's
        # _bind_pivot: the persistent ctx then says the frozen tri-planes are stale while the captured depth branch would still read the
        # previous image's.  Refresh them here, before the first replay that needs them.
        if ctx is getattr(self, '_ctx_persist', None) and not ctx.get('stable_planes_cached', False) and i % self.rot_bs == 0:
            self._refresh_frozen_planes(ctx, w_pivot.detach())
        st['graph'].replay()
        losses = {k: v.clone() for k, v in st['losses'].items()}  # the graph's outputs are overwritten by the next replay
        self.optimizer.step(skip=self._g2['stop'])               # predicated on the device: a stop of THIS iteration (or an earlier one) freezes it
        host = self._g2['host'][i % (self.GRAPH_LAG + 1)]
        host.copy_(self._g2['stop'], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._g2['pending'].append((i, ev, host, losses, steps_before))
        return False, losses

    def train_step(self, i, ctx, w_pivot, rng=None):
        """One iteration of the stage-2 loop.  Returns (stop, losses dict of 0-dim device tensors)."""
        rng = rng or self.rng or DeviceRNG(self.device)
        if self._graph_ok(rng):
            return self._graph_train_step(i, ctx, w_pivot, rng)
        return self._eager_train_step(i, ctx, w_pivot, rng)

    def _eager_train_step(self, i, ctx, w_pivot, rng):
        stop_flag, losses = self._forward_backward(i, ctx, w_pivot, rng, flag_buf=None)
        # the loop's one host read (:148).  The flag was copied to pinned memory right after the main forward, so the
        # wait ends when the GPU has passed THAT point (not the whole iteration): the backward passes keep the GPU busy
        # while the host enqueues the optimiser step and the next forward.
        if stop_flag is not None and stop_flag():
            return True, losses
        self.optimizer.step()
        return False, losses

    @zero_arena.closes_iteration
    def _forward_backward(self, i, ctx, w_pivot, rng, flag_buf):
        """Everything of an iteration up to the early-stop test: forward passes, losses, backward passes, gradients folded into .grad.
        flag_buf None: returns (callable waiting for `lpips <= threshold` on the host, losses); else the flag goes into that device byte."""
        hp = hyperparameters
        G, rot_bs = self.G, self.rot_bs
        ws = w_pivot.detach()
        # one cleared buffer for every accumulator of the iteration (weight / style / bias gradients, split-K outputs, the plane gradient);
        # the pseudo-view iteration needs four times the plain one's, so the two kinds keep their own sizes
        zero_arena.begin(ws.device, key=('stage2', i % rot_bs == 0))
        self.optimizer.zero_grad()
        # ONE backbone pass per iteration.  The reference re-runs G.synthesis -- backbone included -- for the main view and for each
        # pseudo-view branch (:60,92,112,133) and back-propagates each loss through it; w and G do not change inside an iteration,
        # so the tri-planes are the same tensor every time.  They are computed once, every view renders from them (EG3D's own
        # use_cached_backbone switch, triplane.py:64-70), each loss is back-propagated down to the planes, and the summed plane
        # gradient goes through the backbone once at the end: the same gradient (the backbone backward is linear in d planes),
        # 3 backbone forward + 3 backward passes fewer on every 4th iteration.
        bb_params, params = self._trainable_params()
        planes = G._planes(ws, noise_mode='const')
        leaf = planes.detach().requires_grad_(True)
        G._last_planes = leaf
        gen = self._synth(G, ws, ctx['camera'], rng, use_cached_backbone=True)
        losses = {}
        loss = 0.0
        if hp.pt_l2_lambda > 0:
            losses['l2'] = l2_loss(gen['image'], ctx['image'])
            loss = loss + losses['l2'] * hp.pt_l2_lambda
        if hp.pt_lpips_lambda > 0:
            losses['lpips'] = torch.squeeze(self.lpips_loss(gen['image'], y_feats=ctx['target_feats']))
            loss = loss + losses['lpips'] * hp.pt_lpips_lambda
        stop_flag = None
        if 'lpips' in losses:
            if flag_buf is None:
                stop_flag = self._async_flag(losses['lpips'] <= hp.LPIPS_value_threshold)
            else:
                flag_buf.bitwise_or_((losses['lpips'].detach() <= hp.LPIPS_value_threshold).reshape(1).to(torch.uint8))      # sticky
        # The reference calls backward() once per loss (:69,85,105,131): every parameter's .grad is read-modify-written once per
        # call (~150 tiny add_ launches each).  Here each call returns its gradients as fresh tensors (autograd.grad) and they are
        # folded into .grad in the reference's order with one multi-tensor add per call: the same sums.
        pending = []
        d_planes = []

        def branch_backward(branch_loss, sparse):
            with sparse_gradients(sparse and global_config.exploit_sparsity), trace_range('stage2/backward'):    # sparse: d(image) is exactly zero outside the warp mask
                g = torch.autograd.grad(branch_loss, [leaf] + params, allow_unused=True)
            d_planes.append(g[0])
            pending.append(g[1:])
        # The three pseudo-view branches depend on the main FORWARD only (depth_main) and on nothing of each other: each runs on its
        # own HIP stream beside the main backward, so their launch-bound stretches (4^2..64^2 backbone layers, tiny elementwise
        # kernels) fill the CUs the other chains leave idle.  Gradients meet again in the ordered multi-tensor adds below.
        side = self._side_streams() if (i % rot_bs == 0 and global_config.concurrent_branches and gen['image'].is_cuda) else None
        if side:
            main_stream = torch.cuda.current_stream()
            fwd_done = torch.cuda.Event()
            fwd_done.record(main_stream)                            # before the main backward is enqueued
        branch_backward(loss, False)
        if i % rot_bs == 0:
            depth_main = gen['image_depth'].detach()

            def on_stream(k):
                if not side:
                    return contextlib.nullcontext()
                side[k].wait_event(fwd_done)
                return torch.cuda.stream(side[k])

            def rot_branch():
                cams = sample_surrounding_camera(ctx['camera'], batch_size=rot_bs, yaw_range=ctx['yaw_range'], pitch_range=0.1,
                                                 rand=(rng.rand(rot_bs, 1), rng.rand(rot_bs, 1)))
                warp = {}

                def region(out):
                    # the warp needs the rendered depth only, so it runs between the renderer and the super-resolution network: the
                    # loss looks at image * warp_mask, and the SR convs skip the tiles no visible pixel depends on (triplane.synthesis)
                    warp['img'], warp['mask'] = rotate(target_camera=cams, target_depth=out['image_depth'], src_image=ctx['image'].repeat(rot_bs, 1, 1, 1),
                                                       src_camera=ctx['camera'].repeat(rot_bs, 1), src_depth=depth_main.repeat(rot_bs, 1, 1, 1),
                                                       src_mask=ctx['face_mask'].repeat(rot_bs, 1, 1, 1), EPS=5e-2, src_cam2world_inv=ctx['camera_inv'])
                    return warp['mask'] if global_config.exploit_sparsity else None
                gs = self._synth(G, ws, cams, rng, sr_region_fn=region, use_cached_backbone=True)
                losses['rot'] = self.lpips_loss(gs['image'] * warp['mask'], warp['img']) * hp.pt_rot_lambda * rot_bs
                branch_backward(losses['rot'], True)

            def mirror_branch():
                cams_m = sample_surrounding_camera(ctx['camera_m'], batch_size=rot_bs, yaw_range=ctx['yaw_range'], pitch_range=0.1,
                                                   rand=(rng.rand(rot_bs, 1), rng.rand(rot_bs, 1)))
                warp = {}

                def region(out):
                    warp['img'], warp['mask'] = rotate(target_camera=cams_m, target_depth=out['image_depth'], src_image=ctx['image_m'].repeat(rot_bs, 1, 1, 1),
                                                       src_camera=ctx['camera_m'].repeat(rot_bs, 1),
                                                       src_depth=torch.flip(depth_main, dims=[3]).repeat(rot_bs, 1, 1, 1),
                                                       src_mask=ctx['face_mask_m'].repeat(rot_bs, 1, 1, 1), EPS=5e-2, src_cam2world_inv=ctx['camera_m_inv'])
                    return warp['mask'] if global_config.exploit_sparsity else None
                gm = self._synth(G, ws, cams_m, rng, sr_region_fn=region, use_cached_backbone=True)
                flip_warp, flip_mask = torch.flip(warp['img'], dims=[3]), torch.flip(warp['mask'], dims=[3])
                losses['mirror_rot'] = self.box_cx_loss(torch.flip(gm['image'], dims=[3]) * flip_mask, flip_warp,
                                                        ctx['lm'].repeat(rot_bs, 1, 1), plan=ctx.get('box_plan')) * hp.pt_mirror_rot_lambda * rot_bs
                branch_backward(losses['mirror_rot'], True)

            def depth_branch():
                cams_d = sample_camera(batch_size=4, yaw_range=0.7, pitch_range=0.4, device=self.device, rand=(rng.rand(4, 1), rng.rand(4, 1)))
                sample_depth = self._synth(G, ws, cams_d, rng, depth_only=True, use_cached_backbone=True)['image_depth']
                with torch.no_grad():
                    # the frozen generator's tri-planes for this pivot never change: computed on the first use, then
                    # taken from EG3D's own backbone cache (triplane.py:64-70 cache_backbone / use_cached_backbone)
                    cached = ctx.get('stable_planes_cached', False)
                    stable_depth = self._synth(self.original_G, ws, cams_d, rng, depth_only=True, cache_backbone=not cached,
                                               use_cached_backbone=cached)['image_depth']
                    ctx['stable_planes_cached'] = True
                losses['depth'] = l2_loss(stable_depth, sample_depth) * hp.pt_depth_lambda
                branch_backward(losses['depth'], False)

            if hp.pt_rot_lambda > 0:
                with on_stream(0), trace_range('stage2/rot_branch'):
                    rot_branch()
            if hp.pt_mirror_rot_lambda > 0 and ctx['weight_m'] > 0:
                with on_stream(1), trace_range('stage2/mirror_rot_branch'):
                    mirror_branch()
            if hp.pt_depth_lambda > 0:
                with on_stream(2), trace_range('stage2/depth_branch'):
                    depth_branch()
            if side:
                for st in side:
                    main_stream.wait_stream(st)
            if hp.pt_tv_lambda > 0:
                from ...criteria.tv_loss import cal_tv_loss
                losses['tv'] = cal_tv_loss(ws, G) * hp.pt_tv_lambda
                losses['tv'].backward()
        G._last_planes = None
        dpl = d_planes[0]
        for g in d_planes[1:]:
            dpl = dpl + g
        with trace_range('stage2/backbone_backward'):
            pending.append(torch.autograd.grad(planes, bb_params, grad_outputs=dpl, allow_unused=True))
        for grads in pending:
            plist = bb_params if grads is pending[-1] else params
            have = [(p, g) for p, g in zip(plist, grads) if g is not None]
            both = [(p.grad, g) for p, g in have if p.grad is not None]
            if both:
                torch._foreach_add_([a for a, _ in both], [b for _, b in both])
            for p, g in have:
                if p.grad is None:
                    p.grad = g
        # detached: a caller that keeps the dict must not keep the iteration's autograd graph alive with it (with the previous iteration's
        # graph still referenced, ending a HIP-graph capture of the next one crashed inside the runtime)
        zero_arena.finish()
        return stop_flag, {k: v.detach() for k, v in losses.items()}

    def optimise_image(self, ctx, w_pivot, image_name='', rng=None):
        """The per-image loop (:57-157): up to G_1_step iterations, early stop on LPIPS.  -> (iterations counted like the reference's loop -- the one
        that stopped included --, losses of the last one)."""
        iters = completed = 0
        losses = {}
        log_images_counter = 0
        w_pivot = self._bind_pivot(ctx, w_pivot)
        self.reset_pipeline()
        from ...torch_utils.misc import quiet_gc
        with quiet_gc():
            for i in range(hyperparameters.G_1_step):
                stop, losses = self.train_step(i, ctx, w_pivot, rng=rng)
                iters += 1
                if stop:
                    break
                if self.use_wandb and log_images_counter % global_config.log_snapshot == 0:        # (:153-154)
                    self.log_image_from_w(w_pivot, ctx['camera'], self.G, f'{image_name}_G1_inv_{log_images_counter}')
                global_config.training_step += 1
                log_images_counter += 1
                completed = i + 1
            late = self.drain_pipeline()                         # pipelined graph replays report an early stop GRAPH_LAG iterations late
            if late is not None:                                 # count like the reference's loop: it broke in iteration late[0], before that step
                iters, losses = late[0] + 1, late[1]
                global_config.training_step -= completed - late[0]
        return iters, losses

    def train(self):
        paths_config.experiments_output_dir += f'{self.coach_name}'
        output_dir = paths_config.experiments_output_dir
        stats = []
        for idx, data in enumerate(self.data_loader):
            if self.image_counter >= hyperparameters.max_images_to_invert:
                break
            image_name = data['name'][0] if isinstance(data['name'], (list, tuple)) else data['name']
            import time
            t_begin = time.perf_counter()
            cap0 = RotBboxCoach.captures_total
            # SPI_TIME_IMAGE=1: synchronised timestamps around the per-image phases (a debugging aid: the syncs cost a little throughput)
            _tm = os.environ.get('SPI_TIME_IMAGE') == '1' and torch.cuda.is_available()
            def _mark(marks=[]):
                if _tm:
                    torch.cuda.synchronize(); marks.append(time.perf_counter())
                return marks
            _mark()[:] = []
            _mark()
            ctx = self.prepare_image(data)
            _mark()
            paths_config.experiments_output_dir = os.path.join(output_dir, image_name)
            os.makedirs(paths_config.experiments_output_dir, exist_ok=True)
            if self.use_wandb:                                   # (:45-47)
                self.log_target(ctx['image'], 'target_image')
                self.log_target(ctx['image_m'], 'mirror_image')
            self.restart_training()
            _mark()
            embedding_loaded = hyperparameters.load_embedding_coach_name is not None and os.path.isfile(
                f"{paths_config.embedding_base_dir}/{hyperparameters.load_embedding_coach_name}/{image_name}.pt")
            w_pivot = self.get_inversion(image_name, ctx['image'], ctx['camera'], fg_mask=ctx['fg_mask'])
            _mark()
            iters, losses = self.optimise_image(ctx, w_pivot, image_name)
            if _tm:
                m = _mark()
                print('[time] %s: prepare_image %.3f  restart_training %.3f  stage 1 (get_inversion) %.3f  stage 2 (optimise_image) %.3f s' %
                      (image_name, m[1] - m[0], m[2] - m[1], m[3] - m[2], m[4] - m[3]), flush=True)
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            t_loop = time.perf_counter()
            self.image_counter += 1
            self.finish_image(image_name, ctx['image'], ctx['camera'], w_pivot)
            # seconds_loop: per-image set-up + both optimisation loops (what bench.py's it/s is about); seconds_outputs: checkpoint, pictures, video
            st = dict(name=image_name, iters=iters, stage1_iters=0 if embedding_loaded else hyperparameters.first_inv_steps,
                      stage2_graph_captures=RotBboxCoach.captures_total - cap0)
            stats.append(dict(st, **{k: float(v) for k, v in losses.items()}) if iters else st)
            self.post_process(w_pivot, ctx['camera'], self.G, image_name)
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            stats[-1].update(seconds_loop=t_loop - t_begin, seconds_outputs=time.perf_counter() - t_loop)
        paths_config.experiments_output_dir = output_dir
        if self.use_wandb:                                        # (:170-171)
            self.log_metric()
        return stats
