"""Shared machinery of the three stage-1 projectors (W, W+, mirrored W+).

Restates the common body of spi/training/projectors/{w,w_plus,mirror}_projector.py:
  w statistics            mirror_projector.py:36-44   (600 z from RandomState(123) through the mapping net)
  optimiser set           :47-64                      (w_opt + the 13 ``noise_const`` maps, re-initialised to N(0,1))
  lr / noise schedule     :84-95
  noise regulariser       :107-115                    (roll-correlation pyramid down to 8x8, weight 1e5)
  noise renormalisation   :128-131
MI355X-first changes, all result-identical: one fused Adam launch per step; the (fixed) target's
LPIPS features are computed once instead of every step; the unused ``bg_loss`` / dilated-mask
computations (:68-74,117-118) are not performed; no per-step host sync.
"""
import copy
import numpy as np
import torch
import torch.nn.functional as F

from ... import hip
from ...configs import hyperparameters
from ...utils.rng import DeviceRNG
from ..optim import Adam
from ...torch_utils.misc import trace_range, capture_graph
from ...torch_utils import zero_arena
from .schedule import stage1_schedule


def w_statistics(G, c, w_avg_samples, device):
    z = np.random.RandomState(123).randn(w_avg_samples, G.z_dim)
    with torch.no_grad():
        w = G.mapping(torch.from_numpy(z).to(device), c.repeat(w_avg_samples, 1))
    w = w[:, :1, :].cpu().numpy().astype(np.float32)
    w_avg = np.mean(w, axis=0, keepdims=True)
    w_std = (np.sum((w - w_avg) ** 2) / w_avg_samples) ** 0.5
    return w_avg, float(w_std)


class _NoiseReg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, *bufs):
        loss = zero_arena.zeros(1, bufs[0].device)
        pyramid = torch.empty(plan.T * plan.region, device=loss.device, dtype=torch.float32)
        means = torch.empty(plan.T * 16, device=loss.device, dtype=torch.float32)
        hip.call('spi_noise_reg_fwd', hip.ptr(plan.ptrs), hip.ptr(plan.res), plan.T, plan.max_res, hip.ptr(pyramid), hip.ptr(means),
                 hip.ptr(loss), hip.stream())
        ctx.plan = plan
        ctx.save_for_backward(pyramid, means)
        return loss.reshape(())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        plan = ctx.plan
        pyramid, means = ctx.saved_tensors
        grads = torch.empty(plan.total, device=pyramid.device, dtype=torch.float32)
        gpyr = torch.empty_like(pyramid)
        g1 = gout.reshape(1).contiguous().float()
        hip.call('spi_noise_reg_bwd', hip.ptr(plan.ptrs), hip.ptr(plan.res), plan.T, plan.max_res, hip.ptr(pyramid), hip.ptr(means),
                 hip.ptr(g1), hip.ptr(grads), hip.ptr(plan.goff), hip.ptr(gpyr), hip.stream())
        return (None,) + tuple(g.view(r, r) for g, r in zip(torch.split(grads, plan.sizes), plan.res_list))


class NoiseRegulariser:
    """Multi-scale autocorrelation penalty on the StyleGAN2 noise buffers (mirror_projector.py:106-116) and their
    re-normalisation after the optimiser step (:127-131), each as ONE launch over all buffers (``spi_noise_reg_*``,
    ``spi_noise_renorm``) instead of the reference's ~1200 tiny autograd launches per step.  The pointer table is built once:
    the buffers are optimised in place, their addresses never change."""

    def __init__(self, noise_bufs):
        self.bufs = list(noise_bufs)
        dev = self.bufs[0].device
        assert all(b.ndim == 2 and b.shape[0] == b.shape[1] and b.is_contiguous() and b.dtype == torch.float32 for b in self.bufs)
        self.res_list = [int(b.shape[0]) for b in self.bufs]
        self.sizes = [r * r for r in self.res_list]
        self.T, self.max_res, self.total = len(self.bufs), max(self.res_list), sum(self.sizes)
        self.region = self.max_res * self.max_res // 2
        self.ptrs = torch.tensor([b.data_ptr() for b in self.bufs], dtype=torch.int64, device=dev)
        self.res = torch.tensor(self.res_list, dtype=torch.int32, device=dev)
        offs = [0]
        for n in self.sizes[:-1]:
            offs.append(offs[-1] + n)
        self.goff = torch.tensor(offs, dtype=torch.int64, device=dev)

    def _check(self):
        if any(b.data_ptr() != p for b, p in zip(self.bufs, self.ptrs_host())):
            raise RuntimeError('noise buffers were re-allocated after NoiseRegulariser was built')

    def ptrs_host(self):
        if not hasattr(self, '_ptrs_host'):
            self._ptrs_host = [b.data_ptr() for b in self.bufs]
        return self._ptrs_host

    def __call__(self):
        self._check()
        return _NoiseReg.apply(self, *self.bufs)

    def renorm(self):
        self._check()
        hip.call('spi_noise_renorm', hip.ptr(self.ptrs), hip.ptr(self.res), self.T, hip.stream())


def graph_policy(enabled):
    """Whether the stage-1 step is replayed from a HIP graph.  The same answer with and without an initialised process group: the ranks of a
    multi-GPU run (one process per GPU, spi_amd/dist.py) take exactly the code path the single-GPU number is measured on.  (Round 2 switched
    the graph off beside a process group because RCCL's watchdog thread issues HIP calls of its own, which a capture in GLOBAL error mode
    rejects; the capture now runs in thread-local mode there, see capture_mode().)
    Only in a process whose HIP runtime runs with its graph packet capture switched off (spi_amd/__init__.py: with it on, replays after
    ~10^3 eager launches return garbage on ROCm 7) -- checked, not inferred: `hip_graphs_safe()` runs the defect's reproducer once on the
    device -- otherwise the iterations are enqueued eagerly, with a note on stderr, once."""
    if not enabled:
        return False
    import spi_amd
    if not spi_amd.hip_graphs_safe():
        if not getattr(graph_policy, '_warned', False):
            graph_policy._warned = True
            import sys
            print('[spi_amd] HIP-graph replay is off in this process (%r): DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 was not in effect when the HIP runtime '
                  'started, or the replay self-test failed (import spi_amd before the first GPU call, or export the variable); iterations are '
                  'enqueued eagerly' % (spi_amd.hip_graphs_status(),), file=sys.stderr)
        return False
    return True


def capture_mode():
    """hipStreamCaptureMode for torch.cuda.graph: 'global' (any thread's unsafe HIP call invalidates the capture -- the strictest check, kept for
    single-process runs) unless a process group is up: its watchdog thread polls events with hipEventQuery while this thread captures, which
    is harmless (no collective is in flight during a stage-1 step: the path has no data-path collective) and allowed in 'thread_local' mode."""
    import torch.distributed as tdist
    return 'thread_local' if (tdist.is_available() and tdist.is_initialized()) else 'global'


class Projection:
    """State of one stage-1 optimisation; ``step(i)`` is the loop body of mirror_projector.py:81-131."""
    def __init__(self, G, cameras, dist_fn, *, w_mode, initial_w, num_steps, w_avg_samples, device, rng=None,
                 regularize_noise_weight=1e5, schedule_kwargs=None):
        self.rng = rng or DeviceRNG(device)
        self.G = G = copy.deepcopy(G).eval().requires_grad_(False).to(device).float()
        self.num_ws = G.backbone.mapping.num_ws
        self.cameras, self.dist_fn, self.w_mode, self.num_steps = cameras, dist_fn, w_mode, num_steps
        self.reg_weight, self.sched = regularize_noise_weight, (schedule_kwargs or {})
        w_avg, self.w_std = w_statistics(G, cameras[:1], w_avg_samples, device)
        self.noise_bufs = {name: buf for (name, buf) in G.backbone.synthesis.named_buffers() if 'noise_const' in name}
        start_w = initial_w if initial_w is not None else w_avg
        if w_mode == 'w+' and initial_w is None:
            start_w = np.repeat(start_w, self.num_ws, axis=1)
        self.w_opt = torch.tensor(start_w, dtype=torch.float32, device=device, requires_grad=True)
        for buf in self.noise_bufs.values():
            buf[:] = self.rng.randn(*buf.shape)
            buf.requires_grad = True
        self.optimizer = Adam([self.w_opt] + list(self.noise_bufs.values()), betas=(0.9, 0.999), lr=hyperparameters.first_inv_lr)
        self.noise_reg = NoiseRegulariser(self.noise_bufs.values())          # after Adam: it moves the parameters into its flat buffer

    # ---- re-use across images (round 5) -------------------------------------------------------------------------------------------
    # The reference builds a fresh projector per image (deep copy of G, w_avg from 600 mapped samples, new optimiser: mirror_projector.py:28-64),
    # and so did this class -- plus one eager step and a graph capture per image.  Everything a captured step reads lives in device tensors
    # of fixed shape (cameras, the target's LPIPS features, w_opt, the noise maps, Adam's flat state), so the NEXT image overwrites them in
    # place and replays the first image's graph: `run_projection` keeps the projector and calls `rebind`.
    def rebind(self, G, cameras, dist_fn, initial_w, w_avg_samples):
        """Point this projector at another image: same generator instance / shapes / objective type.  -> False when it cannot be re-used."""
        new_state, old_state = getattr(dist_fn, 'state', None), getattr(self.dist_fn, 'state', None)
        if new_state is None or old_state is None or len(new_state) != len(old_state) or getattr(self, '_graph_failed', False):
            return False
        if tuple(cameras.shape) != tuple(self.cameras.shape) or any(a.shape != b.shape or a.dtype != b.dtype for a, b in zip(old_state, new_state)):
            return False
        src, dst = G.state_dict(), self.G.state_dict()
        if src.keys() != dst.keys() or any(src[k].shape != dst[k].shape for k in src):
            return False
        with torch.no_grad():
            for k, v in dst.items():
                v.copy_(src[k])                                   # the caller's generator may have been reset / tuned since (restart_training)
            self.cameras.copy_(cameras)
            for a, b in zip(old_state, new_state):
                a.copy_(b)
            w_avg, self.w_std = w_statistics(self.G, self.cameras[:1], w_avg_samples, self.w_opt.device)
            start_w = initial_w if initial_w is not None else w_avg
            if self.w_mode == 'w+' and initial_w is None:
                start_w = np.repeat(start_w, self.num_ws, axis=1)
            self.w_opt.copy_(torch.as_tensor(start_w, dtype=torch.float32, device=self.w_opt.device))
            for buf in self.noise_bufs.values():
                buf.copy_(self.rng.randn(*buf.shape))
        self.optimizer.reset_state()
        self.images_bound = getattr(self, 'images_bound', 1) + 1
        return True

    # ---- HIP-graph replay -------------------------------------------------------------------------------------------------------
    # A stage-1 step is ~600 launches with no data-dependent host decision: after one eager warm-up step the second is CAPTURED
    # (torch.cuda.graph: forward, backward, Adam, re-normalisation) and every later step is one graph launch.  What changes from step
    # to step lives in device memory: `_hyper` = [lr, 1 - beta1^t, sqrt(1 - beta2^t), w_noise_scale], refreshed by one 16-byte copy
    # before each replay; the random draws are torch's graph-safe philox streams.  Same kernels in the same order as the eager step.
    # Only with the device generator (ReplayRNG draws come from a host list) and only on the GPU; a failed capture falls back to eager
    # steps and says so once.
    GRAPH_WARMUP = 1                                             # eager steps before the capture (lazy initialisations, allocator warm-up)
    captures_total = 0                                           # stage-1 graph captures of this process (all projectors)
    HYPER_RING = 32                                              # pinned slots for the per-step scalars = how far the host may run ahead

    def _graph_ok(self):
        from ...configs import global_config
        return graph_policy(global_config.stage1_hip_graph) and isinstance(self.rng, DeviceRNG) and self.w_opt.is_cuda and not getattr(self, '_graph_failed', False)

    def _set_hyper(self, step):
        lr, w_noise_scale = stage1_schedule(step, self.num_steps, self.w_std, **self.sched)
        self.optimizer.param_groups[0]['lr'] = lr
        lr32, bc1, bc2 = self.optimizer.hyper_values()
        if not hasattr(self, '_hyper'):
            self._hyper = torch.zeros(4, device=self.w_opt.device, dtype=torch.float32)
            self._hyper_ring = torch.zeros(self.HYPER_RING, 4, dtype=torch.float32).pin_memory()
            self._hyper_events = [None] * self.HYPER_RING
            self._hyper_pos = 0
        # the host runs ahead of the GPU: each step's values get their own pinned slot, reused only after its copy has executed
        k = self._hyper_pos % self.HYPER_RING
        self._hyper_pos += 1
        if self._hyper_events[k] is not None:
            self._hyper_events[k].synchronize()
        slot = self._hyper_ring[k]
        slot[0], slot[1], slot[2], slot[3] = lr32, bc1, bc2, w_noise_scale
        self._hyper.copy_(slot, non_blocking=True)
        ev = self._hyper_events[k] or torch.cuda.Event()
        ev.record()
        self._hyper_events[k] = ev

    def _graph_step(self, step):
        from ...configs import global_config
        # the captured launches bake in the arithmetic of the run: another setting drops the graph (one eager step, then a new capture)
        key = (global_config.conv_precision, global_config.conv_winograd, global_config.conv_winograd_f4, global_config.enable_fp16_blocks, global_config.exploit_sparsity)
        if getattr(self, '_graph_key', key) != key:
            self._graph, self._n_eager = None, 0
        self._graph_key = key
        if getattr(self, '_graph', None) is not None:
            self._set_hyper(step)
            self.optimizer.step_count += 1
            self._graph.replay()
            return {k: v.clone() for k, v in self._graph_out.items()}      # the graph's outputs are overwritten by the next replay
        self._n_eager = getattr(self, '_n_eager', 0) + 1
        if self._n_eager <= self.GRAPH_WARMUP:
            return self._body(step, device_hyper=True)
        self._set_hyper(step)
        try:
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with capture_graph(g, capture_error_mode=capture_mode()):
                out = self._body(step, device_hyper=True, hyper_is_set=True)
        except Exception as e:                                   # noqa: BLE001  (capture is an optimisation: the eager step is always valid)
            import sys
            self._graph_failed = True
            torch.cuda.synchronize()
            print(f'[spi_amd] stage-1 HIP-graph capture failed ({type(e).__name__}: {e}); continuing with eager steps', file=sys.stderr)
            import traceback
            print(''.join(traceback.format_tb(e.__traceback__)[-6:]), file=sys.stderr)
            return self._body(step, device_hyper=False)
        self._graph, self._graph_out = g, out
        Projection.captures_total += 1                            # (tests / bench: image k >= 2 must not capture again)
        g.replay()                                               # capture records, it does not execute: run the captured step once
        return {k: v.clone() for k, v in out.items()}

    def step(self, step):
        if self._graph_ok():
            return self._graph_step(step)
        return self._body(step, device_hyper=False)

    @zero_arena.closes_iteration
    def _body(self, step, device_hyper, hyper_is_set=False):
        G, rng, w_opt = self.G, self.rng, self.w_opt
        zero_arena.begin(w_opt.device, key='stage1')             # every accumulator of this step comes out of one buffer cleared by one launch
        if device_hyper:
            if not hyper_is_set:
                self._set_hyper(step)
            w_noise_scale = self._hyper[3]
        else:
            lr, w_noise_scale = stage1_schedule(step, self.num_steps, self.w_std, **self.sched)
            self.optimizer.param_groups[0]['lr'] = lr
        ws = w_opt + rng.randn(*w_opt.shape) * w_noise_scale
        if self.w_mode == 'w':
            ws = ws.repeat([1, self.num_ws, 1])
        batch = self.cameras.shape[0]                            # ws stays [1, L, 512]: G.synthesis shares the backbone across the views
        m = G.neural_rendering_resolution ** 2
        rk = G.rendering_kwargs
        noise = (rng.rand(batch, m, int(rk['depth_resolution']), 1), rng.rand(batch * m, max(int(rk['depth_resolution_importance']), 1)))
        images = G.synthesis(ws, self.cameras, noise_mode='const', render_noise=noise)['image']
        with trace_range('stage1/losses'):
            dist = self.dist_fn(images)
            reg_loss = self.noise_reg()
            loss = dist + reg_loss * self.reg_weight
        self.optimizer.zero_grad()
        with trace_range('stage1/backward'):
            loss.backward()
        with trace_range('stage1/optimizer'):
            self.optimizer.step(hyper=self._hyper[:3] if device_hyper else None)
            self.noise_reg.renorm()
        zero_arena.finish()                                      # nothing outside the step may be handed a view of this step's (possibly graph-owned) buffer
        return dict(dist=dist.detach(), reg=reg_loss.detach(), loss=loss.detach())


_projectors = {}            # (id(G), settings) -> (weakref to G, Projection): the projector of a coach's generator survives from image to image


def _cached_projection(G, cameras, dist_fn, *, w_mode, initial_w, num_steps, w_avg_samples, device, rng, regularize_noise_weight, schedule_kwargs):
    import weakref
    from ...configs import global_config
    # (the device generator only -- rng None means exactly that; DeviceRNG instances are stateless views of torch's default generator)
    if not (global_config.reuse_graphs_across_images and (rng is None or isinstance(rng, DeviceRNG))):
        return None, None
    key = (id(G), w_mode, int(num_steps), int(w_avg_samples), str(device), float(regularize_noise_weight), tuple(sorted((schedule_kwargs or {}).items())))
    hit = _projectors.get(key)
    if hit is not None and hit[0]() is G and hit[1].rebind(G, cameras, dist_fn, initial_w, w_avg_samples):
        return hit[1], key
    for k in [k for k, v in _projectors.items() if v[0]() is None or k[0] == id(G)]:     # one projector per generator; dead generators drop theirs
        del _projectors[k]
    return None, (key, weakref.ref(G))


def run_projection(G, cameras, dist_fn, *, w_mode, initial_w, num_steps, w_avg_samples, device, rng=None, log=None,
                   regularize_noise_weight=1e5, schedule_kwargs=None):
    """Generic stage-1 loop.  ``cameras`` [B,25]; ``dist_fn(images [B,3,R,R]) -> scalar``; w_mode 'w' | 'w+'.
    A ``dist_fn`` that exposes its per-image tensors as ``dist_fn.state`` (list) lets the projector of the previous image of the same
    generator be re-used (`Projection.rebind`: the image's tensors are overwritten in place, the captured step is replayed as is)."""
    proj, slot = _cached_projection(G, cameras, dist_fn, w_mode=w_mode, initial_w=initial_w, num_steps=num_steps, w_avg_samples=w_avg_samples,
                                    device=device, rng=rng, regularize_noise_weight=regularize_noise_weight, schedule_kwargs=schedule_kwargs)
    if proj is None:
        proj = Projection(G, cameras, dist_fn, w_mode=w_mode, initial_w=initial_w, num_steps=num_steps, w_avg_samples=w_avg_samples,
                          device=device, rng=rng, regularize_noise_weight=regularize_noise_weight, schedule_kwargs=schedule_kwargs)
        if slot is not None:
            _projectors[slot[0]] = (slot[1], proj)
    from ...torch_utils.misc import quiet_gc
    with quiet_gc():
        for step in range(num_steps):
            out = proj.step(step)
            if log is not None:
                log.append(dict(out, w=proj.w_opt.detach().clone(), grad_w=proj.w_opt.grad.detach().clone()))
    # a cached projector's w_opt is overwritten by the next image: hand out a copy
    return proj.w_opt.detach().clone() if slot is not None else proj.w_opt
