#!/usr/bin/env python3
"""Benchmark of the SPI inversion hot path on MI355X (contract: see the task description / DESIGN.md).

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / LOCAL_RANK / WORLD_SIZE
in the environment) or from a bare shell -- `python bench.py --gpus N` then spawns its own N ranks (one per GPU, RCCL) and relays
rank 0's JSON line.

Workload (BASELINE.json configs[1]): 1 image per GPU, first_inv_type=mir (500 steps) + G_1_type=RotBbox (1000 steps),
EG3D ffhqrebalanced512-128 architecture at 512^2 with 96 coarse + 96 fine samples per ray, fp32, synthetic inputs
and seeded random-init weights (no checkpoint / dataset exists offline).  A "step" is one iteration of the hot path;
the K timed steps are split between stage-1 ('mir') and stage-2 ('RotBbox') iterations as close to the configuration's 1:2 as K allows,
with the stage-2 part a whole number of 4-iteration super-cycles so the every-4th-step branches are amortised exactly.  `value` does
NOT depend on how K happened to split: it is the rate of the configuration's exact 500:1000 mix computed from the two per-stage rates
measured inside the timed region (max over ranks per stage), value = 3 / (1 / r_stage1 + 2 / r_stage2); `ms_per_step` = 1000 / value per
GPU, and the raw K-step wall time is reported beside it (`timed_region`).
Extra keys of the line, none of them `value`: `sustained` (>= 300 more steps of the same loops in one window of several seconds), `alt`
(split-bf16 convolutions), `dense` (no data-driven skipping), `cfg4` (BASELINE configs[4]: 128+128 samples, fp16 super-resolution) and `pti`
(BASELINE configs[2]: `sg` + `pti`) -- the last two are separate bench.py processes (`--depth 128 --sr-fp16`, `--workload pti`) started after
the benchmark line has been measured (single-GPU default runs only; `--no-legs` skips them).
Prints ONE JSON line on rank 0.
"""
import argparse
import contextlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import spi_amd  # noqa: E402,F401  (before the first GPU call: sets the HIP runtime switch that makes graph replays safe, spi_amd/__init__.py)

# Algorithmic bytes of the FINAL march per ray: read S*(C+1+1)*4 (colours, density, depth), write (C+1+1)*4 (rgb, depth, sum w).
# SURVEY.md 8d's 27 008 B/ray also counts the S-1 per-sample weights; the final march does not write them (only their sum
# is consumed, renderer.py:140), so they are left out: 26 112 + 136 = 26 248 B/ray at S = 192.
RAYMARCH_BYTES_PER_RAY = lambda S, C=32: S * (C + 2) * 4 + (C + 2) * 4
HBM_PEAK_GBS = 8000.0                                                                    # MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=24)
    ap.add_argument('--warmup', type=int, default=6)
    ap.add_argument('--depth', type=int, default=96, help='coarse = fine samples per ray')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline', choices=('sample', 'protocol', 'full'), default='protocol',
                    help="sample (~45 s): 1 warm-up + 1 timed stage-1 'mir' step + 1 timed plain stage-2 iteration of the oracle; "
                         'protocol (default, ~2.5 min): + the every-4th-iteration branch iteration (rot / mirror-rot / depth), i.e. one whole 4-iteration stage-2 '
                         'super-cycle MEASURED in this process -- the value is the same stage mix as the GPU line, not an upper bound; '
                         'full (~4 min): + a thread-scaling line')
    ap.add_argument('--no-wino-f2-leg', action='store_true', help='skip the extra leg that times the same steps with F(2x2,3x3) Winograd on every layer (`wino_f2`)')
    ap.add_argument('--no-winograd', action='store_true', help='3x3 convolutions on the implicit-GEMM kernels only (global_config.conv_winograd = False)')
    ap.add_argument('--dense', action='store_true', help='NOT the benchmark configuration: switch off the data-driven skipping of exactly-zero gradients / '
                                                         'unneeded SR tiles in the masked pseudo-view branches (dense bound of the same step)')
    ap.add_argument('--sr-fp16', action='store_true', help='NOT the benchmark configuration: fp16 MFMA in the super-resolution blocks '
                                                           '(BASELINE config 5); the JSON line then says dtype f32+f16sr')
    ap.add_argument('--narrow', action='store_true', help='debug: reduced-width generator (NOT the benchmark configuration)')
    ap.add_argument('--conv-precision', choices=('f32', 'bf16x6', 'bf16x3'), default='f32',
                    help='arithmetic of the dense convolutions: f32 = exact fp32 MFMA (default, the benchmark configuration); bf16x6 / bf16x3 = fp32 operands '
                         'split into 3 / 2 bf16 pieces, 6 / 3 bf16 MFMAs with fp32 accumulation (opt-in; the JSON line then says so in dtype)')
    ap.add_argument('--alt-conv-precision', choices=('none', 'bf16x6', 'bf16x3'), default='bf16x6',
                    help='after the timed region, time the same K steps once more with this conv arithmetic and report it as `alt` (never as `value`)')
    ap.add_argument('--no-dense-leg', action='store_true', help='skip the extra `dense` measurement (stage 2 once more without the data-driven skipping)')
    ap.add_argument('--only', choices=('stage1', 'stage2'), default=None, help='profiling aid, NOT the benchmark configuration: all K steps from one stage')
    ap.add_argument('--workload', choices=('spi', 'pti'), default='spi',
                    help="spi (default) = BASELINE configs[1] (or configs[4] with --depth 128 --sr-fp16): 'mir' stage 1 + RotBbox stage 2; "
                         "pti = BASELINE configs[2], the PTI baseline: first_inv_type=sg (W projector) + G_1_type=pti (SingleIDCoach)")
    ap.add_argument('--sustained', type=int, default=300,
                    help='after the K timed steps, time this many more steps (1:2 stage mix, whole super-cycles) and report them as `sustained` beside '
                         '`value` (0 = skip): a sub-second window can sit on a boost clock, a 7-second one cannot')
    ap.add_argument('--no-legs', action='store_true', help='skip the `cfg4` (128+128 samples, fp16 super-resolution) and `pti` (configs[2]) legs, which '
                                                           'run as child processes of a default single-GPU run after the benchmark line is measured')
    ap.add_argument('--dry-run', action='store_true', help='plumbing self-test without a GPU: launcher, rendezvous (gloo), barrier and the statistics '
                                                           'all-reduces run as in a real run, the timed steps are replaced by a sleep; prints no metric')
    return ap.parse_args()


def split_steps(k):
    """K steps -> (stage-1 steps, stage-2 steps) in the 500:1000 proportion, stage-2 a multiple of 4."""
    k2 = max(4, int(round(k * 2 / 3 / 4)) * 4) if k >= 6 else max(0, k - k // 3)
    k2 = min(k2, k)
    return k - k2, k2


def cpu_baseline(depth, narrow, mode='sample', k1=8, k2=16):
    """Oracle (CPU restatement of the reference, oracle/loops_ref.py -- pinned bit-exact against the imported reference) timed on this
    box's host cores on the same synthetic inputs as the GPU run, per stage:
      stage 1  one `mir` projector step (view + mirrored view, N = 2 synthesis fwd+bwd, 2 LPIPS, noise regulariser, Adam over w+ / noise maps)
      stage 2  `RotBbox` iterations: plain (main view: synthesis fwd+bwd, L2 + LPIPS, Adam over all G parameters) and, with mode='full', one
               whole 4-iteration super-cycle including the rot / mirror-rot / depth branches of every 4th iteration.
    One untimed warm-up iteration first (thread pool, allocator, first-touch)."""
    from oracle import loops_ref as olp, losses_ref as olo, renderer_ref as orr
    from spi_amd.training.triplane import TriPlaneGenerator, ffhq512_kwargs
    from spi_amd.data.images_dataset import SyntheticDataset
    ncpu = os.cpu_count() or 1
    cores = min(ncpu, 32)        # torch's CPU kernels stop scaling far below the box's hardware threads (scaling line under mode='full')
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    G = TriPlaneGenerator(**ffhq512_kwargs(narrow=narrow))
    P = {k: v.detach().clone() for k, v in G.state_dict().items()}
    pnames = [k for k, _ in G.named_parameters()]
    del G
    W, W19 = olo.make_vgg16_weights(seed=0), olo.make_vgg19_head_weights(seed=1)
    lp = lambda a, b: olo.lpips(W, a, b)
    bx = lambda a, b, l: olo.box_cx_loss(W19, a, b, l)
    d = SyntheticDataset(1)[0]
    mask = d['mask'].reshape(1, 1, 512, 512)
    data = dict(img=d['img'][None], c=torch.as_tensor(d['c']).reshape(1, 25), lm=d['lm'].reshape(1, 68, 2),
                face_mask=olp.face_mask_from_parsing(mask).float())
    opts = dict(orr.DEFAULT_RENDERING, depth_resolution=depth, depth_resolution_importance=depth)
    w = torch.randn(1, 14, 512) * 0.5
    hp = dict(olp.HP, LPIPS_value_threshold=-1.0)

    def timed(fn):
        t0 = time.perf_counter()
        fn()
        return time.perf_counter() - t0

    st = olp.Stage2State(P, pnames)
    t_warm = timed(lambda: olp.stage2_iteration(st, 1, data, w, opts, lp, bx, hp=hp))                       # untimed warm-up
    # stage 1: steps 1.. of a 500-step schedule (step 0 has lr = 0); the loop function runs whole projections, so time 2 steps and 1 step
    t_s1_2 = timed(lambda: olp.project_w_plus(P, data['img'], data['c'], lp, opts, mirror=True, num_steps=2, w_avg_samples=16))
    t_s1_1 = timed(lambda: olp.project_w_plus(P, data['img'], data['c'], lp, opts, mirror=True, num_steps=1, w_avg_samples=16))
    t_s1 = max(t_s1_2 - t_s1_1, 1e-9)                                                                          # one step without the set-up
    t_plain = timed(lambda: olp.stage2_iteration(st, 1, data, w, opts, lp, bx, hp=hp))
    res = dict(unit='iters/s', cores=cores, kind='port', host_threads_available=ncpu,
               stage1_mir_iters_per_s=1.0 / t_s1, stage2_plain_iters_per_s=1.0 / t_plain,
               seconds=dict(warmup=t_warm, stage1_step=t_s1, stage2_plain_iteration=t_plain))
    if mode in ('protocol', 'full'):
        t_branch = timed(lambda: olp.stage2_iteration(st, 0, data, w, opts, lp, bx, hp=hp))                   # i % 4 == 0: all three branches
        t_cycle = t_branch + 3 * t_plain
        res['seconds'].update(stage2_branch_iteration=t_branch, stage2_super_cycle=t_cycle)
        res['stage2_rotbbox_iters_per_s'] = 4.0 / t_cycle
        res['value'] = (k1 + k2) / (k1 * t_s1 + k2 * t_cycle / 4.0)
        if mode == 'full':
            scal = {}
            for th in (8, 16, 32, 64, 128, ncpu):
                if th <= ncpu and th not in scal:
                    torch.set_num_threads(th)
                    scal[th] = timed(lambda: olp.stage2_iteration(st, 1, data, w, opts, lp, bx, hp=hp))
            torch.set_num_threads(cores)
            res['thread_scaling_s_per_plain_iteration'] = scal
        else:
            full = os.path.join(ROOT, 'profiles', 'cpu_baseline_full.json')
            if os.path.exists(full):
                res['thread_scaling_recorded'] = {'s_per_plain_iteration': json.load(open(full)).get('thread_scaling_s_per_plain_iteration'),
                                                  'note': 'profiles/cpu_baseline_full.json (`bench.py --cpu-baseline full`): why 32 of the host threads -- NOT measured in this process; everything else in this object is'}
        res['sample'] = (f"full protocol, measured in this process: 1 warm-up, 1 'mir' step ({t_s1:.1f} s), one 4-iteration RotBbox super-cycle ({t_cycle:.1f} s: branch iteration "
                         f'{t_branch:.1f} s + 3 x plain {t_plain:.1f} s); value = the same {k1}:{k2} stage mix as the `value` of the GPU line')
    else:
        # without the branch iteration the stage-2 rate is an UPPER bound for the CPU (the branches only add work): say so
        res['value'] = (k1 + k2) / (k1 * t_s1 + k2 * t_plain)
        res['sample'] = (f"1 warm-up + 1 timed 'mir' step ({t_s1:.1f} s) + 1 timed PLAIN RotBbox iteration ({t_plain:.1f} s); the every-4th-iteration "
                         f'rot / mirror-rot / depth branches are not in this sample (value = the {k1}:{k2} mix of the two like the GPU line, an upper bound for the CPU); '
                         'full protocol: `bench.py --cpu-baseline full`, kept under profiles/')
        full = os.path.join(ROOT, 'profiles', 'cpu_baseline_full.json')
        if os.path.exists(full):
            rec = json.load(open(full))
            res['full_protocol_recorded'] = rec                   # (recorded once per round by `bench.py --cpu-baseline full`, with that process's own GPU value beside it)
            res['full_protocol_is_a_replayed_record'] = 'profiles/cpu_baseline_full.json: NOT measured in this process (this run timed the 2-iteration sample above)'
    res['sample'] += f'; 512^2, {depth}+{depth} samples, torch {torch.__version__} CPU fp32, {cores} threads of {ncpu}'
    return res


def elementwise_roofline(dev):
    """Roofline line of the HBM-bound elementwise kernels the round-5 profile had no line for (layer tails 4.5 %, 4x4 FIR 4.0 %, ATen elementwise 3.6 %
    of the GPU time): achieved GB/s of ALGORITHMIC bytes at the largest layer they run on (SR b512: 128 channels at 512^2, one image), HIP events on
    the launch stream, outside the timed region.  `aten_elementwise` is a torch add of the same footprint -- what an ATen launch reaches at this size; the
    loop's ATen launches are mostly small (5-6 us on average), so their share is launch-bound, not bandwidth-bound."""
    from spi_amd.torch_utils.ops import bias_act as ba, upfirdn2d as U

    def timed(fn, n=10):
        for _ in range(3):
            fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in ev) / n * 1e-3
    C, H = 128, 512
    peak = 8000.0
    res = {}
    with torch.no_grad():
        dy, y = torch.randn(1, C, H, H, device=dev), torch.randn(1, C, H, H, device=dev)
        noise, st = torch.randn(H, H, device=dev), torch.ones(1, device=dev)
        t = timed(lambda: ba.tail_backward(dy, y, noise, st, 3, 0.2, 1.414, 256.0, False, True, True))
        by = dy.numel() * 12                                   # dy and y read, dz written (the bias / noise sums are O(C + H^2))
        res['tail_bwd_kernel'] = {'shape': f'[1,{C},{H},{H}] lrelu, clamp, bias + noise sums', 'bytes': by, 'avg_launch_us': t * 1e6, 'achieved': by / t / 1e9, 'frac': by / t / 1e9 / peak}
        f = U.setup_filter([1, 3, 3, 1]).to(dev)
        x = torch.randn(1, C, H + 1, H + 1, device=dev)
        b = torch.randn(C, device=dev)
        t = timed(lambda: U.upfirdn2d_bias_act(x, f, noise=noise, noise_strength=torch.ones((), device=dev), bias=b, padding=[1, 1, 1, 1], gain=4, act='lrelu', act_gain=1.414, clamp=256))
        by = (x.numel() + C * H * H) * 4
        res['upfirdn2d_4x4_tiled_kernel'] = {'shape': f'[1,{C},{H + 1},{H + 1}] -> {H}^2, fused noise / bias / lrelu / clamp (the tail of an up-sampling layer)', 'bytes': by,
                                             'avg_launch_us': t * 1e6, 'achieved': by / t / 1e9, 'frac': by / t / 1e9 / peak}
        a2, o2 = torch.randn(1, C, H, H, device=dev), torch.empty(1, C, H, H, device=dev)
        t = timed(lambda: torch.add(dy, a2, out=o2))
        by = dy.numel() * 12
        res['aten_elementwise'] = {'shape': f'torch.add on [1,{C},{H},{H}]', 'bytes': by, 'avg_launch_us': t * 1e6, 'achieved': by / t / 1e9, 'frac': by / t / 1e9 / peak}
    worst = min(v['achieved'] for v in res.values())
    return {'kernel': 'tail_bwd_kernel / upfirdn2d_4x4_tiled_kernel / ATen elementwise at SR b512 (128 channels, 512^2)', 'bound': 'hbm', 'achieved': worst, 'peak': peak, 'unit': 'GB/s',
            'frac': worst / peak, 'kernels': res, 'note': 'achieved = slowest of the three; algorithmic bytes (every tensor element read or written once)'}


def conv_roofline(dev, f16, prec=0):
    """Second roofline line: the matrix-core kernels that hold most of the step's GPU time (igemm_kernel / wgrad_kernel).  Times the
    largest convolution of the loop -- superresolution b512.conv1, 128 -> 128, 3x3 at 512^2, one image -- with HIP events on the
    launch stream, outside the timed region: forward, data gradient, weight gradient."""
    from spi_amd.torch_utils.ops import conv2d_mfma as cm
    from spi_amd import hip
    import ctypes
    n, i, o, h, k = 1, 128, 128, 512, 3
    from spi_amd.configs import global_config as _gc
    half = bool(f16 and _gc.fp16_storage)                       # configs[4]: the SR blocks' activations are fp16 tensors (spi_conv_desc.act_dtype)
    x = torch.randn(n, i, h, h, device=dev)
    w = torch.randn(n, o, k, k, i, device=dev) * 0.03
    y = torch.randn(n, o, h, h, device=dev)
    if half:
        x, y = x.half(), y.half()
    dx, dw = torch.empty_like(x), torch.empty_like(w)
    d = cm._desc(n, i, o, h, h, k, 1, False, True, o * i * k * k, tap_major=1, f16=1 if f16 else prec, half=half)
    flop = 2.0 * n * o * i * k * k * h * h
    # dense MFMA peaks (TFLOP/s), MI355X_MICROARCH.md: fp16 2500, fp32 157.3; split-bf16 modes: the bf16 peak / number of piece products
    peak = 2500.0 if f16 else {0: 157.3, 2: 2500.0 / 3, 3: 2500.0 / 6}[prec]
    res = {}
    for name, fn in (('fwd', lambda: hip.call('spi_conv2d_fwd', ctypes.byref(d), hip.ptr(x), hip.ptr(w), hip.ptr(y), hip.stream())),
                     ('dgrad', lambda: hip.call('spi_conv2d_dgrad', ctypes.byref(d), hip.ptr(y), hip.ptr(w), hip.ptr(dx), hip.stream())),
                     ('wgrad', lambda: hip.call('spi_conv2d_wgrad', ctypes.byref(d), hip.ptr(x), hip.ptr(y), hip.ptr(dw), hip.stream()))):
        for _ in range(3):
            fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev)
        avg = sum(ms) / len(ms)
        res[name] = {'avg_launch_us': avg * 1e3, 'achieved': flop / (avg * 1e-3) / 1e12}
    worst = min(v['achieved'] for v in res.values())
    out = {'kernel': 'igemm_kernel / wgrad_kernel (implicit GEMM, the kernels of the 1x1 / transposed / small layers) on SR b512.conv1 (128->128, 3x3, 512^2, N=1)', 'bound': 'mfma', 'achieved': worst, 'peak': peak,
           'unit': 'TFLOP/s', 'frac': worst / peak, 'flop_per_launch': flop, 'passes': res,
           'note': 'achieved = slowest of the three passes; wgrad includes its memset of dw'}
    if f16:
        out['fp16'] = {'activation_tensors': 'fp16 in HBM (act_dtype)' if half else 'fp32 in HBM, operands rounded on their way into LDS', 'peak': peak,
                       'passes_frac_of_fp16_peak': {k_: v['achieved'] / peak for k_, v in res.items()}}
    # fp16 activation tensors: the same layer on the direct kernels the loop actually takes for its 3x3 / stride-1 layers (hconv.hip: forward / dgrad
    # incl. the launch that converts the weights to their fp16 LDS image, weight gradient incl. its partial-sum reduction)
    if half and _gc.conv_direct_fp16:
        dres = {}
        for name, pid, fn in (('fwd', 0, lambda dd: hip.call('spi_conv2d_fwd', ctypes.byref(dd), hip.ptr(x), hip.ptr(w), hip.ptr(y), hip.stream())),
                              ('dgrad', 1, lambda dd: hip.call('spi_conv2d_dgrad', ctypes.byref(dd), hip.ptr(y), hip.ptr(w), hip.ptr(dx), hip.stream())),
                              ('wgrad', 2, lambda dd: hip.call('spi_conv2d_wgrad', ctypes.byref(dd), hip.ptr(x), hip.ptr(y), hip.ptr(dw), hip.stream()))):
            dd = cm._desc(n, i, o, h, h, k, 1, False, True, o * i * k * k, tap_major=1, f16=1, half=True)
            ws = cm._workspace(dd, pid, x.device)
            if ws is None:
                continue
            for _ in range(3):
                fn(dd)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
            for a, b in ev:
                a.record(); fn(dd); b.record()
            torch.cuda.synchronize()
            avg = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
            dres[name] = {'avg_launch_us': avg * 1e3, 'achieved': flop / (avg * 1e-3) / 1e12, 'frac': flop / (avg * 1e-3) / 1e12 / peak}
        out['fp16']['direct'] = {'kernel': 'hconv_weight_kernel + hconv_kernel (fwd, dgrad), hwgrad_kernel + hwgrad_reduce_kernel (wgrad), same layer',
                                 'passes': dres, 'passes_frac_of_fp16_peak': {k_: v['frac'] for k_, v in dres.items()},
                                 'note': 'these are the kernels the loop runs for this layer; `passes_frac_of_fp16_peak` one level up is the implicit GEMM '
                                         '(transposed / 1x1 / small layers)'}
    # the same layer on the Winograd F(2x2, 3x3) path the loop actually takes for forward / dgrad of the >= 128^2 3x3 layers (exact fp32
    # mode only): `achieved` counts the direct convolution's FLOPs (the algorithmic work), `executed` the MFMA FLOPs issued (/ 2.25)
    from spi_amd.configs import global_config
    if global_config.conv_winograd and not f16 and prec in (0, 3):
        wres = {}
        for name, pid, fn in (('fwd', 0, lambda dd: hip.call('spi_conv2d_fwd', ctypes.byref(dd), hip.ptr(x), hip.ptr(w), hip.ptr(y), hip.stream())),
                              ('dgrad', 1, lambda dd: hip.call('spi_conv2d_dgrad', ctypes.byref(dd), hip.ptr(y), hip.ptr(w), hip.ptr(dx), hip.stream())),
                              ('wgrad', 2, lambda dd: hip.call('spi_conv2d_wgrad', ctypes.byref(dd), hip.ptr(x), hip.ptr(y), hip.ptr(dw), hip.stream()))):
            dd = cm._desc(n, i, o, h, h, k, 1, False, True, o * i * k * k, tap_major=1)
            ws = cm._workspace(dd, pid, x.device)
            if ws is None:
                continue
            for _ in range(3):
                fn(dd)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
            for a, b in ev:
                a.record(); fn(dd); b.record()
            torch.cuda.synchronize()
            avg = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
            # multiplications saved by the minimal-filtering tile the pass runs: F(4x4,3x3) 144 / 36 = 4 (forward / dgrad of this layer with
            # global_config.conv_winograd_f4), F(2x2,3x3) and the F(3x3,2x2) weight gradient 36 / 16 = 2.25
            red = 4.0 if (global_config.conv_winograd_f4 and pid < 2) else 2.25
            wres[name] = {'avg_launch_us': avg * 1e3, 'achieved': flop / (avg * 1e-3) / 1e12, 'executed': flop / red / (avg * 1e-3) / 1e12,
                          'frac_executed': flop / red / (avg * 1e-3) / 1e12 / peak, 'multiplications_saved': red}
        f4 = bool(global_config.conv_winograd_f4)
        out['winograd'] = {'kernel': ('wino4_weight_kernel + wino4_conv_kernel (F(4x4,3x3): fwd, dgrad)' if f4 else 'wino_weight_kernel + wino_conv_kernel (F(2x2,3x3): fwd, dgrad)')
                                     + ', wino_wgrad_kernel (F(3x3,2x2): wgrad, incl. its memset of dw), same layer',
                           'passes': wres,
                           'note': 'fp32 operands and accumulation; ' + ('36 MFMA multiplications per 4x4 tile and channel pair instead of 144' if f4 else '16 MFMA multiplications per 2x2 tile and channel pair instead of 36')
                                   + ' (forward / dgrad); these are the kernels the loop runs for this layer'}
    return out


def child_leg(extra, steps, warmup, timeout=600):
    """Run this script once more as a child process with another configuration and return the fields of its JSON line that a leg reports.
    The child is a complete, independent bench.py run (own process, own warm-up, own barrier + sync bracketing); it runs AFTER the parent's
    measurements, on the same GPU."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--steps', str(steps), '--warmup', str(warmup), '--no-cpu-baseline', '--alt-conv-precision', 'none',
           '--no-dense-leg', '--no-legs', '--sustained', '0'] + list(extra)
    env = dict(os.environ, SPI_BENCH_POOL_GIB='8')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
        if r.returncode != 0 or not line:
            return {'error': f'rc {r.returncode}: ' + (r.stderr or '')[-400:]}
        j = json.loads(line[-1])
        return {k: j.get(k) for k in ('value', 'unit', 'ms_per_step', 'steps', 'warmup', 'dtype', 'stages', 'timed_region') if k in j} | {
            'workload': j['config']['workload'], 'command': 'python bench.py ' + ' '.join(cmd[2:]),
            'note': 'a separate bench.py process run after the benchmark line was measured; NOT the benchmark value'}
    except Exception as e:                                       # noqa: BLE001
        return {'error': repr(e)}


def whole_job_leg(stage_rates, n1=120, n2=120, images=2, timeout=600):
    """The real CLI end to end as a child process: `python -m spi_amd.run_inversion` on `images` synthetic inputs,
    configs[1]'s loops shortened to n1 'mir' steps + n2 RotBbox iterations per image, every output written (checkpoint, embedding, pictures,
    120-frame orbit).  Reports, per image, the rate of its set-up + both loops (`seconds_loop`: prepare_image, the projector, the stage-2 loop)
    beside what the benchmark's two per-stage rates predict for the same n1 : n2 mix -- the second image replays the first image's graphs
    (global_config.reuse_graphs_across_images), so its rate is the one a dataset run sustains."""
    import subprocess
    import tempfile
    out = tempfile.mkdtemp(prefix='spi_wholejob_') + '/'
    cli = ['--output_root', out, '--synthetic', str(images), '--not_use_wandb',
           '--first_inv_type', 'mir', '--first_inv_steps', str(n1), '--G_1_type', 'RotBbox', '--G_1_step', str(n2),
           '--pt_rot_lambda', '0.1', '--pt_mirror_rot_lambda', '0.05', '--pt_depth_lambda', '1.0', '--depth_resolution', '96', '--depth_resolution_importance', '96']
    # (random-init weights give LPIPS distances below the reference's early-stop threshold: the loops must run their full length, as in the timed region)
    cmd = [sys.executable, '-c', 'from spi_amd.configs import hyperparameters as hp; hp.LPIPS_value_threshold = -1.0; '
                                 f'from spi_amd import run_inversion; run_inversion.run({cli!r})']
    env = dict(os.environ, SPI_POOL_GIB='8')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    try:
        t0 = time.perf_counter()
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
        wall = time.perf_counter() - t0
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
        if r.returncode != 0 or not line:
            return {'error': f'rc {r.returncode}: ' + (r.stderr or '')[-400:]}
        j = json.loads(line[-1])
        r1, r2 = stage_rates
        expected = (n1 + n2) / (n1 / r1 + n2 / r2) if r1 and r2 else None
        per = []
        for s in j.get('per_image', []):
            its = s['iters'] + s.get('stage1_iters', 0)
            per.append(dict(name=s['name'], iterations=its, seconds_loop=round(s['seconds_loop'], 3), seconds_outputs=round(s['seconds_outputs'], 3),
                            loop_iters_per_s=round(its / s['seconds_loop'], 2), stage2_graph_captures=s.get('stage2_graph_captures'),
                            loop_rate_over_expected=(round(its / s['seconds_loop'] / expected, 3) if expected else None)))
        first = dict(per[0], what='the single-image job the reference README runs: image 1 pays the eager warm-up iterations and the graph captures of both loops '
                                  '(per process); `loop_rate_over_expected` = its rate over what the two stage rates predict for this mix') if per else None
        return {'first_image': first,
                'images': j['images'], 'iterations': j['iterations'], 'job_seconds_in_train': round(j['seconds'], 2), 'process_wall_seconds': round(wall, 2),
                'job_iters_per_s_including_outputs': round(j['iters_per_sec'], 2), 'hip_graphs': j.get('hip_graphs'), 'per_image': per,
                'expected_iters_per_s_from_this_runs_stage_rates': (round(expected, 2) if expected else None), 'mix': f'{n1} mir steps + {n2} RotBbox iterations per image',
                'command': 'python -m spi_amd.run_inversion ' + ' '.join(cli) + '   (with hyperparameters.LPIPS_value_threshold = -1: no early stop)',
                'note': 'a separate process run after the benchmark line was measured; NOT the benchmark value.  loop = per-image set-up + stage 1 + stage 2 '
                        '(outputs -- checkpoint, pictures, 120-frame orbit -- are listed beside it); image 2 re-uses image 1 graphs'}
    except Exception as e:                                       # noqa: BLE001
        return {'error': repr(e)}


def launch_ranks(n):
    """`python bench.py --gpus N` from a bare shell: spawn one rank per GPU (the reference's analogue: one `--dataset_block i/N` process per
    GPU, images_dataset.py:149-158), relay their output, and never let a dead rank hang the others: the first non-zero exit kills the rest."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'),
                   # like torch.distributed.run (which sets 1): N ranks with one OpenMP pool of all host threads each would oversubscribe the host
                   OMP_NUM_THREADS=os.environ.get('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 1) // (2 * n)))))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        while any(p.poll() is None for p in procs):
            time.sleep(0.2)
            bad = [p.returncode for p in procs if p.poll() not in (None, 0)]
            if bad:
                rc = bad[0]
                break
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate() if rc else p.wait()
        for p in procs:
            try:
                p.wait(timeout=10)
            except Exception:
                p.kill()
    return rc or max((p.returncode or 0) for p in procs)


def reduce_run_stats(sdist, rank, world, t0, dt_rank, ok, dev):
    """Barrier, then ONE max-reduce of the wall time and ONE sum-reduce carrying every rank's own time and done-flag.
    -> (max-over-ranks seconds, per-rank seconds, per-rank done-flags)"""
    sdist.barrier()
    dt = time.perf_counter() - t0
    dt = sdist.reduce_stats([dt], device=dev, op='max')[0]
    slots = [0.0] * (2 * world)
    slots[rank], slots[world + rank] = dt_rank, ok
    slots = sdist.reduce_stats(slots, device=dev)
    return dt, slots[:world], slots[world:]


def device_identity(local):
    """(device index, PCI domain, bus, device) of this rank's GPU as floats for the statistics all-reduce; zeros without a GPU."""
    if not torch.cuda.is_available():
        return [float(local), 0.0, 0.0, 0.0]
    p = torch.cuda.get_device_properties(local)
    return [float(local), float(getattr(p, 'pci_domain_id', 0)), float(getattr(p, 'pci_bus_id', -1)), float(getattr(p, 'pci_device_id', -1))]


def gather_rank_devices(sdist, rank, world, local, dev):
    """One sum-reduce in which every rank fills its own 4 slots -> per-rank device records (same collective kind as the timing reduce)."""
    slots = [0.0] * (4 * world)
    slots[4 * rank:4 * rank + 4] = device_identity(local)
    slots = sdist.reduce_stats(slots, device=dev)
    return [{'rank': r, 'device_index': int(slots[4 * r]), 'pci': '%04x:%02x:%02x' % tuple(int(v) for v in slots[4 * r + 1:4 * r + 4])} for r in range(world)]


def gather_affinity(sdist, rank, world, affinity, dev):
    """One sum-reduce of a per-CPU occupancy vector: how many ranks hold each host CPU after pin_rank_affinity (an entry > 1 = shared core)."""
    ncpu = os.cpu_count() or 1
    held = sorted(affinity) if affinity is not None else sorted(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else []
    vec = [0.0] * (ncpu + world)
    for c in held:
        if c < ncpu:
            vec[c] = 1.0
    vec[ncpu + rank] = float(len(held))
    vec = sdist.reduce_stats(vec, device=dev)
    return {'pinned': affinity is not None, 'cpus_per_rank': [int(v) for v in vec[ncpu:]], 'max_ranks_on_one_cpu': int(max(vec[:ncpu]) if ncpu else 0),
            'disjoint': bool(world == 1 or max(vec[:ncpu]) <= 1.0)}


def process_group_info(world):
    import torch.distributed as td
    if world > 1 and td.is_available() and td.is_initialized():
        return {'world_size': td.get_world_size(), 'backend': td.get_backend()}
    return {'world_size': 1, 'backend': None}


def dry_run(args, sdist, rank, world):
    """--dry-run: everything around the GPU work (used by the world-size-2 CPU test)."""
    sdist.barrier()
    t0 = time.perf_counter()
    ok = 1.0
    try:
        if os.environ.get('SPI_BENCH_FAIL_RANK') == str(rank):
            raise RuntimeError('injected failure')
        time.sleep(0.05 * (rank + 1))
    except Exception:
        ok = 0.0
    dt, rank_s, rank_ok = reduce_run_stats(sdist, rank, world, t0, time.perf_counter() - t0, ok, None)
    devices = gather_rank_devices(sdist, rank, world, int(os.environ.get('LOCAL_RANK', rank)), None)
    affinity = sdist.pin_rank_affinity(int(os.environ.get('LOCAL_RANK', rank)), sdist.local_world_size(world))
    affinity_report = gather_affinity(sdist, rank, world, affinity, None)
    from spi_amd.configs import global_config
    from spi_amd.training.projectors.common import graph_policy, capture_mode
    if rank == 0:
        print(json.dumps({'dry_run': True, 'n_gpus': world, 'steps': args.steps, 'seconds_max_over_ranks': dt,
                          'ranks': {'launched': world, 'completed': int(sum(rank_ok)), 'per_rank_seconds': rank_s, 'devices': devices,
                                    'process_group': process_group_info(world), 'cpu_affinity': affinity_report},
                          'stage1_hip_graph_policy': graph_policy(global_config.stage1_hip_graph), 'graph_capture_mode': capture_mode()}), flush=True)
    sdist.shutdown()
    if int(sum(rank_ok)) != world:
        sys.exit(3)


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    from spi_amd import dist as sdist
    rank, world, local = sdist.init_from_env(backend='gloo' if args.dry_run else None)
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if args.dry_run:
        return dry_run(args, sdist, rank, world)
    if not torch.cuda.is_available():
        raise RuntimeError('bench.py needs an MI355X (the HIP path has no CPU fallback)')
    dev_index = sdist.device_index(local)                        # HIP_VISIBLE_DEVICES honoured
    torch.cuda.set_device(dev_index)
    dev = torch.device(f'cuda:{dev_index}')
    affinity = sdist.pin_rank_affinity(local, sdist.local_world_size(world))
    from spi_amd import hip
    hip.lib()                                                    # fail loudly if libspi_hip.so is missing
    from spi_amd.configs import hyperparameters, paths_config, global_config
    from spi_amd.training.triplane import TriPlaneGenerator, ffhq512_kwargs
    from spi_amd.training.coaches.rot_bbox_cx_coach import RotBboxCoach
    from spi_amd.training.projectors.common import Projection
    from spi_amd.training.projectors.mirror_projector import mirror_setup
    from spi_amd.training.volumetric_rendering import renderer as rmod
    from spi_amd.data.images_dataset import SyntheticDataset
    import tempfile

    global_config.device = str(dev)
    global_config.enable_fp16_blocks = bool(args.sr_fp16)
    global_config.exploit_sparsity = not args.dense
    if args.no_winograd:
        global_config.conv_winograd = False
    global_config.conv_precision = {'f32': 0, 'bf16x6': 3, 'bf16x3': 2}[args.conv_precision]
    tmp = tempfile.mkdtemp(prefix='spi_bench_')
    for k in ('checkpoints_dir', 'embedding_base_dir', 'experiments_output_dir', 'images_output_dir', 'mirror_images_output_dir'):
        setattr(paths_config, k, f'{tmp}/{k}/')
    pti = args.workload == 'pti'
    hyperparameters.first_inv_type, hyperparameters.first_inv_steps = ('sg' if pti else 'mir'), 500
    hyperparameters.G_1_type, hyperparameters.G_1_step = ('pti' if pti else 'RotBbox'), 1000
    hyperparameters.pt_rot_lambda, hyperparameters.pt_mirror_rot_lambda, hyperparameters.pt_depth_lambda, hyperparameters.pt_tv_lambda = 0.1, 0.05, 1.0, 0.0
    hyperparameters.LPIPS_value_threshold = -1.0                 # random weights: never early-stop inside the timed region

    torch.manual_seed(0)
    G = TriPlaneGenerator(**ffhq512_kwargs(narrow=args.narrow, depth_resolution=args.depth, depth_resolution_importance=args.depth))
    G = G.eval().requires_grad_(False).to(dev)
    G.neural_rendering_resolution = 128
    d = SyntheticDataset(world)[rank]                            # one independent image per rank
    data = {k: (v[None] if torch.is_tensor(v) else v) for k, v in d.items()}
    with contextlib.redirect_stdout(sys.stderr):                 # the coach announces its name like the reference does; stdout carries the JSON line only
        if pti:
            from spi_amd.training.coaches.pti_coach import SingleIDCoach
            from spi_amd.training.projectors.w_projector import sg_distance
            coach = SingleIDCoach(None, False, G=G, synthetic=True)
        else:
            coach = RotBboxCoach(None, False, G=G, synthetic=True)
    if pti:
        # BASELINE configs[2]: stage 1 = the W projector (`sg`: one w for all 14 layers, feature distance of the vgg16 extractor), stage 2 = PTI
        image = data['img'].to(dev).float()
        camera = torch.as_tensor(data['c']).to(dev).float().reshape(-1, 25)
        proj = Projection(coach.G, camera, sg_distance(image, coach._sg_vgg16(), dev), w_mode='w', initial_w=None, num_steps=500, w_avg_samples=600, device=dev)
        w_pivot = proj.w_opt.detach().repeat([1, coach.G.backbone.mapping.num_ws, 1]).clone()
        target_feats = coach.lpips_loss.features(image)
        stage2_step = lambda i: coach.train_step(image, camera, w_pivot, target_feats)
    else:
        ctx = coach.prepare_image(data)
        cameras, dist_fn = mirror_setup(ctx['image'], ctx['camera'], coach.lpips_loss, dev)
        proj = Projection(coach.G, cameras, dist_fn, w_mode='w+', initial_w=None, num_steps=500, w_avg_samples=600, device=dev)
        w_pivot = proj.w_opt.detach().clone()
        stage2_step = lambda i: coach.train_step(i, ctx, w_pivot)
    # set-up, not warm-up: the projector replays its step from a HIP graph that it captures on its second call (one eager step first); a
    # real run pays for that once in 500 steps, so it is built here and neither the W warm-up steps nor the K timed steps contain it
    graph_build_steps = 0
    if global_config.stage1_hip_graph:
        while getattr(proj, '_graph', None) is None and not getattr(proj, '_graph_failed', False) and graph_build_steps < 3:
            proj.step(graph_build_steps)
            graph_build_steps += 1
        torch.cuda.synchronize()

    # set-up, not warm-up: reserve the allocator's pool.  Workspace sizes of the masked branches depend on the iteration's random cameras, so
    # a later iteration can ask the caching allocator for a block it has not seen yet, and a first-time hipMalloc of GBs is a host-side stall.
    # One large block allocated and released here stays in torch's cache and is carved up on demand -- what a long-running inversion service
    # does once at start-up (288 GB of HBM per GPU).  (The one-off pause that was actually measured inside the timed region turned out to be
    # Python's garbage collector, see below; this is the precaution against the other source.)
    pool_gib = sdist.reserve_allocator_pool(dev, int(os.environ.get('SPI_BENCH_POOL_GIB', '24')))      # (the same call run_inversion.run makes)

    marks = {}

    def run(n1, n2, s1_base, s2_base):
        ta = time.perf_counter()
        for i in range(n1):
            proj.step(s1_base + i)
        marks['stage1_host_ms_per_step'] = (time.perf_counter() - ta) / max(n1, 1) * 1e3    # stage 1 never syncs: pure host enqueue cost
        torch.cuda.synchronize()                                 # one sync between the stages: SURVEY 8d asks for both rates separately
        marks['stage1_s'] = time.perf_counter() - ta
        marks['stage1_ms_per_step'] = marks['stage1_s'] / max(n1, 1) * 1e3
        tb = time.perf_counter()
        per_iter = [] if os.environ.get('SPI_BENCH_ITER_TIMES') else None      # debugging aid (synchronises every iteration: NOT for the benchmark value)
        for i in range(n2):
            if stage2_step(s2_base + i)[0]:                      # (a stopped loop runs no further iterations -- timing them would time nothing)
                raise RuntimeError('the stage-2 loop early-stopped inside a timed run (LPIPS <= threshold): not a valid measurement')
            if per_iter is not None:
                torch.cuda.synchronize()
                per_iter.append(round((time.perf_counter() - tb) * 1e3, 2))
        if per_iter:
            print('[bench] stage-2 cumulative ms per iteration:', per_iter, file=sys.stderr)
        torch.cuda.synchronize()
        marks['stage2_s'] = time.perf_counter() - tb
        marks['stage2_ms_per_step'] = marks['stage2_s'] / max(n2, 1) * 1e3

    # set-up, not warm-up (like the stage-1 graph above): the stage-2 loop captures one HIP graph per iteration kind on the SECOND occurrence of
    # that kind (plain: iteration 2, with branches: iteration 4).  A real 1000-iteration run pays the two captures once; here they are built
    # before the W warm-up steps, so neither the warm-up nor the K timed steps contain a capture.
    setup_iters = int(os.environ.get('SPI_BENCH_SETUP_ITERS', '8' if (global_config.stage2_hip_graph and not pti and args.only != 'stage1') else '0'))
    if setup_iters:
        run(0, setup_iters, 0, 400)
    w1, w2 = split_steps(args.warmup)
    k1, k2 = split_steps(args.steps)
    if args.only == 'stage1':
        (w1, w2), (k1, k2) = (args.warmup, 0), (args.steps, 0)
    elif args.only == 'stage2':
        (w1, w2), (k1, k2) = (0, (args.warmup + 3) // 4 * 4), (0, (args.steps + 3) // 4 * 4)
        args.steps = k2
    run(w1, w2, 25, 0)                                           # untimed warm-up (past the 5 % lr ramp-up)
    if os.environ.get('SPI_BENCH_GC_FREEZE', '1') != '0':
        # A full (generation-2) collection of the Python heap -- modules, the generator's parameter objects, autograd nodes -- was measured as a
        # one-off 50-80 ms pause inside the timed region of the FIRST process on a fresh box (`SPI_BENCH_ITER_TIMES=1`: the fifth branch
        # iteration took 150 ms instead of 70; a 24-step run read 37.6 instead of 40.2 it/s; gone with the two lines below).  What a
        # long-running service does: collect once, then move everything alive into the permanent generation so later collections are cheap.
        # A real 1500-iteration job pays such a pause a few times = 0.1 %; only a sub-second benchmark window sees it.
        import gc
        gc.collect()
        gc.freeze()
    sdist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    if os.environ.get('SPI_TORCH_PROFILE'):                      # debugging aid: per-op device time / launch counts per stage (stderr)
        from torch.profiler import profile, ProfilerActivity
        for tag, a1, a2 in (('stage 1', k1, 0), ('stage 2', 0, k2)):
            shapes = os.environ.get('SPI_TORCH_PROFILE') == 'shapes'
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=shapes) as prof:
                run(a1, a2, 25 + w1, ((w2 + 3) // 4) * 4)
                torch.cuda.synchronize()
            if shapes:                                            # which tensors the small ATen launches work on
                print(f'==== {tag}: small ATen ops by input shape', file=sys.stderr)
                for e in sorted(prof.key_averages(group_by_input_shape=True), key=lambda e: -e.count):
                    if e.key in ('aten::fill_', 'aten::mul', 'aten::add', 'aten::copy_', 'aten::add_', 'aten::mul_', 'aten::sum', 'aten::div', 'aten::sub',
                                 'aten::neg', 'aten::cat', 'aten::clone', 'aten::contiguous', 'aten::index_select', 'aten::where', 'aten::mean') and e.count >= 3:
                        print(f'{e.count:6d}  dev {e.device_time_total / 1e3:8.2f} ms  {e.key:20s} {str(e.input_shapes)[:150]}', file=sys.stderr)
            ka = prof.key_averages()
            print(f'==== {tag}: {a1 + a2} steps, by device time', file=sys.stderr)
            print(ka.table(sort_by='cuda_time_total', row_limit=45, max_name_column_width=60), file=sys.stderr)
            print(f'==== {tag}: by number of calls', file=sys.stderr)
            for e in sorted(ka, key=lambda e: -e.count)[:45]:
                print(f'{e.count:7d}  dev {e.device_time_total / 1e3:9.2f} ms  cpu {e.cpu_time_total / 1e3:9.2f} ms  {e.key[:90]}', file=sys.stderr)
        ok = 1.0
    else:
        ok = 1.0
        try:
            run(k1, k2, 25 + w1, ((w2 + 3) // 4) * 4)
        except Exception:                                        # a failing rank still reaches the reduce below: it cannot hang the others
            import traceback
            traceback.print_exc()
            ok = 0.0
    torch.cuda.synchronize()
    dt_rank = time.perf_counter() - t0                           # this rank's own K steps
    dt, rank_s, rank_ok = reduce_run_stats(sdist, rank, world, t0, dt_rank, ok, dev)
    n_ok = int(sum(rank_ok))
    rank_devices = gather_rank_devices(sdist, rank, world, local, dev)
    graph_ranks = int(sdist.reduce_stats([1.0 if getattr(proj, '_graph', None) is not None else 0.0], device=dev)[0])
    # the loops' state after the timed steps: a diverged optimisation (NaN latent / weights) would time different work -- reported, so that
    # such a number cannot pass silently.  (Round 4: with the HIP runtime's graph packet capture on, projector replays that followed eager
    # PTI iterations ran with stale kernel arguments and left a NaN latent; spi_amd/__init__.py switches the capture off.)
    state_finite = {'stage1_latent_and_noise': bool(torch.isfinite(proj.optimizer.flat_p).all()),
                    'stage2_generator': bool(torch.isfinite(coach.optimizer.flat_p).all()) if getattr(coach, 'optimizer', None) is not None else None}
    if not all(v is not False for v in state_finite.values()):
        print(f'[bench] WARNING: non-finite optimisation state after the timed steps: {state_finite}', file=sys.stderr)
    affinity_report = gather_affinity(sdist, rank, world, affinity, dev)
    # per-stage wall time, max over ranks: the two rates `value` is computed from
    st_s = sdist.reduce_stats([marks.get('stage1_s', 0.0), marks.get('stage2_s', 0.0)], device=dev, op='max')

    def mix_value(t1, t2):
        """whole-job iters/s of the configuration's exact 500:1000 stage mix from the measured per-stage times (n_ok ranks, one image each)"""
        if k1 and k2:
            return n_ok * 3.0 / (t1 / k1 + 2.0 * t2 / k2)
        return n_ok * (k1 + k2) / max(t1 + t2, 1e-12)
    value = mix_value(st_s[0], st_s[1])
    main_stage_s = list(st_s)
    step1_next, step2_next = 25 + w1 + k1, ((w2 + 3) // 4) * 4 + ((k2 + 3) // 4) * 4      # where later legs continue the two loops
    sustained = None
    if args.sustained > 0 and ok and k1 and k2 and not os.environ.get('SPI_TORCH_PROFILE'):
        # the same loops for >= 300 more steps in ONE window of several seconds (the K-step window above is < 1 s): stage 1 : stage 2 = 1 : 2,
        # stage 2 in whole super-cycles.  Reported beside `value`; `value` stays the K-step measurement the driver asked for.
        n2s = max(4, int(round(args.sustained * 2 / 3 / 4)) * 4)
        n1s = max(1, n2s // 2)
        main_marks = dict(marks)
        sus_err = None
        sdist.barrier(); torch.cuda.synchronize()
        ts = time.perf_counter()
        try:
            run(n1s, n2s, step1_next, step2_next)
        except Exception as e:                                    # noqa: BLE001
            sus_err = repr(e)
        torch.cuda.synchronize(); sdist.barrier()
        dts = sdist.reduce_stats([time.perf_counter() - ts], device=dev, op='max')[0]
        bad = sdist.reduce_stats([0.0 if sus_err is None else 1.0], device=dev)[0]
        s_s = sdist.reduce_stats([marks.get('stage1_s', 0.0), marks.get('stage2_s', 0.0)], device=dev, op='max')
        step1_next, step2_next = step1_next + n1s, step2_next + n2s
        if bad:
            sustained = {'error': sus_err or 'failed on another rank'}
        else:
            sv = n_ok * 3.0 / (s_s[0] / n1s + 2.0 * s_s[1] / n2s)
            sustained = {'value': sv, 'unit': 'iters/s', 'ms_per_step': 1e3 * max(n_ok, 1) / sv, 'steps': n1s + n2s, 'timed_steps': {'stage1': n1s, 'stage2': n2s},
                         'seconds_max_over_ranks': dts, 'stage1_iters_per_s_per_gpu': n1s / s_s[0], 'stage2_iters_per_s_per_gpu': n2s / s_s[1],
                         'note': 'same definition as `value` (exact 1:2 mix of the per-stage rates, max over ranks), measured over one window of several seconds '
                                 'right after the K timed steps, barrier + synchronize on both sides'}
        marks.clear(); marks.update(main_marks)
    # ---- roofline instrumentation.  Both stages of the timed region are HIP-graph replays (round 4: stage 2 too), and a replayed launch cannot be
    # bracketed by HIP events.  So the SAME steps run once more right here, eagerly enqueued (graphs switched off for this pass only), with HIP events
    # on the launch stream around every march / decoder launch: same kernels, same shapes, same process, straight after the timed region.
    events = bwd_events = dfwd_events = dbwd_events = None
    if ok and not os.environ.get('SPI_TORCH_PROFILE'):
        g1, g2 = global_config.stage1_hip_graph, global_config.stage2_hip_graph
        main_marks = dict(marks)
        try:
            if hasattr(coach, 'drain_pipeline'):
                coach.drain_pipeline()
            torch.cuda.synchronize()
            global_config.stage1_hip_graph = global_config.stage2_hip_graph = False
            rmod.MARCH_EVENTS, rmod.MARCH_BWD_EVENTS, rmod.DECODE_FWD_EVENTS, rmod.DECODE_BWD_EVENTS = [], [], [], []
            n2i = min(((k2 + 3) // 4) * 4, 8) if k2 else 0
            run(min(k1, 4), n2i, step1_next, step2_next)
            torch.cuda.synchronize()
            step1_next, step2_next = step1_next + min(k1, 4), step2_next + n2i
        except Exception:                                        # noqa: BLE001  (instrumentation must not cost the benchmark line)
            import traceback
            traceback.print_exc()
        finally:
            events, rmod.MARCH_EVENTS = rmod.MARCH_EVENTS, None
            bwd_events, rmod.MARCH_BWD_EVENTS = rmod.MARCH_BWD_EVENTS, None
            dfwd_events, rmod.DECODE_FWD_EVENTS = rmod.DECODE_FWD_EVENTS, None
            dbwd_events, rmod.DECODE_BWD_EVENTS = rmod.DECODE_BWD_EVENTS, None
            global_config.stage1_hip_graph, global_config.stage2_hip_graph = g1, g2
            marks.clear(); marks.update(main_marks)
    events = events or []
    state_finite['stage1_after_the_eager_instrumentation_pass'] = bool(torch.isfinite(proj.optimizer.flat_p).all())
    alt = None
    if args.alt_conv_precision != 'none' and args.alt_conv_precision != args.conv_precision and ok and not pti and not os.environ.get('SPI_TORCH_PROFILE'):
        # the same K steps once more with the split-bf16 convolutions (opt-in arithmetic; reported beside the benchmark value, never as it).
        # A failure here must not cost the benchmark line: the collectives below run on every rank either way.
        global_config.conv_precision = {'bf16x6': 3, 'bf16x3': 2}[args.alt_conv_precision]
        main_marks = dict(marks)
        alt_err = None
        try:
            leg_warm = 8 if global_config.stage2_hip_graph else 4   # (both stage-2 graphs are re-captured for another arithmetic: second occurrence of each kind)
            run(2 if k1 else 0, leg_warm if k2 else 0, step1_next, step2_next)      # (2 stage-1 steps: the projector re-captures its graph for this arithmetic)
        except Exception as e:                                    # noqa: BLE001
            alt_err = repr(e)
        sdist.barrier(); torch.cuda.synchronize()
        ta = time.perf_counter()
        try:
            if alt_err is None:
                run(k1, k2, step1_next + 2, step2_next + leg_warm)
        except Exception as e:                                    # noqa: BLE001
            alt_err = repr(e)
        torch.cuda.synchronize(); sdist.barrier()
        dta = sdist.reduce_stats([time.perf_counter() - ta], device=dev, op='max')[0]
        bad = sdist.reduce_stats([0.0 if alt_err is None else 1.0], device=dev)[0]
        alt_s = sdist.reduce_stats([marks.get('stage1_s', 0.0), marks.get('stage2_s', 0.0)], device=dev, op='max')
        alt = {'conv_precision': args.alt_conv_precision, 'error': alt_err or 'failed on another rank'} if bad else {'conv_precision': args.alt_conv_precision, 'value': mix_value(alt_s[0], alt_s[1]), 'unit': 'iters/s', 'ms_per_step': 1e3 * world / mix_value(alt_s[0], alt_s[1]),
               'timed_region_s': dta,
               'stage1_mir_iters_per_s_per_gpu': 1e3 / marks['stage1_ms_per_step'] if k1 else None,
               'stage2_rotbbox_iters_per_s_per_gpu': 1e3 / marks['stage2_ms_per_step'] if k2 else None,
               'note': 'same K steps, dense convs with fp32 operands split into bf16 pieces on the bf16 matrix cores (fp32 accumulate; bf16x6: the large 3x3 forward / dgrad passes stay on the fp32 Winograd kernel, which is at least as precise and faster there); '
                       'bf16x6 = 3 pieces / 6 products, error ~2^-23 per product, passes the conv parity tests at the exact kernels\' tolerance; '
                       'NOT the benchmark value'}
        marks.clear(); marks.update(main_marks)
        global_config.conv_precision = {'f32': 0, 'bf16x6': 3, 'bf16x3': 2}[args.conv_precision]
    wino_f2 = None
    if global_config.conv_winograd and global_config.conv_winograd_f4 and not args.no_wino_f2_leg and ok and not pti and not os.environ.get('SPI_TORCH_PROFILE'):
        # the same K steps once more with F(2x2, 3x3) Winograd everywhere (round 5's arithmetic: ~2e-6 of the range per layer instead of F(4x4)'s ~4e-5):
        # reported beside the benchmark value so that the precision trade of the >= 256^2 layers is visible; never `value`.
        global_config.conv_winograd_f4 = False
        main_marks = dict(marks)
        f2_err = None
        try:
            leg_warm = 8 if global_config.stage2_hip_graph else 4
            run(2 if k1 else 0, leg_warm if k2 else 0, step1_next, step2_next)      # graphs are re-captured for another arithmetic
            sdist.barrier(); torch.cuda.synchronize()
            run(k1, k2, step1_next + 2, step2_next + leg_warm)
        except Exception as e:                                    # noqa: BLE001
            f2_err = repr(e)
        torch.cuda.synchronize(); sdist.barrier()
        bad = sdist.reduce_stats([0.0 if f2_err is None else 1.0], device=dev)[0]
        f2_s = sdist.reduce_stats([marks.get('stage1_s', 0.0), marks.get('stage2_s', 0.0)], device=dev, op='max')
        wino_f2 = {'error': f2_err or 'failed on another rank'} if bad else {
            'value': mix_value(f2_s[0], f2_s[1]), 'unit': 'iters/s',
            'stage1_mir_iters_per_s_per_gpu': 1e3 / marks['stage1_ms_per_step'] if k1 else None,
            'stage2_rotbbox_iters_per_s_per_gpu': 1e3 / marks['stage2_ms_per_step'] if k2 else None,
            'note': 'same K steps with global_config.conv_winograd_f4 = False: F(2x2, 3x3) minimal filtering on every Winograd layer (the >= 256^2 layers of `value` run '
                    'F(4x4, 3x3): 1.78x fewer multiplications, result within ~4e-5 of the range of the direct sum instead of ~2e-6; both fp32 operands and accumulation); '
                    'NOT the benchmark value'}
        marks.clear(); marks.update(main_marks)
        global_config.conv_winograd_f4 = True
    dense_leg = None
    if not args.dense and not args.no_dense_leg and ok and k2 and not pti and not os.environ.get('SPI_TORCH_PROFILE'):
        # the dense bound of the same step: the stage-2 iterations once more with the data-driven skipping of exactly-zero gradients /
        # unneeded SR tiles switched off (stage 1 has no masked branch: its rate is the main run's).  Reported beside `value`, never as it.
        global_config.exploit_sparsity = False
        main_marks = dict(marks)
        dense_err = None
        base2 = step2_next + ((k2 + 11) // 4) * 4
        try:
            leg_warm = 8 if global_config.stage2_hip_graph else 4
            run(0, leg_warm, 0, base2)                            # untimed super-cycle(s): allocator / workspace shapes of the dense branches, graph re-capture
            sdist.barrier(); torch.cuda.synchronize()
            run(0, k2, 0, base2 + leg_warm)
        except Exception as e:                                    # noqa: BLE001
            dense_err = repr(e)
        torch.cuda.synchronize(); sdist.barrier()
        d_s = sdist.reduce_stats([marks.get('stage2_s', 0.0)], device=dev, op='max')[0]
        bad = sdist.reduce_stats([0.0 if dense_err is None else 1.0], device=dev)[0]
        dense_leg = {'error': dense_err or 'failed on another rank'} if bad else {
            'value': mix_value(main_stage_s[0], d_s), 'unit': 'iters/s', 'stage2_rotbbox_iters_per_s_per_gpu': k2 / d_s,
            'stage1_mir_iters_per_s_per_gpu': (k1 / main_stage_s[0]) if k1 else None,
            'note': 'dense bound: every ray, gradient segment and SR tile of the masked rot / mirror-rot branches processed (global_config.exploit_sparsity '
                    '= False); same results (tested); stage 1 has no masked branch, its time is the main run\'s; NOT the benchmark value'}
        marks.clear(); marks.update(main_marks)
        global_config.exploit_sparsity = True
    march_ms = [a.elapsed_time(b) for a, b, _ in events]
    march_rays = [r for _, _, r in events]

    if rank == 0:
        S = 2 * args.depth
        per_ray = RAYMARCH_BYTES_PER_RAY(S)
        tot_bytes = sum(march_rays) * per_ray
        tot_s = sum(march_ms) / 1e3
        achieved = tot_bytes / tot_s / 1e9 if tot_s > 0 else 0.0
        traffic = None                                           # HBM bytes per (average) launch from the committed PMC passes
        pmc = os.path.join(ROOT, 'profiles', 'raymarch_pmc.json')
        if os.path.exists(pmc) and march_rays:
            traffic = json.load(open(pmc)).get('hbm_bytes_per_16384_rays') / 16384.0 * (sum(march_rays) / len(march_rays))
        out = {
            'metric': f'SPI inversion iters/sec (512^2, {args.depth}+{args.depth} ray samples)', 'value': value, 'unit': 'iters/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * max(n_ok, 1) / value,
            'value_definition': 'exact 500:1000 (1:2) mix of the two per-stage rates measured in the timed region, max over ranks per stage: '
                                'n_gpus * 3 / (1/r_stage1 + 2/r_stage2); independent of how K splits into stages',
            'timed_region': {'steps': args.steps, 'seconds_max_over_ranks': dt, 'raw_iters_per_s_at_this_k_split': n_ok * args.steps / dt,
                             'stage1_seconds_max_over_ranks': main_stage_s[0], 'stage2_seconds_max_over_ranks': main_stage_s[1]},
            'stage1_host_enqueue_ms_per_step': marks.get('stage1_host_ms_per_step'),
            'ranks': {'launched': world, 'completed': n_ok, 'backend': 'rccl (torch.distributed nccl)' if world > 1 else 'none (single process)',
                      'collectives': 'barrier + 2 all-reduces of <= %d fp64 (timing / done-flags); no data-path collective' % (2 * world),
                      'per_rank_iters_per_s': [args.steps / t if t > 0 else None for t in rank_s], 'devices': rank_devices,
                      'process_group': process_group_info(world), 'ranks_replaying_stage1_graph': graph_ranks, 'cpu_affinity': affinity_report},
            'stages': ({'stage1_sg_iters_per_s_per_gpu': 1e3 / marks['stage1_ms_per_step'] if k1 else None,
                        'stage2_pti_iters_per_s_per_gpu': 1e3 / marks['stage2_ms_per_step'] if k2 else None, 'note': 'rank 0'} if pti else
                       {'stage1_mir_iters_per_s_per_gpu': 1e3 / marks['stage1_ms_per_step'] if k1 else None,
                        'stage2_rotbbox_iters_per_s_per_gpu': 1e3 / marks['stage2_ms_per_step'] if k2 else None,
                        'note': 'rank 0; stage 2 amortises the every-4th-iteration rot / mirror-rot / depth branches over whole super-cycles'}),
            'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': ('f32+f16sr' if args.sr_fp16 else 'f32') + ('' if args.conv_precision == 'f32' else f' (convolutions: fp32 operands split {args.conv_precision}, fp32 accumulate)'), 'data': 'synthetic (seeded 512^2 image / camera / mask / landmarks; '
            'random-init weights of the ffhqrebalanced512-128 architecture)',
            'config': {'workload': (('configs[2]: PTI baseline, 1 image per GPU, first_inv_type=sg (500) + G_1_type=pti (1000), 512^2, ' if pti else
                                    ('configs[4]' if (args.depth == 128 and args.sr_fp16) else 'configs[1]') +
                                    ': 1 image per GPU, first_inv_type=mir (500) + G_1_type=RotBbox (1000), 512^2, ') +
                                   f'{args.depth}+{args.depth} samples' + (', fp16 MFMA super-resolution' if args.sr_fp16 else '')), 'step_mix': {'value_mix': 'stage1:stage2 = 1:2 exact (500:1000), computed from the per-stage rates' if (k1 and k2) else 'single stage', 'timed_steps': {'stage1_mir': k1, 'stage2_rotbbox': k2}},
                       'parallelism': f'{world} independent image(s), no data-path collective', 'narrow_debug_model': bool(args.narrow),
                       'only_stage': args.only, 'stage1_hip_graph': bool(global_config.stage1_hip_graph and getattr(proj, '_graph', None) is not None),
                       'state_finite_after_timed_steps': state_finite, 'hip_graph_packet_capture_env': os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE'), 'hip_graph_self_test': __import__('spi_amd').hip_graphs_status()['self_test'], 'stage1_graph_build_steps_before_warmup': graph_build_steps, 'allocator_pool_reserved_gib_before_warmup': pool_gib, 'python_gc_frozen_after_warmup': os.environ.get('SPI_BENCH_GC_FREEZE', '1') != '0', 'stage2_hip_graph': bool(global_config.stage2_hip_graph and not pti and getattr(coach, '_g2', None) is not None and not getattr(coach, '_graph_failed', False)),
                       'stage2_graph_build_iterations_before_warmup': setup_iters,
                       'conv3x3': ('Winograd F(2x2,3x3) forward / dgrad on the >= 128^2 layers (fp32 operands and accumulation), implicit GEMM elsewhere'
                                   if global_config.conv_winograd and global_config.conv_precision in (0, 3) else 'implicit GEMM'),
                       'sparsity': 'dense (every ray / gradient segment / SR tile processed; NOT the benchmark configuration)' if args.dense else
                                   'data-driven skipping of exactly-zero gradients and unneeded SR tiles in the masked pseudo-view branches (result-identical)'},
            'roofline': {'kernel': 'raymarch_fwd_kernel<3> (final composite, S=%d, C=32)' % S, 'bound': 'hbm', 'achieved': achieved,
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'launches': len(march_ms), 'avg_launch_us': (sum(march_ms) / max(len(march_ms), 1)) * 1e3,
                         'bytes_per_ray': per_ray, 'rays_per_launch': (sum(march_rays) / max(len(march_rays), 1)),
                         'note': 'HIP events on the launch stream around every final-march launch of an eagerly enqueued repeat of the timed steps, run in this '
                                 'process right after the timed region (the timed steps themselves are HIP-graph replays, which cannot carry events); '
                                 'stage 1 (N = 2) and stage 2 (N = 1 main view, N = 4 pseudo-view branches) launches',
                         'traffic_source': 'profiles/raymarch_pmc.json: FETCH_SIZE (doubled, MI355X_MICROARCH.md) + WRITE_SIZE of separate --pmc passes over this kernel '
                                           '(round 4; the kernel has not changed since), scaled from 16 384 rays to this run\'s average launch -- a committed counter record, not re-collected in this process'},
        }
        # second HBM line: the march BACKWARD.  Algorithmic bytes per ACTIVE ray: read S*(C+2)*4 (colours, density, depth)
        # + (C+1)*4 incoming gradients, write S*2*4 (density gradient + colour-gradient scale; the [R,S,C] colour gradient is never
        # materialised).  Rays whose incoming gradient is exactly zero are only flagged (one 128-B read): counted as 0 bytes.
        if bwd_events:
            bwd_ms = [a.elapsed_time(b) for a, b, _, _ in bwd_events]
            act = [int(f.sum().item()) if f is not None else r for _, _, r, f in bwd_events]
            per_ray_b = S * 34 * 4 + 33 * 4 + S * 2 * 4
            ach = sum(act) * per_ray_b / (sum(bwd_ms) / 1e3) / 1e9
            dense = [(m, a) for m, a, (_, _, r, _) in zip(bwd_ms, act, bwd_events) if a == r]
            masked = [(m, a, r) for m, a, (_, _, r, _) in zip(bwd_ms, act, bwd_events) if a != r]

            def part(rows):
                by = sum(a for _, a in rows) * per_ray_b
                return {'launches': len(rows), 'achieved': by / (sum(m for m, _ in rows) / 1e3) / 1e9, 'frac': by / (sum(m for m, _ in rows) / 1e3) / 1e9 / HBM_PEAK_GBS,
                        'avg_launch_us': sum(m for m, _ in rows) / len(rows) * 1e3} if rows else None
            out['roofline_march_bwd'] = {'kernel': 'raymarch_bwd_kernel<%d> (S=%d, C=32)' % ((S + 63) // 64, S), 'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS,
                                         'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS, 'launches': len(bwd_ms), 'avg_launch_us': sum(bwd_ms) / len(bwd_ms) * 1e3,
                                         'bytes_per_active_ray': per_ray_b, 'active_rays_per_launch': sum(act) / len(act),
                                         'dense_launches_only': part(dense),
                                         # the pseudo-view branches' launches (rays with an all-zero incoming gradient are flagged and skipped, 0 bytes): bytes of the live rays
                                         'masked_launches_only': (dict(part([(m, a) for m, a, _ in masked]), live_ray_share=sum(a for _, a, _ in masked) / sum(r for _, _, r in masked))
                                                                  if masked else None),
                                         'note': 'depth-only launches (SPI depth branch: no colour gradient, no colour rows read) carry no march event and are not in this line'}
        # the fused gather + decoder forward is bound by cache-level gather bandwidth + VALU, not by HBM: report the effective gather
        # rate (12 corner rows x 128 B per point, served by L1 / L2 / MALL) against the L2 figure of MI355X_MICROARCH.md and the decoder's
        # vector FLOP rate against the fp32 vector peak.
        if dfwd_events:
            ms = [a.elapsed_time(b) for a, b, _ in dfwd_events]
            pts = [p for _, _, p in dfwd_events]
            gbs = sum(pts) * 1536 / (sum(ms) / 1e3) / 1e9
            tf = sum(pts) * 8320 / (sum(ms) / 1e3) / 1e12
            out['roofline_gather'] = {'kernel': 'decode_fwd_kernel (tri-plane gather + 32-64-33 decoder, colour rows written)', 'bound': 'l2 gather + valu',
                                      'achieved': gbs, 'peak': 34500.0, 'unit': 'GB/s', 'frac': gbs / 34500.0,
                                      'valu': {'achieved': tf, 'peak': 157.3, 'unit': 'TFLOP/s', 'frac': tf / 157.3},
                                      'bytes_per_point_gathered': 1536, 'flop_per_point': 8320, 'launches': len(ms),
                                      'avg_launch_us': sum(ms) / len(ms) * 1e3, 'points_per_launch': sum(pts) / len(pts),
                                      'hbm_compulsory_bytes_per_point': 128 + 4 + 4,
                                      'note': 'effective gather bandwidth at cache level (SURVEY 8d), peak = aggregate L2 bandwidth (MI355X_MICROARCH.md); compulsory HBM traffic '
                                              'is the 128-B colour row + density written per point (planes stay cache-resident)'}
        # the tiled decoder backward (bin_points + decode_bwd_tiled + partial reduce): fp32 MFMA.  `achieved` counts the ALGORITHMIC FLOPs per live
        # point (layer-1 recompute 4096, dY -> dH 4224, dH -> dF 4096; + dW1 4096 + dW2 4224 with decoder gradients); rays whose incoming gradient
        # is exactly zero are skipped and count as 0.
        if dbwd_events:
            ms = [a.elapsed_time(b) for a, b, *_ in dbwd_events]
            live = [(int(f.sum().item()) if f is not None else r) * s_ for _, _, r, s_, f, _, _ in dbwd_events]
            flop = [lv * ((12416 if rgb else 8320) + ((8320 if rgb else 4096) if wg else 0)) for lv, (_, _, _, _, _, wg, rgb) in zip(live, dbwd_events)]
            tf = sum(flop) / (sum(ms) / 1e3) / 1e12
            full = [(m, f) for m, f, lv, e in zip(ms, flop, live, dbwd_events) if lv == e[2] * e[3] and e[5] and e[6]]
            out['roofline_decode_bwd'] = {'kernel': 'spi_triplane_decode_bwd_sorted (bin_points_kernel + decode_bwd_tiled_kernel + decoder_partial_reduce_kernel)',
                                          'bound': 'mfma', 'achieved': tf, 'peak': 157.3, 'unit': 'TFLOP/s', 'frac': tf / 157.3, 'launches': len(ms),
                                          'avg_launch_us': sum(ms) / len(ms) * 1e3, 'live_points_per_launch': sum(live) / len(live),
                                          'dense_wgrad_launches_only': ({'launches': len(full), 'avg_launch_us': sum(m for m, _ in full) / len(full) * 1e3,
                                                                         'achieved': sum(f for _, f in full) / (sum(m for m, _ in full) / 1e3) / 1e12,
                                                                         'frac': sum(f for _, f in full) / (sum(m for m, _ in full) / 1e3) / 1e12 / 157.3} if full else None),
                                          'note': 'algorithmic FLOPs of the decoder backward per live point / call time (gather, MLP on the matrix cores, plane-gradient scatter '
                                                  'all inside the call); the scatter is LDS-atomic / flush bound, see DESIGN.md 3'}
        out['roofline_mfma'] = conv_roofline(dev, bool(args.sr_fp16), global_config.conv_precision)
        try:
            out['roofline_elementwise'] = elementwise_roofline(dev)
        except Exception as e:                                    # noqa: BLE001
            out['roofline_elementwise'] = {'error': repr(e)}
        if sustained is not None:
            out['sustained'] = sustained
        if alt is not None:
            out['alt'] = alt
        if wino_f2 is not None:
            out['wino_f2'] = wino_f2
        if dense_leg is not None:
            out['dense'] = dense_leg
        mb = (out.get('roofline_march_bwd') or {}).get('masked_launches_only') or {}
        out['config']['data_driven_skipping'] = {
            'on': bool(global_config.exploit_sparsity), 'what': 'exactly-zero gradient rays / 16-pixel gradient segments / unneeded SR tiles of the masked pseudo-view branches are not processed (result-identical)',
            'masked_march_bwd_live_ray_share': mb.get('live_ray_share'), 'dense_value_iters_per_s': (dense_leg or {}).get('value'),
            'note': 'the live share comes from the synthetic mask and random-init weights; `dense` is the same step with the skipping off = the robust lower bound of `value`'}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out['cpu_baseline'] = cpu_baseline(args.depth, args.narrow, args.cpu_baseline, 1 if k1 else 0, 2 if k2 else 0)
                if args.cpu_baseline == 'full':
                    # the once-per-round record (profiles/cpu_baseline_full.json): the full CPU protocol with the GPU value of THIS process beside it
                    rec = dict(out['cpu_baseline'], gpu_value_same_process_iters_per_s=value,
                               gpu_sustained_same_process_iters_per_s=(sustained or {}).get('value'), command='python bench.py ' + ' '.join(sys.argv[1:]))
                    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
                    with open(os.path.join(ROOT, 'gpurun_out', 'cpu_baseline_full.json'), 'w') as f:
                        json.dump(rec, f, indent=1)
            except Exception as e:                                # noqa: BLE001  (a host-side failure must not lose the GPU measurement)
                out['cpu_baseline'] = {'error': repr(e)}
        default_cfg = (not pti and args.depth == 96 and not args.sr_fp16 and not args.narrow and not args.dense and args.only is None
                       and args.conv_precision == 'f32' and not args.no_winograd)
        if world == 1 and not args.no_legs and default_cfg and ok:
            # BASELINE configs[4] and configs[2] measured the same way, each as its own bench.py process (VERDICT r03 missing #3)
            del proj, coach
            torch.cuda.empty_cache()
            out['cfg4'] = child_leg(['--depth', '128', '--sr-fp16'], args.steps, args.warmup)
            out['pti'] = child_leg(['--workload', 'pti'], args.steps, args.warmup)
            out['whole_job'] = whole_job_leg((out['stages'].get('stage1_mir_iters_per_s_per_gpu'), out['stages'].get('stage2_rotbbox_iters_per_s_per_gpu')))
            if isinstance(out['whole_job'], dict) and out['whole_job'].get('first_image'):
                out['first_image'] = out['whole_job'].pop('first_image')
        print(json.dumps(out), flush=True)
    sdist.shutdown()                                             # ranks leave together (rank 0 is still timing its roofline lines)
    if n_ok != world:
        sys.exit(3)


if __name__ == '__main__':
    main()
