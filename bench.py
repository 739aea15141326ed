#!/usr/bin/env python3
"""Benchmark of the SPI inversion hot path on MI355X (contract: see the task description / DESIGN.md).

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: launched through torch.distributed.run)

Workload (BASELINE.json configs[1]): 1 image per GPU, first_inv_type=mir (500 steps) + G_1_type=RotBbox (1000 steps),
EG3D ffhqrebalanced512-128 architecture at 512^2 with 96 coarse + 96 fine samples per ray, fp32, synthetic inputs
and seeded random-init weights (no checkpoint / dataset exists offline).  A "step" is one iteration of the hot path;
the K timed steps keep the configuration's 1:2 mix of stage-1 ('mir') and stage-2 ('RotBbox') iterations, with the
stage-2 part a whole number of 4-iteration super-cycles so the every-4th-step branches are amortised exactly.
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
    ap.add_argument('--sr-fp16', action='store_true', help='NOT the benchmark configuration: fp16 MFMA in the super-resolution blocks '
                                                           '(BASELINE config 5); the JSON line then says dtype f32+f16sr')
    ap.add_argument('--narrow', action='store_true', help='debug: reduced-width generator (NOT the benchmark configuration)')
    return ap.parse_args()


def split_steps(k):
    """K steps -> (stage-1 steps, stage-2 steps) in the 500:1000 proportion, stage-2 a multiple of 4."""
    k2 = max(4, int(round(k * 2 / 3 / 4)) * 4) if k >= 6 else max(0, k - k // 3)
    k2 = min(k2, k)
    return k - k2, k2


def cpu_baseline(depth, narrow):
    """Oracle (CPU restatement of the reference, oracle/loops_ref.py) timed on this box's host cores:
    ONE stage-2 main-branch iteration (G.synthesis fwd+bwd, L2 + LPIPS, Adam over all G parameters)."""
    from oracle import loops_ref as olp, losses_ref as olo, renderer_ref as orr
    from spi_amd.training.triplane import TriPlaneGenerator, ffhq512_kwargs
    from spi_amd.data.images_dataset import SyntheticDataset
    cores = min(os.cpu_count() or 1, 32)        # torch's CPU kernels stop scaling (and thrash) far below the box's 256 hardware threads
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    G = TriPlaneGenerator(**ffhq512_kwargs(narrow=narrow))
    P = {k: v.detach().clone() for k, v in G.state_dict().items()}
    pnames = [k for k, _ in G.named_parameters()]
    del G
    st = olp.Stage2State(P, pnames)
    W = olo.make_vgg16_weights(seed=0)
    d = SyntheticDataset(1)[0]
    data = dict(img=d['img'][None], c=torch.as_tensor(d['c']).reshape(1, 25))
    opts = dict(orr.DEFAULT_RENDERING, depth_resolution=depth, depth_resolution_importance=depth)
    w = torch.randn(1, 14, 512) * 0.5
    t0 = time.perf_counter()
    olp.stage2_iteration(st, 1, data, w, opts, lambda a, b: olo.lpips(W, a, b), None, pti_only=True)
    dt = time.perf_counter() - t0
    return dict(value=1.0 / dt, unit='iters/s', cores=cores, kind='port',
                sample=f'1 stage-2 main-branch iteration (1 synthesis fwd+bwd at 512^2 / {depth}+{depth} samples, L2+LPIPS, Adam over '
                       f'all G parameters; the every-4th-step rot / mirror-rot / depth branches are NOT in the sample), {dt:.1f} s, '
                       f'torch {torch.__version__} CPU fp32, {cores} threads of {os.cpu_count()}')


def conv_roofline(dev, f16):
    """Second roofline line: the matrix-core kernels that hold most of the step's GPU time (igemm_kernel / wgrad_kernel).  Times the
    largest convolution of the loop -- superresolution b512.conv1, 128 -> 128, 3x3 at 512^2, one image -- with HIP events on the
    launch stream, outside the timed region: forward, data gradient, weight gradient."""
    from spi_amd.torch_utils.ops import conv2d_mfma as cm
    from spi_amd import hip
    import ctypes
    n, i, o, h, k = 1, 128, 128, 512, 3
    x = torch.randn(n, i, h, h, device=dev)
    w = torch.randn(n, o, k, k, i, device=dev) * 0.03
    y = torch.empty(n, o, h, h, device=dev)
    dx, dw = torch.empty_like(x), torch.empty_like(w)
    d = cm._desc(n, i, o, h, h, k, 1, False, True, o * i * k * k, tap_major=1, f16=int(f16))
    flop = 2.0 * n * o * i * k * k * h * h
    peak = 2500.0 if f16 else 157.3                              # dense MFMA peaks (TFLOP/s): fp16 / fp32, MI355X_MICROARCH.md
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
    return {'kernel': 'igemm_kernel / wgrad_kernel on SR b512.conv1 (128->128, 3x3, 512^2, N=1)', 'bound': 'mfma', 'achieved': worst, 'peak': peak,
            'unit': 'TFLOP/s', 'frac': worst / peak, 'flop_per_launch': flop, 'passes': res,
            'note': 'achieved = slowest of the three passes; wgrad includes its memset of dw'}


def main():
    args = parse()
    from spi_amd import dist as sdist
    rank, world, local = sdist.init_from_env()
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}'
    if not torch.cuda.is_available():
        raise RuntimeError('bench.py needs an MI355X (the HIP path has no CPU fallback)')
    torch.cuda.set_device(local)
    dev = torch.device(f'cuda:{local}')
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
    tmp = tempfile.mkdtemp(prefix='spi_bench_')
    for k in ('checkpoints_dir', 'embedding_base_dir', 'experiments_output_dir', 'images_output_dir', 'mirror_images_output_dir'):
        setattr(paths_config, k, f'{tmp}/{k}/')
    hyperparameters.first_inv_type, hyperparameters.first_inv_steps = 'mir', 500
    hyperparameters.G_1_type, hyperparameters.G_1_step = 'RotBbox', 1000
    hyperparameters.pt_rot_lambda, hyperparameters.pt_mirror_rot_lambda, hyperparameters.pt_depth_lambda, hyperparameters.pt_tv_lambda = 0.1, 0.05, 1.0, 0.0
    hyperparameters.LPIPS_value_threshold = -1.0                 # random weights: never early-stop inside the timed region

    torch.manual_seed(0)
    G = TriPlaneGenerator(**ffhq512_kwargs(narrow=args.narrow, depth_resolution=args.depth, depth_resolution_importance=args.depth))
    G = G.eval().requires_grad_(False).to(dev)
    G.neural_rendering_resolution = 128
    with contextlib.redirect_stdout(sys.stderr):                 # the coach announces its name like the reference does; stdout carries the JSON line only
        coach = RotBboxCoach(None, False, G=G)
    d = SyntheticDataset(world)[rank]                            # one independent image per rank
    data = {k: (v[None] if torch.is_tensor(v) else v) for k, v in d.items()}
    ctx = coach.prepare_image(data)
    cameras, dist_fn = mirror_setup(ctx['image'], ctx['camera'], coach.lpips_loss, dev)
    proj = Projection(coach.G, cameras, dist_fn, w_mode='w+', initial_w=None, num_steps=500, w_avg_samples=600, device=dev)
    w_pivot = proj.w_opt.detach().clone()

    marks = {}

    def run(n1, n2, s1_base, s2_base):
        ta = time.perf_counter()
        for i in range(n1):
            proj.step(s1_base + i)
        marks['stage1_host_ms_per_step'] = (time.perf_counter() - ta) / max(n1, 1) * 1e3    # stage 1 never syncs: pure host enqueue cost
        torch.cuda.synchronize()                                 # one sync between the stages: SURVEY 8d asks for both rates separately
        marks['stage1_ms_per_step'] = (time.perf_counter() - ta) / max(n1, 1) * 1e3
        tb = time.perf_counter()
        for i in range(n2):
            coach.train_step(s2_base + i, ctx, w_pivot)
        torch.cuda.synchronize()
        marks['stage2_ms_per_step'] = (time.perf_counter() - tb) / max(n2, 1) * 1e3

    w1, w2 = split_steps(args.warmup)
    k1, k2 = split_steps(args.steps)
    run(w1, w2, 25, 0)                                           # untimed warm-up (past the 5 % lr ramp-up)
    rmod.MARCH_EVENTS = []                                       # HIP events around every final-march launch in the timed region
    sdist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    if os.environ.get('SPI_TORCH_PROFILE'):                      # debugging aid: per-op device time / launch counts per stage (stderr)
        from torch.profiler import profile, ProfilerActivity
        for tag, a1, a2 in (('stage 1', k1, 0), ('stage 2', 0, k2)):
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
                run(a1, a2, 25 + w1, ((w2 + 3) // 4) * 4)
                torch.cuda.synchronize()
            ka = prof.key_averages()
            print(f'==== {tag}: {a1 + a2} steps, by device time', file=sys.stderr)
            print(ka.table(sort_by='cuda_time_total', row_limit=45, max_name_column_width=60), file=sys.stderr)
            print(f'==== {tag}: by number of calls', file=sys.stderr)
            for e in sorted(ka, key=lambda e: -e.count)[:45]:
                print(f'{e.count:7d}  dev {e.device_time_total / 1e3:9.2f} ms  cpu {e.cpu_time_total / 1e3:9.2f} ms  {e.key[:90]}', file=sys.stderr)
    else:
        run(k1, k2, 25 + w1, ((w2 + 3) // 4) * 4)
    t_enq = time.perf_counter() - t0                             # host time to enqueue the K steps (== dt when host-bound)
    torch.cuda.synchronize(); sdist.barrier()
    dt = time.perf_counter() - t0
    events, rmod.MARCH_EVENTS = rmod.MARCH_EVENTS, None
    dt = sdist.reduce_stats([dt], device=dev, op='max')[0]
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
            'metric': f'SPI inversion iters/sec (512^2, {args.depth}+{args.depth} ray samples)', 'value': world * args.steps / dt, 'unit': 'iters/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'host_enqueue_ms_per_step': t_enq / args.steps * 1e3,
            'stage1_host_enqueue_ms_per_step': marks.get('stage1_host_ms_per_step'),
            'stages': {'stage1_mir_iters_per_s_per_gpu': 1e3 / marks['stage1_ms_per_step'] if k1 else None,
                       'stage2_rotbbox_iters_per_s_per_gpu': 1e3 / marks['stage2_ms_per_step'] if k2 else None,
                       'note': 'rank 0; stage 2 amortises the every-4th-iteration rot / mirror-rot / depth branches over whole super-cycles'},
            'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32+f16sr' if args.sr_fp16 else 'f32', 'data': 'synthetic (seeded 512^2 image / camera / mask / landmarks; '
            'random-init weights of the ffhqrebalanced512-128 architecture)',
            'config': {'workload': ('configs[4]' if (args.depth == 128 and args.sr_fp16) else 'configs[1]') +
                                   ': 1 image per GPU, first_inv_type=mir (500) + G_1_type=RotBbox (1000), 512^2, '
                                   f'{args.depth}+{args.depth} samples' + (', fp16 MFMA super-resolution' if args.sr_fp16 else ''), 'step_mix': {'stage1_mir': k1, 'stage2_rotbbox': k2},
                       'parallelism': f'{world} independent image(s), no data-path collective', 'narrow_debug_model': bool(args.narrow)},
            'roofline': {'kernel': 'raymarch_fwd_kernel<3> (final composite, S=%d, C=32)' % S, 'bound': 'hbm', 'achieved': achieved,
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'launches': len(march_ms), 'avg_launch_us': (sum(march_ms) / max(len(march_ms), 1)) * 1e3,
                         'bytes_per_ray': per_ray, 'rays_per_launch': (sum(march_rays) / max(len(march_rays), 1))},
        }
        out['roofline_mfma'] = conv_roofline(dev, bool(args.sr_fp16))
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.depth, args.narrow)
        print(json.dumps(out), flush=True)
    sdist.shutdown()                                             # ranks leave together (rank 0 is still timing its roofline lines)


if __name__ == '__main__':
    main()
