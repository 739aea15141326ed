#!/usr/bin/env python3
"""Per-shape efficiency table of every convolution launch of the benchmark loop.

Records every spi_conv2d_fwd / _dgrad / _wgrad call of ONE stage-1 step and ONE stage-2 super-cycle (4 iterations, the first with the three
pseudo-view branches) of the benchmark configuration (BASELINE configs[1], eager iterations), groups them by (pass, shape, path), re-times
every group in isolation with HIP events and prints: launches per stage-1 step / per stage-2 super-cycle, us per launch, TFLOP/s, fraction of
the fp32 matrix peak (157.3 TF/s; Winograd rows: direct-equivalent FLOPs, i.e. / 2.25 of them are executed), the launch's workgroups and how
many rounds of the chip's 256 CUs that is, and the share of the loop's convolution time (1:2 stage mix).

    gpurun -- 'python tools/igemm_shapes.py > gpurun_out/r05_igemm_shapes.txt'
"""
import ctypes, os, sys, collections
os.environ.setdefault('SPI_STAGE1_GRAPH', '0'); os.environ.setdefault('SPI_STAGE2_GRAPH', '0')
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spi_amd import hip
from spi_amd.configs import hyperparameters, paths_config, global_config
from spi_amd.training.triplane import TriPlaneGenerator, ffhq512_kwargs
from spi_amd.training.coaches.rot_bbox_cx_coach import RotBboxCoach
from spi_amd.training.projectors.common import Projection
from spi_amd.training.projectors.mirror_projector import mirror_setup
from spi_amd.data.images_dataset import SyntheticDataset
import contextlib, tempfile

PEAK = 157.3
FIELDS = ('N', 'I', 'O', 'H', 'W', 'kh', 'pad', 'transposed', 'flip', 'w_tap_major', 'compute_f16')


def main():
    dev = torch.device('cuda:0')
    global_config.device = str(dev)
    tmp = tempfile.mkdtemp(prefix='spi_shapes_')
    for k in ('checkpoints_dir', 'embedding_base_dir', 'experiments_output_dir', 'images_output_dir', 'mirror_images_output_dir'):
        setattr(paths_config, k, f'{tmp}/{k}/')
    hyperparameters.first_inv_type, hyperparameters.first_inv_steps = 'mir', 500
    hyperparameters.G_1_type, hyperparameters.G_1_step = 'RotBbox', 1000
    hyperparameters.pt_rot_lambda, hyperparameters.pt_mirror_rot_lambda, hyperparameters.pt_depth_lambda, hyperparameters.pt_tv_lambda = 0.1, 0.05, 1.0, 0.0
    hyperparameters.LPIPS_value_threshold = -1.0
    torch.manual_seed(0)
    G = TriPlaneGenerator(**ffhq512_kwargs(narrow=False, depth_resolution=96, depth_resolution_importance=96)).eval().requires_grad_(False).to(dev)
    G.neural_rendering_resolution = 128
    d = SyntheticDataset(1)[0]
    data = {k: (v[None] if torch.is_tensor(v) else v) for k, v in d.items()}
    with contextlib.redirect_stdout(sys.stderr):
        coach = RotBboxCoach(None, False, G=G, synthetic=True)
    ctx = coach.prepare_image(data)
    cameras, dist_fn = mirror_setup(ctx['image'], ctx['camera'], coach.lpips_loss, dev)
    proj = Projection(coach.G, cameras, dist_fn, w_mode='w+', initial_w=None, num_steps=500, w_avg_samples=600, device=dev)
    w_pivot = proj.w_opt.detach().clone()
    proj.step(0); proj.step(1)
    for i in range(4):
        coach.train_step(400 + i, ctx, w_pivot)
    torch.cuda.synchronize()

    # ---- record
    calls = {'s1': collections.Counter(), 's2': collections.Counter()}
    descs = {}
    phase = ['s1']
    orig = hip.call

    def spy(name, *args):
        if name in ('spi_conv2d_fwd', 'spi_conv2d_dgrad', 'spi_conv2d_wgrad'):
            dsc = args[0]._obj
            pid = ('spi_conv2d_fwd', 'spi_conv2d_dgrad', 'spi_conv2d_wgrad').index(name)
            key = (pid,) + tuple(int(getattr(dsc, f)) for f in FIELDS) + (int(dsc.w_batch_stride != 0), int(bool(dsc.workspace)), int(bool(dsc.dy_seg_flags)),
                                                                           int(bool(dsc.out_seg_flags)), int(bool(dsc.bias) or bool(dsc.noise) or dsc.act > 1))
            calls[phase[0]][key] += 1
            descs.setdefault(key, None)
        return orig(name, *args)
    hip.call = spy
    import spi_amd.torch_utils.ops.conv2d_mfma as cm
    import spi_amd.training.networks_stylegan2 as ns
    for m in (cm, ns):
        if getattr(m, 'hip', None) is hip:
            pass                                                   # (modules call hip.call through the module attribute: the patch is seen)
    proj.step(2)
    torch.cuda.synchronize()
    phase[0] = 's2'
    for i in range(4):
        coach.train_step(404 + i, ctx, w_pivot)
    torch.cuda.synchronize()
    hip.call = orig

    # ---- re-time every group in isolation
    rows = []
    names = ('fwd', 'dgrad', 'wgrad')
    for key in descs:
        pid = key[0]
        f = dict(zip(FIELDS, key[1:1 + len(FIELDS)]))
        per_sample, wino, dyflags, outflags, epi = key[1 + len(FIELDS):]
        n, i, o, h, w, k = f['N'], f['I'], f['O'], f['H'], f['W'], f['kh']
        tr = f['transposed']
        oh = 2 * h + k - 2 if tr else h + 2 * f['pad'] - k + 1
        ow = 2 * w + k - 2 if tr else w + 2 * f['pad'] - k + 1
        nw = n if per_sample else 1
        x = torch.randn(n, i, h, w, device=dev)
        wt = torch.randn(nw, o, k, k, i, device=dev) * 0.05
        y = torch.randn(n, o, oh, ow, device=dev)
        bias = torch.zeros(o, device=dev) if epi else None
        dsc = hip.ConvDesc(n, i, o, h, w, k, k, f['pad'], tr, f['flip'], f['w_tap_major'], f['compute_f16'], (o * i * k * k if per_sample else 0), hip.ptr(bias), None, None,
                           (3 if epi else 0), 0.2, 1.4142, -1.0, None, None, 0, 0)
        ws = None
        if wino:
            nb = hip.lib().spi_conv2d_workspace_bytes(ctypes.byref(dsc), pid)
            if nb > 0:
                ws = torch.empty(nb, device=dev, dtype=torch.uint8)
                dsc.workspace, dsc.workspace_bytes = ws.data_ptr(), nb
        plan = (ctypes.c_int32 * 8)()
        hip.lib().spi_conv2d_plan(ctypes.byref(dsc), pid, plan)
        dw = torch.zeros_like(wt)
        dx = torch.empty_like(x)

        def launch():
            if pid == 0:
                hip.call('spi_conv2d_fwd', ctypes.byref(dsc), hip.ptr(x), hip.ptr(wt), hip.ptr(y), hip.stream())
            elif pid == 1:
                hip.call('spi_conv2d_dgrad', ctypes.byref(dsc), hip.ptr(y), hip.ptr(wt), hip.ptr(dx), hip.stream())
            else:
                hip.call('spi_conv2d_wgrad', ctypes.byref(dsc), hip.ptr(x), hip.ptr(y), hip.ptr(dw), hip.stream())
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            launch()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        flop = 2.0 * n * o * i * k * k * (h * w if tr else oh * ow)
        rows.append(dict(key=key, name=names[pid], n=n, i=i, o=o, h=h, w=w, k=k, tr=tr, wino=bool(ws is not None), masked=bool(dyflags or outflags), f16=f['compute_f16'],
                         shared=not per_sample, us=us, tf=flop / us * 1e-6, c1=calls['s1'][key], c2=calls['s2'][key], plan=list(plan)))
    # loop time in the configuration's 1:2 mix: one stage-1 step + 2 x (super-cycle / 4) per 3 iterations (masked launches are timed dense here: upper bound)
    for r in rows:
        r['mix_us'] = r['us'] * (r['c1'] + 2.0 * r['c2'] / 4.0)
    tot = sum(r['mix_us'] for r in rows)
    tot_ig = sum(r['mix_us'] for r in rows if not r['wino'])
    rows.sort(key=lambda r: -r['mix_us'])
    print(f'# convolution launches of the benchmark loop (configs[1]); time = isolated launch x launches in the 1:2 stage mix (3 iterations); masked launches timed dense')
    print(f'# all convs {tot / 3e3:.2f} ms per iteration, implicit-GEMM paths {tot_ig / 3e3:.2f} ms ({100 * tot_ig / tot:.0f} %)')
    print(f'{"pass":6s} {"shape":34s} {"path":9s} {"tile":9s} {"K-rng":>5s} {"blocks":>7s} {"rounds":>6s} {"st1":>4s} {"st2x4":>5s} {"us":>8s} {"TF/s":>7s} {"of peak":>7s} {"share":>6s}')
    for r in rows:
        shape = f'{r["n"]}x{r["i"]}->{r["o"]} k{r["k"]} @{r["h"]}x{r["w"]}' + (' up2' if r['tr'] else '') + (' shW' if r['shared'] else '') + (' msk' if r['masked'] else '')
        p = r['plan']
        path = ('wino' if r['wino'] else 'igemm') + ({0: '', 1: '-f16', 2: '-bf3', 3: '-bf6'}[r['f16']])
        blocks = p[4]
        print(f'{r["name"]:6s} {shape:34s} {path:9s} {p[1]:4d}x{p[2]:<4d} {p[3]:5d} {blocks:7d} {blocks / 256 if blocks > 0 else float("nan"):6.1f} {r["c1"]:4d} {r["c2"]:5d} '
              f'{r["us"]:8.1f} {r["tf"]:7.1f} {r["tf"] / PEAK:7.2f} {100 * r["mix_us"] / tot:5.1f}%')


if __name__ == '__main__':
    main()
