#!/usr/bin/env python3
"""Quick check of the F(4x4, 3x3) Winograd kernel against F(2x2, 3x3) and the implicit GEMM on the same inputs (all on the GPU, seconds):
   python tools/wino_f4_check.py            (run it under `timeout`: a kernel under development may hang)"""
import os, sys, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spi_amd import hip
from spi_amd.configs import global_config
from spi_amd.torch_utils.ops import conv2d_mfma as cm

DEV = 'cuda'
L = hip.lib()


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max()).item()


for (N, I, O, H, W, per, epi) in ((1, 128, 64, 256, 512, False, False), (1, 128, 128, 256, 256, True, True), (2, 256, 64, 256, 512, False, False), (1, 128, 128, 512, 512, True, True), (1, 256, 256, 256, 256, True, False)):
    gen = torch.Generator().manual_seed(I + O)
    x = torch.randn(N, I, H, W, generator=gen).to(DEV)
    w = (torch.randn(*((N,) if per else ()), O, I, 3, 3, generator=gen) / (I * 9) ** 0.5).to(DEV)
    kw = dict(padding=1, flip=True)
    if epi:
        kw.update(bias=torch.randn(O, generator=gen).to(DEV), noise=torch.randn(H, W, generator=gen).to(DEV), noise_strength=torch.tensor(0.3).to(DEV), act='lrelu', gain=1.3, clamp=2.0)
    ref = torch.nn.functional.conv2d(x[0:1].double(), (w[0] if per else w).double().flip([2, 3]), padding=1)
    if epi:
        z = ref + (kw['noise'] * 0.3).double() + kw['bias'].double().view(1, -1, 1, 1)
        ref = (torch.nn.functional.leaky_relu(z, 0.2) * 1.3).clamp(-2.0, 2.0)
    outs = {}
    for mode in ('f4', 'f2', 'igemm'):
        global_config.conv_winograd_f4 = mode == 'f4'
        cm._sync_wino_f4()
        global_config.conv_winograd = mode != 'igemm'
        cm._frozen_ws.clear()
        d = cm._desc(N, I, O, H, W, 3, 1, False, True, O * I * 9 if per else 0, tap_major=1)
        nb = L.spi_conv2d_workspace_bytes(ctypes.byref(d), 0)
        with torch.no_grad():
            outs[mode] = cm.conv2d(x, w, **kw)
        torch.cuda.synchronize()
        print(f'  {mode}: workspace {nb} bytes ({nb // (4 * I * ((O + 63) // 64 * 64) * (N if per else 1))} frequencies), vs fp64 torch conv (sample 0): {rel(outs[mode][0:1], ref):.2e}', flush=True)
    print(f'{(N, I, O, H, W, per, epi)}: f4 vs f2 {rel(outs["f4"], outs["f2"]):.2e}, f4 vs igemm {rel(outs["f4"], outs["igemm"]):.2e}', flush=True)
global_config.conv_winograd_f4 = True
cm._sync_wino_f4()
global_config.conv_winograd = True
