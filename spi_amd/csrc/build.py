"""Build libspi_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).

    python -m spi_amd.csrc.build [--force]
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SOURCES = ['elementwise.hip', 'render.hip', 'conv.hip', 'winograd.hip', 'losses.hip', 'flrelu.hip']
# winograd.hip keeps its 256 accumulators per lane in AGPRs (the other 256 registers hold operands and the loaders' state)
NO_VGPR_FORM = {'winograd.hip'}
LIB = os.path.join(HERE, 'libspi_hip.so')
STAMP = os.path.join(HERE, '.build_stamp')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics', '-Wno-comment',
         # MFMA results stay in VGPRs: the default AGPR form costs 2 v_accvgpr moves per MFMA across the igemm loop back-edge, and on
         # gfx950 every VALU instruction beside an fp32 MFMA steals ~2.6 matrix-pipe cycles (tools/ubench/mfma_valu.hip)
         '-mllvm', '-amdgpu-mfma-vgpr-form=1',
         '-Wno-pass-failed', '-I' + os.path.join(ROOT, 'include'), '-I' + HERE]


def _digest():
    h = hashlib.sha256(' '.join(FLAGS).encode())
    for name in SOURCES + ['common.hpp', os.path.join(ROOT, 'include', 'spi_hip.h')]:
        with open(os.path.join(HERE, name), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def build(force=False, verbose=True):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read() == dig:
        return LIB
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(HERE, src.replace('.hip', '.o'))
        objs.append(obj)
        flags = [f for f in FLAGS if not (src in NO_VGPR_FORM and f in ('-mllvm', '-amdgpu-mfma-vgpr-form=1'))]
        cmd = [hipcc] + flags + ['-c', os.path.join(HERE, src), '-o', obj]
        if verbose:
            print('[spi_amd build]', ' '.join(cmd), flush=True)
        procs.append(subprocess.Popen(cmd))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed')
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print('[spi_amd build]', ' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, 'w') as f:
        f.write(dig)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
