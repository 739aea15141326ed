"""Build libspi_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).

    python -m spi_amd.csrc.build [--force]
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SOURCES = ['elementwise.hip', 'render.hip', 'conv.hip', 'winograd.hip', 'losses.hip', 'flrelu.hip', 'hconv.hip']
# winograd.hip keeps its 256 accumulators per lane in AGPRs (the other 256 registers hold operands and the loaders' state)
NO_VGPR_FORM = {'winograd.hip'}
LIB = os.path.join(HERE, 'libspi_hip.so')
STAMP = os.path.join(HERE, '.build_stamp')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics', '-Wno-comment',
         # MFMA results stay in VGPRs: the default AGPR form costs 2 v_accvgpr moves per MFMA across the igemm loop back-edge, and on
         # gfx950 every VALU instruction beside an fp32 MFMA steals ~2.6 matrix-pipe cycles (tools/ubench/mfma_valu.hip)
         '-mllvm', '-amdgpu-mfma-vgpr-form=1',
         '-Wno-pass-failed', '-I' + os.path.join(ROOT, 'include'), '-I' + HERE]


# The sources carry inline asm for LDS-DMA loads and for raw_buffer_load_b64 / b128 (the builtins of this compiler mis-handle them, see
# render.hip / winograd.hip) and rely on its register allocation for the 512-register Winograd kernels: they were written against and
# verified with ROCm 7.2's hipcc.  Another compiler may build something that runs differently, so it must be asked for explicitly.
HIPCC_SERIES = '7.2'


def check_compiler(hipcc):
    """-> the compiler's 'HIP version' string; raises unless it is the series the kernels were verified with (SPI_ALLOW_ANY_HIPCC=1 overrides)."""
    try:
        out = subprocess.run([hipcc, '--version'], capture_output=True, text=True).stdout
    except OSError as e:
        raise RuntimeError(f'cannot run {hipcc} ({e}): libspi_hip.so must be (re)built with ROCm {HIPCC_SERIES}.x hipcc -- set HIPCC=/path/to/hipcc '
                           '(SPI_ALLOW_ANY_HIPCC=1 accepts another series; re-run the GPU tests then)') from e
    ver = next((ln.split(':', 1)[1].strip() for ln in out.splitlines() if ln.startswith('HIP version')), '')
    if not ver.startswith(HIPCC_SERIES + '.') and os.environ.get('SPI_ALLOW_ANY_HIPCC') != '1':
        raise RuntimeError(f'{hipcc} reports HIP version {ver or "?"}; the kernels are verified with {HIPCC_SERIES}.x '
                           '(set SPI_ALLOW_ANY_HIPCC=1 to build with another compiler and re-run the GPU tests)')
    return ver


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
    ver = check_compiler(hipcc)
    if verbose:
        print(f'[spi_amd build] hipcc: HIP version {ver}', flush=True)
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
    with open(STAMP + '.compiler', 'w') as f:                    # which compiler built the library that the stamp vouches for
        f.write(ver + '\n')
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
