"""Debug aid: find kernels that read memory nobody wrote.

Runs a callable once (the caching allocator now holds every block the work needs), overwrites every INACTIVE block of torch's caching allocator
with a NaN pattern (hipMemsetD32 through libamdhip64), runs it again and compares: a kernel that reads an uninitialised `torch.empty` / a
workspace region it never wrote turns its outputs into NaN (or into something else than the first run's) instead of silently depending on what
the previous tenant of the block left behind.

    python tools/poison_check.py stage2        # five RotBbox iterations of the narrow generator (the reference-pin scenario)
    python tools/poison_check.py stage1 | pti
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]


def free_blocks(default_pool_only=False):
    """[(address, bytes)] of the caching allocator's inactive blocks (optionally: of the default pool only, not of graph-private pools)."""
    torch.cuda.synchronize()
    out = []
    for seg in torch.cuda.memory_snapshot():
        if default_pool_only and tuple(seg.get('segment_pool_id', (0, 0))) != (0, 0):
            continue
        addr = seg['address']
        for blk in seg['blocks']:
            if blk['state'] == 'inactive' and blk['size'] >= 4:
                out.append((addr, blk['size']))
            addr += blk['size']
    return out


def poison_blocks(blocks, pattern=0x7FC00000):
    hipl = ctypes.CDLL('libamdhip64.so')
    hipl.hipMemsetD32.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
    hipl.hipMemsetD32.restype = ctypes.c_int
    for addr, size in blocks:
        rc = hipl.hipMemsetD32(ctypes.c_void_p(addr), ctypes.c_int(pattern - (1 << 32) if pattern >= (1 << 31) else pattern), size // 4)
        assert rc == 0, rc
    torch.cuda.synchronize()
    return sum(b[1] for b in blocks)


def poison_free_blocks(pattern=0x7FC00000):
    """Fill every inactive block of the caching allocator with `pattern` (default: a quiet NaN).  -> bytes written."""
    torch.cuda.synchronize()
    hipl = ctypes.CDLL('libamdhip64.so')
    hipl.hipMemsetD32.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
    hipl.hipMemsetD32.restype = ctypes.c_int
    total = 0
    for seg in torch.cuda.memory_snapshot():
        addr = seg['address']
        for blk in seg['blocks']:
            if blk['state'] == 'inactive' and blk['size'] >= 4:
                rc = hipl.hipMemsetD32(ctypes.c_void_p(addr), ctypes.c_int(pattern - (1 << 32) if pattern >= (1 << 31) else pattern), blk['size'] // 4)
                assert rc == 0, rc
                total += blk['size']
            addr += blk['size']
    torch.cuda.synchronize()
    return total


def scenario(name):
    import conftest
    import test_hip_reference_pins_gpu as pins

    golden = conftest.Golden
    if name == 'stage2':
        g = golden('trajectory_stage2')

        def run():
            coach, log, stats, rng, p0 = pins._run_product_coach('RotBbox', g, 5, -1.0, int(g['n_draws']))
            return {f'it{i}/{k}': v for i, e in enumerate(log) for k, v in e['grads'].items()}
        return run
    if name == 'pti':
        g = golden('trajectory_pti')

        def run():
            coach, log, stats, rng, p0 = pins._run_product_coach('PTI', g, 3, -1.0, int(g['n_draws']))
            return {f'it{i}/{k}': v for i, e in enumerate(log) for k, v in e['grads'].items()}
        return run
    raise SystemExit(f'unknown scenario {name}')


def main():
    run = scenario(sys.argv[1] if len(sys.argv) > 1 else 'stage2')
    a = run()
    a = {k: v.clone() for k, v in a.items()}
    worst = 0
    # a quiet NaN (floats that are read before they are written), zeros / ones / all-ones (flag, count and index buffers: a stale zero or a
    # stale non-zero may each be the harmless value)
    for pattern in (0x7FC00000, 0x0, 0x3F800000, 0xFFFFFFFF, 0x1):
        n = poison_free_blocks(pattern)
        b = run()
        bad = 0
        for k in a:
            nan = int(torch.isnan(b[k]).sum())
            err = float((a[k] - b[k]).abs().max() / a[k].abs().max().clamp_min(1e-30)) if not nan else float('nan')
            if nan or err > 2e-5:
                bad += 1
                print(f'  {k:70s} nan {nan:8d}  max diff vs first run / max {err:.2e}')
        print(f'pattern {pattern:#010x}: {n / 2**20:.0f} MiB of cached blocks overwritten, {"clean" if not bad else f"{bad} tensors differ"}')
        worst = max(worst, bad)
    sys.exit(1 if worst else 0)


if __name__ == '__main__':
    main()
