"""spi_amd: MI355X-native implementation of SPI's per-image inversion inner loop.

Host code is Python on PyTorch-ROCm (device memory, streams, autograd plumbing); all hot-path
arithmetic runs in hand-written HIP kernels for gfx950 behind the C ABI declared in
``include/spi_hip.h`` (built to ``spi_amd/csrc/libspi_hip.so``).  There is no CPU fallback:
calling a compute op without the library / a GPU raises.
"""
import os as _os

__version__ = '0.1.0'

# ---- HIP graphs on ROCm 7: switch off the runtime's AQL packet capture -------------------------------------------------------------------
# The loops replay their iterations from HIP graphs.  With `DEBUG_CLR_GRAPH_PACKET_CAPTURE` at its default (on) the runtime pre-builds
# the AQL packets of a graph once; after ~10^3 ordinary (eager) launches on the same device a replay of such a graph was found (round 4) to
# run some of its kernel nodes with stale arguments -- garbage results, NaN latents, once a GPU memory fault; reproduced with ATen ops
# alone (tools/ubench/graph_after_eager.py: capture 5 x (mul, sum over dim 1, sum), replay, 1500 x `t.add_(1)`, replay -> another number).
# With the switch off every replay is correct and the measured iteration rate is unchanged (the steps are GPU-bound; the host-side cost of a
# stage-1 replay goes from 0.2 ms to 6 ms of a 15.7 ms step).  The runtime reads the variable when it initialises (first HIP call), so it
# is set here, at import -- `hip_graphs_safe()` says whether that was early enough; if not, the loops enqueue their iterations eagerly.
_PACKET_CAPTURE = 'DEBUG_CLR_GRAPH_PACKET_CAPTURE'
_env_ok = None          # the variable reads '0' in this process (necessary, not sufficient: the runtime may have started before it was set)
_canary = None          # result of the one-time replay check on the device (None = not run yet)


def _guard_hip_graphs():
    global _env_ok
    import torch
    started = torch.cuda.is_initialized()
    if _os.environ.get(_PACKET_CAPTURE) is None and not started:
        _os.environ[_PACKET_CAPTURE] = '0'          # (inherited by child processes, e.g. bench.py's legs and ranks: they want it too)
    # an explicit setting is respected ('0' = safe, anything else = the user wants the capture: no graph replay here); a runtime that is
    # already up without the variable cannot be changed any more
    _env_ok = _os.environ.get(_PACKET_CAPTURE) == '0'


def _run_graph_canary(n_eager=1500):
    """The defect's own reproducer (tools/ubench/graph_after_eager.py) as a one-time self-test: capture a small graph, replay it, run
    `n_eager` ordinary launches, replay again -- the two replays must agree bit for bit.  ~15 ms, once per process, before the first capture
    of a loop.  `torch.cuda.is_initialized() == False` at import does NOT prove the runtime had not read its flags yet
    (`is_available()`, `device_count()`, RCCL ... start HIP without setting torch's lazy-init flag; round-4 advisor), so the environment
    alone is not trusted."""
    import torch
    from .torch_utils.misc import capture_graph
    import torch.distributed as tdist
    mode = 'thread_local' if (tdist.is_available() and tdist.is_initialized()) else 'global'
    dev = torch.cuda.current_device()
    gen = torch.Generator(device='cuda').manual_seed(1234)
    feats = [torch.randn(1, c, s, s, device='cuda', generator=gen) for c, s in ((64, 64), (128, 32), (256, 16), (512, 8), (512, 4))]
    out = torch.zeros((), device='cuda')

    def body():
        acc = None
        for f in feats:
            v = torch.sum(f * f, dim=1, keepdim=True).sum()
            acc = v if acc is None else acc + v
        out.copy_(acc)
    body()
    torch.cuda.synchronize(dev)
    ref = float(out)
    g = torch.cuda.CUDAGraph()
    with capture_graph(g, capture_error_mode=mode):
        body()
    g.replay()
    torch.cuda.synchronize(dev)
    first = float(out)
    t = torch.zeros(1024, device='cuda')
    for _ in range(n_eager):
        t.add_(1.0)
    out.zero_()
    g.replay()
    torch.cuda.synchronize(dev)
    second = float(out)
    del g
    return first == ref and second == ref


def hip_graphs_safe():
    """True when HIP-graph replays can be trusted in this process: the AQL packet capture is switched off in the environment (see above) AND the
    one-time replay self-test on the device agrees with eager execution.  Only then do the loops replay graphs; otherwise they enqueue their
    iterations eagerly and say so once on stderr."""
    global _canary
    if not _env_ok:
        return False
    if _canary is None:
        import torch
        if not torch.cuda.is_available() or torch.cuda.is_current_stream_capturing():
            return bool(_env_ok)                     # nothing to test on (CPU-only process) / not now: the answer is not cached
        try:
            _canary = bool(_run_graph_canary())
        except Exception as e:                       # a capture that fails outright is as much a "no" as a wrong replay
            import sys
            print(f'[spi_amd] HIP-graph self-test raised {type(e).__name__}: {e}', file=sys.stderr)
            _canary = False
        if not _canary:
            import sys
            print('[spi_amd] HIP-graph self-test FAILED: a captured graph replayed after 1500 eager launches did not reproduce its result '
                  '(ROCm graph packet capture active? the HIP runtime was probably initialised before `import spi_amd` set '
                  f'{_PACKET_CAPTURE}=0); iterations are enqueued eagerly', file=sys.stderr)
    return bool(_canary)


def hip_graphs_status():
    """{'env': the variable reads '0', 'self_test': True / False / None (not run yet)} -- reported by bench.py and the loops' stats line."""
    return {'env': bool(_env_ok), 'self_test': _canary}


_guard_hip_graphs()
