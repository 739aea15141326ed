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
_graphs_safe = None


def _guard_hip_graphs():
    global _graphs_safe
    import torch
    started = torch.cuda.is_initialized()
    if _os.environ.get(_PACKET_CAPTURE) is None and not started:
        _os.environ[_PACKET_CAPTURE] = '0'
    # an explicit setting is respected ('0' = safe, anything else = the user wants the capture: no graph replay here); a runtime that is
    # already up without the variable cannot be changed any more
    _graphs_safe = _os.environ.get(_PACKET_CAPTURE) == '0'


def hip_graphs_safe():
    """True when this process runs with the AQL packet capture of HIP graphs switched off (see above): only then do the loops replay graphs."""
    return bool(_graphs_safe)


_guard_hip_graphs()
