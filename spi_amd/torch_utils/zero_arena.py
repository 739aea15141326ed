"""One zero fill per optimisation iteration.

The HIP kernels of the inversion loops accumulate through atomics -- split-K / channel-split convolution outputs, weight and style
gradients, the layer-tail sums (bias / noise gradients), the per-sample loss sums, the tri-plane gradient -- and every such
accumulator has to start at zero.  Allocated one by one that is ~110 `fill_` launches of ATen plus ~55 of the library's own
`zero_kernel` per stage-2 iteration (63 + 40 in stage 1; `SPI_TORCH_PROFILE=1 python bench.py`): 4 us of device time each for a few
hundred bytes to a few MB, 0.7 ms of a 24 ms iteration.  MI355X-first: the loops announce the start of an iteration with `begin()`,
which allocates ONE buffer as large as the zeros the last iteration of the same kind asked for and clears it with one launch at full
HBM rate; `zeros()` hands out 256-byte aligned views of it.  The views keep the storage alive, so the arena is freed when the last
accumulator of the iteration dies (torch's caching allocator recycles the block: no device allocation after warm-up), and a capture
of the iteration in a HIP graph records the one fill like any other launch.

The loops close the iteration with `finish()` (the next `begin()` implies it): after it nothing is handed a view of that buffer any more -- which
matters when the iteration was captured into a HIP graph and the buffer belongs to the graph's memory pool.
Outside an iteration (`begin()` never called or `finish()`ed: the operator tests, inference) and whenever a request does not fit (the first
iteration of a kind, a larger batch) `zeros()` is `torch.zeros` and `take()` returns None -- same values either way.
"""
import contextlib
import os
import torch

ALIGN = 64                                   # floats: every view starts on a 256-byte boundary (float4 stores, cache-line atomics)


class _State:
    buf = None                               # this iteration's arena (fp32, 1-D) or None
    off = 0                                  # floats handed out
    demand = 0                               # floats asked for since begin() (whether they fitted or not)
    key = None
    open = False                             # between begin() and finish()
    peaks = {}                               # iteration kind -> largest demand seen


_s = _State()
enabled = os.environ.get('SPI_ZERO_ARENA', '1') != '0'       # A-B measurements: off = every request is its own torch.zeros


def begin(device, key=None):
    """Start of an optimisation iteration of kind `key` (iterations that differ in how many accumulators they need -- the plain and the
    pseudo-view iterations of stage 2 -- keep separate sizes, so the plain one does not clear memory it will not use)."""
    finish()
    _s.key, _s.off, _s.demand, _s.open = key, 0, 0, True
    n = _s.peaks.get(key, 0)
    _s.buf = torch.zeros(n, device=device, dtype=torch.float32) if (enabled and n > 0) else None


def finish():
    """End of the iteration begun last (implied by the next `begin()`): remember what it needed, drop the arena reference."""
    if _s.demand > _s.peaks.get(_s.key, 0):
        _s.peaks[_s.key] = _s.demand
    _s.buf, _s.off, _s.demand, _s.open = None, 0, 0, False


@contextlib.contextmanager
def iteration(device, key=None):
    """``with zero_arena.iteration(device, key):`` = `begin()` ... `finish()` with the `finish()` guaranteed: an exception (or a failed HIP-graph
    capture) inside an iteration must not leave the arena open -- later work in the process (inference, operator tests, video export) would draw
    its 'zeros' from a stale buffer that may belong to the failed capture's memory pool."""
    begin(device, key)
    try:
        yield
    finally:
        finish()


def closes_iteration(fn):
    """Decorator for the loops' iteration bodies (they call `begin()` themselves, where the key is known): whatever happens inside, the
    iteration is `finish()`ed on the way out."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **kw):
        try:
            return fn(*a, **kw)
        finally:
            finish()
    return wrapped


def reset():
    """Forget every recorded size (a new run with other shapes)."""
    finish()
    _s.peaks.clear()
    _s.key = None


def in_iteration():
    """True between `begin()` and `finish()`: requests are being recorded (and served, once the kind's size is known)."""
    return _s.open and enabled


def take(n, device):
    """A zeroed fp32 view of `n` elements from the current arena, or None if there is none / it is full / it lives on another device."""
    n = int(n)
    n_al = (n + ALIGN - 1) // ALIGN * ALIGN
    if _s.open:
        _s.demand += n_al
    buf = _s.buf
    if buf is None or n == 0 or _s.off + n_al > buf.numel():
        return None
    d = torch.device(device)
    if d.type != buf.device.type or (d.index is not None and d.index != buf.device.index):
        return None
    v = buf[_s.off:_s.off + n]
    _s.off += n_al
    return v


def zeros(shape, device):
    """`torch.zeros(shape, device=device, dtype=torch.float32)`, from the arena when there is one."""
    if isinstance(shape, int):
        shape = (shape,)
    n = 1
    for d in shape:
        n *= int(d)
    v = take(n, device) if n else None
    if v is None:
        return torch.zeros(shape, device=device, dtype=torch.float32)
    return v.view(shape)


def zeros_like(t):
    return zeros(tuple(t.shape), t.device)
