"""Tracing helpers of the reference's ``torch_utils/misc.py`` that the hot path uses.

``profiled_function`` (misc.py:102-107 wraps a function in ``torch.autograd.profiler.record_function``) marks the same two functions as
the reference does -- ``normalize_2nd_moment`` and ``modulated_conv2d`` (networks_stylegan2.py:27,33) -- and ``trace_range`` marks the
phases of one synthesis (backbone / renderer / super-resolution) and of the two loops (losses, optimiser).  On MI355X the ranges are
rocTX ranges (``torch.cuda.nvtx`` is rocTX on ROCm): ``rocprofv3 --marker-trace --kernel-trace`` shows them beside the kernels.
They cost a Python call each, so they are OFF unless ``SPI_TRACE=1`` (or ``enable_tracing(True)``): the default run has no marker
overhead and a captured HIP graph contains none.
"""
import contextlib
import functools
import os

import torch

_enabled = [os.environ.get('SPI_TRACE', '0') == '1']


def enable_tracing(on=True):
    _enabled[0] = bool(on)


def tracing_enabled():
    return _enabled[0]


@contextlib.contextmanager
def trace_range(name):
    """rocTX range + autograd-profiler range around a phase; a no-op context when tracing is off."""
    if not _enabled[0]:
        yield
        return
    use_roctx = torch.cuda.is_available()
    if use_roctx:
        torch.cuda.nvtx.range_push(name)
    try:
        with torch.autograd.profiler.record_function(name):
            yield
    finally:
        if use_roctx:
            torch.cuda.nvtx.range_pop()


def profiled_function(fn):
    """Same contract as the reference's decorator (the range carries the function's name); free when tracing is off."""
    @functools.wraps(fn)
    def decorator(*args, **kwargs):
        if not _enabled[0]:
            return fn(*args, **kwargs)
        with trace_range(fn.__name__):
            return fn(*args, **kwargs)
    return decorator


@contextlib.contextmanager
def quiet_gc():
    """Around an optimisation loop: collect once, then move everything alive (modules, parameters, the loss networks) into Python's
    permanent generation, so that the generation-2 collections triggered by the loop's own churn of autograd nodes stay cheap.  A full
    collection over the whole heap was measured as a 50-80 ms pause -- two or three iterations' worth -- whenever it struck inside the loop
    (bench.py, `SPI_BENCH_ITER_TIMES=1`).  ``global_config.freeze_gc_in_loops = False`` / ``SPI_GC_FREEZE=0`` leaves the collector alone."""
    from ..configs import global_config
    if not getattr(global_config, 'freeze_gc_in_loops', True):
        yield
        return
    import gc
    gc.collect()
    gc.freeze()
    try:
        yield
    finally:
        gc.unfreeze()


_capture_streams = {}


@contextlib.contextmanager
def capture_graph(g, capture_error_mode='global'):
    """``with torch.cuda.graph(g, capture_error_mode=...)`` without its ``torch.cuda.empty_cache()``.  torch empties the caching allocator before
    every capture "to free as much memory as we can for the graph"; on a 288 GB part that buys nothing and costs the loops their warm allocator:
    the pool reserved at start-up (dist.reserve_allocator_pool) and every cached block go back to the driver, and the eager iterations after the
    capture pay hipMalloc for GBs again -- most of the ~1 s per image the two stage-2 captures used to cost (tools/soak_cli.py).  Same stream
    discipline as torch's context manager: a dedicated capture stream per device, device-wide synchronisation first."""
    import torch
    dev = torch.cuda.current_device()
    stream = _capture_streams.get(dev)
    if stream is None:
        stream = _capture_streams[dev] = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        g.capture_begin(capture_error_mode=capture_error_mode)
        try:
            yield
        finally:
            g.capture_end()
