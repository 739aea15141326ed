"""Dense 2-D convolutions on the MI355X matrix cores (fp32 MFMA), with autograd.

Replaces what the reference gets from cuDNN through ``conv2d_gradfix.conv2d`` / ``conv_transpose2d``
(eg3d/torch_utils/ops/conv2d_gradfix.py:37-45) in the two shapes the generator uses
(eg3d/torch_utils/ops/conv2d_resample.py:114-136):

  * ``groups = batch`` convs with per-sample weights ``w [N, O, I, k, k]`` -- the fused modulated
    convolution of ``modulated_conv2d`` (networks_stylegan2.py:85-88).  No [1, N*I, H, W] reshapes:
    the kernel indexes the sample's weights directly.
  * shared-weight convs ``w [O, I, k, k]`` (VGG feature extractors of the losses).

``transposed=True`` is ``conv_transpose2d(stride=2, padding=0)``: out[o, 2y+ky, 2x+kx] += x[i,y,x] w[o,i,ky,kx]
(note the [O, I] weight order also in this mode).  ``flip=True`` flips the kernel spatially.
An optional fused epilogue applies ``+ noise * strength``, ``+ bias``, activation, gain, clamp to the
accumulators (SynthesisLayer.forward tail, networks_stylegan2.py:320-329) for stride-1 convs.
"""
import contextlib
import weakref
import ctypes
import torch
from ... import hip
from . import bias_act as _ba
from .. import zero_arena

# Sparse output gradients.  SPI's pseudo-view losses are masked (visibility / foreground masks, rot_bbox_cx_coach.py:94-140 of the
# reference): most pixels of d(image) are exactly zero, and so are the gradients arriving at the super-resolution convolutions.
# Inside ``with sparse_gradients():`` the backward of large convs first maps the all-zero 16-pixel segments of dy
# (spi_seg_flags) and hands the map to dgrad / wgrad, which skip them.  Results equal the dense kernels' up to fp32 summation order.
_sparse = [False]
SPARSE_MIN_PIXELS = 128 * 128


@contextlib.contextmanager
def sparse_gradients(enabled=True):
    old = _sparse[0]
    _sparse[0] = bool(enabled)
    try:
        yield
    finally:
        _sparse[0] = old


# Needed-output regions.  When the consumer of a generator forward looks at the image only inside a known region (the warp mask of
# the pseudo-view losses), the convs that opted in (sparse_grad=True) skip the output tiles that hold no flagged 16-pixel segment:
# ``with needed_output({(OH, OW): int32 flags [N, ceil(OH*OW/16)]})`` -- the caller provides one map per output resolution, dilated by
# what the layers between that conv and the consumer read (superresolution.needed_output_maps).
_needed = [None]


@contextlib.contextmanager
def needed_output(flags_by_shape):
    old = _needed[0]
    _needed[0] = flags_by_shape
    try:
        yield
    finally:
        _needed[0] = old


def seg_flags(x):
    """[N, C, H, W] (fp32 or fp16) -> int32 [N, ceil(H*W/16)]: 1 where the 16-pixel segment (flat index) holds a non-zero in any channel."""
    n, c = x.shape[0], x.shape[1]
    hw = x[0, 0].numel()
    flags = torch.empty(n, (hw + 15) // 16, device=x.device, dtype=torch.int32)
    if x.dtype == torch.float16:
        hip.call('spi_seg_flags_t', hip.ptr(x), hip.ptr(flags), n, c, hw, hip.DTYPE_IDS[torch.float16], hip.stream())
    else:
        hip.call('spi_seg_flags', hip.ptr(x), hip.ptr(flags), n, c, hw, hip.stream())
    return flags


def half_io(x, f16):
    """Does this conv run on fp16 ACTIVATION TENSORS?  When its input arrives as a half tensor and it computes with fp16 operands (the reference's
    use_fp16 blocks cast x once at the block entry, networks_stylegan2.py:423-436, and everything after it stays half) and the reduction axis
    is made of whole 16-channel slabs in both directions (the buffer-descriptor path)."""
    return f16 == 1 and x.dtype == torch.float16


def _desc(n, i, o, h, w, k, pad, transposed, flip, wbs, bias=None, noise=None, ng=None, act=0, alpha=0.0, gain=1.0, clamp=-1.0, tap_major=0,
          f16=0, dy_flags=None, out_flags=None, dw_zeroed=0, half=False):
    d = hip.ConvDesc(n, i, o, h, w, k, k, pad, int(transposed), int(flip), int(tap_major), int(f16), wbs, hip.ptr(bias), hip.ptr(noise), hip.ptr(ng),
                     act, alpha, gain, clamp, hip.ptr(dy_flags), hip.ptr(out_flags), int(dw_zeroed))
    d.act_dtype = 1 if half else 0
    return d


# Transformed weights of FROZEN shared-weight convolutions (the VGG feature extractors of LPIPS / BoxCX): the Winograd passes transform their weights
# into the workspace with a launch of their own (`wino_weight_kernel`, 9 us) -- 36 launches per stage-2 iteration for weights that never change.
# The workspace of such a conv is kept per (weight tensor, pass) and handed back with `workspace_ready = 1` (round 5); an in-place update of the
# weights (`_version`) or another tensor at the same address drops it.
_frozen_ws = {}
_FROZEN_SWEEP_AT = 512
_pinned_retired = []


def _sweep_frozen():
    """Drop the entries whose weight tensor is gone and whose workspace no captured launch has been handed (`pinned`).  Nothing else is ever
    freed: a live graph replays `workspace_ready = 1` launches that read the workspace by address (clearing the whole table, or
    replacing an entry after an in-place weight update, left such graphs reading freed memory)."""
    for key in [k for k, e in _frozen_ws.items() if e[2]() is None and not e[3]]:
        del _frozen_ws[key]


def _frozen_key(w, pass_id, d):
    return (id(w), w.data_ptr(), tuple(w.shape), pass_id, int(d.flip), int(d.w_tap_major), int(d.compute_f16), int(d.act_dtype), str(w.device))


_wino_f4_state = None


def _sync_wino_f4():
    """`global_config.conv_winograd_f4` -> the library's process-wide switch (spi_conv_wino_f4_set), before a workspace is sized with it."""
    global _wino_f4_state
    from ...configs import global_config
    want = bool(global_config.conv_winograd_f4)
    if want != _wino_f4_state:
        hip.lib().spi_conv_wino_f4_set(1 if want else 0)
        _wino_f4_state = want


def _workspace(d, pass_id, device, w=None, frozen=False):
    """Scratch memory with which `pass_id` (0 forward, 1 dgrad, 2 wgrad) of the conv `d` takes its Winograd path (None: it has none, or
    ``global_config.conv_winograd`` is off).  Comes from torch's caching allocator: no device allocation after warm-up.
    `w` + `frozen` (shared weights that take no gradient): the transformed weights are cached across calls, see `_frozen_ws`."""
    from ...configs import global_config
    direct_f16 = d.compute_f16 == 1 and d.act_dtype == 1 and global_config.conv_direct_fp16     # hconv.hip: fp16 tensors, 3x3, stride 1
    # (a transposed conv has direct kernels for its forward and its data gradient -- a stride-2 conv of the fp16 gradient --, not for its weight gradient)
    if d.kh != 3 or (d.transposed and not (direct_f16 and pass_id < 2)) or not (direct_f16 or (global_config.conv_winograd and d.compute_f16 in (0, 3))):
        return None
    _sync_wino_f4()
    nbytes = hip.lib().spi_conv2d_workspace_bytes(ctypes.byref(d), pass_id)
    if nbytes <= 0:
        return None
    if frozen and w is not None and pass_id < 2 and d.w_batch_stride == 0:
        key = _frozen_key(w, pass_id, d)
        hit = _frozen_ws.get(key)
        capturing = torch.cuda.is_current_stream_capturing()
        # (the entry holds a weak reference to the tensor OBJECT: a temporary -- e.g. a freshly modulated weight under no_grad -- dies, and the
        #  next temporary of that shape that the caching allocator puts at the same address must not inherit its transform)
        if hit is not None and hit[2]() is w and hit[1].numel() == nbytes:
            # A workspace that was handed to a CAPTURED launch with `workspace_ready = 1` is baked into that graph by address: it is pinned
            # (never freed, never replaced) from then on.  An in-place update of the weights re-transforms INTO the same workspace -- graphs
            # that skip the transform then read the current weights' transform, which is what an eager call would compute.
            if capturing:
                hit[3] = True
            d.workspace, d.workspace_bytes = hit[1].data_ptr(), nbytes
            if hit[0] == w._version:
                d.workspace_ready = 1
            else:
                hit[0] = w._version
            return hit[1]
        if not capturing:       # (a workspace born inside a capture lives in the graph's pool: not cacheable)
            if len(_frozen_ws) > _FROZEN_SWEEP_AT:
                _sweep_frozen()
            if hit is not None and hit[3]:
                _pinned_retired.append(hit[1])      # (a dead tensor's entry whose workspace a graph still reads: the key is reused, the memory is not)
            ws = torch.empty(nbytes, device=device, dtype=torch.uint8)
            _frozen_ws[key] = [w._version, ws, weakref.ref(w), False]
            d.workspace, d.workspace_bytes = ws.data_ptr(), nbytes
            return ws
    ws = torch.empty(nbytes, device=device, dtype=torch.uint8)
    d.workspace, d.workspace_bytes = ws.data_ptr(), nbytes
    return ws


_accumulates = {}


def _out_tensor(d, pass_id, shape, device):
    """The output tensor of pass `pass_id` (0 forward, 1 dgrad) of the conv `d`.  Launches that ACCUMULATE into their output (split-K implicit
    GEMM of the 4^2..32^2 layers, channel-split Winograd) clear it first; inside an optimisation iteration the output is instead a view of the
    iteration's zero arena (`d.out_zeroed = 1`: the library skips its fill launch -- ~40 per generator pass)."""
    if zero_arena.in_iteration():
        key = (pass_id, d.N, d.I, d.O, d.H, d.W, d.kh, d.pad, d.transposed, d.compute_f16, d.w_batch_stride != 0, bool(d.out_seg_flags), bool(d.workspace))
        acc = _accumulates.get(key)
        if acc is None:
            acc = _accumulates[key] = hip.lib().spi_conv2d_out_accumulates(ctypes.byref(d), pass_id) == 1
        if acc:
            n = 1
            for s in shape:
                n *= int(s)
            v = zero_arena.take(n, device)
            if v is not None:
                d.out_zeroed = 1
                return v.view(shape)
    return torch.empty(shape, device=device, dtype=torch.float16 if d.act_dtype == 1 else torch.float32)


def pad_o16(dz, w, o):
    """fp16 activation tensors: the data-gradient pass reduces over the OUTPUT channels in whole 16-channel slabs (buffer-descriptor path).  The
    3-channel torgb layers get 13 zero channels (gradient) / zero rows (tap-major weights [.., O, k, k, I]): 16 x H x W halves beside the
    layer's 128+ x H x W input -- a few percent of its traffic.  -> (dz, w, padded O)"""
    o16 = (o + 15) // 16 * 16
    if o16 == o:
        return dz, w, o
    dzp = torch.zeros(dz.shape[0], o16, dz.shape[2], dz.shape[3], device=dz.device, dtype=dz.dtype)
    dzp[:, :o] = dz
    shape = list(w.shape)
    shape[-4] = o16
    wp = torch.zeros(shape, device=w.device, dtype=w.dtype)
    wp.narrow(-4, 0, o).copy_(w)
    return dzp, wp, o16


def out_size(h, k, pad, transposed):
    return 2 * h + k - 2 if transposed else h + 2 * pad - k + 1


class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, noise, strength, pad, transposed, flip, act_id, alpha, gain, clamp, f16=False, may_be_sparse=False):
        # w is TAP-MAJOR here: [O, k, k, I] or [N, O, k, k, I]
        half = half_io(x, f16)                             # fp16 activation tensors in, fp16 out (weights stay fp32)
        x = x.contiguous() if half else x.contiguous().float()
        w = w.contiguous().float()
        n, i, h, wd = x.shape
        per_sample = (w.ndim == 5)
        o, k = w.shape[-4], w.shape[-2]
        assert w.shape[-1] == i and (not per_sample or w.shape[0] == n)
        wbs = o * i * k * k if per_sample else 0
        oh, ow = out_size(h, k, pad, transposed), out_size(wd, k, pad, transposed)
        bb = bias.contiguous().float() if bias is not None else None
        nz = noise.contiguous().float() if noise is not None else None
        ng = strength.detach().reshape(1).contiguous().float() if (noise is not None and strength is not None) else None
        of = _needed[0].get((oh, ow)) if (may_be_sparse and _needed[0]) else None
        if of is not None:
            assert of.dtype == torch.int32 and tuple(of.shape) == (n, (oh * ow + 15) // 16), 'needed_output: flags must be int32 [N, ceil(OH*OW/16)]'
        d = _desc(n, i, o, h, wd, k, pad, transposed, flip, wbs, bb, nz, ng, act_id, alpha, gain, clamp, tap_major=1, f16=f16, out_flags=of, half=half)
        frozen = (not per_sample) and not w.requires_grad          # shared weights that take no gradient (VGG extractors): cache their Winograd transform
        ws = _workspace(d, 0, x.device, w, frozen)      # noqa: F841  (keeps the scratch tensor alive until the launch is enqueued)
        y = _out_tensor(d, 0, (n, o, oh, ow), x.device)
        hip.call('spi_conv2d_fwd', ctypes.byref(d), hip.ptr(x), hip.ptr(w), hip.ptr(y), hip.stream())
        has_epi = act_id != 0 and (act_id != 1 or gain != 1 or clamp >= 0)
        ctx.save_for_backward(x, w, y if has_epi else None, nz, ng)
        ctx.cfg = (pad, transposed, flip, act_id, alpha, gain, clamp, has_epi, wbs, f16, may_be_sparse)
        ctx.frozen_w = frozen
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, w, y, nz, ng = ctx.saved_tensors
        pad, transposed, flip, act_id, alpha, gain, clamp, has_epi, wbs, f16, may_be_sparse = ctx.cfg
        n, i, h, wd = x.shape
        o, k = w.shape[-4], w.shape[-2]
        # every small accumulator of this backward (bias gradient, per-pixel noise sums, weight gradient) is carved from ONE zeroed
        # buffer: one fill launch instead of three
        n_tail = _ba.tail_zero_elems(dy, nz, ctx.needs_input_grad[3], ctx.needs_input_grad[4], ctx.needs_input_grad[2])
        n_dw = w.numel() if ctx.needs_input_grad[1] else 0
        zbuf = zero_arena.zeros(n_tail + n_dw, x.device) if n_tail + n_dw else None
        dz, d_noise, d_strength, d_bias = _ba.tail_backward(dy, y if has_epi else None, nz, ng, act_id, alpha, gain, clamp,
                                                            ctx.needs_input_grad[3], ctx.needs_input_grad[4], ctx.needs_input_grad[2],
                                                            zero_buf=zbuf)
        # (only convs that opted in -- the generator's; the loss networks behind the masks see dense gradients)
        flags = seg_flags(dz) if (_sparse[0] and may_be_sparse and dz.shape[2] * dz.shape[3] >= SPARSE_MIN_PIXELS) else None
        d = _desc(n, i, o, h, wd, k, pad, transposed, flip, wbs, tap_major=1, f16=f16, dy_flags=flags, dw_zeroed=1, half=(x.dtype == torch.float16))
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dd, dzd, wd_ = d, dz, w
            if d.act_dtype == 1 and o % 16 != 0:       # (torgb: 3 output channels)
                dzd, wd_, o16 = pad_o16(dz, w, o)
                dd = _desc(n, i, o16, h, wd, k, pad, transposed, flip, (o16 * i * k * k if wbs else 0), tap_major=1, f16=f16, dy_flags=flags, half=True)
            ws = _workspace(dd, 1, x.device, w, getattr(ctx, 'frozen_w', False) and dd is d)            # noqa: F841
            dx = _out_tensor(dd, 1, tuple(x.shape), x.device)
            hip.call('spi_conv2d_dgrad', ctypes.byref(dd), hip.ptr(dzd), hip.ptr(wd_), hip.ptr(dx), hip.stream())
        if ctx.needs_input_grad[1]:
            dw = zbuf[n_tail:].view(w.shape)
            ws2 = _workspace(d, 2, x.device)            # noqa: F841  (opt-in to the F(3x3, 2x2) weight-gradient kernel)
            hip.call('spi_conv2d_wgrad', ctypes.byref(d), hip.ptr(x), hip.ptr(dz), hip.ptr(dw), hip.stream())
        return dx, dw, d_bias, d_noise, d_strength, None, None, None, None, None, None, None, None, None


def to_tap_major(w):
    """[..., O, I, k, k] -> [..., O, k, k, I] (channels innermost), the layout the kernels consume."""
    return w.movedim(-3, -1).contiguous()


def conv2d(x, w, bias=None, noise=None, noise_strength=None, padding=0, transposed=False, flip=False, act=None, alpha=None,
           gain=None, clamp=None, tap_major=False, fp16=False, sparse_grad=False):
    """x [N,I,H,W]; w [O,I,k,k] (shared) or [N,O,I,k,k] (per sample) -- or already [.., O,k,k,I] with tap_major=True.
    act=None -> no activation/gain/clamp.  fp16: operands rounded to fp16 on their way into the matrix cores (fp32 accumulate,
    fp32 tensors) in all three passes -- the reference's `use_fp16` blocks.  sparse_grad: this conv's output gradient may hold large
    exactly-zero regions; its backward maps and skips them when run inside ``with sparse_gradients():``."""
    if not tap_major:
        w = to_tap_major(w)
    if act is None:
        act_id, a, g, c = (1 if (bias is not None or noise is not None) else 0), 0.0, 1.0, -1.0
    else:
        act_id, d_alpha, d_gain, _ = _ba.activation_funcs[act]
        assert act_id in (1, 2, 3), 'fused epilogue supports linear / relu / lrelu'
        a = float(d_alpha if alpha is None else alpha)
        g = float(d_gain if gain is None else gain)
        c = float(-1 if clamp is None else clamp)
    return _Conv2d.apply(x, w, bias, noise, noise_strength, int(padding), bool(transposed), bool(flip), act_id, a, g, c, precision(fp16), bool(sparse_grad))


def precision(fp16=False):
    """spi_conv_desc.compute_f16 for a conv: 1 = fp16 operands (the reference's use_fp16 blocks, opt-in), else the run-wide setting
    ``global_config.conv_precision``: 0 = exact fp32 MFMA, 2 / 3 = fp32 operands split into 2 / 3 bf16 pieces, 3 / 6 bf16 MFMAs."""
    from ...configs import global_config
    return 1 if fp16 else int(global_config.conv_precision)
