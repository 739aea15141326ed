"""ctypes binding of libspi_hip.so (the C ABI in include/spi_hip.h).

Thin by design: tensors go down as raw device pointers + sizes + the current HIP stream; outputs
are allocated by the caller with ``torch.empty``.  There is NO fallback: if the library cannot be
loaded, or a tensor is not a contiguous fp32 GPU tensor, the call raises.
"""
import ctypes
import os
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libspi_hip.so')
_lib = None
ABI_VERSION = 13          # SPI_ABI_VERSION of include/spi_hip.h this binding was written against

c_f = ctypes.c_float
c_i = ctypes.c_int
c_l = ctypes.c_int64
c_p = ctypes.c_void_p


class ConvDesc(ctypes.Structure):
    _fields_ = [('N', c_i), ('I', c_i), ('O', c_i), ('H', c_i), ('W', c_i), ('kh', c_i), ('kw', c_i), ('pad', c_i),
                ('transposed', c_i), ('flip', c_i), ('w_tap_major', c_i), ('compute_f16', c_i), ('w_batch_stride', c_l), ('bias', c_p), ('noise', c_p),
                ('noise_gain', c_p), ('act', c_i), ('alpha', c_f), ('gain', c_f), ('clamp', c_f), ('dy_seg_flags', c_p), ('out_seg_flags', c_p), ('dw_zeroed', c_i), ('out_zeroed', c_i),
                ('workspace', c_p), ('workspace_bytes', c_l), ('act_dtype', c_i), ('workspace_ready', c_i)]


class AffineJob(ctypes.Structure):
    """spi_affine_job of include/spi_hip.h"""
    _fields_ = [('x', c_p), ('w', c_p), ('b', c_p), ('y', c_p), ('g', c_p), ('dx_acc', c_p), ('dw', c_p), ('gain', c_f), ('O', c_i)]


AFFINE_MAX_JOBS = 32


class ModulateJob(ctypes.Structure):
    """spi_modulate_job of include/spi_hip.h"""
    _fields_ = [('weight', c_p), ('styles', c_p), ('w_out', c_p), ('dcoef', c_p), ('g', c_p), ('d_weight', c_p), ('d_styles', c_p),
                ('style_gain', c_f), ('O', c_i), ('I', c_i), ('T', c_i), ('demodulate', c_i)]


MODULATE_MAX_JOBS = 32

_SIGS = {
    'spi_abi_version': ([], c_i),
    'spi_sizeof_conv_desc': ([], c_i),
    'spi_ray_sampler': ([c_p, c_p, c_i, c_i, c_p, c_p, c_p], c_i),
    'spi_coarse_depths': ([c_p, c_l, c_i, c_f, c_f, c_p, c_p], c_i),
    'spi_nchw_to_nhwc': ([c_p, c_p, c_i, c_i, c_i, c_i, c_p], c_i),
    'spi_nhwc_to_nchw': ([c_p, c_p, c_i, c_i, c_i, c_i, c_p], c_i),
    'spi_sample_from_planes_fwd': ([c_p, c_p, c_i, c_l, c_i, c_i, c_f, c_p, c_p], c_i),
    'spi_sample_from_planes_bwd': ([c_p, c_p, c_i, c_l, c_i, c_i, c_f, c_p, c_p], c_i),
    'spi_triplane_decode_fwd': ([c_p] * 9 + [c_i, c_l, c_i, c_i, c_i, c_f, c_i, c_i, c_p, c_p, c_p], c_i),
    'spi_triplane_decode_bwd': ([c_p] * 11 + [c_i, c_l, c_i, c_i, c_i, c_f, c_i, c_i, c_p, c_p, c_p], c_i),
    'spi_triplane_decode_bwd_sorted': ([c_p] * 13 + [c_i, c_i, c_i, c_i, c_i, c_i, c_f] + [c_p] * 8, c_i),
    'spi_triplane_decode_bwd_sorted_ws': ([c_i, c_i, c_i, c_i], c_l),
    'spi_decoder_wgrad': ([c_p, c_l, c_p, c_p, c_p, c_p, c_p], c_i),
    'spi_minmax': ([c_p, c_l, c_p, c_p], c_i),
    'spi_raymarch_fwd': ([c_p] * 5 + [c_l, c_i, c_i, c_i, c_i] + [c_p] * 5, c_i),
    'spi_raymarch_bwd': ([c_p] * 8 + [c_l, c_i, c_i, c_i, c_i] + [c_p] * 5, c_i),
    'spi_importance_sample': ([c_p, c_p, c_p, c_l, c_i, c_i, c_p, c_i, c_p], c_i),
    'spi_merge_sort_depths': ([c_p, c_p, c_l, c_i, c_i, c_p, c_p, c_p], c_i),
    'spi_style_grad': ([c_p] * 6 + [c_i, c_i, c_i, c_i, c_f, c_p], c_i),
    'spi_seg_flags': ([c_p, c_p, c_i, c_i, c_l, c_p], c_i),
    'spi_bias_act': ([c_p] * 6 + [c_l, c_i, c_l, c_i, c_i, c_f, c_f, c_f, c_p], c_i),
    'spi_upfirdn2d': ([c_p] * 3 + [c_i] * 15 + [c_f, c_i, c_i] + [c_p] * 3 + [c_i, c_f, c_f, c_f, c_p], c_i),
    'spi_bias_act_t': ([c_p] * 6 + [c_l, c_i, c_l, c_i, c_i, c_f, c_f, c_f, c_i, c_p], c_i),
    'spi_upfirdn2d_t': ([c_p] * 3 + [c_i] * 4 + [c_p, c_p] + [c_i] * 11 + [c_f, c_i, c_i, c_i, c_p], c_i),
    'spi_upfirdn2d_fused_t': ([c_p] * 3 + [c_i] * 15 + [c_f, c_i, c_i] + [c_p] * 3 + [c_i, c_f, c_f, c_f, c_i, c_p], c_i),
    'spi_tail_bwd_t': ([c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_l, c_i, c_f, c_f, c_f, c_i, c_p], c_i),
    'spi_tail_bwd_dot_t': ([c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_l, c_i, c_f, c_f, c_f, c_p, c_p, c_p, c_p, c_i, c_p], c_i),
    'spi_chan_dot_t': ([c_p, c_p, c_p, c_l, c_i, c_l, c_p, c_p, c_p, c_i, c_f, c_f, c_i, c_p], c_i),
    'spi_seg_flags_t': ([c_p, c_p, c_i, c_i, c_l, c_i, c_p], c_i),
    'spi_filtered_lrelu_t': ([c_p] * 6 + [c_i] * 14 + [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_p], c_i),
    'spi_filtered_lrelu': ([c_p] * 6 + [c_i] * 14 + [c_f, c_f, c_f, c_i, c_i, c_i, c_p], c_i),
    'spi_filtered_lrelu_fused': ([c_p] * 6 + [c_i] * 14 + [c_f, c_f, c_f] + [c_i] * 8 + [c_p], c_i),
    'spi_filtered_lrelu_act': ([c_p, c_p, c_l, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_i, c_p], c_i),
    'spi_conv2d_workspace_bytes': ([ctypes.POINTER(ConvDesc), c_i], c_l),
    'spi_conv2d_out_accumulates': ([ctypes.POINTER(ConvDesc), c_i], c_i),
    'spi_conv2d_plan': ([ctypes.POINTER(ConvDesc), c_i, c_p], c_i),
    'spi_conv_wino_f4_set': ([c_i], None),
    'spi_conv2d_fwd': ([ctypes.POINTER(ConvDesc), c_p, c_p, c_p, c_p], c_i),
    'spi_conv2d_dgrad': ([ctypes.POINTER(ConvDesc), c_p, c_p, c_p, c_p], c_i),
    'spi_conv2d_wgrad': ([ctypes.POINTER(ConvDesc), c_p, c_p, c_p, c_p], c_i),
    'spi_rotate_warp': ([c_p] * 7 + [c_i, c_i, c_i, c_f, c_p, c_p, c_p], c_i),
    'spi_chan_dot': ([c_p, c_p, c_p, c_l, c_i, c_l, c_p, c_p, c_p, c_i, c_f, c_f, c_p], c_i),
    'spi_tail_bwd': ([c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_l, c_i, c_f, c_f, c_f, c_p], c_i),
    'spi_modulate_fwd': ([c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_p], c_i),
    'spi_modulate_bwd': ([c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_p], c_i),
    'spi_noise_reg_fwd': ([c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p], c_i),
    'spi_noise_reg_bwd': ([c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p], c_i),
    'spi_noise_renorm': ([c_p, c_p, c_i, c_p], c_i),
    'spi_lpips_layer_fwd': ([c_p, c_p, c_p, c_i, c_i, c_l, c_p, c_p], c_i),
    'spi_lpips_layer_bwd': ([c_p, c_p, c_p, c_p, c_i, c_i, c_l, c_p, c_p], c_i),
    'spi_roi_align_fwd': ([c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p], c_i),
    'spi_roi_align_bwd': ([c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p], c_i),
    'spi_contextual_workspace_bytes': ([c_i, c_i, c_i], c_l),
    'spi_contextual_fwd': ([c_p, c_i, c_i, c_i, c_f] + [c_p] * 7, c_i),
    'spi_contextual_bwd': ([c_p, c_p, c_i, c_i, c_i, c_f] + [c_p] * 6, c_i),
    'spi_adam_multi': ([c_p, c_p, c_i, c_l, c_f, c_f, c_f, c_f, c_i, c_p], c_i),
    'spi_adam_multi_pred': ([c_p, c_p, c_i, c_l, c_f, c_f, c_f, c_f, c_i, c_p, c_p], c_i),
    'spi_decoder_gains': ([c_p] * 4 + [c_f] * 4 + [c_p] * 4 + [c_i, c_p], c_i),
    'spi_affine_fwd': ([c_p, c_p, c_p, c_f, c_p, c_i, c_i, c_i, c_p], c_i),
    'spi_affine_bwd': ([c_p, c_p, c_p, c_f, c_p, c_p, c_i, c_i, c_i, c_p], c_i),
    'spi_modulate_multi_fwd': ([ctypes.POINTER(ModulateJob), c_i, c_i, c_p], c_i),
    'spi_modulate_multi_bwd': ([ctypes.POINTER(ModulateJob), c_i, c_i, c_p], c_i),
    'spi_affine_multi_fwd': ([ctypes.POINTER(AffineJob), c_i, c_i, c_i, c_l, c_p], c_i),
    'spi_affine_multi_bwd': ([ctypes.POINTER(AffineJob), c_i, c_i, c_i, c_l, c_p], c_i),
    'spi_adam_multi_dev': ([c_p, c_p, c_i, c_l, c_p, c_f, c_f, c_f, c_p], c_i),
}
EXPORTS = sorted(list(_SIGS) + ['spi_last_error'])


def lib():
    """Load (once) and return the ctypes handle.  Raises if the shared library is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f'{LIB_PATH} not found: build it with `python -m spi_amd.csrc.build` '
                               '(spi_amd has no CPU / PyTorch fallback for its kernels)')
        L = ctypes.CDLL(LIB_PATH)
        for name, (args, res) in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = res
        L.spi_last_error.restype = ctypes.c_char_p
        if L.spi_abi_version() != ABI_VERSION or L.spi_sizeof_conv_desc() != ctypes.sizeof(ConvDesc):
            raise RuntimeError('libspi_hip.so ABI mismatch (version %d, spi_conv_desc %d bytes; this binding: version %d, %d bytes): rebuild with '
                               '`python -m spi_amd.csrc.build`' % (L.spi_abi_version(), L.spi_sizeof_conv_desc(), ABI_VERSION, ctypes.sizeof(ConvDesc)))
        _lib = L
    return _lib


_PTR_DTYPES = (torch.float32, torch.int32, torch.int64, torch.uint8, torch.float16, torch.float64)      # (float16: the typed entry points / spi_conv_desc.act_dtype only)
DTYPE_IDS = {torch.float32: 0, torch.float16: 1, torch.float64: 2}          # SPI_DTYPE_* of the typed plugin entry points (spi_bias_act_t, spi_upfirdn2d_t)


def ptr_any(t):
    """Device pointer of a GPU tensor of any dtype / dense layout for the typed entry points (the caller passes dtype and strides)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('spi_amd kernels need GPU tensors (no CPU fallback); got a %s tensor' % t.device.type)
    return t.data_ptr()



def ptr(t):
    """Device pointer of a contiguous fp32 / int32 GPU tensor (None -> NULL).  (On the launch path ~250 times per step: the checks are
    ordered so that the good case costs three attribute reads.)"""
    if t is None:
        return None
    if t.is_cuda and t.is_contiguous() and t.dtype in _PTR_DTYPES:
        return t.data_ptr()
    if not t.is_cuda:
        raise RuntimeError('spi_amd kernels need GPU tensors (no CPU fallback); got a %s tensor' % t.device.type)
    if not t.is_contiguous():
        raise RuntimeError('spi_amd kernels need contiguous tensors')
    raise RuntimeError(f'unsupported dtype {t.dtype}')


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream():
    """Raw hipStream_t of torch's current stream on the current device."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def check(rc, name):
    if rc != 0:
        raise RuntimeError(f'{name} failed ({rc}): {lib().spi_last_error().decode()}')


_fns = {}


_TRACE_CALLS = bool(os.environ.get('SPI_TRACE_CALLS'))       # debugging aid: name every library call on stderr and wait for it (a GPU memory
                                                              # fault aborts the process: the last name printed is the faulting launch)


def call(name, *args):
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(lib(), name)
    if _TRACE_CALLS:
        import sys
        print(f'[spi call] {name} {[a if isinstance(a, (int, float)) else type(a).__name__ for a in args]}', file=sys.stderr, flush=True)
    rc = fn(*args)
    if rc != 0:
        check(rc, name)
    if _TRACE_CALLS:
        import torch
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.synchronize()

