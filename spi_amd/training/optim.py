"""Adam for the inversion loops: one HIP launch per step over flat parameter / gradient / moment buffers.

Semantics of ``torch.optim.Adam(params, lr, betas=(0.9, 0.999), eps=1e-8)`` as both SPI loops use it
(spi/training/projectors/mirror_projector.py:58, spi/training/coaches/base_coach.py:133-135).
MI355X-first: at construction every parameter's storage is moved into ONE contiguous fp32 buffer
(``p.data`` becomes a view) and so is its gradient; ``zero_grad`` is one memset, ``step`` is one
kernel (``spi_adam_multi``) streaming 4 flat arrays -- 28 B/parameter, HBM-bound -- instead of
O(#tensors) launches.  A parameter that NEVER receives a gradient sees g = 0, for which the update is
exactly 0 (m = v = 0): the same end state as torch skipping it.  Both SPI loops satisfy "always or never":
every tensor the main loss reaches gets a gradient on every step (W+ and the 13 noise maps in stage 1; all of
backbone.synthesis / superresolution / decoder in stage 2) and the mapping network never does.  A parameter that
received gradients on SOME steps only would differ from torch.optim.Adam(set_to_none=True), which skips it with its
own step counter while this kernel keeps decaying its moments under the shared step count -- not a case either
loop produces, and `step()` asserts it in debug runs (SPI_ADAM_CHECK=1).
"""
import os
import torch
from .. import hip

_CHECK = bool(os.environ.get('SPI_ADAM_CHECK'))


class Adam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params]
        assert len(self.params) > 0
        dev = self.params[0].device
        self.param_groups = [dict(params=self.params, lr=lr, betas=betas, eps=eps)]
        total = sum(p.numel() for p in self.params)
        self.flat_p = torch.empty(total, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(total, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(total, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(total, device=dev, dtype=torch.float32)
        off = 0
        for p in self.params:
            n = p.numel()
            self.flat_p[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + n].view(p.shape)
            p.grad = self.flat_g[off:off + n].view(p.shape)
            off += n
        self.step_count = 0
        self._table = torch.tensor([self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr()],
                                   dtype=torch.int64, device=dev)
        self._sizes = torch.tensor([total], dtype=torch.int64, device=dev)
        self._total = total

    def zero_grad(self, set_to_none=False):
        self.flat_g.zero_()
        off = 0
        for p in self.params:               # re-attach in case autograd replaced a .grad tensor
            n = p.numel()
            g = p.grad
            if g is None or g.data_ptr() != self.flat_g.data_ptr() + 4 * off:
                p.grad = self.flat_g[off:off + n].view(p.shape)
            off += n

    def reset_state(self):
        self.exp_avg.zero_(); self.exp_avg_sq.zero_(); self.flat_g.zero_()
        self.step_count = 0

    def hyper_values(self, step_count=None):
        """(lr, 1 - beta1^t, sqrt(1 - beta2^t)) exactly as spi_adam_multi forms them on the host (fp32 powf / sqrtf), for `step(hyper=...)`."""
        import ctypes
        import ctypes.util
        libm = getattr(Adam, '_libm', None)
        if libm is None:
            libm = Adam._libm = ctypes.CDLL(ctypes.util.find_library('m') or 'libm.so.6')
            libm.powf.restype, libm.powf.argtypes = ctypes.c_float, [ctypes.c_float, ctypes.c_float]
            libm.sqrtf.restype, libm.sqrtf.argtypes = ctypes.c_float, [ctypes.c_float]
        grp = self.param_groups[0]
        t = float(self.step_count + 1 if step_count is None else step_count)
        f32 = lambda v: ctypes.c_float(v).value
        bc1 = f32(1.0 - libm.powf(grp['betas'][0], t))
        bc2 = libm.sqrtf(f32(1.0 - libm.powf(grp['betas'][1], t)))
        return f32(grp['lr']), bc1, bc2

    @torch.no_grad()
    def step(self, hyper=None, skip=None):
        """hyper: optional DEVICE tensor [lr, 1 - beta1^t, sqrt(1 - beta2^t)] (see `hyper_values`): the launch then carries no step-dependent
        host scalar and can be replayed from a captured HIP graph.
        skip: optional DEVICE uint8 [1]; non-zero when the kernel runs -> this step changes nothing (`spi_adam_multi_pred`: the early-stop decision
        of a loop whose host runs ahead of the GPU)."""
        off = 0
        for p in self.params:               # gradients that autograd placed elsewhere are folded back in
            n = p.numel()
            g = p.grad
            if g is not None and g.data_ptr() != self.flat_g.data_ptr() + 4 * off:
                self.flat_g[off:off + n].add_(g.reshape(-1))
                p.grad = self.flat_g[off:off + n].view(p.shape)
            off += n
        self.step_count += 1
        if _CHECK:                                       # debug: the always-or-never assumption above
            off = 0
            had = getattr(self, '_had_grad', None)
            now = []
            for p in self.params:
                n = p.numel()
                now.append(bool((self.flat_g[off:off + n] != 0).any()))
                off += n
            if had is not None and any(h and not n_ for h, n_ in zip(had, now)):
                raise AssertionError('a parameter that had a gradient before has none now: flat Adam differs from torch.optim.Adam here')
            self._had_grad = [h or n_ for h, n_ in zip(had, now)] if had is not None else now
        grp = self.param_groups[0]
        if hyper is not None:
            hip.call('spi_adam_multi_dev', hip.ptr(self._table), hip.ptr(self._sizes), 1, self._total, hip.ptr(hyper), float(grp['betas'][0]),
                     float(grp['betas'][1]), float(grp['eps']), hip.stream())
            return
        if skip is not None:
            assert skip.dtype == torch.uint8 and skip.is_cuda
            hip.call('spi_adam_multi_pred', hip.ptr(self._table), hip.ptr(self._sizes), 1, self._total, float(grp['lr']), float(grp['betas'][0]),
                     float(grp['betas'][1]), float(grp['eps']), self.step_count, skip.data_ptr(), hip.stream())
            return
        hip.call('spi_adam_multi', hip.ptr(self._table), hip.ptr(self._sizes), 1, self._total, float(grp['lr']), float(grp['betas'][0]),
                 float(grp['betas'][1]), float(grp['eps']), self.step_count, hip.stream())
