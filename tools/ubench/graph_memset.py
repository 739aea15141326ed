#!/usr/bin/env python3
"""Is a memset node of a captured HIP graph reliable on replay?  zero (hipMemsetAsync, n bytes) -> add 1 twice -> compare, replayed.
ROCm 7.0 / torch 2.10: byte counts that are not a multiple of 16 leave garbage behind from the second replay on -- which is why
libspi_hip.so zeroes with a kernel (spi_zero_async) and why torch's own multi-block reductions (Reduce.cuh clears its semaphore
buffer with cudaMemsetAsync, 4 bytes per block) can misbehave inside replayed graphs."""
import ctypes, torch
dev = 'cuda'
hiprt = ctypes.CDLL('libamdhip64.so')
raw = torch._C._cuda_getCurrentRawStream
bad_sizes, ok_sizes = [], []
for nbytes in list(range(4, 300, 4)) + [1000, 1004, 1008, 4096, 4100, 65536, 65540, (1 << 20) + 4, (1 << 20) + 16]:
    n = nbytes // 4
    a = torch.empty(n, device=dev); ones = torch.ones(n, device=dev)
    def body():
        rc = hiprt.hipMemsetAsync(ctypes.c_void_p(a.data_ptr()), 0, ctypes.c_size_t(nbytes), ctypes.c_void_p(raw(torch.cuda.current_device())))
        assert rc == 0
        a.add_(ones); a.add_(ones)
    body(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    wrong = 0
    for rep in range(6):
        g.replay(); torch.cuda.synchronize()
        wrong += int((a != 2).sum().item())
    (bad_sizes if wrong else ok_sizes).append(nbytes)
print('memset-node byte counts that FAIL on replay:', bad_sizes)
print('byte counts that are fine:', ok_sizes)
