"""spi_amd: MI355X-native implementation of SPI's per-image inversion inner loop.

Host code is Python on PyTorch-ROCm (device memory, streams, autograd plumbing); all hot-path
arithmetic runs in hand-written HIP kernels for gfx950 behind the C ABI declared in
``include/spi_hip.h`` (built to ``spi_amd/csrc/libspi_hip.so``).  There is no CPU fallback:
calling a compute op without the library / a GPU raises.
"""
__version__ = '0.1.0'
