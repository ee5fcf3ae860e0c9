"""dvd_gan_amd -- MI355X-native DVD-GAN training hot path.

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed); every
arithmetic op of the G / D_s / D_t step runs in hand-written gfx950 kernels behind the C ABI
declared in include/dvdgan_hip.h (csrc/libdvdgan_hip.so).  There is no CPU or eager fallback:
`dvd_gan_amd.lib.lib()` raises if the library is missing.
"""
__version__ = "0.1.0"
