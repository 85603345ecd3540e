"""audiocraft_amd -- MI355X-native implementation of AudioCraft's generation hot path
(EnCodec SEANet + RVQ, MusicGen LM decode) behind the reference's Python API.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all arithmetic on
the path runs in hand-written gfx950 HIP kernels in libacmi.so (C ABI: include/acmi.h).  There is no
CPU or PyTorch-eager fallback: using a model without the built library or without a GPU raises.
"""
__version__ = '0.1.0'
