"""Audio-rate helpers of the generation API -- mirrors the parts of `audiocraft.data.audio_utils` the path touches
(reference audiocraft/data/audio_utils.py:18-59 `convert_audio_channels`, `convert_audio`).

Resampling in the reference is `julius.resample_frac`, a third-party routine; here the polyphase FIR it defines runs in a
HIP kernel (`acmi_resample_frac`), its filter bank built once per rate pair on the host."""
import math
import typing as tp

import torch

from . import _C

_KERNELS: tp.Dict[tp.Tuple[int, int, str], tp.Tuple[torch.Tensor, int]] = {}


def _resample_kernel(old_sr: int, new_sr: int, device, zeros: int = 24, rolloff: float = 0.945):
    """Filter bank of julius.ResampleFrac (old_sr, new_sr already reduced by their gcd): for output phase i the taps
    sinc(t) * cos^2(t / zeros / 2) at t = (-i / new_sr + idx / old_sr) * sr * pi clamped to +-zeros * pi, with
    sr = rolloff * min(old, new); rows normalised to unit sum.  float32 arithmetic like the original."""
    key = (old_sr, new_sr, str(device))
    if key not in _KERNELS:
        sr = min(new_sr, old_sr) * rolloff
        width = math.ceil(zeros * old_sr / sr)
        idx = torch.arange(-width, width + old_sr).float()
        rows = []
        for i in range(new_sr):
            t = (-i / new_sr + idx / old_sr) * sr
            t = t.clamp(-zeros, zeros) * math.pi
            window = torch.cos(t / zeros / 2) ** 2
            sinc = torch.where(t == 0, torch.ones_like(t), torch.sin(t) / t)
            k = sinc * window
            rows.append(k / k.sum())
        _KERNELS[key] = (torch.stack(rows).contiguous().to(device), width)
    return _KERNELS[key]


def resample_frac(x: torch.Tensor, old_sr: int, new_sr: int) -> torch.Tensor:
    """[..., T] at old_sr -> [..., floor(T * new_sr / old_sr)] at new_sr."""
    old_sr, new_sr = int(old_sr), int(new_sr)
    if old_sr == new_sr:
        return x
    g = math.gcd(old_sr, new_sr)
    o, n = old_sr // g, new_sr // g
    # The filter runs on the MI355X only (acmi_resample_frac; there is no CPU implementation in the product).  Melodies and
    # prompts are usually loaded on the host (MusicGen.generate_with_chroma calls convert_audio on them): such an input
    # makes the round trip through the accelerator and comes back on its own device, like the reference's call would.
    src_dev = x.device
    if not x.is_cuda:
        if not torch.cuda.is_available():
            raise RuntimeError("resample_frac runs on the MI355X (acmi_resample_frac): no accelerator is visible")
        x = x.cuda()
    kernel, width = _resample_kernel(o, n, x.device)
    shape = x.shape
    flat = x.reshape(-1, shape[-1]).float().contiguous()
    out_len = int(math.floor(n * shape[-1] / o))
    y = _C.resample_frac(flat, kernel, o, n, width, out_len)
    return y.reshape(*shape[:-1], out_len).to(src_dev)


def convert_audio_channels(wav: torch.Tensor, channels: int = 2) -> torch.Tensor:
    """reference audio_utils.py:18-51"""
    *shape, src_channels, length = wav.shape
    if src_channels == channels:
        return wav
    if channels == 1:
        return wav.mean(dim=-2, keepdim=True)
    if src_channels == 1:
        return wav.expand(*shape, channels, length)
    if src_channels >= channels:
        return wav[..., :channels, :]
    raise ValueError('The audio file has less channels than requested but is not mono.')


def convert_audio(wav: torch.Tensor, from_rate: float, to_rate: float, to_channels: int) -> torch.Tensor:
    """reference audio_utils.py:54-59: resample, then convert the channels."""
    wav = resample_frac(wav, int(from_rate), int(to_rate))
    return convert_audio_channels(wav, to_channels)
