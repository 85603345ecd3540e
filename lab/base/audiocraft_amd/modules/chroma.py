"""Chroma extraction and quantisation on the device -- API mirror of `audiocraft.modules.chroma.ChromaExtractor`
(reference audiocraft/modules/chroma.py:16-66).

The reference composes two third-party pieces: `torchaudio.transforms.Spectrogram` and the `librosa.filters.chroma`
table.  Here the framed power spectrum, the filterbank contraction, the inf-norm and the argmax one-hot are ONE HIP
kernel (`acmi_chroma`, include/acmi.h); the two constant tables it reads -- the chroma filterbank and the FFT twiddles --
are computed once on the host in double precision from their published definitions.
"""
import math
import typing as tp

import numpy as np
import torch
from torch import nn

from .. import _C


def _chroma_filterbank(sr: int, n_fft: int, n_chroma: int) -> np.ndarray:
    """The table `librosa.filters.chroma(sr=sr, n_fft=n_fft, tuning=0, n_chroma=n_chroma)` returns (its defaults:
    Gaussian bump of each FFT bin on the circular chroma axis, unit L2 norm per bin, Gaussian weighting over octaves
    centred on octave 5 with width 2, row 0 = C): [n_chroma, n_fft // 2 + 1] float32."""
    k = np.arange(1, n_fft, dtype=np.float64)
    pitch = n_chroma * np.log2(k * (sr / n_fft) / (440.0 / 16.0))       # bin centre in chroma bins above A0 / 16... (octs * n)
    pitch = np.concatenate(([pitch[0] - 1.5 * n_chroma], pitch))         # the DC bin sits 1.5 octaves below bin 1
    width = np.concatenate((np.maximum(np.diff(pitch), 1.0), [1.0]))
    half = np.round(n_chroma / 2.0)
    dist = pitch[None, :] - np.arange(n_chroma, dtype=np.float64)[:, None]
    dist = np.remainder(dist + half + 10 * n_chroma, n_chroma) - half     # wrapped distance on the chroma circle
    w = np.exp(-0.5 * (2.0 * dist / width[None, :]) ** 2)
    col = np.sqrt((w * w).sum(axis=0, keepdims=True))
    col[col < np.finfo(np.float64).tiny] = 1.0
    w = w / col
    w = w * np.exp(-0.5 * ((pitch / n_chroma - 5.0) / 2.0) ** 2)[None, :]
    w = np.roll(w, -3 * (n_chroma // 12), axis=0)
    return np.ascontiguousarray(w[:, : n_fft // 2 + 1], dtype=np.float32)


class ChromaExtractor(nn.Module):
    """Same constructor and output as the reference class (chroma.py:16-66) for the configuration MusicGen uses
    (winlen = nfft = 2 ** radix2_exp, winhop = winlen // 4, norm = inf); other combinations raise."""

    def __init__(self, sample_rate: int, n_chroma: int = 12, radix2_exp: int = 12, nfft: tp.Optional[int] = None,
                 winlen: tp.Optional[int] = None, winhop: tp.Optional[int] = None, argmax: bool = False,
                 norm: float = torch.inf, device=None):
        super().__init__()
        self.winlen = winlen or 2 ** radix2_exp
        self.nfft = nfft or self.winlen
        self.winhop = winhop or (self.winlen // 4)
        if self.nfft != self.winlen or self.winhop * 4 != self.winlen or self.nfft & (self.nfft - 1) or norm != torch.inf:
            raise NotImplementedError("acmi_chroma implements nfft == winlen == 4 * winhop (a power of two) with the "
                                      "inf-norm: the ChromaStemConditioner configuration")
        self.radix2_exp = int(math.log2(self.nfft))
        self.sample_rate = sample_rate
        self.n_chroma = n_chroma
        self.norm = norm
        self.argmax = argmax
        k = np.arange(self.nfft // 2, dtype=np.float64) * (2.0 * np.pi / self.nfft)
        tw = np.stack([np.cos(k), -np.sin(k)], axis=1).astype(np.float32)
        # not part of the state dict (the reference's `fbanks` is non-persistent too)
        self.register_buffer('fbanks', torch.from_numpy(_chroma_filterbank(sample_rate, self.nfft, n_chroma)).to(device),
                             persistent=False)
        self.register_buffer('twiddle', torch.from_numpy(tw).to(device), persistent=False)

    def forward(self, wav: torch.Tensor, return_raw: bool = False):
        """wav [B, T] or [B, 1, T] -> chroma [B, frames, n_chroma] (one-hot if `argmax`)."""
        if wav.dim() == 3:
            assert wav.shape[1] == 1, "mono input expected"
            wav = wav[:, 0]
        wav = wav.to(device=self.fbanks.device, dtype=torch.float32).contiguous()
        return _C.chroma(wav, self.radix2_exp, self.twiddle, self.fbanks, self.argmax, want_raw=return_raw)
