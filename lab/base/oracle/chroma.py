"""Oracle (TEST INFRASTRUCTURE) -- melody front-end: ChromaExtractor (audiocraft/modules/chroma.py:16-66) and the
length matching of ChromaStemConditioner._get_wav_embedding (audiocraft/modules/conditioners.py:737-748).

PARITY UNPINNED against the reference binaries: the arithmetic of this row lives in two third-party packages that are
absent from /root/reference and not installed here (requirements.txt: `librosa` -- unpinned -- and
`torchaudio>=2.0.0,<2.1.2`), and the reference's tests hold no vector for it.  What follows
restates their published algorithms, anchored on the reference's call site:

  * `librosa.filters.chroma(sr, n_fft, tuning=0, n_chroma)` (chroma.py:41-42; librosa 0.10 `filters.chroma`, defaults
    ctroct=5.0, octwidth=2, norm=2, base_c=True, dtype float32):  Gaussian bumps on the log-frequency (chroma) axis, one
    per FFT bin, L2-normalised per bin, weighted by a Gaussian in octaves around ctroct, rolled so that row 0 is C.
  * `torchaudio.transforms.Spectrogram(n_fft, win_length, hop_length, power=2, center=True, pad=0, normalized=True)`
    (chroma.py:43-45; torchaudio 2.1 `functional.spectrogram`): `torch.stft` with a periodic Hann window, reflect
    padding of n_fft // 2 on both sides, frame normalisation "window" (divide by sqrt(sum w^2)), then |.|^2.

It is pinned instead (tests/test_oracle_golden.py) to an independent implementation of the same STFT definition
(`scipy.signal.stft`, a different code path) and to closed-form cases (a pure tone lands on its pitch class; a
null wav is one-hot on class 0, as `argmax` of an all-zero frame gives in the reference).
"""
import math
import typing as tp

import numpy as np


def chroma_filterbank(sr: int, n_fft: int, n_chroma: int = 12, tuning: float = 0.0, ctroct: float = 5.0,
                      octwidth: tp.Optional[float] = 2.0, base_c: bool = True) -> np.ndarray:
    """librosa.filters.chroma -> [n_chroma, 1 + n_fft // 2] float32 (computed in float64 like librosa)."""
    frequencies = np.linspace(0, sr, n_fft, endpoint=False)[1:]
    a440 = 440.0 * 2.0 ** (tuning / n_chroma)
    frqbins = n_chroma * np.log2(frequencies / (a440 / 16.0))          # hz_to_octs * bins_per_octave
    # the 0 Hz bin: 1.5 octaves below bin 1 (chroma 50 % rotated from bin 1, broad)
    frqbins = np.concatenate(([frqbins[0] - 1.5 * n_chroma], frqbins))
    binwidthbins = np.concatenate((np.maximum(frqbins[1:] - frqbins[:-1], 1.0), [1]))
    D = np.subtract.outer(frqbins, np.arange(0, n_chroma, dtype="d")).T
    n_chroma2 = np.round(float(n_chroma) / 2)
    D = np.remainder(D + n_chroma2 + 10 * n_chroma, n_chroma) - n_chroma2
    wts = np.exp(-0.5 * (2 * D / np.tile(binwidthbins, (n_chroma, 1))) ** 2)
    # util.normalize(norm=2, axis=0): columns to unit L2 norm (columns below `tiny` are left alone)
    length = np.sqrt(np.sum(np.abs(wts) ** 2, axis=0, keepdims=True))
    length[length < np.finfo(wts.dtype).tiny] = 1.0
    wts = wts / length
    if octwidth is not None:
        wts *= np.tile(np.exp(-0.5 * (((frqbins / n_chroma - ctroct) / octwidth) ** 2)), (n_chroma, 1))
    if base_c:
        wts = np.roll(wts, -3 * (n_chroma // 12), axis=0)
    return np.ascontiguousarray(wts[:, : int(1 + n_fft / 2)], dtype=np.float32)


def hann_periodic(n: int) -> np.ndarray:
    """torch.hann_window(n) (periodic=True): 0.5 - 0.5 cos(2 pi k / n)."""
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)).astype(np.float32)


def power_spectrogram(wav: np.ndarray, n_fft: int, hop: int) -> np.ndarray:
    """Spectrogram(power=2, center=True, normalized=True): wav [B, T] -> [B, 1 + n_fft // 2, 1 + T // hop] float32."""
    wav = np.asarray(wav, dtype=np.float32)
    B, T = wav.shape
    pad = n_fft // 2
    x = np.pad(wav, ((0, 0), (pad, pad)), mode='reflect')
    n_frames = 1 + T // hop
    w = hann_periodic(n_fft)
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    frames = x[:, idx] * w                                              # [B, frames, n_fft]
    spec = np.fft.rfft(frames.astype(np.float64), axis=-1)
    spec = spec / math.sqrt(float(np.sum(w.astype(np.float64) ** 2)))   # normalized="window"
    return (np.abs(spec) ** 2).astype(np.float32).transpose(0, 2, 1)


def chroma_extract(wav: np.ndarray, sample_rate: int, n_chroma: int = 12, radix2_exp: int = 12, argmax: bool = False,
                   return_raw: bool = False):
    """ChromaExtractor.forward (chroma.py:46-66): wav [B, T] (or [B, 1, T]) -> [B, frames, n_chroma] float32."""
    wav = np.asarray(wav, dtype=np.float32)
    if wav.ndim == 3:
        wav = wav[:, 0]
    n_fft = 2 ** radix2_exp
    hop = n_fft // 4
    T = wav.shape[-1]
    if T < n_fft:   # a wav nullified by the conditioner (or simply short): zero pad, centred
        p = n_fft - T
        r = 0 if p % 2 == 0 else 1
        wav = np.pad(wav, ((0, 0), (p // 2, p // 2 + r)))
    spec = power_spectrogram(wav, n_fft, hop)                           # [B, F, t]
    fb = chroma_filterbank(sample_rate, n_fft, n_chroma)
    raw = np.einsum('cf,bft->bct', fb.astype(np.float64), spec.astype(np.float64)).astype(np.float32)
    denom = np.maximum(np.abs(raw).max(axis=1, keepdims=True), 1e-6)    # F.normalize(p=inf, dim=-2, eps=1e-6)
    norm = (raw / denom).transpose(0, 2, 1)                             # b d t -> b t d
    if argmax:
        idx = norm.argmax(-1)
        out = np.zeros_like(norm)
        np.put_along_axis(out, idx[..., None], 1.0, axis=-1)
        norm = out
    if return_raw:
        return norm, raw.transpose(0, 2, 1)
    return norm


def match_length(chroma: np.ndarray, chroma_len: int) -> np.ndarray:
    """ChromaStemConditioner._get_wav_embedding with match_len_on_eval (conditioners.py:737-748): truncate, or tile
    periodically and truncate."""
    T = chroma.shape[1]
    if T > chroma_len:
        return chroma[:, :chroma_len]
    if T < chroma_len:
        n_repeat = int(math.ceil(chroma_len / T))
        return np.tile(chroma, (1, n_repeat, 1))[:, :chroma_len]
    return chroma
