"""Oracle (TEST INFRASTRUCTURE) -- fractional resampling as `convert_audio` does it (audiocraft/data/audio_utils.py:54-59
calls `julius.resample_frac(wav, from_rate, to_rate)`).

PARITY UNPINNED against the reference binary: julius (requirements.txt: `julius`, unpinned) is a third-party package that
is neither in /root/reference nor installed here.  This restates the published algorithm of `julius.ResampleFrac`
(julius 0.2.x, resample.py): rates reduced by their gcd; for each of the new_sr output phases a windowed-sinc low-pass
(zeros = 24 crossings, cut-off 0.945 x min(old, new), Hann-squared-cosine window) evaluated at the fractional delay of
that phase and normalised to unit sum; the input replicate-padded; output length floor(T * new / old).
Pinned in tests/test_oracle_golden.py through its defining properties (identity at equal rates, DC and in-band tones
preserved to 1e-3, out-of-band tones suppressed, agreement with scipy.signal.resample_poly away from the edges)."""
import math

import numpy as np


def resample_frac(x: np.ndarray, old_sr: int, new_sr: int, zeros: int = 24, rolloff: float = 0.945) -> np.ndarray:
    x = np.asarray(x, dtype=np.float32)
    if old_sr == new_sr:
        return x
    g = math.gcd(old_sr, new_sr)
    old, new = old_sr // g, new_sr // g
    sr = min(new, old) * rolloff
    width = math.ceil(zeros * old / sr)
    idx = np.arange(-width, width + old, dtype=np.float32)
    kernels = []
    for i in range(new):
        t = ((-i / new + idx / old) * sr).astype(np.float32)
        t = np.clip(t, -zeros, zeros) * np.float32(math.pi)
        window = np.cos(t / zeros / 2) ** 2
        sinc = np.where(t == 0, np.float32(1), np.sin(t) / np.where(t == 0, np.float32(1), t))
        k = (sinc * window).astype(np.float32)
        kernels.append(k / k.sum())
    kernel = np.stack(kernels)                                    # [new, 2 width + old]
    shape = x.shape
    T = shape[-1]
    flat = x.reshape(-1, T)
    xp = np.pad(flat, ((0, 0), (width, width + old)), mode='edge')
    K = kernel.shape[1]
    n_frames = (xp.shape[1] - K) // old + 1
    frames = np.stack([xp[:, n * old:n * old + K] for n in range(n_frames)], axis=1)     # [rows, frames, K]
    y = np.einsum('rfk,ik->rfi', frames.astype(np.float64), kernel.astype(np.float64)).reshape(flat.shape[0], -1)
    out_len = int(math.floor(new * T / old))
    return y[:, :out_len].astype(np.float32).reshape(*shape[:-1], out_len)
