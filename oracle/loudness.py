"""Oracle (TEST INFRASTRUCTURE) -- integrated loudness as `normalize_loudness` measures it (audiocraft/data/audio_utils.py:62-88
calls `torchaudio.transforms.Loudness(sample_rate)(wav)` = `torchaudio.functional.loudness`).

PARITY UNPINNED against the reference binary: torchaudio (requirements.txt: `torchaudio>=2.0.0`) is a third-party package that
is neither in /root/reference nor installed here.  This restates the published algorithm of torchaudio 2.x
(`functional/functional.py::loudness`, `functional/filtering.py::treble_biquad / highpass_biquad / biquad / lfilter`), which
follows ITU-R BS.1770-4:

  1. K-weighting, two biquads from the RBJ audio-EQ cookbook forms torchaudio uses:
       high shelf  +4 dB at 1500 Hz, Q = 1 / sqrt(2)        (treble_biquad)
       high-pass   38 Hz, Q = 0.5                            (highpass_biquad)
     each run as a direct-form recursion whose OUTPUT IS CLAMPED to [-1, 1] (lfilter's default `clamp=True`);
  2. mean square of 400 ms blocks with 75 % overlap (`unfold(gate, step)`: only whole blocks), per channel;
  3. channel weights 1, 1, 1, 1.41, 1.41; block loudness -0.691 + 10 log10(sum_c g_c z_c);
  4. absolute gate -70 LKFS, then the relative gate 10 LU below the loudness of the absolutely gated blocks;
  5. LKFS = -0.691 + 10 log10(sum_c g_c mean_{gated blocks} z_c).

Written independently of `audiocraft_amd/data_audio.py` (which filters with scipy.signal.lfilter and gathers the blocks by
index): the recursions and the block means below are explicit loops.  Pinned in tests/test_oracle_golden.py through what the
recommendation itself fixes -- a 997 Hz sine of amplitude a in one channel reads -3.01 + 20 log10(a) LKFS, a second equal
channel adds 3.01 dB, blocks below the gates do not count -- and through the K-weighting coefficients BS.1770-4 tabulates at
48 kHz (the cookbook forms reproduce them to ~1e-3)."""
import math

import numpy as np

GATE_S, OVERLAP, GAMMA_ABS, KWEIGHT_BIAS = 0.4, 0.75, -70.0, -0.691
CHANNEL_GAINS = (1.0, 1.0, 1.0, 1.41, 1.41)


def treble_coefficients(sample_rate: float, gain_db: float = 4.0, central_freq: float = 1500.0, Q: float = 1 / math.sqrt(2)):
    """-> (b, a), a[0] = 1: torchaudio's treble_biquad (high shelf)."""
    w0 = 2 * math.pi * central_freq / sample_rate
    alpha = math.sin(w0) / 2 / Q
    A = 10.0 ** (gain_db / 40.0)
    c = math.cos(w0)
    sq = 2 * math.sqrt(A) * alpha
    b = [A * ((A + 1) + (A - 1) * c + sq), -2 * A * ((A - 1) + (A + 1) * c), A * ((A + 1) + (A - 1) * c - sq)]
    a = [(A + 1) - (A - 1) * c + sq, 2 * ((A - 1) - (A + 1) * c), (A + 1) - (A - 1) * c - sq]
    return [v / a[0] for v in b], [v / a[0] for v in a]


def highpass_coefficients(sample_rate: float, cutoff: float = 38.0, Q: float = 0.5):
    """-> (b, a), a[0] = 1: torchaudio's highpass_biquad."""
    w0 = 2 * math.pi * cutoff / sample_rate
    alpha = math.sin(w0) / 2 / Q
    c = math.cos(w0)
    b = [(1 + c) / 2, -1 - c, (1 + c) / 2]
    a = [1 + alpha, -2 * c, 1 - alpha]
    return [v / a[0] for v in b], [v / a[0] for v in a]


def biquad_clamped(x: np.ndarray, b, a) -> np.ndarray:
    """y[n] = b0 x[n] + b1 x[n-1] + b2 x[n-2] - a1 y[n-1] - a2 y[n-2] over the last axis, then clamp to [-1, 1] (the clamp is
    applied to the finished output, as torchaudio's lfilter does -- not inside the recursion)."""
    x = np.asarray(x, dtype=np.float64)
    y = np.zeros_like(x)
    x1 = np.zeros(x.shape[:-1]); x2 = np.zeros(x.shape[:-1]); y1 = np.zeros(x.shape[:-1]); y2 = np.zeros(x.shape[:-1])
    for n in range(x.shape[-1]):
        xn = x[..., n]
        yn = b[0] * xn + b[1] * x1 + b[2] * x2 - a[1] * y1 - a[2] * y2
        y[..., n] = yn
        x2, x1, y2, y1 = x1, xn, y1, yn
    return np.clip(y, -1.0, 1.0)


def loudness(wav: np.ndarray, sample_rate: int) -> float:
    """Integrated loudness in LKFS of wav [C, T] (or [T]); -inf when no block passes the absolute gate."""
    x = np.asarray(wav, dtype=np.float64)
    if x.ndim == 1:
        x = x[None]
    C = x.shape[0]
    if C > 5:
        raise ValueError("only up to 5 channels are supported")
    gate = int(round(GATE_S * sample_rate))
    step = int(round(gate * (1 - OVERLAP)))
    if x.shape[-1] < gate:
        raise ValueError("loudness needs at least one 400 ms block")
    x = biquad_clamped(x, *treble_coefficients(sample_rate))
    x = biquad_clamped(x, *highpass_coefficients(sample_rate))
    blocks = []
    start = 0
    while start + gate <= x.shape[-1]:
        seg = x[:, start:start + gate]
        blocks.append((seg * seg).sum(axis=-1) / gate)
        start += step
    z = np.stack(blocks, axis=-1)                       # [C, blocks]
    g = np.asarray(CHANNEL_GAINS[:C])
    with np.errstate(divide='ignore'):
        block_l = KWEIGHT_BIAS + 10 * np.log10((g[:, None] * z).sum(axis=0))
    keep = block_l > GAMMA_ABS
    if not keep.any():
        return -math.inf
    gamma_rel = KWEIGHT_BIAS + 10 * math.log10(float((g * z[:, keep].mean(axis=-1)).sum())) - 10.0
    keep = keep & (block_l > gamma_rel)
    if not keep.any():   # cannot happen for gamma_rel 10 LU below the gated mean; kept for symmetry with torchaudio's 0 / 0
        return math.nan
    return KWEIGHT_BIAS + 10 * math.log10(float((g * z[:, keep].mean(axis=-1)).sum()))
