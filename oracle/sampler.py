"""Test infrastructure (CPU): host restatement of the device sampler's random draw, so that SAMPLED tokens can be compared
token by token instead of only in distribution.

The reference draws with torch.multinomial on its own generator (audiocraft/utils/utils.py:88-105, :108-122;
audiocraft/models/lm.py:402-418), whose stream a device kernel cannot share.  libacmi's `sample_kernel`
(audiocraft_amd/csrc/acmi_lm.hip) uses the same ALGORITHM as torch.multinomial -- an exponential race, argmax_i p_i / q_i with
q_i ~ Exp(1) -- on a counter-based generator (Philox4x32-10), which a host can replay exactly:

    counter = (i, b * K + k, step_lo, step_hi)      i = vocabulary index, (b, k) = sample / codebook, step = stream position
    key     = (seed_lo, seed_hi)
    u_i     = ((philox(counter, key)[0] >> 8) + 0.5) / 2^24
    token   = argmax over {i : p_i >= kth largest p, p_i > 0} of p_i / (-log u_i)      (first index on ties)

`race` takes the ORACLE's probabilities, so a test that feeds it the device's seed and positions checks the device's sampled
tokens against the reference arithmetic (softmax, top-k support incl. ties) with the random stream held fixed.
Philox4x32-10: Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (SC'11); constants as in Random123 / cuRAND.
"""
import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85


def philox4x32_10(c0, c1, c2, c3, k0: int, k1: int):
    """Counters: uint32 arrays of one shape (or scalars, broadcast); key: two python ints.  -> four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in np.broadcast_arrays(c0, c1, c2, c3))
    k0 &= 0xffffffff
    k1 &= 0xffffffff
    for _ in range(10):
        p0 = _M0 * c0.astype(np.uint64)
        p1 = _M1 * c2.astype(np.uint64)
        n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ np.uint32(k0)
        n1 = p1.astype(np.uint32)
        n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ np.uint32(k1)
        n3 = p0.astype(np.uint32)
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + _W0) & 0xffffffff
        k1 = (k1 + _W1) & 0xffffffff
    return c0, c1, c2, c3


def uniforms(card: int, row: int, step: int, seed: int) -> np.ndarray:
    """u_i in (0, 1) for i < card, as f64 (every value is exactly representable in f32, like on the device)."""
    i = np.arange(card, dtype=np.uint32)
    r0, _, _, _ = philox4x32_10(i, np.uint32(row), np.uint32(step & 0xffffffff), np.uint32((step >> 32) & 0xffffffff),
                                seed & 0xffffffff, (seed >> 32) & 0xffffffff)
    return ((r0 >> np.uint32(8)).astype(np.float64) + 0.5) / 16777216.0


def race(probs: np.ndarray, top_k: int, row: int, step: int, seed: int):
    """probs: [card] probabilities of one (sample, codebook) at one position (f32 or f64).
    -> (token, margin, boundary): margin = 1 - second best / best of p_i / q_i (a margin below ~1e-5 is a NEAR TIE: the
    device's f32 rounding of p, log and the division may pick the runner-up); boundary = True when the winner or the
    runner-up sits within 1e-6 (relative) of the top-k threshold, i.e. the support itself is a near tie."""
    p = np.asarray(probs, dtype=np.float64)
    card = p.shape[0]
    thr = 0.0
    if 0 < top_k < card:
        thr = np.partition(p, card - top_k)[card - top_k]      # k-th largest; ties are kept (p >= thr)
    q = -np.log(uniforms(card, row, step, seed))
    r = np.where((p >= thr) & (p > 0), p / q, -1.0)
    order = np.argsort(-r, kind='stable')
    best, second = int(order[0]), int(order[1])
    margin = 1.0 - r[second] / r[best] if r[second] > 0 else 1.0
    boundary = bool(thr > 0 and (abs(p[best] - thr) <= 1e-6 * thr or abs(p[second] - thr) <= 1e-6 * thr))
    return best, float(margin), boundary
