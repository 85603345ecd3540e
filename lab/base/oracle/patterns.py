"""Oracle (TEST INFRASTRUCTURE) -- delay-pattern codebook interleaving, as naive python loops.

Follows audiocraft/modules/codebooks_patterns.py:339-356 (DelayedPatternProvider.get_pattern),
:154-179 (build_pattern_sequence), :225-248 (revert_pattern_sequence), :116-118
(get_first_step_with_timesteps); the loop form mirrors the reference's own independent checks in
tests/modules/test_codebooks_patterns.py:107-146.
"""
import typing as tp

import torch


def delayed_layout(n_q: int, timesteps: int, delays: tp.Optional[tp.List[int]] = None):
    """layout[s] = list of (t, q) placed at sequence step s; layout[0] == [] (special-token step)."""
    delays = list(range(n_q)) if delays is None else delays
    out: tp.List[tp.List[tp.Tuple[int, int]]] = [[]]
    for t in range(timesteps + max(delays)):
        v = []
        for q, d in enumerate(delays):
            if t - d >= 0:
                v.append((t - d, q))
        out.append(v)
    return out


def build_pattern_sequence(z: torch.Tensor, special_token: int, delays=None):
    """[B, K, T] -> values [B, K, S], mask [K, S]."""
    B, K, T = z.shape
    layout = delayed_layout(K, T, delays)
    S = len(layout)
    values = torch.full((B, K, S), special_token, dtype=z.dtype)
    mask = torch.zeros(K, S, dtype=torch.bool)
    for s, coords in enumerate(layout):
        for (t, q) in coords:
            if t < T:
                values[:, q, s] = z[:, q, t]
                mask[q, s] = True
    return values, mask


def revert_pattern_sequence(s: torch.Tensor, special_token: int, timesteps: int, delays=None):
    """[B, K, S] -> values [B, K, T], mask [K, T]."""
    B, K, S = s.shape
    layout = delayed_layout(K, timesteps, delays)
    values = torch.full((B, K, timesteps), special_token, dtype=s.dtype)
    mask = torch.zeros(K, timesteps, dtype=torch.bool)
    for step, coords in enumerate(layout):
        if step < S:
            for (t, q) in coords:
                if t < timesteps:
                    values[:, q, t] = s[:, q, step]
                    mask[q, t] = True
    return values, mask


def first_step_with_timestep(n_q: int, timesteps: int, t: int, delays=None) -> tp.Optional[int]:
    for s, coords in enumerate(delayed_layout(n_q, timesteps, delays)):
        for (tt, q) in coords:
            if tt == t:
                return s
    return None
