"""Oracle (TEST INFRASTRUCTURE) -- codebook interleaving patterns, as naive python loops.

Follows audiocraft/modules/codebooks_patterns.py:339-356 (DelayedPatternProvider.get_pattern),
:154-179 (build_pattern_sequence), :225-248 (revert_pattern_sequence), :116-118
(get_first_step_with_timesteps); the loop form mirrors the reference's own independent checks in
tests/modules/test_codebooks_patterns.py:107-146.  The other providers of the reference's builder
(:359-552) are restated as layouts too (`provider_layout`); every function takes `layout=` in place of the delays.
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


def provider_layout(name: str, n_q: int, timesteps: int, **kw):
    """Layout of the reference's provider `name` (builders.py:244-250): list over sequence steps of (t, q) lists."""
    if name == 'parallel':       # :359-369
        name, kw = 'delay', dict(delays=[0] * n_q, empty_initial=kw.get('empty_initial', 0))
    if name == 'delay':          # :339-356
        delays = kw.get('delays') or list(range(n_q))
        flat, empty = kw.get('flatten_first', 0), kw.get('empty_initial', 0)
        out: tp.List[tp.List[tp.Tuple[int, int]]] = [[] for _ in range(1 + empty)]
        for t in range(min(timesteps, flat)):
            for q in range(n_q):
                out.append([(t, q)])
        for t in range(flat, timesteps + max(delays)):
            out.append([(t - d, q) for q, d in enumerate(delays) if t - d >= flat])
        return out
    if name == 'coarse_first':   # :515-530
        delays = kw.get('delays') or [0] * (n_q - 1)
        out = [[]] + [[(t, 0)] for t in range(timesteps)]
        for t in range(timesteps + max(delays)):
            out.append([(t - d, q + 1) for q, d in enumerate(delays) if t - d >= 0])
        return out
    if name == 'musiclm':        # :543-552
        g = kw.get('group_by', 2)
        return [[]] + [[(t, q)] for off in range(0, n_q, g) for t in range(timesteps) for q in range(off, off + g)]
    if name == 'unroll':         # :413-491: (position, coordinates) pairs, sorted
        flattening = kw.get('flattening') or list(range(n_q))
        delays = kw.get('delays') or [0] * n_q
        groups: tp.Dict[int, tp.Tuple[tp.List[int], int]] = {}
        for q, (inner, d) in enumerate(zip(flattening, delays)):
            groups.setdefault(inner, ([], d))[0].append(q)
        total = timesteps + max(delays)
        indexed: list = [(-1, [])]
        for t in range(total):
            for inner in range(max(groups) + 1):
                if inner in groups:
                    qs, d = groups[inner]
                    if t + d < total:
                        indexed.append((t + d, [(t, q) for q in qs]))
                else:
                    indexed.append((t, []))
        return [coords for _, coords in sorted(indexed)]
    raise ValueError(name)


def build_pattern_sequence(z: torch.Tensor, special_token: int, delays=None, layout=None):
    """[B, K, T] -> values [B, K, S], mask [K, S]."""
    B, K, T = z.shape
    layout = delayed_layout(K, T, delays) if layout is None else layout
    S = len(layout)
    values = torch.full((B, K, S), special_token, dtype=z.dtype)
    mask = torch.zeros(K, S, dtype=torch.bool)
    for s, coords in enumerate(layout):
        for (t, q) in coords:
            if t < T:
                values[:, q, s] = z[:, q, t]
                mask[q, s] = True
    return values, mask


def revert_pattern_sequence(s: torch.Tensor, special_token: int, timesteps: int, delays=None, layout=None):
    """[B, K, S] -> values [B, K, T], mask [K, T]."""
    B, K, S = s.shape
    layout = delayed_layout(K, timesteps, delays) if layout is None else layout
    values = torch.full((B, K, timesteps), special_token, dtype=s.dtype)
    mask = torch.zeros(K, timesteps, dtype=torch.bool)
    for step, coords in enumerate(layout):
        if step < S:
            for (t, q) in coords:
                if t < timesteps:
                    values[:, q, t] = s[:, q, step]
                    mask[q, t] = True
    return values, mask


def first_step_with_timestep(n_q: int, timesteps: int, t: int, delays=None, layout=None) -> tp.Optional[int]:
    for s, coords in enumerate(delayed_layout(n_q, timesteps, delays) if layout is None else layout):
        for (tt, q) in coords:
            if tt == t:
                return s
    return None
