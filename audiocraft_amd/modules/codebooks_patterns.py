"""Codebook interleaving patterns (host side).

API mirror of `audiocraft.modules.codebooks_patterns`: the pattern MusicGen uses (`DelayedPatternProvider`, reference
audiocraft/modules/codebooks_patterns.py:305-356) with the `Pattern` gather helpers (:116-118, :154-179, :225-269), and
the other providers of the reference's builder (`parallel`, `unroll`, `coarse_first`, `musiclm`; `LayoutPattern`).  Unlike
the reference, which materialises a python list of coordinates per sequence step and loops over it, the delay pattern is
closed form:

    sequence step s >= 1 holds timestep  t = s - 1 - delays[q]  of codebook q   (if 0 <= t < T)
    S = T + max(delays) + 1,  step 0 is the special-token step

so every index map below is a broadcasted arithmetic expression.  In the generation loop the same
rule is evaluated on the device by the sampling kernel through the `[K, S]` validity mask.
"""
import typing as tp
from collections import namedtuple
from functools import lru_cache

import torch

LayoutCoord = namedtuple('LayoutCoord', ['t', 'q'])


class Pattern:
    """Delay pattern over `timesteps` steps and `n_q` codebooks."""

    def __init__(self, n_q: int, timesteps: int, delays: tp.Sequence[int]):
        assert len(delays) == n_q and list(delays) == sorted(delays) and min(delays) >= 0
        self.n_q = n_q
        self.timesteps = timesteps
        self.delays = list(delays)

    # -- layout views kept for API compatibility ---------------------------------------------------
    @property
    def layout(self) -> tp.List[tp.List[LayoutCoord]]:
        out: tp.List[tp.List[LayoutCoord]] = [[]]
        for s in range(1, self.timesteps + max(self.delays) + 1):
            out.append([LayoutCoord(s - 1 - d, q) for q, d in enumerate(self.delays) if s - 1 - d >= 0])
        return out

    @property
    def num_sequence_steps(self) -> int:
        return self.timesteps + max(self.delays)

    @property
    def max_delay(self) -> int:
        # the reference's definition (codebooks_patterns.py:84-90): how far the layout's timesteps run past `timesteps`.  The
        # layout has timesteps + max(delays) steps after the special one, its last holds timestep T + max - 1 - min(delays)
        return max(self.delays) - min(self.delays)

    @property
    def valid_layout(self):
        lay = self.layout
        return lay[:len(lay) - self.max_delay]

    def starts_with_special_token(self) -> bool:
        return True

    def get_first_step_with_timesteps(self, t: int, q: tp.Optional[int] = None) -> tp.Optional[int]:
        assert t <= self.timesteps, "provided timesteps is greater than the pattern's number of timesteps"
        # (t == timesteps: the layout runs max_delay steps past the last timestep and holds the coordinates (timesteps, q) of
        # the codebooks delayed by less than max_delay, codebooks_patterns.py:347-353)
        last = self.num_sequence_steps
        steps = [t + 1 + d for d in (self.delays if q is None else [self.delays[q]]) if t + 1 + d <= last]
        return min(steps) if steps else None

    # -- index maps --------------------------------------------------------------------------------
    def _seq_len(self, keep_only_valid_steps: bool) -> int:
        return self.num_sequence_steps + 1 - (self.max_delay if keep_only_valid_steps else 0)

    def sequence_map(self, timesteps: int, keep_only_valid_steps: bool = False, device='cpu'):
        """-> (t_index [K, S] int64 clamped, mask [K, S] bool) with t_index[q, s] = s - 1 - delays[q]."""
        assert timesteps <= self.timesteps
        S = self._seq_len(keep_only_valid_steps)
        s = torch.arange(S, device=device).view(1, -1)
        d = torch.tensor(self.delays, device=device).view(-1, 1)
        t = s - 1 - d
        mask = (t >= 0) & (t < timesteps)
        return t.clamp(0, max(timesteps - 1, 0)), mask

    def build_pattern_sequence(self, z: torch.Tensor, special_token: int, keep_only_valid_steps: bool = False):
        """[B, K, T] -> (values [B, K, S], indexes [K, S], mask [K, S]); see reference :154-179.
        `indexes` uses the reference's flattened convention (q * T + t, or K * T for the special token)."""
        B, K, T = z.shape
        assert K == self.n_q
        t, mask = self.sequence_map(T, keep_only_valid_steps, z.device)
        if T == 0:
            values = torch.full((B, K, t.shape[1]), special_token, dtype=z.dtype, device=z.device)
        else:
            values = torch.where(mask[None], z.gather(2, t[None].expand(B, -1, -1)),
                                 torch.full((), special_token, dtype=z.dtype, device=z.device))
        q = torch.arange(K, device=z.device).view(-1, 1)
        indexes = torch.where(mask, t + q * T, torch.full_like(t, K * T))
        return values, indexes, mask

    def _revert_map(self, sequence_steps: int, keep_only_valid_steps: bool, is_model_output: bool, device):
        assert sequence_steps <= self._seq_len(keep_only_valid_steps), \
            "sequence to revert is longer than the defined pattern"
        T = self.timesteps
        tt = torch.arange(T, device=device).view(1, -1)
        d = torch.tensor(self.delays, device=device).view(-1, 1)
        s = tt + d + (0 if is_model_output else 1)
        limit = min(sequence_steps, self._seq_len(keep_only_valid_steps) - (1 if is_model_output else 0))
        mask = s < limit
        return s.clamp(max=max(sequence_steps - 1, 0)), mask

    def revert_pattern_sequence(self, s: torch.Tensor, special_token: int, keep_only_valid_steps: bool = False):
        """[B, K, S] -> (values [B, K, T], indexes [K, T], mask [K, T]); reference :225-248."""
        B, K, S = s.shape
        assert K == self.n_q
        idx, mask = self._revert_map(S, keep_only_valid_steps, False, s.device)
        values = torch.where(mask[None], s.gather(2, idx[None].expand(B, -1, -1)),
                             torch.full((), special_token, dtype=s.dtype, device=s.device))
        q = torch.arange(K, device=s.device).view(-1, 1)
        indexes = torch.where(mask, idx + q * S, torch.full_like(idx, K * S))
        return values, indexes, mask

    def revert_pattern_logits(self, logits: torch.Tensor, special_token: float, keep_only_valid_steps: bool = False):
        """[B, card, K, S] -> (values [B, card, K, T], indexes, mask); reference :250-269."""
        B, card, K, S = logits.shape
        idx, mask = self._revert_map(S, keep_only_valid_steps, True, logits.device)
        g = logits.gather(3, idx[None, None].expand(B, card, -1, -1))
        values = torch.where(mask[None, None], g, torch.full((), special_token, dtype=logits.dtype,
                                                             device=logits.device))
        q = torch.arange(K, device=logits.device).view(-1, 1)
        indexes = torch.where(mask, idx + q * S, torch.full_like(idx, K * S))
        return values, indexes, mask


class LayoutPattern:
    """A codebook interleaving pattern given by its layout (reference `Pattern`, codebooks_patterns.py:20-269), for the
    providers whose layout is not the plain delay rule above.  The layout is held as three parallel int64 arrays -- the
    sequence step, timestep and codebook of every placed coordinate -- and `n_steps` (layout entries, the special-token step
    included); every index map is one numpy scatter over those arrays instead of the reference's loops over a list of lists.
    Same methods, arguments and return values as `Pattern`."""

    def __init__(self, n_q: int, timesteps: int, n_steps: int, step, t, q, starts_empty: bool = True):
        import numpy as np
        self.n_q, self.timesteps, self.n_steps = n_q, timesteps, int(n_steps)
        self._step, self._t, self._q = (np.asarray(a, dtype=np.int64) for a in (step, t, q))
        self._starts_empty = starts_empty
        self._validate()

    def _validate(self):
        """codebooks_patterns.py:52-77: the sequence starts with the special-token step, a codebook appears at most once per
        step, a codebook's timesteps never go back as the sequence advances."""
        import numpy as np
        assert self._starts_empty and (self._step.size == 0 or self._step.min() >= 1), \
            "pattern layout must start with an empty step (the special token)"
        assert self.n_steps - 1 <= self.n_q * max(self.timesteps, 1) + self.n_steps, "layout longer than its coordinates"
        key = self._step * self.n_q + self._q
        assert np.unique(key).size == key.size, "Multiple entries for a same codebook are found at one step"
        for qq in range(self.n_q):
            sel = self._q == qq
            order = np.argsort(self._step[sel], kind='stable')
            tt = self._t[sel][order]
            assert (np.diff(tt) >= 0).all(), f"Past timesteps are found in the sequence for codebook = {qq}"

    # -- layout views ------------------------------------------------------------------------------
    @property
    def layout(self) -> tp.List[tp.List[LayoutCoord]]:
        import numpy as np
        out: tp.List[tp.List[LayoutCoord]] = [[] for _ in range(self.n_steps)]
        for i in np.argsort(self._step, kind='stable'):      # within a step: the order the provider emitted
            out[int(self._step[i])].append(LayoutCoord(int(self._t[i]), int(self._q[i])))
        return out

    @property
    def num_sequence_steps(self) -> int:
        return self.n_steps - 1

    @property
    def max_delay(self) -> int:
        return (int(self._t.max()) + 1 if self._t.size else 0) - self.timesteps

    @property
    def valid_layout(self):
        lay = self.layout
        return lay[:len(lay) - self.max_delay]

    def starts_with_special_token(self) -> bool:
        return self._starts_empty

    def get_sequence_coords_with_timestep(self, t: int, q: tp.Optional[int] = None):
        assert t <= self.timesteps, "provided timesteps is greater than the pattern's number of timesteps"
        if q is not None:
            assert q <= self.n_q, "provided number of codebooks is greater than the pattern's number of codebooks"
        import numpy as np
        sel = (self._t == t) if q is None else ((self._t == t) & (self._q == q))
        idx = np.nonzero(sel)[0]
        idx = idx[np.argsort(self._step[idx], kind='stable')]
        return [(int(self._step[i]), LayoutCoord(int(self._t[i]), int(self._q[i]))) for i in idx]

    def get_steps_with_timestep(self, t: int, q: tp.Optional[int] = None) -> tp.List[int]:
        return [step for step, _ in self.get_sequence_coords_with_timestep(t, q)]

    def get_first_step_with_timesteps(self, t: int, q: tp.Optional[int] = None) -> tp.Optional[int]:
        steps = self.get_steps_with_timestep(t, q)
        return steps[0] if steps else None

    # -- index maps --------------------------------------------------------------------------------
    def _ref_len(self, keep_only_valid_steps: bool) -> int:
        return self.n_steps - (self.max_delay if keep_only_valid_steps else 0)

    def _sequence_indexes(self, timesteps: int, n_q: int, keep_only_valid_steps: bool, device):
        """-> (indexes [K, S] into the flattened [K * T | special] values, mask [K, S]); codebooks_patterns.py:120-152"""
        import numpy as np
        assert n_q == self.n_q, f"invalid number of codebooks for the sequence and the pattern: {n_q} != {self.n_q}"
        assert timesteps <= self.timesteps, "invalid number of timesteps used to build the sequence from the pattern"
        S = self._ref_len(keep_only_valid_steps)
        indexes = np.full((n_q, S), n_q * timesteps, dtype=np.int64)
        mask = np.zeros((n_q, S), dtype=bool)
        sel = (self._step < S) & (self._t < timesteps)
        indexes[self._q[sel], self._step[sel]] = self._t[sel] + self._q[sel] * timesteps
        mask[self._q[sel], self._step[sel]] = True
        return torch.from_numpy(indexes).to(device), torch.from_numpy(mask).to(device)

    def _reverted_indexes(self, sequence_steps: int, n_q: int, keep_only_valid_steps: bool, is_model_output: bool, device):
        """-> (indexes [K, T] into the flattened [K * S | special] sequence, mask [K, T]); codebooks_patterns.py:181-223"""
        import numpy as np
        assert n_q == self.n_q, f"invalid number of codebooks for the sequence and the pattern: {n_q} != {self.n_q}"
        ref_len = self._ref_len(keep_only_valid_steps)
        assert sequence_steps <= ref_len, f"sequence to revert is longer than the defined pattern: {sequence_steps} > {ref_len}"
        shift = 1 if (is_model_output and self.starts_with_special_token()) else 0   # logits of step s predict step s + 1
        T = self.timesteps
        indexes = np.full((n_q, T), n_q * sequence_steps, dtype=np.int64)
        mask = np.zeros((n_q, T), dtype=bool)
        s = self._step - shift
        sel = (self._step < ref_len) & (s >= 0) & (s < sequence_steps) & (self._t < T)
        order = np.argsort(s[sel], kind='stable')      # a later step overwrites an earlier one, like the reference's loop
        qq, tt, ss = self._q[sel][order], self._t[sel][order], s[sel][order]
        indexes[qq, tt] = ss + qq * sequence_steps
        mask[qq, tt] = True
        return torch.from_numpy(indexes).to(device), torch.from_numpy(mask).to(device)

    @staticmethod
    def _take(flat: torch.Tensor, indexes: torch.Tensor, special_token) -> torch.Tensor:
        """flat [..., N] -> [..., K, S]: entry N of `indexes` reads the special token."""
        N = flat.shape[-1]
        safe = indexes.clamp(max=max(N - 1, 0)).reshape(-1)
        if N == 0:
            vals = torch.zeros(*flat.shape[:-1], safe.numel(), dtype=flat.dtype, device=flat.device)
        else:
            vals = flat.index_select(-1, safe)
        vals = torch.where(indexes.reshape(-1) >= N, torch.full((), special_token, dtype=flat.dtype, device=flat.device), vals)
        return vals.reshape(*flat.shape[:-1], *indexes.shape)

    def build_pattern_sequence(self, z: torch.Tensor, special_token: int, keep_only_valid_steps: bool = False):
        B, K, T = z.shape
        indexes, mask = self._sequence_indexes(T, K, keep_only_valid_steps, z.device)
        return self._take(z.reshape(B, K * T), indexes, special_token), indexes, mask

    def revert_pattern_sequence(self, s: torch.Tensor, special_token: int, keep_only_valid_steps: bool = False):
        B, K, S = s.shape
        indexes, mask = self._reverted_indexes(S, K, keep_only_valid_steps, False, s.device)
        return self._take(s.reshape(B, K * S), indexes, special_token), indexes, mask

    def revert_pattern_logits(self, logits: torch.Tensor, special_token: float, keep_only_valid_steps: bool = False):
        B, card, K, S = logits.shape
        indexes, mask = self._reverted_indexes(S, K, keep_only_valid_steps, True, logits.device)
        return self._take(logits.reshape(B, card, K * S), indexes, special_token), indexes, mask


class CodebooksPatternProvider:
    def __init__(self, n_q: int, cached: bool = True):
        assert n_q > 0
        self.n_q = n_q
        self.get_pattern = lru_cache(100)(self.get_pattern)  # type: ignore

    def get_pattern(self, timesteps: int):
        raise NotImplementedError()


def _delayed_coords(n_q: int, delays: tp.Sequence[int], timesteps: int, first_t: int, first_step: int, q_offset: int = 0):
    """Coordinates of the delay rule from timestep `first_t` on: step first_step + (t + delays[q] - first_t) holds (t, q) for
    first_t <= t < timesteps + max(delays) - delays[q] (the reference's layouts run max(delays) steps past the last timestep,
    codebooks_patterns.py:347-353).  -> (step, t, q arrays, steps used)"""
    import numpy as np
    md = max(delays) if len(delays) else 0
    steps, ts, qs = [], [], []
    for qi, d in enumerate(delays):
        t = np.arange(first_t, max(first_t, timesteps + md - d), dtype=np.int64)
        steps.append(first_step + t + d - first_t)
        ts.append(t)
        qs.append(np.full_like(t, qi + q_offset))
    cat = lambda parts: np.concatenate(parts) if parts else np.zeros(0, dtype=np.int64)   # noqa: E731
    return cat(steps), cat(ts), cat(qs), max(0, timesteps + md - first_t)


class DelayedPatternProvider(CodebooksPatternProvider):
    """Codebook q is delayed by `delays[q]` steps (default q): the MusicGen pattern (closed form: `Pattern`).  With
    `flatten_first` (the first N timesteps one codebook per step) or `empty_initial` (N more empty steps in front) the layout
    form `LayoutPattern` is used (codebooks_patterns.py:305-356)."""

    def __init__(self, n_q: int, delays: tp.Optional[tp.List[int]] = None, flatten_first: int = 0,
                 empty_initial: int = 0):
        super().__init__(n_q)
        self.delays = list(range(n_q)) if delays is None else list(delays)
        self.flatten_first, self.empty_initial = flatten_first, empty_initial
        assert len(self.delays) == self.n_q
        assert sorted(self.delays) == self.delays
        if empty_initial < 0:
            raise NotImplementedError("empty_initial < 0 (a sequence without the special-token step) is not supported by "
                                      "LMModel.generate in the reference either")

    def get_pattern(self, timesteps: int):
        if not self.flatten_first and not self.empty_initial:
            return Pattern(self.n_q, timesteps, self.delays)
        import numpy as np
        first = 1 + self.empty_initial
        nf = min(timesteps, self.flatten_first)
        ft = np.repeat(np.arange(nf, dtype=np.int64), self.n_q)
        fq = np.tile(np.arange(self.n_q, dtype=np.int64), nf)
        fstep = first + np.arange(nf * self.n_q, dtype=np.int64)
        dstep, dt, dq, used = _delayed_coords(self.n_q, self.delays, timesteps, self.flatten_first, first + nf * self.n_q)
        return LayoutPattern(self.n_q, timesteps, first + nf * self.n_q + used, np.concatenate([fstep, dstep]),
                             np.concatenate([ft, dt]), np.concatenate([fq, dq]))


class ParallelPatternProvider(DelayedPatternProvider):
    """All codebooks of a timestep in one step: the delay rule with no delay (codebooks_patterns.py:359-369)."""

    def __init__(self, n_q: int, empty_initial: int = 0):
        super().__init__(n_q, [0] * n_q, empty_initial=empty_initial)


class UnrolledPatternProvider(CodebooksPatternProvider):
    """Codebooks flattened over `flattening[q]` inner steps per timestep, each inner step optionally delayed
    (codebooks_patterns.py:372-491): inner step i of timestep t is emitted at position t + delay_i of the sequence of
    (position, coordinates) pairs, which is then sorted -- ties by the coordinates themselves, empty steps first."""

    def __init__(self, n_q: int, flattening: tp.Optional[tp.List[int]] = None, delays: tp.Optional[tp.List[int]] = None):
        super().__init__(n_q)
        flattening = list(range(n_q)) if flattening is None else list(flattening)
        delays = [0] * n_q if delays is None else list(delays)
        assert len(flattening) == n_q and len(delays) == n_q
        assert sorted(flattening) == flattening and sorted(delays) == delays
        self._groups: tp.Dict[int, tp.Tuple[tp.List[int], int]] = {}
        for qi, (inner, d) in enumerate(zip(flattening, delays)):
            if inner in self._groups:
                assert self._groups[inner][1] == d, ("Delay and flattening between codebooks is inconsistent: ",
                                                     "two codebooks flattened to the same position should have the same delay.")
                self._groups[inner][0].append(qi)
            else:
                self._groups[inner] = ([qi], d)
        self.max_delay = max(delays)

    @property
    def _num_inner_steps(self) -> int:
        return max(self._groups) + 1

    def num_virtual_steps(self, timesteps: int) -> int:
        return timesteps * self._num_inner_steps + 1

    def get_pattern(self, timesteps: int):
        import numpy as np
        T = timesteps + self.max_delay
        # one record per emitted layout entry: (position, empty-first flag, timestep, first codebook) is the reference's sort
        # key (a tuple compare of (position, [LayoutCoord(t, q), ...]); an empty list sorts before any other)
        pos, nonempty, tt, inner = [], [], [], []
        t = np.arange(T, dtype=np.int64)
        for i in range(self._num_inner_steps):
            if i in self._groups:
                d = self._groups[i][1]
                keep = t + d < T
                pos.append(t[keep] + d); nonempty.append(np.ones(keep.sum(), dtype=np.int64)); tt.append(t[keep])
                inner.append(np.full(keep.sum(), i, dtype=np.int64))
            else:
                pos.append(t); nonempty.append(np.zeros(T, dtype=np.int64)); tt.append(t); inner.append(np.full(T, i, dtype=np.int64))
        pos, nonempty, tt, inner = (np.concatenate(a) for a in (pos, nonempty, tt, inner))
        first_q = np.array([self._groups[i][0][0] if i in self._groups else 0 for i in inner], dtype=np.int64)
        order = np.lexsort((first_q, tt, nonempty, pos))
        step_of = np.empty_like(order)
        step_of[order] = 1 + np.arange(order.size)            # entry 0 is the special-token step
        steps, ts, qs = [], [], []
        for e in range(order.size):
            if nonempty[e]:
                for qi in self._groups[int(inner[e])][0]:
                    steps.append(step_of[e]); ts.append(tt[e]); qs.append(qi)
        return LayoutPattern(self.n_q, timesteps, 1 + order.size, steps, ts, qs)


class CoarseFirstPattern(CodebooksPatternProvider):
    """First the whole first codebook, then the others in parallel (VALL-E style), optionally delayed
    (codebooks_patterns.py:494-530)."""

    def __init__(self, n_q: int, delays: tp.Optional[tp.List[int]] = None):
        super().__init__(n_q)
        self.delays = [0] * (n_q - 1) if delays is None else list(delays)
        assert len(self.delays) == self.n_q - 1
        assert sorted(self.delays) == self.delays

    def get_pattern(self, timesteps: int):
        import numpy as np
        t0 = np.arange(timesteps, dtype=np.int64)
        dstep, dt, dq, used = _delayed_coords(self.n_q - 1, self.delays, timesteps, 0, 1 + timesteps, q_offset=1)
        return LayoutPattern(self.n_q, timesteps, 1 + timesteps + used, np.concatenate([1 + t0, dstep]),
                             np.concatenate([t0, dt]), np.concatenate([np.zeros_like(t0), dq]))


class MusicLMPattern(CodebooksPatternProvider):
    """Groups of `group_by` codebooks, each group flattened over all timesteps before the next one starts
    (codebooks_patterns.py:533-552)."""

    def __init__(self, n_q: int, group_by: int = 2):
        super().__init__(n_q)
        assert n_q % group_by == 0, "MusicLMPattern: group_by must divide n_q"
        self.group_by = group_by

    def get_pattern(self, timesteps: int):
        import numpy as np
        g = self.group_by
        offset, t, j = np.meshgrid(np.arange(0, self.n_q, g), np.arange(timesteps), np.arange(g), indexing='ij')
        n = offset.size
        return LayoutPattern(self.n_q, timesteps, 1 + n, 1 + np.arange(n, dtype=np.int64), t.reshape(-1), (offset + j).reshape(-1))
