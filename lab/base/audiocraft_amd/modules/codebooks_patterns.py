"""Delay-pattern codebook interleaving (host side).

API mirror of `audiocraft.modules.codebooks_patterns` for the one pattern MusicGen uses
(`DelayedPatternProvider`, reference audiocraft/modules/codebooks_patterns.py:305-356) and of the
`Pattern` gather helpers (:116-118, :154-179, :225-269).  Unlike the reference, which materialises a
python list of coordinates per sequence step and loops over it, the delay pattern is closed form:

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
        return max(self.delays)

    @property
    def valid_layout(self):
        lay = self.layout
        return lay[:len(lay) - self.max_delay]

    def starts_with_special_token(self) -> bool:
        return True

    def get_first_step_with_timesteps(self, t: int, q: tp.Optional[int] = None) -> tp.Optional[int]:
        assert t <= self.timesteps, "provided timesteps is greater than the pattern's number of timesteps"
        if t >= self.timesteps:
            return None
        return t + 1 + (self.delays[q] if q is not None else min(self.delays))

    # -- index maps --------------------------------------------------------------------------------
    def _seq_len(self, keep_only_valid_steps: bool) -> int:
        return self.timesteps + 1 + (0 if keep_only_valid_steps else self.max_delay)

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


class CodebooksPatternProvider:
    def __init__(self, n_q: int, cached: bool = True):
        assert n_q > 0
        self.n_q = n_q
        self.get_pattern = lru_cache(100)(self.get_pattern)  # type: ignore

    def get_pattern(self, timesteps: int) -> Pattern:
        raise NotImplementedError()


class DelayedPatternProvider(CodebooksPatternProvider):
    """Codebook q is delayed by `delays[q]` steps (default q).  `flatten_first` / `empty_initial`
    variants of the reference are not used by any MusicGen config and are not implemented."""

    def __init__(self, n_q: int, delays: tp.Optional[tp.List[int]] = None, flatten_first: int = 0,
                 empty_initial: int = 0):
        super().__init__(n_q)
        if flatten_first or empty_initial:
            raise NotImplementedError("flatten_first / empty_initial are outside the MusicGen path")
        self.delays = list(range(n_q)) if delays is None else list(delays)
        assert len(self.delays) == self.n_q
        assert sorted(self.delays) == self.delays

    def get_pattern(self, timesteps: int) -> Pattern:
        return Pattern(self.n_q, timesteps, self.delays)
