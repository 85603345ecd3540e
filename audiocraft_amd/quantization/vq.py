"""Residual vector quantizer (inference) on MI355X.

API mirror of `audiocraft.quantization.vq.ResidualVectorQuantizer` / `base.BaseQuantizer`
(reference audiocraft/quantization/vq.py:16-115, base.py:27-60) for encode / decode; buffer names
mirror `core_vq.py` (`vq.layers.{q}._codebook.{inited,cluster_size,embed,embed_avg}`) so EnCodec
checkpoints load unchanged.  k-means init, EMA updates, dead-code expiry and the straight-through gradient are
training-only and out of scope; `forward` is the reference's in eval mode.
"""
import math
import typing as tp
from dataclasses import dataclass, field

import torch
from torch import nn

from .. import _C


@dataclass
class QuantizedResult:
    """quantization/base.py:18-24"""
    x: torch.Tensor
    codes: torch.Tensor
    bandwidth: torch.Tensor          # kb/s used, per batch item
    penalty: tp.Optional[torch.Tensor] = None
    metrics: dict = field(default_factory=dict)


class _Codebook(nn.Module):
    def __init__(self, dim: int, codebook_size: int, device=None):
        super().__init__()
        bound = math.sqrt(3.0 / dim)  # any non-degenerate init; real values come from the checkpoint
        embed = torch.empty(codebook_size, dim, device=device).uniform_(-bound, bound)
        self.register_buffer("inited", torch.Tensor([True]).to(embed.device))
        self.register_buffer("cluster_size", torch.zeros(codebook_size, device=device))
        self.register_buffer("embed", embed)
        self.register_buffer("embed_avg", embed.clone())


class _VQLayer(nn.Module):
    def __init__(self, dim, codebook_size, device=None):
        super().__init__()
        self._codebook = _Codebook(dim, codebook_size, device)


class _RVQ(nn.Module):
    def __init__(self, num_quantizers, dim, codebook_size, device=None):
        super().__init__()
        self.layers = nn.ModuleList([_VQLayer(dim, codebook_size, device) for _ in range(num_quantizers)])


class BaseQuantizer(nn.Module):
    def forward(self, x: torch.Tensor, frame_rate: int) -> QuantizedResult:
        raise NotImplementedError()

    def encode(self, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError()

    def decode(self, codes: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError()

    @property
    def total_codebooks(self):
        raise NotImplementedError()

    @property
    def num_codebooks(self):
        raise NotImplementedError()

    def set_num_codebooks(self, n: int):
        raise NotImplementedError()


class ResidualVectorQuantizer(BaseQuantizer):
    def __init__(self, dimension: int = 256, n_q: int = 8, q_dropout: bool = False, bins: int = 1024,
                 decay: float = 0.99, kmeans_init: bool = True, kmeans_iters: int = 10,
                 threshold_ema_dead_code: int = 2, orthogonal_reg_weight: float = 0.0,
                 orthogonal_reg_active_codes_only: bool = False, orthogonal_reg_max_codes: tp.Optional[int] = None,
                 device=None):
        super().__init__()
        self.max_n_q = n_q
        self.n_q = n_q
        self.q_dropout = q_dropout
        self.dimension = dimension
        self.bins = bins
        self.vq = _RVQ(n_q, dimension, bins, device)
        self._prep: tp.Optional[tp.Tuple[torch.Tensor, torch.Tensor]] = None
        self.check_codes = True   # decode() raises on out-of-range code values like the reference's F.embedding

    def _codebooks(self):
        """Stacked [K, bins, D] codebooks + their squared norms (`embed.pow(2).sum(0)`, core_vq.py:169)."""
        if self._prep is None:
            cb = torch.stack([layer._codebook.embed.detach().float() for layer in self.vq.layers]).contiguous()
            self._prep = (cb, _C.rvq_codebook_norms(cb))
        return self._prep

    @torch.no_grad()
    def forward(self, x: torch.Tensor, frame_rate: int) -> QuantizedResult:
        """vq.py:76-85 in eval mode: the quantized latents (sum of the selected codebook entries), the codes [B, K, T], the
        bandwidth n_q log2(bins) frame_rate / 1000 kb/s and a zero penalty (the commitment loss only exists while training,
        core_vq.py:347-364); no q_dropout, no straight-through estimator, no EMA updates."""
        assert not self.training, "audiocraft_amd quantizers run in eval mode only (no EMA / k-means / dropout)"
        codes = self.encode(x)
        bw = torch.tensor(self.n_q * math.log2(self.bins) * frame_rate / 1000).to(x)
        return QuantizedResult(self.decode(codes), codes, bw, penalty=torch.zeros((), device=x.device, dtype=x.dtype))

    def encode(self, x: torch.Tensor) -> torch.Tensor:
        """latents [B, D, T] -> codes [B, K, T] int64 (reference vq.py:87-96, core_vq.py:386-396)."""
        cb, norms = self._codebooks()
        return _C.rvq_encode(x.float().contiguous(), cb, norms, self.n_q)

    def decode(self, codes: torch.Tensor) -> torch.Tensor:
        """codes [B, K, T] -> quantized latents [B, D, T] (reference vq.py:98-103)."""
        cb, _ = self._codebooks()
        codes = codes.to(torch.int64).contiguous()
        if codes.shape[1] > cb.shape[0]:
            raise IndexError(f"decode: codes carry {codes.shape[1]} codebooks, the quantizer holds {cb.shape[0]}")
        # F.embedding raises on an out-of-range index (core_vq.py:177-179); the kernel would clamp silently
        # (e.g. the LM's special token `card` leaking into the codes), so check here: one tiny reduction per decode
        if self.check_codes and codes.numel() and bool(((codes < 0) | (codes >= cb.shape[1])).any()):
            raise IndexError(f"decode: code values outside [0, {cb.shape[1]})")
        return _C.rvq_decode(codes, cb)

    @property
    def total_codebooks(self):
        return self.max_n_q

    @property
    def num_codebooks(self):
        return self.n_q

    def set_num_codebooks(self, n: int):
        assert n > 0 and n <= self.max_n_q
        self.n_q = n
