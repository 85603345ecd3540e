"""Oracle (TEST INFRASTRUCTURE) -- MusicGen LMModel: streaming transformer forward + generate loop.

Functional fp32 restatement over a reference-format state dict (keys as dumped from
`audiocraft.models.lm.LMModel.state_dict()`).  The KV cache grows by `torch.cat` exactly like the
reference (audiocraft/modules/transformer.py:266-298) so the timed CPU baseline pays the same
O(T^2) copies the reference pays.
"""
import math
import typing as tp
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F

from . import patterns


@dataclass
class LMConfig:
    """LMModel / StreamingTransformer constructor arguments used by MusicGen
    (config/model/lm/musicgen_lm.yaml, config/model/lm/default.yaml)."""
    dim: int = 1024
    num_heads: int = 16
    num_layers: int = 24
    hidden_scale: int = 4
    n_q: int = 4
    card: int = 2048
    cross_attention: bool = True
    delays: tp.List[int] = field(default_factory=lambda: [0, 1, 2, 3])
    max_period: float = 10000.
    positional_scale: float = 1.0
    cfg_coef: float = 3.0
    eps: float = 1e-5
    # options of StreamingTransformer no MusicGen release uses (config/model/lm/default.yaml:25-33), part of the
    # decode step all the same: rotary positions, xPos decay, a bounded receptive field
    positional_embedding: str = 'sin'      # 'sin' | 'rope' | 'sin_rope' (transformer.py:632-637, 701-704)
    xpos: bool = False
    past_context: tp.Optional[int] = None
    # attention options of config/model/lm/default.yaml:43-46 (off in every release)
    kv_repeat: int = 1                      # H / kv_repeat key / value heads (transformer.py:196-200, 373-386, 398-400)
    qk_layer_norm: bool = False             # LayerNorm on the projected queries / keys (transformer.py:216-222, 388-392)
    qk_layer_norm_cross: bool = False       # same in the cross-attention (transformer.py:358-360, 526-529)
    # codebook pattern other than the delay rule: (provider name, its kwargs) of builders.get_codebooks_pattern_provider
    pattern: tp.Optional[tp.Tuple[str, dict]] = None
    # post-norm layers (transformer.py:567-573; the constructor default of LMModel / StreamingTransformer, lm.py:147; every
    # release sets norm_first: true): x = norm1(x + sa(x)); x = norm_cross(x + ca(layer INPUT)); x = norm2(x + ff(x)); no out_norm
    norm_first: bool = True


def create_sin_embedding(positions: torch.Tensor, dim: int, max_period: float = 10000.) -> torch.Tensor:
    """audiocraft/modules/transformer.py:70-89: cos first, then sin; divisor half_dim - 1."""
    half = dim // 2
    positions = positions.to(torch.float32)
    adim = torch.arange(half, dtype=torch.float32).view(1, 1, -1)
    phase = positions / (torch.full([], max_period) ** (adim / (half - 1)))
    return torch.cat([torch.cos(phase), torch.sin(phase)], dim=-1)


def rope_rotate(x: torch.Tensor, start: int, max_period: float, scale: float, xpos: bool, invert_decay: bool) -> torch.Tensor:
    """RotaryEmbedding.rotate (audiocraft/modules/rope.py:75-96) on x [B, H, T, hd] (time_dim = 2): pairs (2i, 2i+1) of the
    head dimension are complex numbers, multiplied by  (e^{i pos f_i} * decay_i(pos)) * scale + (1 - scale)  with
    f_i = max_period^(-2i / hd) (:59-61) and the xPos decay ((i / (hd/2) + 0.4) / 1.4)^(pos / 512) (rope.py:26-44), inverted
    for keys (:111-114)."""
    hd, T = x.shape[-1], x.shape[2]
    adim = torch.arange(0, hd, 2, dtype=torch.float32)[: hd // 2]
    freqs = 1.0 / (max_period ** (adim / hd))
    idx = torch.arange(start + T, dtype=torch.float32)
    angles = torch.outer(idx, freqs)
    rotation = torch.polar(torch.ones_like(angles), angles)[start:start + T].view(1, 1, T, -1)
    decay: tp.Union[float, torch.Tensor] = 1.0
    if xpos:
        half = hd // 2
        rates = (torch.arange(half, dtype=torch.float32) / half + 0.4) / 1.4
        sc = rates ** (idx / 512).unsqueeze(-1)
        decay = torch.polar(sc, torch.zeros_like(sc))[start:start + T].view(1, 1, T, -1)
        if invert_decay:
            decay = decay ** -1
    xc = torch.view_as_complex(x.to(torch.float32).reshape(*x.shape[:-1], -1, 2))
    out = torch.view_as_real(xc * ((rotation * decay) * scale + (1.0 - scale))).view_as(x)
    return out.type_as(x)


def _ln(x, sd, prefix, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + '.weight'], sd[prefix + '.bias'], eps)


def _attention(q, k, v, causal: bool, past_context: tp.Optional[int] = None):
    """F.scaled_dot_product_attention semantics (transformer.py:412-414): scale 1/sqrt(hd),
    lower-triangular mask aligned top-left when `causal` (only used when #q == #k).  past_context: the custom
    attention's mask (_get_mask, transformer.py:249-264): a query at position p sees keys p - past_context .. p."""
    scale = 1.0 / math.sqrt(q.shape[-1])
    w = (q @ k.transpose(-1, -2)) * scale
    if causal:
        Tq, Tk = w.shape[-2:]
        m = torch.ones(Tq, Tk, dtype=torch.bool).tril()
        if past_context is not None:
            m &= torch.ones(Tq, Tk, dtype=torch.bool).triu(-past_context)
        w = w.masked_fill(~m, float('-inf'))
    return torch.softmax(w, dim=-1) @ v


class LMState:
    """Streaming state (audiocraft/modules/streaming.py:20-119): per-layer past K/V + offsets."""
    def __init__(self, num_layers: int):
        self.past_k: tp.List[tp.Optional[torch.Tensor]] = [None] * num_layers
        self.past_v: tp.List[tp.Optional[torch.Tensor]] = [None] * num_layers
        self.ctx_offset: tp.List[tp.Optional[int]] = [None] * num_layers   # attention 'offset': keys dropped by past_context
        self.offset = 0          # transformer.offsets (same for every row)
        self.first_step = True   # fuser: 'offsets' not yet in state (conditioners.py:1722-1727)


def transformer_forward(sd: dict, cfg: LMConfig, x: torch.Tensor, cross_src: tp.Optional[torch.Tensor],
                        state: tp.Optional[LMState]) -> torch.Tensor:
    """StreamingTransformer.forward (transformer.py:693-713) + StreamingTransformerLayer.forward
    (:550-574, norm_first) + StreamingMultiheadAttention.forward (:315-451, custom / memory-efficient
    torch-SDPA branch, layout "b h t d").  x: [B, T, C]."""
    B, T, C = x.shape
    H = cfg.num_heads
    hd = C // H
    offset = state.offset if state is not None else 0
    pos = torch.arange(T).view(1, -1, 1) + offset
    if cfg.positional_embedding in ('sin', 'sin_rope'):
        x = x + cfg.positional_scale * create_sin_embedding(pos, C, cfg.max_period)
    use_rope = cfg.positional_embedding in ('rope', 'sin_rope')

    def ls(name):   # LayerScale (transformer.py:92-110, 526-538): a per-channel gain on the residual branch, or identity
        return sd.get(f'{name}.scale', 1.0)
    for li in range(cfg.num_layers):
        p = f'transformer.layers.{li}'
        # --- self attention (pre-norm: on norm1(x); post-norm: on x itself, norm1 follows the residual add)
        src = x
        h = _ln(x, sd, p + '.norm1', cfg.eps) if cfg.norm_first else x
        proj = F.linear(h, sd[p + '.self_attn.in_proj_weight'], sd.get(p + '.self_attn.in_proj_bias'))
        if cfg.kv_repeat == 1:
            packed = proj.view(B, T, 3, H, hd).permute(2, 0, 3, 1, 4)  # "b t (p h d) -> p b h t d"
            q, k, v = packed[0], packed[1], packed[2]
        else:
            # transformer.py:373-386: the projection emits C query features, then kv_heads * hd key and as many value features
            kvh = H // cfg.kv_repeat
            q = proj[..., :C].view(B, T, H, hd).transpose(1, 2)
            k = proj[..., C:C + kvh * hd].view(B, T, kvh, hd).transpose(1, 2)
            v = proj[..., C + kvh * hd:].view(B, T, kvh, hd).transpose(1, 2)
        if cfg.qk_layer_norm:
            # transformer.py:388-392: over the full model dimension ("b t (h d)"), before the rotary positions and the cache
            assert cfg.kv_repeat == 1
            q = _ln(q.transpose(1, 2).reshape(B, T, C), sd, p + '.self_attn.q_layer_norm', cfg.eps).view(B, T, H, hd).transpose(1, 2)
            k = _ln(k.transpose(1, 2).reshape(B, T, C), sd, p + '.self_attn.k_layer_norm', cfg.eps).view(B, T, H, hd).transpose(1, 2)
        # _get_mask (transformer.py:233-247): no mask for one step; lower-triangular for T>1, which
        # the reference only supports when there is no past (raises otherwise).
        causal = T > 1
        if use_rope:
            # _apply_rope (transformer.py:300-313, 394-395): the new q / k are rotated BEFORE they join the cache, at
            # start = (keys dropped so far) + (keys cached).  That is the absolute position -- except after a FIRST
            # streaming call longer than past_context: _complete_kv (:286-297) drops the surplus keys but initialises
            # the dropped-keys counter to 0, so every later position lags by (first length - past_context).  Restated
            # as is (tests/golden/lm_rope.npz: a 9-step prompt against past_context 6).
            start = 0
            if state is not None and state.past_k[li] is not None:
                start = (state.ctx_offset[li] or 0) + state.past_k[li].shape[2]
            q = rope_rotate(q, start, cfg.max_period, cfg.positional_scale, cfg.xpos, False)
            k = rope_rotate(k, start, cfg.max_period, cfg.positional_scale, cfg.xpos, True)
        if state is not None:
            if state.past_k[li] is not None:
                assert T == 1, "reference raises 'Not supported at the moment' here"
                k = torch.cat([state.past_k[li], k], dim=2)   # _complete_kv, transformer.py:274-281
                v = torch.cat([state.past_v[li], v], dim=2)
            keep = 0 if cfg.past_context is None else max(0, k.shape[2] - cfg.past_context)   # :286-293
            state.past_k[li], state.past_v[li] = k[:, :, keep:], v[:, :, keep:]
            state.ctx_offset[li] = 0 if state.ctx_offset[li] is None else state.ctx_offset[li] + keep
        if cfg.kv_repeat > 1:   # expand_repeated_kv (transformer.py:90-107, 398-400): AFTER the cache, which keeps kv_heads heads
            k, v = k.repeat_interleave(cfg.kv_repeat, dim=1), v.repeat_interleave(cfg.kv_repeat, dim=1)
        a = _attention(q, k, v, causal, cfg.past_context)
        a = a.permute(0, 2, 1, 3).reshape(B, T, C)
        x = x + ls(p + '.layer_scale_1') * F.linear(a, sd[p + '.self_attn.out_proj.weight'], sd.get(p + '.self_attn.out_proj.bias'))
        if not cfg.norm_first:
            x = _ln(x, sd, p + '.norm1', cfg.eps)
        # --- cross attention: q/k/v projections with the three slices of in_proj_weight, k/v
        # re-projected at every call, no key-padding mask (transformer.py:344-361, 542-548)
        if cfg.cross_attention:
            assert cross_src is not None
            # post-norm: the reference passes `src`, the LAYER INPUT, as the query source (transformer.py:569-572), not norm1's output
            h = _ln(x, sd, p + '.norm_cross', cfg.eps) if cfg.norm_first else src
            w = sd[p + '.cross_attention.in_proj_weight']
            bq = bk = bv = None
            if p + '.cross_attention.in_proj_bias' in sd:
                bq, bk, bv = sd[p + '.cross_attention.in_proj_bias'].chunk(3)
            qc, kc = F.linear(h, w[:C], bq), F.linear(cross_src, w[C:2 * C], bk)
            if cfg.qk_layer_norm_cross:   # transformer.py:358-360
                qc = _ln(qc, sd, p + '.cross_attention.q_layer_norm', cfg.eps)
                kc = _ln(kc, sd, p + '.cross_attention.k_layer_norm', cfg.eps)
            qc = qc.view(B, T, H, hd).transpose(1, 2)
            kc = kc.view(B, -1, H, hd).transpose(1, 2)
            vc = F.linear(cross_src, w[2 * C:], bv).view(B, -1, H, hd).transpose(1, 2)
            a = _attention(qc, kc, vc, False).transpose(1, 2).reshape(B, T, C)
            x = x + ls(p + '.layer_scale_cross') * F.linear(a, sd[p + '.cross_attention.out_proj.weight'],
                                                            sd.get(p + '.cross_attention.out_proj.bias'))
            if not cfg.norm_first:
                x = _ln(x, sd, p + '.norm_cross', cfg.eps)
        # --- feed forward, exact (erf) GELU
        h = _ln(x, sd, p + '.norm2', cfg.eps) if cfg.norm_first else x
        h = F.gelu(F.linear(h, sd[p + '.linear1.weight'], sd.get(p + '.linear1.bias')))
        x = x + ls(p + '.layer_scale_2') * F.linear(h, sd[p + '.linear2.weight'], sd.get(p + '.linear2.bias'))
        if not cfg.norm_first:
            x = _ln(x, sd, p + '.norm2', cfg.eps)
    if state is not None:
        state.offset = offset + T
    return x


def cross_pos_emb(cross_src: torch.Tensor, scale: float) -> torch.Tensor:
    """ConditionFuser's cross_attention_pos_emb (conditioners.py:1750-1757): a sinusoidal embedding of the source positions,
    times cross_attention_pos_emb_scale, added to the concatenated cross-attention source."""
    positions = torch.arange(cross_src.shape[1]).view(1, -1, 1)
    return cross_src + scale * create_sin_embedding(positions, cross_src.shape[-1])


def lm_forward(sd: dict, cfg: LMConfig, sequence: torch.Tensor, cross_src: tp.Optional[torch.Tensor],
               prepend_src: tp.Optional[torch.Tensor] = None,
               state: tp.Optional[LMState] = None,
               input_ops: tp.Sequence[tp.Tuple[str, torch.Tensor]] = ()) -> torch.Tensor:
    """LMModel.forward (audiocraft/models/lm.py:221-268) with precomputed condition tensors.
    sequence [B, K, S] int64 -> logits [B, K, S, card].  `prepend_src` [B, P, C] is concatenated
    before the tokens on the first (or non-streaming) call only (ConditionFuser.forward,
    conditioners.py:1739-1741) and the logits are cropped back to the last S steps (lm.py:265-266).
    `input_ops`: the fuser's 'sum' / 'input_interpolate' conditions in dict order (and, when a prepend condition precedes one of
    them in that order, ('prepend', cond) entries instead of `prepend_src`), applied to the embedded input of EVERY
    call before `prepend_src` joins it (conditioners.py:1733-1737): ('sum', cond [B, 1 | T, C]) is added (broadcast like
    the reference's in-place `input += cond`), ('input_interpolate', cond [B, Tc, C]) is nearest-resampled to the call's
    length first -- a one-step streaming call therefore always receives its frame 0."""
    B, K, S = sequence.shape
    x = sum(F.embedding(sequence[:, k], sd[f'emb.{k}.weight']) for k in range(K))
    first = state.first_step if state is not None else True
    # the reference walks the conditions in dict order (conditioners.py:1730-1748): an op that comes AFTER a 'prepend' one sees
    # the prepended rows as part of the input.  ('prepend', cond) entries in `input_ops` keep that order; `prepend_src` (the
    # common case: every prepend after every sum / interpolate) is the last entry.
    ops = list(input_ops) + ([('prepend', prepend_src)] if prepend_src is not None else [])
    for op, cond in ops:
        if op == 'sum':
            assert cond.shape[1] in (1, x.shape[1]), "the reference's in-place add cannot broadcast this"
            x = x + cond
        elif op == 'input_interpolate':
            x = x + F.interpolate(cond.transpose(1, 2), size=x.shape[1]).transpose(1, 2)
        elif op == 'prepend':
            if first:
                x = torch.cat([cond, x], dim=1)
        else:
            raise ValueError(op)
    if state is not None:
        state.first_step = False
    out = transformer_forward(sd, cfg, x, cross_src, state)
    if cfg.norm_first:   # lm.py:171-173, 260-261: out_norm only exists on pre-norm models
        out = _ln(out, sd, 'out_norm', cfg.eps)
    logits = torch.stack([F.linear(out, sd[f'linears.{k}.weight'], sd.get(f'linears.{k}.bias'))
                          for k in range(K)], dim=1)
    return logits[:, :, -S:]


# ----------------------------------------------------------------------------- sampling

def multinomial(probs: torch.Tensor, generator=None) -> torch.Tensor:
    """utils.multinomial (audiocraft/utils/utils.py:88-105), num_samples=1."""
    flat = probs.reshape(-1, probs.shape[-1])
    out = torch.multinomial(flat, num_samples=1, generator=generator)
    return out.reshape(*probs.shape[:-1], 1)


def top_k_filter(probs: torch.Tensor, k: int) -> torch.Tensor:
    """The deterministic half of utils.sample_top_k (utils.py:108-122): keep every prob >= the
    k-th largest (ties included), renormalise."""
    kth = torch.topk(probs, k, dim=-1)[0][..., [-1]]
    probs = probs * (probs >= kth).float()
    return probs / probs.sum(dim=-1, keepdim=True)


def top_p_filter(probs: torch.Tensor, p: float):
    """Deterministic half of utils.sample_top_p (utils.py:125-141) -> (sorted probs, sort index)."""
    ps, pi = torch.sort(probs, dim=-1, descending=True)
    cum = torch.cumsum(ps, dim=-1)
    ps = ps * (~(cum - ps > p)).float()
    return ps / ps.sum(dim=-1, keepdim=True), pi


def cfg_mix(all_logits: torch.Tensor, cfg_coef: float) -> torch.Tensor:
    """lm.py:391-399: rows [cond; uncond] -> uncond + (cond - uncond) * coef."""
    B = all_logits.shape[0] // 2
    cond, uncond = all_logits.split(B, dim=0)
    return uncond + (cond - uncond) * cfg_coef


def double_cfg_mix(all_logits: torch.Tensor, cfg_coef: float, cfg_coef_beta: float) -> torch.Tensor:
    """lm.py:372-376 (MusicGen-Style double CFG): rows [text + wav; wav only; null] ->
    uncond + coef * (wav + beta * (cond - wav) - uncond)."""
    B = all_logits.shape[0] // 3
    cond, wav, uncond = all_logits.split(B, dim=0)
    return uncond + cfg_coef * (wav + cfg_coef_beta * (cond - wav) - uncond)


def sample_next_token(logits: torch.Tensor, use_sampling: bool, temp: float, top_k: int, top_p: float,
                      generator=None) -> torch.Tensor:
    """lm.py:402-418 on logits [B, K, card] (last step) -> [B, K, 1]."""
    if use_sampling and temp > 0.0:
        probs = torch.softmax(logits / temp, dim=-1)
        if top_p > 0.0:
            ps, pi = top_p_filter(probs, top_p)
            return torch.gather(pi, -1, multinomial(ps, generator))
        if top_k > 0:
            return multinomial(top_k_filter(probs, top_k), generator)
        return multinomial(probs, generator)
    return torch.argmax(logits, dim=-1, keepdim=True)


# ----------------------------------------------------------------------------- generate

@torch.no_grad()
def generate(sd: dict, cfg: LMConfig, prompt: tp.Optional[torch.Tensor], num_samples: int,
             cross_src: tp.Optional[torch.Tensor], prepend_src: tp.Optional[torch.Tensor] = None,
             max_gen_len: int = 256, use_sampling: bool = True, temp: float = 1.0, top_k: int = 250,
             top_p: float = 0.0, cfg_coef: tp.Optional[float] = None, remove_prompts: bool = False,
             generator=None, callback=None, return_logits: bool = False, max_steps: tp.Optional[int] = None,
             cfg_coef_beta: tp.Optional[float] = None, null_cross_src: tp.Optional[torch.Tensor] = None,
             null_prepend_src: tp.Optional[torch.Tensor] = None,
             input_ops: tp.Sequence[tp.Tuple[str, torch.Tensor]] = (),
             null_input_ops: tp.Sequence[tp.Tuple[str, torch.Tensor]] = ()):
    """LMModel.generate (audiocraft/models/lm.py:420-587).

    Default one-forward CFG mode: `cross_src` / `prepend_src` hold the already-batched `[cond; uncond]` condition
    tensors ([2B, L, C]); pass both as None for unconditional generation (no CFG).
    `cfg_coef_beta` (double CFG, lm.py:362-376): the condition tensors hold `[text + wav; wav only; null]` (3B rows).
    `null_cross_src` / `null_prepend_src` (two_step_cfg, lm.py:377-386): the conditional tensors hold B rows and
    the unconditional pass runs separately on these, with its own condition length and streaming state; the mix
    then uses the model's `cfg.cfg_coef` -- the reference ignores the `cfg_coef` argument on that branch.
    `input_ops` / `null_input_ops`: see lm_forward (rows batched like the other condition tensors).
    """
    coef = cfg.cfg_coef if cfg_coef is None else cfg_coef
    use_cfg = cross_src is not None or prepend_src is not None or len(input_ops) > 0
    two_step = null_cross_src is not None or null_prepend_src is not None
    assert not (two_step and cfg_coef_beta is not None)
    null_state = LMState(cfg.num_layers) if two_step else None
    K, special, unknown = cfg.n_q, cfg.card, -1
    if prompt is None:
        prompt = torch.zeros((num_samples, K, 0), dtype=torch.long)
    B, _, T0 = prompt.shape
    assert T0 < max_gen_len
    gen_codes = torch.full((B, K, max_gen_len), unknown, dtype=torch.long)
    gen_codes[..., :T0] = prompt
    layout = None if cfg.pattern is None else patterns.provider_layout(cfg.pattern[0], K, max_gen_len, **cfg.pattern[1])
    gen_sequence, mask = patterns.build_pattern_sequence(gen_codes, special, cfg.delays, layout)
    start = patterns.first_step_with_timestep(K, max_gen_len, T0, cfg.delays, layout)
    assert start is not None
    state = LMState(cfg.num_layers)
    S = gen_sequence.shape[-1]
    prev = 0
    all_logits = []
    for offset in range(start, S):
        if max_steps is not None and offset - start >= max_steps:
            break
        curr = gen_sequence[..., prev:offset]
        if two_step:
            cond = lm_forward(sd, cfg, curr, cross_src, prepend_src, state, input_ops)
            uncond = lm_forward(sd, cfg, curr, null_cross_src, null_prepend_src, null_state, null_input_ops)
            logits = uncond + (cond - uncond) * cfg.cfg_coef
        elif cfg_coef_beta is not None:
            logits = lm_forward(sd, cfg, torch.cat([curr, curr, curr], dim=0), cross_src, prepend_src, state, input_ops)
            logits = double_cfg_mix(logits, coef, cfg_coef_beta)
        else:
            seq = torch.cat([curr, curr], dim=0) if use_cfg else curr
            logits = lm_forward(sd, cfg, seq, cross_src, prepend_src, state, input_ops)
            if use_cfg:
                logits = cfg_mix(logits, coef)
        logits = logits[:, :, -1]  # [B, K, card]
        if return_logits:
            all_logits.append(logits)
        nxt = sample_next_token(logits, use_sampling, temp, top_k, top_p, generator)
        valid = mask[..., offset:offset + 1].expand(B, -1, -1)
        nxt[~valid] = special
        cur = gen_sequence[..., offset:offset + 1]
        gen_sequence[..., offset:offset + 1] = torch.where(cur == unknown, nxt, cur)
        prev = offset
        if callback is not None:
            callback(1 + offset - start, S - start)
    if max_steps is not None:
        return gen_sequence, (torch.stack(all_logits, dim=2) if return_logits else None)
    assert not (gen_sequence == unknown).any()
    out_codes, out_mask = patterns.revert_pattern_sequence(gen_sequence, unknown, max_gen_len, cfg.delays, layout)
    assert (out_codes != unknown).all() and out_mask.all()
    out = out_codes[..., (T0 if remove_prompts else 0):max_gen_len]
    assert (out >= 0).all() and (out <= cfg.card).all()
    if return_logits:
        return out, torch.stack(all_logits, dim=2)  # [B, K, steps, card]
    return out
