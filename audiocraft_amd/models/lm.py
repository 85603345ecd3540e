"""MusicGen language model on MI355X -- host side.

API mirror of `audiocraft.models.lm.LMModel` (reference audiocraft/models/lm.py:120-587): same
constructor arguments for the options MusicGen uses, same parameter names (so reference
checkpoints load with `load_state_dict`), same `generate(...)` signature and return value.

What differs is the execution model.  The reference walks ~25 ATen ops per layer per step from a
Python loop and re-concatenates the KV cache every step.  Here one decode position is a single
C-ABI call (`acmi_lm_step`, include/acmi.h) that enqueues ~9 fused HIP kernels per layer; the call
is captured once into a hipGraph and replayed for every step: the position counter, the token
sequence, the KV cache (written in place) and the sampler all live on the device, so the host does
nothing between steps except replay (and the optional progress callback).

 * weights: packed once (`_pack`) into bf16 (default on GPU, `BASELINE.json` config) or f32 (parity
   mode) [out, in] matrices; the 4 output heads are stacked into one [K*card, d] matrix;
 * cross-attention keys/values are projected ONCE per generate() instead of every step
   (reference transformer.py:344-361 recomputes them);
 * prompt / prepended-condition prefill runs through the same step, PREFILL_CHUNK consecutive positions per
   call as extra rows (streaming == batch, reference tests/modules/test_transformer.py:16-49).
"""
import ctypes as C
import os
import math
import typing as tp
from contextlib import contextmanager

import torch
from torch import nn

from .. import _C
from ..modules.codebooks_patterns import CodebooksPatternProvider
from ..modules.conditioners import (ClassifierFreeGuidanceDropout, ConditionFuser, ConditioningAttributes,
                                    ConditioningProvider, ConditionType, _drop_description_condition)

ConditionTensors = tp.Dict[str, ConditionType]


def _trunc_normal_(t: torch.Tensor, std: float):
    # reference get_init_fn('gaussian') (lm.py:35-59): N(0, std) truncated at 3 std
    return nn.init.trunc_normal_(t, mean=0.0, std=std, a=-3 * std, b=3 * std)


def expand_kv_in_proj(w: torch.Tensor, b: tp.Optional[torch.Tensor], d: int, num_heads: int, kv_repeat: int):
    """in_proj_weight / bias of a kv_repeat self-attention ([d + 2 kv_dim, d] / [d + 2 kv_dim], transformer.py:196-200) as the
    [3d, d] / [3d] of an ordinary one: every stored key / value head is laid out once per query head that shares it (query
    head h reads kv head h // kv_repeat: expand_repeated_kv, transformer.py:90-107, 398-400), so the QKV launch appends
    ordinary per-head K / V rows to the caches and no kernel knows about the option."""
    if kv_repeat == 1:
        return w, b
    hd, kvd = d // num_heads, d // kv_repeat

    def ex(t):
        return t.reshape(kvd // hd, hd, *t.shape[1:]).repeat_interleave(kv_repeat, dim=0).reshape(d, *t.shape[1:])
    w = torch.cat([w[:d], ex(w[d:d + kvd]), ex(w[d + kvd:])], dim=0)
    if b is not None:
        b = torch.cat([b[:d], ex(b[d:d + kvd]), ex(b[d + kvd:])], dim=0)
    return w, b


class _Attn(nn.Module):
    """Parameter container named like StreamingMultiheadAttention (custom / memory-efficient layout).
    kv_dim < dim: kv_repeat (transformer.py:196-200: the in-projection emits dim query features and kv_dim key / value
    features each); qk_layer_norm: `q_layer_norm` / `k_layer_norm` over the model dimension (transformer.py:216-222)."""
    def __init__(self, dim: int, bias: bool, device=None, kv_dim: tp.Optional[int] = None, qk_layer_norm: bool = False):
        super().__init__()
        kv_dim = dim if kv_dim is None else kv_dim
        self.in_proj_weight = nn.Parameter(torch.empty(dim + 2 * kv_dim, dim, device=device))
        self.in_proj_bias = nn.Parameter(torch.zeros(dim + 2 * kv_dim, device=device)) if bias else None
        self.out_proj = nn.Linear(dim, dim, bias=bias, device=device)
        if qk_layer_norm:
            self.q_layer_norm = nn.LayerNorm(dim, eps=1e-5, device=device)
            self.k_layer_norm = nn.LayerNorm(dim, eps=1e-5, device=device)


class _Rope(nn.Module):
    """Buffer container named like RotaryEmbedding / XPos (rope.py:14-61): `frequencies` [hd / 2], `xpos.decay_rates`
    [hd / 2]; the reference shares one instance between the transformer and every self-attention, so its state dict
    lists the buffers under `transformer.rope.*` and `transformer.layers.{i}.self_attn.rope.*`."""
    def __init__(self, hd: int, max_period: float, xpos: bool, device=None):
        super().__init__()
        adim = torch.arange(0, hd, 2, device=device, dtype=torch.float32)[: hd // 2]
        self.register_buffer('frequencies', 1.0 / (max_period ** (adim / hd)))
        self.xpos: tp.Optional[nn.Module] = None
        if xpos:
            self.xpos = nn.Module()
            half = hd // 2
            self.xpos.register_buffer('decay_rates', (torch.arange(half, device=device, dtype=torch.float32) / half + 0.4) / 1.4)


class _LayerScale(nn.Module):
    """Parameter container named like LayerScale (transformer.py:92-110): `scale` [channels]."""
    def __init__(self, channels: int, init: float, device=None):
        super().__init__()
        self.scale = nn.Parameter(torch.full((channels,), float(init), device=device))


class _Layer(nn.Module):
    """Parameter container named like StreamingTransformerLayer (transformer.py:454-574)."""
    def __init__(self, dim: int, ffn: int, cross_attention: bool, bias_ff: bool, bias_attn: bool, device=None,
                 layer_scale: tp.Optional[float] = None, kv_dim: tp.Optional[int] = None, qk_layer_norm: bool = False,
                 qk_layer_norm_cross: bool = False):
        super().__init__()
        self.self_attn = _Attn(dim, bias_attn, device, kv_dim, qk_layer_norm)
        self.linear1 = nn.Linear(dim, ffn, bias=bias_ff, device=device)
        self.linear2 = nn.Linear(ffn, dim, bias=bias_ff, device=device)
        self.norm1 = nn.LayerNorm(dim, eps=1e-5, device=device)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5, device=device)
        self.cross_attention: tp.Optional[_Attn] = None
        if cross_attention:
            self.cross_attention = _Attn(dim, bias_attn, device, None, qk_layer_norm_cross)
            self.norm_cross = nn.LayerNorm(dim, eps=1e-5, device=device)
        if layer_scale is not None:   # LayerScale (transformer.py:92-110, 526-538): folded into the branch's last matrix at pack time
            self.layer_scale_1 = _LayerScale(dim, layer_scale, device)
            self.layer_scale_2 = _LayerScale(dim, layer_scale, device)
            if cross_attention:
                self.layer_scale_cross = _LayerScale(dim, layer_scale, device)


class _Transformer(nn.Module):
    def __init__(self, dim, ffn, num_layers, cross_attention, bias_ff, bias_attn, device=None, layer_scale=None,
                 rope: tp.Optional[_Rope] = None, kv_dim: tp.Optional[int] = None, qk_layer_norm: bool = False,
                 qk_layer_norm_cross: bool = False):
        super().__init__()
        self.rope = rope
        self.layers = nn.ModuleList([_Layer(dim, ffn, cross_attention, bias_ff, bias_attn, device, layer_scale, kv_dim,
                                            qk_layer_norm, qk_layer_norm_cross)
                                     for _ in range(num_layers)])
        if rope is not None:
            for layer in self.layers:
                layer.self_attn.rope = rope


class LMModel(nn.Module):
    """Transformer LM over K parallel codebook streams, MI355X execution.

    Args mirror `audiocraft.models.lm.LMModel.__init__` (lm.py:145-177) plus:
        weight_dtype: torch.bfloat16 (bench / serving) or torch.float32 (parity mode) for the packed matrices.
        kv_dtype: dtype of the KV caches (defaults to weight_dtype).
    Beyond the MusicGen configuration the decode step also implements the transformer options no release uses
    (config/model/lm/default.yaml:25-46): positional_embedding 'rope' / 'sin_rope' (+ xpos), past_context, layer_scale,
    kv_repeat (the shared key / value heads are laid out per query head when the weights are packed: no kernel knows),
    qk_layer_norm / qk_layer_norm_cross (a launch of their own after the projections), and post-norm layers
    (norm_first=False, the constructor default here as in the reference; transformer.py:567-573 -- a correctness path: plain
    matrices, every LayerNorm a launch of its own, no out_norm, see include/acmi.h acmi_lm_model.post_norm).  Unsupported
    reference options raise: non-causal / non-GELU transformers (and, like the reference, norms other than 'layer_norm').
    """

    def __init__(self, pattern_provider: CodebooksPatternProvider, condition_provider: ConditioningProvider,
                 fuser: ConditionFuser, n_q: int = 8, card: int = 1024, dim: int = 128, num_heads: int = 8,
                 hidden_scale: int = 4, norm: str = 'layer_norm', norm_first: bool = False,
                 emb_lr: tp.Optional[float] = None, bias_proj: bool = True,
                 weight_init: tp.Optional[str] = None, depthwise_init: tp.Optional[str] = None,
                 zero_bias_init: bool = False, cfg_dropout: float = 0, cfg_coef: float = 1.0,
                 attribute_dropout: tp.Dict[str, tp.Dict[str, float]] = {}, two_step_cfg: bool = False,
                 num_layers: int = 2, cross_attention: bool = False, bias_ff: bool = True, bias_attn: bool = True,
                 positional_embedding: str = 'sin', max_period: float = 10000., positional_scale: float = 1.0,
                 causal: bool = True, activation: str = 'gelu', dropout: float = 0.0,
                 weight_dtype: torch.dtype = torch.bfloat16, kv_dtype: tp.Optional[torch.dtype] = None,
                 device=None, **kwargs):
        super().__init__()
        if norm != 'layer_norm':
            raise ValueError(f"Unknown norm type: {norm}")   # create_norm_fn, transformer.py:54-67: the reference knows no other
        if not norm_first and dim > 2048:
            raise NotImplementedError("norm_first=False for dim > 2048 (acmi_layer_norm_rows)")
        self.norm_first = bool(norm_first)
        if positional_embedding not in ('sin', 'rope', 'sin_rope'):
            raise ValueError(f"positional_embedding {positional_embedding!r} (transformer.py:632)")
        if activation != 'gelu' or not causal:
            raise NotImplementedError("only causal, GELU transformers")
        assert dim % num_heads == 0 and dim % 8 == 0
        self.kv_repeat = int(kwargs.get('kv_repeat', 1))
        self.qk_layer_norm = bool(kwargs.get('qk_layer_norm', False))
        self.qk_layer_norm_cross = bool(kwargs.get('qk_layer_norm_cross', False)) and cross_attention   # (only exists with one)
        assert self.kv_repeat >= 1 and num_heads % self.kv_repeat == 0, "kv_repeat must divide num_heads (transformer.py:197)"
        assert not self.qk_layer_norm or self.kv_repeat == 1, "qk_layer_norm needs kv_repeat == 1 (transformer.py:219)"
        if dim > 2048 and (self.qk_layer_norm or self.qk_layer_norm_cross):
            raise NotImplementedError("qk_layer_norm for dim > 2048 (acmi_layer_norm_rows)")
        self.positional_embedding = positional_embedding
        self.xpos = bool(kwargs.get('xpos', False))
        self.past_context: tp.Optional[int] = kwargs.get('past_context')
        layer_scale = kwargs.get('layer_scale')
        assert self.past_context is None or self.past_context > 0
        self.cfg_coef = cfg_coef
        self.cfg_dropout = ClassifierFreeGuidanceDropout(p=cfg_dropout)
        self.condition_provider = condition_provider
        self.fuser = fuser
        self.card = card
        self.n_q = n_q
        self.dim = dim
        self.num_heads = num_heads
        self.num_layers = num_layers
        self.ffn_dim = int(hidden_scale * dim)
        self.pattern_provider = pattern_provider
        self.two_step_cfg = two_step_cfg
        self.max_period = max_period
        self.positional_scale = positional_scale
        self.has_cross_attention = cross_attention
        self.weight_dtype = weight_dtype
        self.kv_dtype = kv_dtype or weight_dtype
        self.emb = nn.ModuleList([nn.Embedding(card + 1, dim, device=device) for _ in range(n_q)])
        rope = None
        if positional_embedding in ('rope', 'sin_rope'):
            rope = _Rope(dim // num_heads, max_period, self.xpos, device)
        kv_dim = (dim // num_heads) * (num_heads // self.kv_repeat)
        self.transformer = _Transformer(dim, self.ffn_dim, num_layers, cross_attention, bias_ff, bias_attn, device, layer_scale,
                                        rope, kv_dim, self.qk_layer_norm, self.qk_layer_norm_cross)
        # lm.py:171-173: out_norm only exists on pre-norm models (a post-norm stack ends with its last layer's norm2)
        self.out_norm: tp.Optional[nn.LayerNorm] = nn.LayerNorm(dim, eps=1e-5, device=device) if norm_first else None
        self.linears = nn.ModuleList([nn.Linear(dim, card, bias=bias_proj, device=device) for _ in range(n_q)])
        self._init_weights(weight_init, depthwise_init, zero_bias_init)
        self._packed: tp.Optional[dict] = None
        self._run: tp.Optional[dict] = None
        self._graph_keepalive = None
        self._stream = None
        self._is_streaming = False
        self.eval()

    # ------------------------------------------------------------------------------------- init
    def _init_weights(self, weight_init, depthwise_init, zero_bias_init):
        """Same distributions as the reference (lm.py:179-211); never used for parity (tests copy weights)."""
        with torch.no_grad():
            for li, layer in enumerate(self.transformer.layers):
                nn.init.kaiming_uniform_(layer.self_attn.in_proj_weight, a=math.sqrt(5))
                if layer.cross_attention is not None:
                    nn.init.kaiming_uniform_(layer.cross_attention.in_proj_weight, a=math.sqrt(5))
            if weight_init is None:
                return
            assert weight_init == 'gaussian', "only 'gaussian' init is implemented"
            for e in self.emb:
                _trunc_normal_(e.weight, 1 / math.sqrt(self.dim))
            for li, layer in enumerate(self.transformer.layers):
                depth = {'current': li + 1, 'global': self.num_layers, None: None}[depthwise_init]
                for m in layer.modules():
                    if isinstance(m, nn.Linear):
                        std = 1 / math.sqrt(m.in_features)
                        if depth is not None:
                            std /= math.sqrt(2 * depth)
                        _trunc_normal_(m.weight, std)
                        if zero_bias_init and m.bias is not None:
                            nn.init.zeros_(m.bias)
            for lin in self.linears:
                _trunc_normal_(lin.weight, 1 / math.sqrt(self.dim))

    @property
    def special_token_id(self) -> int:
        return self.card

    @property
    def num_codebooks(self) -> int:
        return self.n_q

    @property
    def device(self):
        return next(iter(self.parameters())).device

    def _invalidate(self):
        """Drop the packed kernel weights, the run state and any captured graph (they hold raw device pointers)."""
        if getattr(self, '_masters_released', False) and not getattr(self, '_loading', False):
            raise RuntimeError("the f32 master weights were released (release_master_weights): load_state_dict() before moving "
                               "or re-packing the model")
        self._packed = None
        self._run = None
        self._graph_keepalive = None
        self._stream = None

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        self._loading = True
        try:
            if getattr(self, '_masters_released', False):   # give the parameters their storage back first
                for name, p in self.named_parameters():
                    if name in self._master_shapes:
                        shape, dtype = self._master_shapes[name]
                        p.data = torch.empty(shape, device=p.device, dtype=dtype)
                self._masters_released = False
            self._invalidate()
            return super().load_state_dict(state_dict, strict=strict, **kw)
        finally:
            self._loading = False

    def _apply(self, fn, *args, **kw):   # .to() / .cuda() / .float(): parameters move, the packs must be rebuilt
        result = super()._apply(fn, *args, **kw)
        self._invalidate()
        return result

    # ------------------------------------------------------------------------------------- packing
    def _pack(self):
        """Kernel-side weight layout + ctypes descriptors (rebuilt after load_state_dict / .to())."""
        dev = self.device
        if dev.type != 'cuda':
            raise RuntimeError("audiocraft_amd.LMModel runs on an MI355X only (no CPU fallback); move it to 'cuda'")
        wd = self.weight_dtype
        keep: tp.List[torch.Tensor] = []

        def W(t, half=False):  # nn.Linear weight [N, K] -> MFMA B-fragment order ("tiled weight", include/acmi.h)
            tw = _C.TiledWeight(t.detach().to(dev), wd, half=half)
            keep.append(tw)
            return tw

        def E(t):  # embedding tables stay row-major
            t = t.detach().to(device=dev, dtype=wd).contiguous()
            keep.append(t)
            return t

        def Fp(t):
            if t is None:
                return None
            t = t.detach().to(device=dev, dtype=torch.float32).contiguous()
            keep.append(t)
            return t

        post = not self.norm_first

        def folded(w, norm, own_bias=None):
            """LN(x) W^T = standardise(x) W'^T + W beta, W' = W diag(gamma)  ->  (tiled W', f32 bias W beta,
            f32 column sums of the dtype-rounded W': standardise(x) W'^T = rstd (x W'^T - mean colsum)).
            Post-norm models (norm_first=False): the GEMM consumes x itself -> (tiled W, the projection's own bias or zeros,
            no column sums: acmi_lm_step then runs the unfolded form)."""
            w32 = w.detach().to(device=dev, dtype=torch.float32)
            if post:
                own = torch.zeros(w32.shape[0], device=dev) if own_bias is None else own_bias
                return W(w32), Fp(own), None
            g = norm.weight.detach().to(device=dev, dtype=torch.float32)
            b = norm.bias.detach().to(device=dev, dtype=torch.float32)
            bias = w32 @ b
            if own_bias is not None:
                bias = bias + own_bias.detach().to(device=dev, dtype=torch.float32)
            wf = w32 * g[None, :]
            return W(wf), Fp(bias), Fp(wf.to(wd).double().sum(dim=1).float())

        d = self.dim

        def self_in_proj(attn):
            return expand_kv_in_proj(attn.in_proj_weight.detach(), None if attn.in_proj_bias is None else attn.in_proj_bias.detach(),
                                     d, self.num_heads, self.kv_repeat)

        layers = (_C.LMLayer * self.num_layers)()
        pk: dict = {'keep': keep, 'layers': layers, 'per_layer': []}
        for li, layer in enumerate(self.transformer.layers):
            def scaled_bias(b, name):   # bias of a residual branch's last projection, times the branch's LayerScale
                if b is None:
                    return None
                ls = getattr(layer, name, None)
                b32 = b.detach().to(device=dev, dtype=torch.float32)
                return Fp(b32 if ls is None else b32 * ls.scale.detach().to(device=dev, dtype=torch.float32))

            def scaled(w, name):   # LayerScale: x + s * (a W^T) = x + a (diag(s) W)^T, folded into the matrix (f32, then rounded)
                ls = getattr(layer, name, None)
                w32 = w.detach().to(device=dev, dtype=torch.float32)
                return w32 if ls is None else w32 * ls.scale.detach().to(device=dev, dtype=torch.float32)[:, None]
            w_out32, w_ff2_32 = scaled(layer.self_attn.out_proj.weight, 'layer_scale_1'), scaled(layer.linear2.weight, 'layer_scale_2')
            ent = {'w_out': W(w_out32), 'w_ff2': W(w_ff2_32)}
            for key, b, name in (('b_out', layer.self_attn.out_proj.bias, 'layer_scale_1'), ('b_ff2', layer.linear2.bias, 'layer_scale_2')):
                sb = scaled_bias(b, name)
                if sb is not None:
                    ent[key] = sb
            kt2 = 64 if wd == torch.bfloat16 else 32
            if d % 8 == 0 and d // 8 <= 256 and self.ffn_dim % kt2 == 0:
                # 8-feature workgroups for FFN2 in calls of <= 32 rows (acmi_lm_layer.w_ff2h): a second copy of the weight
                ent['w_ff2h'] = W(w_ff2_32, half=True)
            sa_w, sa_b = self_in_proj(layer.self_attn)
            ent['w_qkv'], ent['b_qkv'], ent['cs_qkv'] = folded(sa_w, layer.norm1, sa_b)
            if self.qk_layer_norm:
                for key, mod in (('q_ln', layer.self_attn.q_layer_norm), ('k_ln', layer.self_attn.k_layer_norm)):
                    ent[key + '_g'], ent[key + '_b'] = Fp(mod.weight), Fp(mod.bias)
            ent['w_ff1'], ent['b_ff1'], ent['cs_ff1'] = folded(layer.linear1.weight, layer.norm2, layer.linear1.bias)
            if layer.cross_attention is not None:
                ca = layer.cross_attention
                ipw, ipb = ca.in_proj_weight, ca.in_proj_bias
                ent['w_cq'], ent['b_cq'], ent['cs_cq'] = folded(ipw[:d], layer.norm_cross, None if ipb is None else ipb[:d])
                if ipb is not None:   # k / v biases: added when the cross-attention caches are filled (_project_cross_kv)
                    ent['b_ck'], ent['b_cv'] = Fp(ipb[d:2 * d]), Fp(ipb[2 * d:])
                    pk['cross_kv_bias'] = pk.get('cross_kv_bias', False) or bool((ipb[d:] != 0).any())
                if self.qk_layer_norm_cross:
                    # the query goes through q_layer_norm: it is no longer linear in x1, so it keeps a projection launch of
                    # its own (no w_qkvx / w_mq); the keys are normalised when the cross-attention caches are filled
                    ent['cq_ln_g'], ent['cq_ln_b'] = Fp(ca.q_layer_norm.weight), Fp(ca.q_layer_norm.bias)
                    ent['ck_ln_g'], ent['ck_ln_b'] = Fp(ca.k_layer_norm.weight), Fp(ca.k_layer_norm.bias)
                if self.qk_layer_norm_cross or post:
                    pass    # (post-norm: the query is projected from the layer input by a launch of its own as well)
                else:
                    # x1 W_cq'^T = x0 W_cq'^T + att (W_cq' W_out)^T with x1 = x0 + att W_out^T: the x0 part is a fourth block of
                    # output features of the QKV launch (raw, no LayerNorm epilogue), the att part rides in the out-projection
                    # launch (include/acmi.h, acmi_lm_layer.w_qkvx / w_mq)
                    g_c = layer.norm_cross.weight.detach().to(device=dev, dtype=torch.float32)
                    wq = ipw[:d].detach().to(device=dev, dtype=torch.float32) * g_c[None, :]
                    g_1 = layer.norm1.weight.detach().to(device=dev, dtype=torch.float32)
                    wqkv = sa_w.to(device=dev, dtype=torch.float32) * g_1[None, :]
                    ent['w_qkvx'] = W(torch.cat([wqkv, wq], dim=0))
                    ent['b_qkvx'] = Fp(torch.cat([ent['b_qkv'], torch.zeros(d, device=dev)]))
                    ent['cs_qkvx'] = Fp(torch.cat([ent['cs_qkv'], torch.zeros(d, device=dev)]))
                    ent['w_mq'] = W(wq @ w_out32)
                    if 'b_out' in ent:   # r = x1 W_cq'^T with x1 = x0 + att W_out^T + b_out
                        ent['b_mq'] = Fp(wq @ ent['b_out'])
                sb = scaled_bias(ca.out_proj.bias, 'layer_scale_cross')
                if sb is not None:
                    ent['b_cout'] = sb
                ent.update({'w_ck': W(ipw[d:2 * d]), 'w_cv': W(ipw[2 * d:]),
                            'w_cout': W(scaled(ca.out_proj.weight, 'layer_scale_cross'))})
            if post:   # the LayerNorms themselves: applied in place after each block's residual add
                for key, mod in (('n1', layer.norm1), ('n2', layer.norm2)) + \
                        ((('nc', layer.norm_cross),) if layer.cross_attention is not None else ()):
                    ent[key + '_g'], ent[key + '_b'] = Fp(mod.weight), Fp(mod.bias)
            ent = {k: v for k, v in ent.items() if v is not None}
            L = layers[li]
            for k in ('w_qkv', 'w_out', 'w_cq', 'w_cout', 'w_xcq', 'w_ff1', 'w_ff2', 'b_qkv', 'b_cq', 'b_ff1', 'cs_qkv',
                      'cs_cq', 'cs_ff1', 'w_qkvx', 'b_qkvx', 'cs_qkvx', 'w_mq', 'w_ff2h', 'b_out', 'b_cout', 'b_ff2', 'b_mq',
                      'q_ln_g', 'q_ln_b', 'k_ln_g', 'k_ln_b', 'cq_ln_g', 'cq_ln_b', 'n1_g', 'n1_b', 'nc_g', 'nc_b', 'n2_g', 'n2_b'):
                setattr(L, k, ent[k].data_ptr() if k in ent else None)
            pk['per_layer'].append(ent)
        embs = [E(e.weight) for e in self.emb]
        emb_arr = (_C.vp * self.n_q)(*[e.data_ptr() for e in embs])
        head_bias = None
        if self.linears[0].bias is not None:
            head_bias = torch.cat([lin.bias for lin in self.linears], dim=0)
        w_head, b_head, cs_head = folded(torch.cat([lin.weight for lin in self.linears], dim=0), self.out_norm, head_bias)
        desc_post = int(post)
        half = d // 2
        # divisor table of create_sin_embedding, computed exactly like the reference does
        # (transformer.py:83-88: f32 tensor ops on the host); cos/sin are evaluated on the device
        adim = torch.arange(half, dtype=torch.float32)
        pos_freq = Fp(torch.full([], self.max_period, dtype=torch.float32) ** (adim / max(half - 1, 1)))
        desc = _C.LMModelDesc()
        desc.dim, desc.num_heads, desc.num_layers, desc.ffn_dim = d, self.num_heads, self.num_layers, self.ffn_dim
        desc.n_q, desc.card = self.n_q, self.card
        desc.wdtype, desc.kvdtype = _C.dtype_code(wd), _C.dtype_code(self.kv_dtype)
        desc.cross_attention = int(self.has_cross_attention)
        # 'rope' alone: no sinusoidal embedding (transformer.py:701-704) -> its scale is 0 for the embedding kernel
        desc.eps, desc.positional_scale = 1e-5, (0.0 if self.positional_embedding == 'rope' else self.positional_scale)
        desc.rope_freq = desc.rope_decay = None
        desc.rope_scale, desc.rope_base = float(self.positional_scale), 512.0
        desc.past_context = int(self.past_context or 0)
        desc.post_norm = desc_post
        if self.positional_embedding in ('rope', 'sin_rope'):
            # the tables of RotaryEmbedding / XPos: buffers computed like the reference does (rope.py:26-30, 59-61)
            pk['rope_freq'] = Fp(self.transformer.rope.frequencies)
            desc.rope_freq = pk['rope_freq'].data_ptr()
            if self.xpos:
                pk['rope_decay'] = Fp(self.transformer.rope.xpos.decay_rates)
                desc.rope_decay = pk['rope_decay'].data_ptr()
        desc.layers = C.cast(layers, C.POINTER(_C.LMLayer))
        desc.emb = C.cast(emb_arr, C.POINTER(_C.vp))
        desc.pos_table = None
        desc.w_head, desc.b_head = w_head.data_ptr(), b_head.data_ptr()
        desc.cs_head = None if cs_head is None else cs_head.data_ptr()
        pk.update({'desc': desc, 'emb_arr': emb_arr, 'w_head': w_head, 'b_head': b_head, 'cs_head': cs_head,
                   'pos_freq': pos_freq, 'embs': embs})
        self._packed = pk
        self._run = None
        return pk

    # ------------------------------------------------------------------------------------- opt-in: drop the f32 masters
    def release_master_weights(self):
        """Opt-in memory saving for serving: free the f32 master copies of the big matrices (the transformer's projections,
        the embedding tables, the heads: 7.4 GB for MusicGen-medium) once the kernel-side packs exist.  The model keeps
        generating from the packs; `state_dict()` re-materialises the matrices from them on demand -- exact to the packed
        element type (bf16 packs give back the bf16-rounded weights the kernels compute with; the LayerNorm / LayerScale
        folds are divided out again) -- and `load_state_dict` restores full masters.  Moving the model (`.to()`) or
        changing dtypes afterwards needs a fresh `load_state_dict` first."""
        if getattr(self, '_masters_released', False):
            return
        self._packed or self._pack()
        self._master_shapes = {}
        for name, p in self.named_parameters():
            if p.dim() == 2 and not name.startswith('condition_provider.'):
                self._master_shapes[name] = (tuple(p.shape), p.dtype)
                p.data = torch.empty(0, device=p.device, dtype=p.dtype)
        self._masters_released = True

    def _rematerialise(self, name: str) -> torch.Tensor:
        """The matrix `name` of the reference's state dict, rebuilt from the packs (see release_master_weights)."""
        pk = self._packed
        assert pk is not None, "the packs are gone: load_state_dict() first"
        d = self.dim
        shape, dtype = self._master_shapes[name]

        def plain(tw, rows=None):
            m = _C.untile_matrix(tw.data, tw.N, tw.K).float()
            return m if rows is None else m[rows]

        def norm_w(mod):   # the LayerNorm gain folded into the consuming matrix (post-norm models fold nothing)
            return mod.weight.detach().float() if self.norm_first else torch.ones(d, device=self.device)

        parts = name.split('.')
        if parts[0] == 'emb':
            return pk['embs'][int(parts[1])].float().to(dtype)
        if parts[0] == 'linears':
            k = int(parts[1])
            return (plain(pk['w_head'], slice(k * self.card, (k + 1) * self.card)) / norm_w(self.out_norm)[None, :]).to(dtype)
        li = int(parts[2])
        layer, ent = self.transformer.layers[li], pk['per_layer'][li]
        key = '.'.join(parts[3:])

        def unscale(m, ls_name):
            ls = getattr(layer, ls_name, None)
            return m if ls is None else m / ls.scale.detach().float()[:, None]
        if key == 'self_attn.in_proj_weight':
            m = plain(ent['w_qkv']) / norm_w(layer.norm1)[None, :]
            if self.kv_repeat > 1:   # the packs hold every shared key / value head once per query head: keep the first copy
                hd, rep = d // self.num_heads, self.kv_repeat
                first = lambda t: t.reshape(self.num_heads // rep, rep, hd, d)[:, 0].reshape(-1, d)  # noqa: E731
                m = torch.cat([m[:d], first(m[d:2 * d]), first(m[2 * d:])], dim=0)
            return m.to(dtype)
        if key == 'self_attn.out_proj.weight':
            return unscale(plain(ent['w_out']), 'layer_scale_1').to(dtype)
        if key == 'linear1.weight':
            return (plain(ent['w_ff1']) / norm_w(layer.norm2)[None, :]).to(dtype)
        if key == 'linear2.weight':
            return unscale(plain(ent['w_ff2']), 'layer_scale_2').to(dtype)
        if key == 'cross_attention.in_proj_weight':
            q = plain(ent['w_cq']) / norm_w(layer.norm_cross)[None, :]
            return torch.cat([q, plain(ent['w_ck']), plain(ent['w_cv'])], dim=0).to(dtype)
        if key == 'cross_attention.out_proj.weight':
            return unscale(plain(ent['w_cout']), 'layer_scale_cross').to(dtype)
        raise KeyError(name)

    def state_dict(self, *args, **kwargs):
        sd = super().state_dict(*args, **kwargs)
        if getattr(self, '_masters_released', False):
            prefix = kwargs.get('prefix', args[1] if len(args) > 1 else '')
            for name in self._master_shapes:
                sd[prefix + name] = self._rematerialise(name)
        return sd

    # ------------------------------------------------------------------------------------- run state
    def _prepare_run(self, B: int, use_cfg: int, Tmax: int, Lc: int, S: int):
        """(Re)allocate KV caches / activations for this batch geometry; reused across generate() calls.
        use_cfg: _C.CFG_NONE / CFG_PAIR / CFG_DOUBLE -> 1, 2 or 3 row groups of B rows."""
        pk = self._packed or self._pack()
        use_cfg = int(use_cfg)
        key = (B, use_cfg, Tmax, Lc, S)
        if self._run is not None and self._run['key'] == key:
            return self._run
        self._stream = None   # a stream holds pointers into the buffers replaced below
        for L in pk['layers']:   # ... and so do the layer descriptors' score-folded cross-attention tables (ACMI_CROSS_FOLD)
            L.w_qkvs = L.b_qkvs = L.cs_qkvs = L.w_g2 = L.b_gs = L.xs_u = L.xs_cs = L.xs_bs = None
        dev = self.device
        Beff = B * (use_cfg + 1)
        H, hd, d = self.num_heads, self.dim // self.num_heads, self.dim
        # the decode kernels take their geometry in packed 16-bit launch words (include/acmi.h: acmi_attn_desc / acmi_linear_desc)
        if max(Tmax, Lc, Beff * self.PREFILL_CHUNK) > 65535:
            raise ValueError(f"stream of {Tmax} positions / {Lc} condition rows / {Beff * self.PREFILL_CHUNK} activation rows: libacmi's "
                             f"decode kernels address at most 65535 cache positions and rows (packed launch words); generate "
                             f"longer audio through MusicGen's windowed generation (extend_stride)")
        f32 = dict(device=dev, dtype=torch.float32)
        run: dict = {'key': key, 'Beff': Beff, 'graphs': {}}
        run['k'] = torch.zeros(self.num_layers, Beff, H, Tmax, hd, device=dev, dtype=self.kv_dtype)
        run['v'] = torch.zeros(self.num_layers, Beff, H, Tmax, hd, device=dev, dtype=self.kv_dtype)
        if self.has_cross_attention:
            run['ck'] = torch.zeros(self.num_layers, Beff, H, max(Lc, 1), hd, device=dev, dtype=self.kv_dtype)
            run['cv'] = torch.zeros(self.num_layers, Beff, H, max(Lc, 1), hd, device=dev, dtype=self.kv_dtype)
            # the values once more, time-minor and zero padded to 32 positions: the prefill's cross-attention (acmi_lm_layer.cvt_cache)
            run['cvt_tcap'] = -(-max(Lc, 1) // 32) * 32
            run['cvt'] = torch.zeros(self.num_layers, Beff, H, hd, run['cvt_tcap'], device=dev, dtype=self.kv_dtype)
        for li in range(self.num_layers):
            L = pk['layers'][li]
            L.k_cache, L.v_cache = run['k'][li].data_ptr(), run['v'][li].data_ptr()
            if self.has_cross_attention:
                L.ck_cache, L.cv_cache = run['ck'][li].data_ptr(), run['cv'][li].data_ptr()
                L.cvt_cache = run['cvt'][li].data_ptr()
        # activation rows: one decode position = Beff rows; a prefill call runs PREFILL_CHUNK consecutive positions
        # (prompt / prepended-condition rows) at once through the same kernels
        rows = Beff * self.PREFILL_CHUNK
        run['x'] = torch.zeros(rows, d, **f32)
        run['q'] = torch.zeros(rows, d, **f32)
        # activations that feed a GEMM directly live in A-fragment order, zero padded
        run['stats'] = torch.zeros(rows, max(1, d // 8), 2, **f32)
        # x as raw fragments: two (hi, lo) pairs, the hi buffers twice as wide so that the self-attention output
        # sits next to x ([x | att], the operand of the paired out-projection / cross-query launch)
        kt = _C._tile_params(self.weight_dtype)[1]
        dp = -(-d // kt) * kt
        for name in ('xn', 'xn2'):
            run[name] = _C.tiled_activation_buffer(rows, 2 * dp, self.weight_dtype, dev)
        for name in ('xlo', 'xlo2'):
            run[name] = _C.tiled_activation_buffer(rows, d, self.weight_dtype, dev)
        run['x_rbs'] = 2 * dp // kt
        # single-term fragments (bf16 weights): per-row shifts, two alternating halves (acmi_lm_state.xshift)
        run['xshift'] = torch.zeros(2 * rows, **f32)
        run['r'] = torch.zeros(rows, d, **f32)
        run['att'] = _C.tiled_activation_buffer(rows, d, self.weight_dtype, dev)
        run['hidden'] = _C.tiled_activation_buffer(rows, self.ffn_dim, self.weight_dtype, dev)
        run['pos_table'] = _C.pos_table(pk['pos_freq'], Tmax, d)
        run['logits'] = torch.zeros(Beff, self.n_q * self.card, **f32)
        run['step_logits'] = torch.zeros(B, self.n_q, self.card, **f32)
        run['pos'] = torch.zeros(4, device=dev, dtype=torch.int32)
        run['cross_len_rows'] = torch.full((Beff,), max(Lc, 1), device=dev, dtype=torch.int32)
        run['gen_sequence'] = torch.zeros(B, self.n_q, S, device=dev, dtype=torch.int64)
        run['seq_mask'] = torch.zeros(self.n_q, S, device=dev, dtype=torch.uint8)
        # QKV -> self-attention as one launch (acmi_lm_state.qkv_hand): the hand-off row, every word the sentinel, + the error word
        if self.QKV_ATTN_FUSED:
            run['qkv_hand'] = torch.full((Beff, 3 * d), self.HAND_SENTINEL, device=dev, dtype=torch.int32)
            run['hand_err'] = torch.zeros(4, device=dev, dtype=torch.int32)
        self._run = run
        return run

    # The decode step's QKV GEMM + self-attention as ONE launch where the geometry allows (include/acmi.h, acmi_lm_state.qkv_hand;
    # same arithmetic as the two launches; DESIGN.md section 5.11: +3.0 % medium B = 8, +4.0 % large, +4.6 % small B = 1, +5.8 % medium
    # B = 4 on one box, whole GPU suite green in both modes).  ACMI_QKV_ATTN=0 restores the two launches.
    QKV_ATTN_FUSED = os.environ.get('ACMI_QKV_ATTN', '1') != '0'
    HAND_SENTINEL = 0x7fc0dead

    def _hand_arm(self, run):
        """Before a sequence of decode steps: every hand-off slot armed, the error word cleared."""
        if 'qkv_hand' in run:
            run['qkv_hand'].fill_(self.HAND_SENTINEL)
            run['hand_err'].zero_()

    def _hand_check(self, run):
        """After them: a poll that gave up means a step consumed garbage (one host read per generate)."""
        if 'qkv_hand' in run and int(run['hand_err'][0]) != 0:
            n = int(run['hand_err'][0])
            self._hand_arm(run)
            raise _C.AcmiError(f"acmi_lm_step: {n} hand-off polls of the fused QKV + self-attention launch gave up "
                               "(set ACMI_QKV_ATTN=0 for the two launches)")

    def _make_state(self, run, B, use_cfg, Tmax, Lc, S, prepend, record_logits, use_sampling, temp, top_k, top_p,
                    cfg_coef, seed, cfg_coef_beta: float = 0.0, cross_lens: tp.Optional[torch.Tensor] = None,
                    row_off: tp.Optional[torch.Tensor] = None, input_add: tp.Optional[torch.Tensor] = None) -> _C.LMState:
        st = _C.LMState()
        if input_add is not None:    # fuser 'sum' / 'input_interpolate' (acmi_lm_state.input_add): [Beff, n_add, d] f32
            assert input_add.dim() == 3 and input_add.shape[0] == run['Beff'] and input_add.shape[2] == self.dim, input_add.shape
            run['input_add'] = input_add.to(device=self.device, dtype=torch.float32).contiguous()
            st.input_add, st.n_add = run['input_add'].data_ptr(), run['input_add'].shape[1]
        else:
            run.pop('input_add', None)
            st.input_add, st.n_add = None, 0
        if row_off is not None:      # left padding of the rows' streams (two_step_cfg with unequal prepend lengths)
            if self.positional_embedding in ('rope', 'sin_rope') and self.past_context is not None and self.past_context > 0:
                # (the rotary lag after a first call longer than past_context would differ per row group: rope_first / rope_shift)
                raise NotImplementedError("two_step_cfg with prepended conditions of different lengths on a rotary model "
                                          "with a bounded context (past_context)")
            run['row_off'] = row_off.to(device=self.device, dtype=torch.int32).contiguous()
            st.row_off = run['row_off'].data_ptr()
        else:
            run.pop('row_off', None)
            st.row_off = None
        st.rope_first = st.rope_shift = 0
        st.cfg_coef_beta = float(cfg_coef_beta)
        if cross_lens is not None:   # per-row cross-attention length (two_step_cfg)
            run['cross_len_rows'].copy_(cross_lens.to(torch.int32))
            st.cross_len_rows = run['cross_len_rows'].data_ptr()
        else:
            st.cross_len_rows = None
        st.Beff, st.B, st.use_cfg, st.Tmax, st.Lc = run['Beff'], B, int(use_cfg), Tmax, Lc
        st.n_prepend = 0 if prepend is None else prepend.shape[1]
        self._last_n_prepend = st.n_prepend   # (bench.py: the prefix rows a generate put in front of the token stream)
        st.S = S
        st.gen_sequence, st.seq_mask = run['gen_sequence'].data_ptr(), run['seq_mask'].data_ptr()
        st.prepend = None if prepend is None else prepend.data_ptr()
        st.pos = run['pos'].data_ptr()
        st.x, st.q, st.att = run['x'].data_ptr(), run['q'].data_ptr(), run['att'].data_ptr()
        st.stats = run['stats'].data_ptr()
        st.xn, st.xlo, st.xn2, st.xlo2 = (run[k].data_ptr() for k in ('xn', 'xlo', 'xn2', 'xlo2'))
        st.x_rbs, st.r = run['x_rbs'], run['r'].data_ptr()
        st.xshift = run['xshift'].data_ptr() if self.weight_dtype == torch.bfloat16 else None
        st.cross_active_rows = 0
        st.pf_xn, st.pf_vt, st.pf_tcap = None, None, 0
        st.cvt_tcap = run.get('cvt_tcap', 0) if self.has_cross_attention else 0
        st.hidden, st.logits = run['hidden'].data_ptr(), run['logits'].data_ptr()
        st.step_logits = run['step_logits'].data_ptr() if record_logits else None
        st.use_sampling, st.temp, st.top_k, st.top_p = int(use_sampling), float(temp), int(top_k), float(top_p)
        st.cfg_coef, st.seed = float(cfg_coef), int(seed) & ((1 << 64) - 1)
        st.qkv_hand = run['qkv_hand'].data_ptr() if 'qkv_hand' in run else None
        st.hand_err = run['hand_err'].data_ptr() if 'qkv_hand' in run else None
        return st

    def _project_cross_kv(self, run, cross_src: torch.Tensor):
        """K/V projection of the cross-attention source, once per generate (vs. every step in the
        reference, transformer.py:344-361).  cross_src [Beff, Lc, d] f32."""
        pk = self._packed
        Beff, Lc, d = cross_src.shape
        flat = cross_src.reshape(Beff * Lc, d).contiguous()
        tmp = torch.empty(Beff * Lc, d, device=flat.device, dtype=torch.float32)
        for li in range(self.num_layers):
            ent = pk['per_layer'][li]
            _C.linear(flat, ent['w_ck'], tmp, bias=ent.get('b_ck'))
            if 'ck_ln_g' in ent:     # qk_layer_norm_cross: k_layer_norm on the projected keys (transformer.py:358-360)
                _C.layer_norm_rows(tmp, ent['ck_ln_g'], ent['ck_ln_b'], 1e-5, out=tmp)
            _C.kv_store(tmp.view(Beff, Lc, d), run['ck'][li], 0)
            _C.linear(flat, ent['w_cv'], tmp, bias=ent.get('b_cv'))
            _C.kv_store(tmp.view(Beff, Lc, d), run['cv'][li], 0)
        run['cvt'][..., :Lc].copy_(run['cv'].transpose(3, 4))   # [L, Beff, H, hd, Lc]: one strided copy per generate

    # ------------------------------------------------------------------------------------- score-folded cross-attention
    def _cross_fold_rows(self, run, n_live: int, Lc: int, two_step: bool) -> int:
        """Rows R for which the decode step runs the score-folded cross-attention (include/acmi.h, acmi_lm_state.xs_rows;
        modules/cross_fold.py), or 0 = the separate launches.  The per-generate tables hold R H Lc d elements -- three of them
        per layer -- where the launches they replace stream d d: taken while R H Lc <= 2 d (MusicGen-medium: 8 conditioned rows x
        24 heads x 16 text positions = 2 d), with the paired launches available and one source length for all rows."""
        import os
        # OPT-IN (ACMI_CROSS_FOLD=1): parity-green but measured slower than the separate launches at every batch size on MI355X
        # (B = 8 x 30 s: RTF 60.5 against 65.8; B = 1 / 2 / 4 x 10 s: 1.96 / 2.01 / 2.07 against 1.88 / 1.90 / 1.95 ms per position;
        # DESIGN.md 5.9) -- one launch less per layer does not pay for wider QKV / paired launches and the table build
        if os.environ.get('ACMI_CROSS_FOLD', '') != '1' or not self.has_cross_attention or two_step or Lc <= 0:
            return 0
        pk = self._packed
        d, H, Beff = self.dim, self.num_heads, run['Beff']
        kt = _C._tile_params(self.weight_dtype)[1]
        if 'w_qkvx' not in pk['per_layer'][0] or getattr(self, '_masters_released', False):
            return 0          # no paired launches (qk_layer_norm_cross, post-norm) / the f32 matrices the tables are folded from are gone
        R = n_live if 0 < n_live < Beff else Beff
        HL = H * Lc
        if R * HL > 2 * d or HL > 1024 or d % kt or (3 * d) % 16 or d // 16 > 128:
            return 0
        if Beff * (-(-R * HL // 16) * 16) > run['r'].numel():
            return 0
        return R

    def _build_cross_fold(self, run, R: int, Lc: int):
        """Fill the per-generate tables of the score-folded cross-attention from the cross-attention caches just projected
        (torch tensor algebra on the device, once per generate: modules/cross_fold.py) and point the layers at them."""
        from ..modules import cross_fold
        pk = self._packed
        dev, wd = self.device, self.weight_dtype
        d, H = self.dim, self.num_heads
        HL, N = H * Lc, R * H * Lc
        Np = -(-N // 16) * 16        # whole 16-feature tiles (tile_matrix pads the rows with zeros)
        nt3 = 3 * d // 16
        xs = run.get('xs')
        if xs is None or xs['key'] != (R, Lc):
            xs = {'key': (R, Lc), 'layers': []}
            for li in range(self.num_layers):
                ent = pk['per_layer'][li]
                w3 = ent['w_qkv'].data                                   # [3d / 16, K tiles, 4, 16, e]: n-tile major
                w_qkvs = torch.empty((nt3 + Np // 16,) + tuple(w3.shape[1:]), device=dev, dtype=wd)
                w_qkvs[:nt3].copy_(w3)
                zeros = torch.zeros(Np, device=dev)
                xs['layers'].append({'w_qkvs': w_qkvs, 'b_qkvs': torch.cat([ent['b_qkv'], zeros]).contiguous(),
                                     'cs_qkvs': torch.cat([ent['cs_qkv'], zeros]).contiguous()})
            run['xs'] = xs
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32)  # noqa: E731
        for li, layer in enumerate(self.transformer.layers):
            ent, bufs, ca = pk['per_layer'][li], xs['layers'][li], layer.cross_attention

            def scaled(w, name):
                ls = getattr(layer, name, None)
                return f32(w) if ls is None else f32(w) * f32(ls.scale)[:, None]
            kc, vc = run['ck'][li][:R].float(), run['cv'][li][:R].float()     # the VALUES the attention kernels would read
            wq = f32(ca.in_proj_weight[:d]) * f32(layer.norm_cross.weight)[None, :]
            t = cross_fold.fold_tables(kc, vc, wq, ent['b_cq'], scaled(layer.self_attn.out_proj.weight, 'layer_scale_1'),
                                       ent.get('b_out'), scaled(ca.out_proj.weight, 'layer_scale_cross'), wd)
            bufs['w_qkvs'][nt3:].copy_(_C.tile_matrix(t['G'].reshape(N, d), wd))
            bufs['w_g2'] = _C.tile_matrix(t['G2'].reshape(N, d), wd)
            bufs['u'] = _C.cross_fold_u_layout(t['U'], wd)
            bufs['cs'], bufs['bs'] = t['CS'].contiguous(), t['BS'].contiguous()
            bufs['b_gs'] = None if t['b_gs'] is None else torch.cat([t['b_gs'].reshape(N), torch.zeros(Np - N, device=dev)]).contiguous()
            L = pk['layers'][li]
            L.w_qkvs, L.b_qkvs, L.cs_qkvs = bufs['w_qkvs'].data_ptr(), bufs['b_qkvs'].data_ptr(), bufs['cs_qkvs'].data_ptr()
            L.w_g2, L.b_gs = bufs['w_g2'].data_ptr(), (None if bufs['b_gs'] is None else bufs['b_gs'].data_ptr())
            L.xs_u, L.xs_cs, L.xs_bs = bufs['u'].data_ptr(), bufs['cs'].data_ptr(), bufs['bs'].data_ptr()

    # ------------------------------------------------------------------------------------- conditions
    def _cfg_condition_tensors(self, conditions: tp.List[ConditioningAttributes], cfg_coef_beta=None,
                               two_step_cfg: bool = False):
        """The condition tensors of the three CFG modes of the reference's generate (lm.py:486-511):
        default      conditions + null conditions -> ONE batch [cond; uncond]                       (dict)
        double CFG   conditions + wav-only conditions + null conditions -> [cond; wav; uncond]       (dict, 3B rows)
        two_step_cfg conditions and null conditions encoded SEPARATELY, each with its own padding    (tuple of dicts)"""
        null_conditions = ClassifierFreeGuidanceDropout(p=1.0)(conditions)
        provider = self.condition_provider
        if cfg_coef_beta is not None:
            allc = conditions + _drop_description_condition(conditions) + null_conditions
            return provider(provider.tokenize(allc))
        if two_step_cfg:
            return (provider(provider.tokenize(conditions)), provider(provider.tokenize(null_conditions)))
        return provider(provider.tokenize(conditions + null_conditions))

    def _fuse_two_step(self, cond: ConditionTensors, null: ConditionTensors):
        """two_step_cfg on the device: the reference runs the conditional and the unconditional forward one after the
        other, each with its own condition tensors and its own streaming state (lm.py:377-387).  Rows are independent,
        so the two passes are the two row groups of ONE step; what must be kept apart is the cross-attention source
        length of each pass (padding is not masked in cross-attention: SURVEY.md section 7) -> per-row lengths."""
        p_c, x_c = self.fuser.fuse(cond)
        p_n, x_n = self.fuser.fuse(null)
        cross_src, lens = None, None
        if x_c is not None or x_n is not None:
            assert x_c is not None and x_n is not None
            Lc, Ln = x_c.shape[1], x_n.shape[1]
            L = max(Lc, Ln)
            pad = lambda t: torch.nn.functional.pad(t.float(), (0, 0, 0, L - t.shape[1]))  # noqa: E731
            cross_src = torch.cat([pad(x_c), pad(x_n)], dim=0)
            lens = torch.tensor([Lc] * x_c.shape[0] + [Ln] * x_n.shape[0], dtype=torch.int32)
        prepend, row_off = None, None
        if p_c is not None or p_n is not None:
            # The two passes may prepend different numbers of condition rows (a melody model: 5 text positions against the
            # 1 of an all-null batch), i.e. the same token sits at different transformer positions in the two streams.  The
            # shorter stream is padded ON THE LEFT (zeros, never attended: acmi_lm_state.row_off), so that both reach their
            # first token at the same stream position and one sampler launch serves both row groups.
            d = (p_c if p_c is not None else p_n).shape[2]
            rows_c = p_c.shape[0] if p_c is not None else x_c.shape[0] if x_c is not None else p_n.shape[0]
            rows_n = p_n.shape[0] if p_n is not None else rows_c
            dev_ = (p_c if p_c is not None else p_n).device
            p_c = p_c.float() if p_c is not None else torch.zeros(rows_c, 0, d, device=dev_)
            p_n = p_n.float() if p_n is not None else torch.zeros(rows_n, 0, d, device=dev_)
            Pc, Pn = p_c.shape[1], p_n.shape[1]
            P = max(Pc, Pn)
            lpad = lambda t: torch.nn.functional.pad(t, (0, 0, P - t.shape[1], 0))  # noqa: E731
            prepend = torch.cat([lpad(p_c), lpad(p_n)], dim=0)
            if Pc != Pn:
                row_off = torch.tensor([P - Pc] * p_c.shape[0] + [P - Pn] * p_n.shape[0], dtype=torch.int32)
        return prepend, cross_src, lens, row_off

    def _input_add_table(self, ops_groups, lengths: tp.Sequence[int]) -> tp.Optional[torch.Tensor]:
        """acmi_lm_state.input_add for consecutive reference calls of `lengths` token steps: [rows, sum(lengths), d], rows =
        the row groups' conditions stacked like the other condition tensors (`ops_groups`: one `ConditionFuser.input_ops`
        list per separately encoded group -- one for the batched CFG modes, two for two_step_cfg).  None without ops."""
        if not any(ops_groups):
            return None
        assert all(ops_groups), "every row group needs the same 'sum' / 'input_interpolate' conditions"
        per_call = [torch.cat([ConditionFuser.input_add_rows(ops, T) for ops in ops_groups], dim=0) for T in lengths]
        return torch.cat(per_call, dim=1)

    # ------------------------------------------------------------------------------------- generate
    @_C.exclusive
    @torch.no_grad()
    def generate(self,
                 prompt: tp.Optional[torch.Tensor] = None,
                 conditions: tp.List[ConditioningAttributes] = [],
                 num_samples: tp.Optional[int] = None,
                 max_gen_len: int = 256,
                 use_sampling: bool = True,
                 temp: float = 1.0,
                 top_k: int = 250,
                 top_p: float = 0.0,
                 cfg_coef: tp.Optional[float] = None,
                 cfg_coef_beta: tp.Optional[float] = None,
                 two_step_cfg: tp.Optional[bool] = None,
                 remove_prompts: bool = False,
                 check: bool = False,
                 callback: tp.Optional[tp.Callable[[int, int], None]] = None,
                 condition_tensors: tp.Optional[tp.Union[ConditionTensors, tp.Tuple[ConditionTensors, ConditionTensors]]] = None,
                 seed: tp.Optional[int] = None,
                 return_logits: bool = False,
                 use_graph: bool = True,
                 ) -> torch.Tensor:
        """Same contract as the reference `LMModel.generate` (lm.py:420-587) -> LongTensor [B, K, T], including its three
        classifier-free-guidance modes: one batched forward on `[cond; uncond]` (default), `two_step_cfg` (separate
        conditional / unconditional passes, mixed with the MODEL's cfg_coef like the reference does, lm.py:386) and
        double CFG (`cfg_coef_beta`, rows `[text + wav; wav; null]`, lm.py:362-376).

        Extra keyword-only conveniences (not in the reference): `condition_tensors` (the already encoded conditions, as
        `_cfg_condition_tensors` returns them for the mode in use -- the multi-GPU path broadcasts these), `seed` (device
        Philox stream; default drawn from torch's global generator), `return_logits` (also return the CFG-mixed logits
        of every step, [B, K, steps, card]) and `use_graph`.
        """
        assert not self.training, "generation shouldn't be used in training mode."
        dev = self.device
        two_step = self.two_step_cfg if two_step_cfg is None else two_step_cfg
        cfg_conditions: tp.Any = {}
        if condition_tensors is not None:
            assert not conditions, "Shouldn't pass both conditions and condition_tensors."
            cfg_conditions = condition_tensors
        elif conditions:
            cfg_conditions = self._cfg_condition_tensors(conditions, cfg_coef_beta, bool(two_step))
        two_step = isinstance(cfg_conditions, tuple)
        groups = 1
        if two_step or cfg_conditions:
            groups = 3 if (cfg_coef_beta is not None and not two_step) else 2
        if num_samples is None:
            if prompt is not None:
                num_samples = prompt.shape[0]
            elif conditions:
                num_samples = len(conditions)
            elif two_step:
                num_samples = next(iter(cfg_conditions[0].values()))[0].shape[0]
            elif cfg_conditions:
                num_samples = next(iter(cfg_conditions.values()))[0].shape[0] // groups
            else:
                num_samples = 1
        use_cfg = {1: _C.CFG_NONE, 2: _C.CFG_PAIR, 3: _C.CFG_DOUBLE}[groups]
        # the two-step branch of the reference ignores the cfg_coef argument (lm.py:386)
        coef = self.cfg_coef if (cfg_coef is None or two_step) else cfg_coef

        if prompt is None:
            assert num_samples > 0
            prompt = torch.zeros((num_samples, self.num_codebooks, 0), dtype=torch.long, device=dev)
        prompt = prompt.to(dev)
        B, K, T0 = prompt.shape
        assert K == self.n_q
        start_offset = T0
        assert start_offset < max_gen_len

        pattern = self.pattern_provider.get_pattern(max_gen_len)
        unknown_token = -1
        gen_codes = torch.full((B, K, max_gen_len), unknown_token, dtype=torch.long, device=dev)
        gen_codes[..., :start_offset] = prompt
        gen_sequence, _, mask = pattern.build_pattern_sequence(gen_codes, self.special_token_id)
        start_offset_sequence = pattern.get_first_step_with_timesteps(start_offset)
        assert start_offset_sequence is not None
        S = gen_sequence.shape[-1]

        # fuse conditions: what is prepended to the token stream, what is cross-attended to
        cross_lens, row_off = None, None
        if two_step:
            prepend, cross_src, cross_lens, row_off = self._fuse_two_step(*cfg_conditions)
            ops_groups = [self.fuser.input_ops(c) for c in cfg_conditions]
        else:
            mixed = self.fuser.mixed_order(cfg_conditions)
            prepend, cross_src = self.fuser.fuse(cfg_conditions, allow_mixed=mixed)
            ops_groups = [self.fuser.input_ops(cfg_conditions)]
        # 'sum' / 'input_interpolate' conditions: the reference's first call covers the steps before the first generated one,
        # every later call is one step (lm.py:536-543)
        if not two_step and mixed:
            # such a condition after a 'prepend' one in the provider's order: the reference's loop adds it to the prepended rows
            # of the first call too, and resamples an interpolated condition over prepend + token positions
            prepend, add_first = self.fuser.first_call_inputs(cfg_conditions, start_offset_sequence)
            input_add = torch.cat([add_first, ConditionFuser.input_add_rows(ops_groups[0], 1)], dim=1)
        else:
            input_add = self._input_add_table(ops_groups, [start_offset_sequence, 1])
        if self.has_cross_attention:
            assert cross_src is not None, "this model cross-attends to a condition but none was given"
        else:
            assert cross_src is None, "this model has no cross-attention layers"
        for t in (prepend, cross_src):
            assert t is None or t.shape[0] == B * groups, f"condition rows {t.shape[0]} != {B} samples x {groups} row groups"
        P = 0 if prepend is None else prepend.shape[1]
        Lc = 0 if cross_src is None else cross_src.shape[1]
        Tmax = P + S  # positions g = 0 .. P + S - 2 are ever run

        run = self._prepare_run(B, use_cfg, Tmax, Lc, S)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        if prepend is not None:
            prepend = prepend.to(device=dev, dtype=torch.float32).contiguous()
        state = self._make_state(run, B, use_cfg, Tmax, Lc, S, prepend, return_logits, use_sampling, temp, top_k,
                                 top_p, coef, seed, cfg_coef_beta=0.0 if cfg_coef_beta is None else cfg_coef_beta,
                                 cross_lens=cross_lens, row_off=row_off, input_add=input_add)
        desc = self._packed['desc']
        desc.pos_table = run['pos_table'].data_ptr()
        run['gen_sequence'].copy_(gen_sequence)
        run['seq_mask'].copy_(mask.to(torch.uint8))
        run['pos'].zero_()
        if cross_src is not None:
            cross_src = cross_src.to(device=dev, dtype=torch.float32).contiguous()
            self._project_cross_kv(run, cross_src)
            # Rows whose source is all zero (null conditions: the `uncond` half of the CFG batch, conditioners.py:492-506)
            # have K = V = 0 (no in_proj bias, checked at pack time): their cross-attention block adds exactly 0.  When
            # they sit at the tail of the batch the attention launch skips them; `att` must then read as zeros there.
            live = (cross_src != 0).flatten(1).any(dim=1).nonzero()
            n_live = int(live.max()) + 1 if live.numel() else 0
            if self._packed.get('cross_kv_bias', False):
                n_live = run['Beff']     # k / v biases: a null source no longer means K = V = 0
            state.cross_active_rows = n_live if 0 < n_live < run['Beff'] else 0
            run['att'].zero_()
            # the decode step's cross-attention in its score-folded form where the geometry allows (one launch less per layer)
            state.xs_rows = self._cross_fold_rows(run, n_live, Lc, two_step)
            if state.xs_rows > 0:
                self._build_cross_fold(run, state.xs_rows, Lc)

        # ---- prefill: prepended condition rows and prompt steps, PREFILL_CHUNK positions per call, no sampling
        self._set_first_call(state, P + start_offset_sequence)
        self._prefill(desc, state, P + start_offset_sequence - 1)

        # ---- decode: one hipGraph replay per position
        self._hand_arm(run)
        n_steps = S - start_offset_sequence
        all_logits = []
        graph = None
        if use_graph and n_steps > 2:
            graph = self._capture(desc, state)
        for i in range(n_steps):
            if graph is not None:
                graph.replay()
            else:
                _C.lm_step(desc, state, _C.STEP_DECODE)
            if return_logits:
                all_logits.append(run['step_logits'].clone())
            if callback is not None:
                callback(1 + i, n_steps)
        gen_sequence = run['gen_sequence'].clone()
        self._hand_check(run)

        if check:
            assert not (gen_sequence == unknown_token).any()
            assert (gen_sequence == torch.where(mask[None, ...].expand(B, -1, -1), gen_sequence,
                                                self.special_token_id)).all()
        out_codes, _, out_mask = pattern.revert_pattern_sequence(gen_sequence, special_token=unknown_token)
        out_start_offset = start_offset if remove_prompts else 0
        out_codes = out_codes[..., out_start_offset:max_gen_len]
        if check:
            assert (out_mask[..., :max_gen_len] == 1).all()
            assert (out_codes >= 0).all() and (out_codes <= self.card).all()
        if return_logits:
            return out_codes, torch.stack(all_logits, dim=2)
        return out_codes

    def _capture(self, desc, state):
        """Capture one decode position into a hipGraph (torch.cuda.CUDAGraph is only the capture/replay
        plumbing; every node is an acmi kernel).  Device-side position counter => replayable."""
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            g.capture_begin()   # (global capture mode: one generate at a time per process -- see DESIGN.md section 7.1, two-stream lab)
            try:
                _C.lm_step(desc, state, _C.STEP_DECODE)
            finally:
                g.capture_end()
        torch.cuda.current_stream().wait_stream(s)
        self._graph_keepalive = (g, desc, state)
        return g

    PREFILL_CHUNK = 8   # consecutive positions per prefill call (activation buffers hold Beff * PREFILL_CHUNK rows)

    def _set_first_call(self, state, n_first: int):
        """The reference's first streaming forward covers `n_first` positions at once; with rotary positions AND a
        past_context shorter than that, its later positions lag by the keys that call dropped (acmi_lm_state.rope_first /
        rope_shift; oracle/lm.py, transformer_forward)."""
        state.rope_first, state.rope_shift = 0, 0
        if self.past_context and self.positional_embedding != 'sin' and n_first > self.past_context:
            state.rope_first, state.rope_shift = int(n_first), int(n_first - self.past_context)

    PREFILL_MAX_POSITIONS = 4096   # one MFMA-tiled call covers at most this many positions (activations: Beff x this rows)

    def _big_prefill_ok(self, n_positions: int) -> bool:
        """The MFMA-tiled prefill (acmi_lm_step with pf_xn: ONE forward over all prompt / prefix positions, like the
        reference's first streaming call, lm.py:540-543) serves a prefill that starts an empty stream when the geometry
        fits its tiles; ACMI_PREFILL=chunk forces the decode kernels on PREFILL_CHUNK positions per call (A/B, tests)."""
        import os
        mode = os.environ.get('ACMI_PREFILL', '')
        if mode == 'chunk':
            return False
        kt = _C._tile_params(self.weight_dtype)[1]
        hd = self.dim // self.num_heads
        rows = (self._run['Beff'] if self._run else 1) * (-(-n_positions // 16) * 16)
        fits = (self.dim % (2 * kt) == 0 and self.ffn_dim % (2 * kt) == 0 and hd in (8, 16, 32, 64, 128)
                and self.kv_dtype == self.weight_dtype and n_positions <= self.PREFILL_MAX_POSITIONS
                and rows <= 65535)   # one launch row per grid.y entry of the cross-attention kernel
        return fits and n_positions >= (2 if mode == 'big' else 4)

    def _prefill_big(self, desc, state, run, n_positions: int):
        dev = self.device
        Beff, d, H = run['Beff'], self.dim, self.num_heads
        npp = -(-n_positions // 16) * 16
        tcap = -(-n_positions // 32) * 32
        rows = Beff * npp
        pf = run.get('pf')
        if pf is None or pf['rows'] < rows or pf['tcap'] < tcap:
            f32 = dict(device=dev, dtype=torch.float32)
            rows_a, tcap_a = max(rows, pf['rows'] if pf else 0), max(tcap, pf['tcap'] if pf else 0)
            run['pf'] = pf = None   # release the previous scratch first
            pf = {'rows': rows_a, 'tcap': tcap_a,
                  'x': torch.zeros(rows_a, d, **f32), 'q': torch.zeros(rows_a, d, **f32),
                  'stats': torch.zeros(rows_a, 2, **f32),
                  'xn': _C.tiled_activation_buffer(rows_a, d, self.weight_dtype, dev),
                  'att': _C.tiled_activation_buffer(rows_a, d, self.weight_dtype, dev),
                  'hidden': _C.tiled_activation_buffer(rows_a, self.ffn_dim, self.weight_dtype, dev),
                  'vt': torch.zeros(Beff * H * (d // H) * tcap_a, device=dev, dtype=self.kv_dtype)}
            run['pf'] = pf
        saved = (state.x, state.q, state.stats, state.att, state.hidden)
        state.x, state.q, state.stats = pf['x'].data_ptr(), pf['q'].data_ptr(), pf['stats'].data_ptr()
        state.att, state.hidden = pf['att'].data_ptr(), pf['hidden'].data_ptr()
        state.pf_xn, state.pf_vt, state.pf_tcap = pf['xn'].data_ptr(), pf['vt'].data_ptr(), tcap
        state.n_pos = n_positions
        try:
            _C.lm_step(desc, state, _C.STEP_PREFILL)
        finally:
            state.x, state.q, state.stats, state.att, state.hidden = saved
            state.pf_xn, state.pf_vt, state.pf_tcap = None, None, 0
            state.n_pos = 1

    def _prefill(self, desc, state, n_positions: int, start: int = 0):
        """Run `n_positions` input-only positions (prepended conditions, prompt tokens) beginning at stream position `start`
        (= what the device position counter holds; every caller today has just zeroed it): one MFMA-tiled forward over all
        of them when the stream is EMPTY and the geometry allows (`_big_prefill_ok`) -- lm_prefill_big sizes its time-minor V
        scratch for positions [0, n_positions) only, so a non-empty stream must take the other path -- else the same kernels
        as a decode position, several consecutive positions per call as extra rows (acmi_lm_state.n_pos)."""
        if n_positions > 0 and start == 0 and not state.row_off and self._big_prefill_ok(n_positions):
            return self._prefill_big(desc, state, self._run, n_positions)
        done = 0
        while done < n_positions:
            state.n_pos = min(self.PREFILL_CHUNK, n_positions - done)
            _C.lm_step(desc, state, _C.STEP_PREFILL)
            done += state.n_pos
        state.n_pos = 1

    # ------------------------------------------------------------------------------------- streaming protocol
    # `StreamingModule` of the reference (audiocraft/modules/streaming.py:20-119): `with lm.streaming():` makes successive
    # `forward` calls continue one stream (KV caches, position offsets); the state is a dict of tensors with the
    # reference's key names.  Here the state lives in the device buffers `acmi_lm_step` works on; the dict entries are
    # VIEWS of them (the caches are append-only, so a state taken earlier stays valid and `set_streaming_state` of it
    # rewinds the stream).
    streaming_capacity = 2048   # positions a stream can hold (KV cache rows allocated at the first streaming forward)

    @contextmanager
    def streaming(self):
        """Context manager to enter streaming mode; the streaming state is reset on exit (streaming.py:59-67)."""
        self._is_streaming = True
        try:
            yield
        finally:
            self._is_streaming = False
            self.reset_streaming()

    def reset_streaming(self):
        self._stream = None

    def _stream_begin(self, B: int, condition_tensors: ConditionTensors, first_len: int = 1):
        dev = self.device
        mixed = self.fuser.mixed_order(condition_tensors)
        prepend, cross_src = self.fuser.fuse(condition_tensors, allow_mixed=mixed)
        add_first = None
        if mixed:   # a 'sum' / 'input_interpolate' condition after a 'prepend' one: the first call's length shapes the prepended rows
            prepend, add_first = self.fuser.first_call_inputs(condition_tensors, first_len)
        P = 0 if prepend is None else prepend.shape[1]
        Lc = 0 if cross_src is None else cross_src.shape[1]
        cap = self.streaming_capacity
        run = self._prepare_run(B, _C.CFG_NONE, P + cap + 1, Lc, cap + 1)
        if prepend is not None:
            prepend = prepend.to(device=dev, dtype=torch.float32).contiguous()
        ops = self.fuser.input_ops(condition_tensors)
        # 'sum' / 'input_interpolate': one entry per stream step, filled call by call (each call's length is what the
        # reference's fuser resamples to: _streaming_forward)
        input_add = torch.zeros(B, cap + 1, self.dim, device=dev) if ops else None
        state = self._make_state(run, B, _C.CFG_NONE, P + cap + 1, Lc, cap + 1, prepend, True, False, 1.0, 0, 0.0, 1.0, 0,
                                 input_add=input_add)
        desc = self._packed['desc']
        desc.pos_table = run['pos_table'].data_ptr()
        run['gen_sequence'].fill_(-1)
        run['seq_mask'].fill_(1)
        run['pos'].zero_()
        if cross_src is not None:
            self._project_cross_kv(run, cross_src.to(device=dev, dtype=torch.float32).contiguous())
        self._prefill(desc, state, P)   # the fuser prepends on the first call only (conditioners.py:1722-1741)
        self._stream = {'run': run, 'state': state, 'desc': desc, 'prepend': prepend, 'P': P, 'steps': 0, 'B': B, 'ops': ops,
                        'add_first': add_first}
        return self._stream

    def _streaming_forward(self, sequence: torch.Tensor, condition_tensors: ConditionTensors) -> torch.Tensor:
        B, K, S = sequence.shape
        st = getattr(self, '_stream', None)
        if st is None:
            st = self._stream_begin(B, condition_tensors, S)
            self._set_first_call(st['state'], st['P'] + S)   # the first call's length fixes the rotary offsets
            st['first_len'] = st['P'] + S
        assert st['B'] == B, "the batch size of a stream cannot change"
        assert st['steps'] + S <= self.streaming_capacity, "stream longer than LMModel.streaming_capacity"
        run, off = st['run'], st['steps']
        run['gen_sequence'][:, :, off:off + S] = sequence.to(self.device)
        if st.get('ops'):
            add = st.pop('add_first', None) if off == 0 else None    # (mixed order: what the first call adds to its token rows)
            run['input_add'][:, off:off + S] = (add if add is not None else ConditionFuser.input_add_rows(st['ops'], S)).to(self.device)
        outs = []
        if off == 0:
            self._hand_arm(run)
        for _ in range(S):
            _C.lm_step(st['desc'], st['state'], _C.STEP_DECODE)
            outs.append(run['step_logits'].clone())
        self._hand_check(run)
        st['steps'] = off + S
        return torch.stack(outs, dim=2)

    def get_streaming_state(self) -> tp.Dict[str, torch.Tensor]:
        """Key names of the reference (streaming.py:75-86): `transformer.offsets`,
        `fuser.offsets`, `transformer.layers.{i}.self_attn.past_keys|past_values|offset`.  K / V: [B, H, t, hd] views."""
        st = getattr(self, '_stream', None)
        if st is None:
            return {}
        run, B = st['run'], st['B']
        t = st['P'] + st['steps']
        dev = self.device
        state = {'transformer.offsets': torch.full((B,), t, dtype=torch.long, device=dev),
                 'fuser.offsets': torch.full((B,), st['steps'], dtype=torch.long, device=dev)}
        # `offset` of the reference = keys dropped by past_context SINCE the first call (transformer.py:289-297: the entry is
        # created as tensor(0) by the first call, whatever that call dropped, and incremented by every later trim); the
        # rotary position of the next step is offset + stored keys (transformer.py:300-313).  The caches here are never
        # trimmed (the window is applied by the attention kernel), so K / V are returned at full length.
        pc = self.past_context
        dropped = 0 if not pc else max(0, t - pc) - max(0, st.get('first_len', t) - pc)
        for li in range(self.num_layers):
            pre = f'transformer.layers.{li}.self_attn.'
            # kv_repeat: the caches hold every shared head once per query head; the state lists the H / kv_repeat stored heads
            state[pre + 'past_keys'] = run['k'][li][:, ::self.kv_repeat, :t]
            state[pre + 'past_values'] = run['v'][li][:, ::self.kv_repeat, :t]
            state[pre + 'offset'] = torch.tensor(dropped, dtype=torch.long, device=dev)
        return state

    def set_streaming_state(self, state: tp.Dict[str, torch.Tensor]):
        """Inverse of get_streaming_state (streaming.py:88-104): rewinds / restores the stream.  K / V tensors that are
        not views of this model's own caches are copied in."""
        st = getattr(self, '_stream', None)
        if not state:
            self._stream = None
            return
        assert st is not None, "set_streaming_state: no stream to restore into (run a streaming forward first)"
        run = st['run']
        t = int(state['transformer.offsets'][0])
        assert st['P'] <= t <= st['P'] + self.streaming_capacity
        known = {'transformer.offsets', 'fuser.offsets'}
        pc = self.past_context
        for li in range(self.num_layers):
            pre = f'transformer.layers.{li}.self_attn.'
            for name, cache in (('past_keys', run['k'][li]), ('past_values', run['v'][li])):
                src = state[pre + name]
                # full length (this class's own states) or trimmed to the last past_context keys (a reference-format state)
                n = src.shape[2]
                assert n == t or (pc and n == min(t, pc)), (src.shape, t, pc)
                if self.kv_repeat > 1:
                    cache[:, :, t - n:t].copy_(src.repeat_interleave(self.kv_repeat, dim=1) if src.shape[1] != cache.shape[1] else src)
                elif src.data_ptr() != cache.data_ptr():
                    cache[:, :, t - n:t].copy_(src)
                known.add(pre + name)
            known.add(pre + 'offset')
        assert set(state.keys()) <= known, sorted(set(state.keys()) - known)
        # rotary lag: the reference's next rotary position is offset + stored keys (stored = min(t, past_context))
        off_key = 'transformer.layers.0.self_attn.offset'
        if pc and self.positional_embedding != 'sin' and off_key in state:
            lag = t - (int(state[off_key]) + min(t, pc))
            cst = st['state']
            if lag > 0:
                cst.rope_first, cst.rope_shift = (cst.rope_first if 0 < cst.rope_first <= t else t), lag
            else:
                cst.rope_first, cst.rope_shift = 0, 0
            st['first_len'] = lag + pc if lag > 0 else min(st.get('first_len', t), pc)
        st['steps'] = t - st['P']
        run['pos'][:2] = torch.tensor([t, 0], dtype=torch.int32, device=run['pos'].device)
        run['gen_sequence'][:, :, st['steps']:].fill_(-1)

    # ------------------------------------------------------------------------------------- teacher forcing
    @_C.exclusive
    @torch.no_grad()
    def forward_steps(self, sequence: torch.Tensor, condition_tensors: ConditionTensors) -> torch.Tensor:
        """Teacher-forced streaming forward for parity tests: runs the pattern `sequence` [B, K, S]
        through the decode step one position at a time (no CFG mixing: rows are taken as given) and
        returns logits [B, K, S, card] like `LMModel.forward` (lm.py:221-268)."""
        dev = self.device
        B, K, S = sequence.shape
        mixed = self.fuser.mixed_order(condition_tensors)
        prepend, cross_src = self.fuser.fuse(condition_tensors, allow_mixed=mixed)
        # the reference's forward is ONE call of S steps: that is the length its fuser interpolates a condition to
        if mixed:   # ... and, for a condition that follows a 'prepend' one in the provider's order, the prepended rows count too
            prepend, input_add = self.fuser.first_call_inputs(condition_tensors, S)
        else:
            input_add = self._input_add_table([self.fuser.input_ops(condition_tensors)], [S])
        P = 0 if prepend is None else prepend.shape[1]
        Lc = 0 if cross_src is None else cross_src.shape[1]
        run = self._prepare_run(B, _C.CFG_NONE, P + S + 1, Lc, S + 1)
        if prepend is not None:
            prepend = prepend.to(device=dev, dtype=torch.float32).contiguous()
        state = self._make_state(run, B, _C.CFG_NONE, P + S + 1, Lc, S + 1, prepend, True, False, 1.0, 0, 0.0, 1.0, 0,
                                 input_add=input_add)
        desc = self._packed['desc']
        desc.pos_table = run['pos_table'].data_ptr()
        seq = torch.full((B, K, S + 1), -1, dtype=torch.long, device=dev)
        seq[..., :S] = sequence.to(dev)
        run['gen_sequence'].copy_(seq)
        run['seq_mask'].fill_(1)
        run['pos'].zero_()
        if cross_src is not None:
            self._project_cross_kv(run, cross_src.to(device=dev, dtype=torch.float32).contiguous())
        self._prefill(desc, state, P)
        outs = torch.empty(B, K, S, self.card, device=dev, dtype=torch.float32)
        self._hand_arm(run)
        graph = self._capture(desc, state) if S > 2 else None   # one captured position, replayed S times
        for i in range(S):
            if graph is not None:
                graph.replay()
            else:
                _C.lm_step(desc, state, _C.STEP_DECODE)
            outs[:, :, i].copy_(run['step_logits'])  # the sampler only writes slots still at -1
        self._hand_check(run)
        return outs

    # ------------------------------------------------------------------------------------- reference forward API
    @_C.exclusive
    @torch.no_grad()
    def forward(self, sequence: torch.Tensor, conditions: tp.List[ConditioningAttributes] = [],
                condition_tensors: tp.Optional[ConditionTensors] = None, stage: int = -1) -> torch.Tensor:
        """`LMModel.forward` of the reference (lm.py:221-268): pattern sequence [B, K, S] -> logits
        [B, K, S, card].  Evaluated causally position by position through the decode kernels
        (streaming == batch); conditions are encoded here when `condition_tensors` is not given
        (no CFG / attribute dropout: inference only)."""
        if condition_tensors is None:
            tokenized = self.condition_provider.tokenize(conditions)
            condition_tensors = self.condition_provider(tokenized)
        else:
            assert not conditions, "Shouldn't pass both conditions and condition_tensors."
        if getattr(self, '_is_streaming', False):
            return self._streaming_forward(sequence, condition_tensors)
        return self.forward_steps(sequence, condition_tensors)

    @_C.exclusive
    @torch.no_grad()
    def compute_predictions(self, codes: torch.Tensor, conditions: tp.List[ConditioningAttributes] = [],
                            condition_tensors: tp.Optional[ConditionTensors] = None, stage: int = -1,
                            keep_only_valid_steps: bool = True):
        """`LMModel.compute_predictions` (lm.py:270-321): codes [B, K, T] -> (logits [B, K, T, card]
        re-aligned with the codes, mask [B, K, T] of valid positions)."""
        B, K, T = codes.shape
        codes = codes.contiguous()
        pattern = self.pattern_provider.get_pattern(T)
        sequence_codes, _, _ = pattern.build_pattern_sequence(codes, self.special_token_id,
                                                              keep_only_valid_steps=keep_only_valid_steps)
        logits = self.forward(sequence_codes, conditions, condition_tensors, stage=stage)  # [B, K, S, card]
        logits = logits.permute(0, 3, 1, 2)                                                 # [B, card, K, S]
        logits, _, logits_mask = pattern.revert_pattern_logits(logits, float('nan'),
                                                               keep_only_valid_steps=keep_only_valid_steps)
        logits = logits.permute(0, 2, 3, 1)                                                 # [B, K, T, card]
        return LMOutput(logits, logits_mask[None, :, :].expand(B, -1, -1))


class LMOutput(tp.NamedTuple):
    logits: torch.Tensor  # [B, K, T, card], already re-aligned with the input codes
    mask: torch.Tensor    # [B, K, T]
