"""Golden fixtures for the transformer / fuser options of the reference that no MusicGen release switches on
(config/model/lm/default.yaml:43-46, conditioners.py:1690, 1733-1737), generated from the unmodified reference:

  lm_kv_repeat.npz   kv_repeat = 2 (transformer.py:196-200, 373-386, 398-400: the in-projection emits H / 2 key / value
                     heads, each shared by two consecutive query heads), attention and feed-forward biases on
  lm_qk_ln.npz       qk_layer_norm + qk_layer_norm_cross (transformer.py:216-222, 358-360, 388-392: LayerNorm over the
                     full model dimension of the projected queries and keys, before the head split / rotary positions)
  lm_fuser_sum.npz   fuser {'sum': ['genre'], 'input_interpolate': ['curve'], 'cross': ['description']} with
                     cross_attention_pos_emb (conditioners.py:1733-1737, 1750-1757): a one-frame condition added to
                     every input step, a 5-frame condition nearest-resampled to each call's length, a sinusoidal
                     embedding added to the cross-attention source

The reference runs with custom=True (kv_repeat / qk_layer_norm assert the custom attention, transformer.py:210-219).

  lm_fuser_prepend_sum.npz  fuser {'prepend': ['description'], 'sum': ['genre'], 'input_interpolate': ['curve']}: in the provider's
                     dict order the prepend comes FIRST, so the reference adds the later conditions to the prepended rows as
                     well (its loop works on the concatenated input, conditioners.py:1730-1748): the one-frame condition to all
                     P + T rows of the first call, the 5-frame condition resampled over P + T positions

  lm_post_norm.npz   norm_first=False -- the constructor default of LMModel / StreamingTransformer (lm.py:147, default.yaml:21), which
                     every release overrides: x = norm1(x + sa(x)); x = norm_cross(x + ca(src)) with the LAYER INPUT `src` as the
                     query source (transformer.py:567-573); x = norm2(x + ff(x)); no out_norm (lm.py:171-173).  Biases on

  lm_patterns.npz    one model generating through the other codebook patterns of the reference's builder
                     (codebooks_patterns.py:359-552): parallel, unroll (partly flattened, delayed), coarse_first, musiclm,
                     delay with flatten_first / empty_initial -- greedy tokens and per-step logits of each

Run in the build container only:   python tests/golden/make_options_golden.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (installs the import stubs, defines the tiny-model builders)
from make_golden import (ConditionFuser, ConditioningAttributes, ConditioningProvider, DelayedPatternProvider,  # noqa: E402
                         LMModel, TextConditioner)


class SynthFrames(TextConditioner):
    """A text-keyed condition of `frames` frames (1 for a 'sum' condition): seeded randn through the real output_proj."""
    def __init__(self, dim, output_dim, frames, seed):
        super().__init__(dim, output_dim)
        self.frames, self.seed = frames, seed

    def tokenize(self, x):
        return x

    def forward(self, x):
        g = torch.Generator().manual_seed(self.seed)
        B = len(x)
        mask = torch.tensor([[1] * self.frames if xi is not None else [0] * self.frames for xi in x])
        e = torch.randn(B, self.frames, self.dim, generator=g)
        return self.output_proj(e) * mask.unsqueeze(-1), mask


def build(cfg, conditioners, fuse, extra, fuser_kw=None, cross_attention=True, norm_first=True):
    torch.manual_seed(cfg['seed'])
    lm = LMModel(DelayedPatternProvider(cfg['n_q'], delays=cfg['delays']), ConditioningProvider(conditioners),
                 ConditionFuser(fuse, **(fuser_kw or {})),
                 n_q=cfg['n_q'], card=cfg['card'], dim=cfg['dim'], num_heads=cfg['num_heads'],
                 hidden_scale=cfg['hidden_scale'], norm='layer_norm', norm_first=norm_first, bias_proj=cfg.get('bias_proj', False),
                 weight_init='gaussian', depthwise_init='current', zero_bias_init=True, cfg_coef=cfg['cfg_coef'],
                 num_layers=cfg['num_layers'], dropout=0., activation='gelu', bias_ff=cfg.get('bias_ff', False),
                 bias_attn=cfg.get('bias_attn', False), causal=True, custom=True, memory_efficient=False,
                 attention_as_float32=False, cross_attention=cross_attention, **extra).eval()
    with torch.no_grad():
        for k, p in lm.named_parameters():
            if '.norm' in k or k.startswith('out_norm') or '_layer_norm.' in k:
                p.add_(0.1 * torch.randn_like(p))
            if k.endswith('.bias') and 'norm' not in k and 'output_proj' not in k:   # zero_bias_init: make the biases count
                p.add_(0.05 * torch.randn_like(p))
    return lm


def run(name, cfg, lm, conds, extra_arrays=None):
    arrays = dict(extra_arrays or {})
    tokens, rec, ct = mg.run_lm(lm, conds, None, 12, use_sampling=False)
    for key, (t, _m) in ct.items():
        arrays['cond_' + key] = t
    arrays['greedy_tokens'] = tokens
    arrays['greedy_step_logits'] = torch.stack([r[:, :, -1] for r in rec], dim=2)
    g = torch.Generator().manual_seed(6)
    prompt = torch.randint(0, cfg['card'], (3, cfg['n_q'], 4), generator=g)
    tokens, rec, _ = mg.run_lm(lm, conds, prompt, 11, use_sampling=False)
    arrays['prompt'] = prompt
    arrays['cont_tokens'] = tokens
    arrays['cont_first_logits'] = rec[0]
    seq = torch.randint(0, cfg['card'] + 1, (6, cfg['n_q'], 10), generator=g)
    with torch.no_grad():
        arrays['tf_sequence'] = seq
        arrays['tf_logits'] = lm(seq, [], ct)                                   # full (non streaming) forward
    mg.save(name, cfg, lm.state_dict(), **arrays)
    print(name, 'tokens', tuple(tokens.shape), 'keys', len(lm.state_dict()))


BASE = dict(dim=32, num_heads=4, num_layers=2, hidden_scale=4, n_q=4, card=32, cross_attention=True,
            delays=[0, 1, 2, 3], cfg_coef=3.0, cond_dim=8, Lc=5)
CROSS_ONLY = {'cross': ['description'], 'prepend': [], 'sum': [], 'input_interpolate': []}


if __name__ == '__main__':
    conds = [ConditioningAttributes(text={'description': f'p{i}'}) for i in range(3)]
    ONLY = len(sys.argv) > 1 and sys.argv[1] == 'prepend_sum_only'   # regenerate just that fixture
    if len(sys.argv) > 1 and sys.argv[1] == 'post_norm_only':
        cfg = dict(BASE, seed=25, bias_attn=True, bias_ff=True, bias_proj=True, norm_first=False)
        torch.manual_seed(3004)
        text = {'description': mg.SynthText(cfg['cond_dim'], cfg['dim'], cfg['Lc'])}
        lm = build(cfg, text, CROSS_ONLY, {}, norm_first=False)
        assert lm.out_norm is None and not lm.transformer.layers[0].norm_first
        run('lm_post_norm', cfg, lm, conds)
        sys.exit(0)

    extra = dict(kv_repeat=2)
    cfg = dict(BASE, seed=21, bias_attn=True, bias_ff=True, **extra)
    torch.manual_seed(3000)
    text = {'description': mg.SynthText(cfg['cond_dim'], cfg['dim'], cfg['Lc'])}
    if not ONLY:
        run('lm_kv_repeat', cfg, build(cfg, text, CROSS_ONLY, extra), conds)

    extra = dict(qk_layer_norm=True, qk_layer_norm_cross=True)
    cfg = dict(BASE, seed=22, bias_attn=True, **extra)
    torch.manual_seed(3001)
    text = {'description': mg.SynthText(cfg['cond_dim'], cfg['dim'], cfg['Lc'])}
    if not ONLY:
        run('lm_qk_ln', cfg, build(cfg, text, CROSS_ONLY, extra), conds)

    cfg = dict(BASE, seed=23, curve_frames=5, cross_attention_pos_emb=True, cross_attention_pos_emb_scale=0.7)
    torch.manual_seed(3002)
    cds = {'description': mg.SynthText(cfg['cond_dim'], cfg['dim'], cfg['Lc']),
           'genre': SynthFrames(cfg['cond_dim'], cfg['dim'], 1, 77),
           'curve': SynthFrames(cfg['cond_dim'], cfg['dim'], cfg['curve_frames'], 78)}
    fuse = {'cross': ['description'], 'prepend': [], 'sum': ['genre'], 'input_interpolate': ['curve']}
    conds3 = [ConditioningAttributes(text={'description': f'p{i}', 'genre': f'g{i}', 'curve': f'c{i}'}) for i in range(3)]
    if not ONLY:
        lm = build(cfg, cds, fuse, {}, dict(cross_attention_pos_emb=True, cross_attention_pos_emb_scale=0.7))
        run('lm_fuser_sum', cfg, lm, conds3)

    cfg = dict(BASE, seed=24, curve_frames=5, cross_attention=False)
    torch.manual_seed(3003)
    cds = {'description': mg.SynthText(cfg['cond_dim'], cfg['dim'], cfg['Lc']),
           'genre': SynthFrames(cfg['cond_dim'], cfg['dim'], 1, 79),
           'curve': SynthFrames(cfg['cond_dim'], cfg['dim'], cfg['curve_frames'], 80)}
    fuse = {'cross': [], 'prepend': ['description'], 'sum': ['genre'], 'input_interpolate': ['curve']}
    lm = build(cfg, cds, fuse, {}, cross_attention=False)
    assert list(lm.condition_provider.conditioners.keys()) == ['description', 'genre', 'curve']   # the order that matters
    run('lm_fuser_prepend_sum', cfg, lm, conds3)
    if len(sys.argv) > 1 and sys.argv[1] == 'prepend_sum_only':
        sys.exit(0)

    # ---- codebook patterns: the provider is an attribute of the LM, the weights do not depend on it
    import audiocraft.modules.codebooks_patterns as cbp
    PATTERNS = [('parallel', {}), ('unroll', dict(flattening=[0, 1, 1, 2], delays=[0, 0, 0, 1])), ('coarse_first', dict(delays=[0, 1, 1])),
                ('musiclm', dict(group_by=2)), ('delay', dict(delays=[0, 1, 2, 3], flatten_first=2, empty_initial=1))]
    klass = {'parallel': cbp.ParallelPatternProvider, 'unroll': cbp.UnrolledPatternProvider, 'coarse_first': cbp.CoarseFirstPattern,
             'musiclm': cbp.MusicLMPattern, 'delay': cbp.DelayedPatternProvider}
    cfg = dict(BASE, seed=24, patterns=[[n, k] for n, k in PATTERNS])
    torch.manual_seed(3003)
    text = {'description': mg.SynthText(cfg['cond_dim'], cfg['dim'], cfg['Lc'])}
    lm = build(cfg, text, CROSS_ONLY, {})
    arrays = {}
    g = torch.Generator().manual_seed(9)
    prompt = torch.randint(0, cfg['card'], (3, cfg['n_q'], 3), generator=g)
    arrays['prompt'] = prompt
    for i, (name, kw) in enumerate(PATTERNS):
        lm.pattern_provider = klass[name](cfg['n_q'], **kw)
        tokens, rec, ct = mg.run_lm(lm, conds, None, 7, use_sampling=False)
        arrays['cond_description'] = ct['description'][0]
        arrays[f'tokens_{i}'] = tokens
        arrays[f'step_logits_{i}'] = torch.stack([r[:, :, -1] for r in rec], dim=2)
        if name != 'coarse_first':   # (a prompt shorter than the sequence makes coarse_first's first call span every coarse step)
            tokens, _, _ = mg.run_lm(lm, conds, prompt, 7, use_sampling=False)
            arrays[f'cont_tokens_{i}'] = tokens
    mg.save('lm_patterns', cfg, lm.state_dict(), **arrays)
    print('lm_patterns', [n for n, _ in PATTERNS])
