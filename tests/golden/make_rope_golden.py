"""Golden fixtures for the StreamingTransformer options no MusicGen release uses but the decode step supports
(BASELINE.json north_star "RoPE"): rotary positions (+ xPos decay), a bounded receptive
field (past_context) and LayerScale, generated from the unmodified reference.

The reference is run with custom=True / memory_efficient=False: its streaming rotary offset reads
`past_keys.shape[1]` (transformer.py:305), which is the time axis only in the custom attention's "b t h d" layout --
with the memory-efficient torch backend ("b h t d") that is the number of heads, and streaming no longer equals the
full forward.  The custom path is the one the reference's own tests pin (tests/modules/test_rope.py:66-121:
streaming == full forward, with and without past_context), so it is the semantics restated and implemented here.

  lm_rope.npz      positional_embedding='rope', xpos=True, past_context=6, layer_scale=0.3, positional_scale=0.8
  lm_sin_rope.npz  positional_embedding='sin_rope' (sinusoidal embedding AND rotary q / k), everything else default
  lm_rope_two_step_prepend.npz  positional_embedding='rope' on a PREPEND fuser with two_step_cfg=True: the conditional pass
                   prepends 5 condition positions, the unconditional pass 1, each with its own streaming state (lm.py:378-390),
                   so the rotary position of a token differs between the two streams; with and without a prompt

Run in the build container only:   python tests/golden/make_rope_golden.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (installs the import stubs, defines the tiny-model builders)
from make_golden import (ConditionFuser, ConditioningAttributes, ConditioningProvider, DelayedPatternProvider,  # noqa: E402
                         LMModel)


def build(cfg, extra):
    torch.manual_seed(2000 + cfg['seed'])
    cond = {'description': mg.SynthText(cfg['cond_dim'], cfg['dim'], cfg['Lc'])}
    torch.manual_seed(cfg['seed'])
    lm = LMModel(DelayedPatternProvider(cfg['n_q'], delays=cfg['delays']), ConditioningProvider(cond),
                 ConditionFuser({'cross': ['description'], 'prepend': [], 'sum': [], 'input_interpolate': []}),
                 n_q=cfg['n_q'], card=cfg['card'], dim=cfg['dim'], num_heads=cfg['num_heads'],
                 hidden_scale=cfg['hidden_scale'], norm='layer_norm', norm_first=True, bias_proj=False,
                 weight_init='gaussian', depthwise_init='current', zero_bias_init=True, cfg_coef=cfg['cfg_coef'],
                 num_layers=cfg['num_layers'], dropout=0., activation='gelu', bias_ff=False, bias_attn=False,
                 causal=True, custom=True, memory_efficient=False, attention_as_float32=False,
                 cross_attention=True, **extra).eval()
    with torch.no_grad():
        for k, p in lm.named_parameters():
            if '.norm' in k or k.startswith('out_norm'):
                p.add_(0.1 * torch.randn_like(p))
            if 'layer_scale' in k:                 # a constant gain would hide a channel mix-up
                p.mul_(1.0 + 0.5 * torch.rand_like(p))
    return lm


def make(name, extra):
    cfg = dict(dim=32, num_heads=4, num_layers=2, hidden_scale=4, n_q=4, card=32, cross_attention=True,
               delays=[0, 1, 2, 3], cfg_coef=3.0, seed=11, cond_dim=8, Lc=5, **extra)
    lm = build(cfg, extra)
    conds = [ConditioningAttributes(text={'description': f'p{i}'}) for i in range(3)]
    arrays = {}
    tokens, rec, ct = mg.run_lm(lm, conds, None, 14, use_sampling=False)
    arrays['cross_src'] = ct['description'][0]
    arrays['greedy_tokens'] = tokens
    arrays['greedy_step_logits'] = torch.stack([r[:, :, -1] for r in rec], dim=2)
    g = torch.Generator().manual_seed(6)
    prompt = torch.randint(0, cfg['card'], (3, cfg['n_q'], 9), generator=g)   # longer than past_context: windowed prefill
    tokens, rec, _ = mg.run_lm(lm, conds, prompt, 14, use_sampling=False)
    arrays['prompt'] = prompt
    arrays['cont_tokens'] = tokens
    arrays['cont_first_logits'] = rec[0]
    seq = torch.randint(0, cfg['card'] + 1, (6, cfg['n_q'], 12), generator=g)
    with torch.no_grad():
        arrays['tf_sequence'] = seq
        arrays['tf_logits'] = lm(seq, [], ct)                                   # full (non streaming) forward
    mg.save(name, cfg, lm.state_dict(), **arrays)
    print(name, 'tokens', tuple(tokens.shape), 'keys', len(lm.state_dict()))


def make_two_step_prepend_rope():
    from make_cfg_modes_golden import SynthTextRagged
    cfg = dict(dim=32, num_heads=4, num_layers=2, hidden_scale=4, n_q=4, card=32, cross_attention=False,
               delays=[0, 1, 2, 3], cfg_coef=3.0, seed=13, cond_dim=8, Lc=5, positional_embedding='rope', positional_scale=0.7)
    torch.manual_seed(2000 + cfg['seed'])
    cond = {'description': SynthTextRagged(cfg['cond_dim'], cfg['dim'], cfg['Lc'])}
    torch.manual_seed(cfg['seed'])
    lm = LMModel(DelayedPatternProvider(cfg['n_q'], delays=cfg['delays']), ConditioningProvider(cond),
                 ConditionFuser({'cross': [], 'prepend': ['description'], 'sum': [], 'input_interpolate': []}),
                 n_q=cfg['n_q'], card=cfg['card'], dim=cfg['dim'], num_heads=cfg['num_heads'],
                 hidden_scale=cfg['hidden_scale'], norm='layer_norm', norm_first=True, bias_proj=False,
                 weight_init='gaussian', depthwise_init='current', zero_bias_init=True, cfg_coef=cfg['cfg_coef'],
                 num_layers=cfg['num_layers'], dropout=0., activation='gelu', bias_ff=False, bias_attn=False,
                 causal=True, custom=True, memory_efficient=False, attention_as_float32=False,
                 cross_attention=False, positional_embedding='rope', positional_scale=cfg['positional_scale']).eval()
    with torch.no_grad():
        for k, p in lm.named_parameters():
            if '.norm' in k or k.startswith('out_norm'):
                p.add_(0.1 * torch.randn_like(p))
    conds = [ConditioningAttributes(text={'description': f'r{i}'}) for i in range(3)]
    rec = []
    h = lm.register_forward_hook(lambda mod, inp, out: rec.append(out.detach().clone()))
    tokens = lm.generate(None, conds, max_gen_len=11, use_sampling=False, two_step_cfg=True)
    h.remove()
    null = mg.ClassifierFreeGuidanceDropout(p=1.0)(conds)
    ct = lm.condition_provider(lm.condition_provider.tokenize(conds))
    nt = lm.condition_provider(lm.condition_provider.tokenize(null))
    assert ct['description'][0].shape[1] == 5 and nt['description'][0].shape[1] == 1
    gp = torch.Generator().manual_seed(19)
    prompt = torch.randint(0, cfg['card'], (3, cfg['n_q'], 4), generator=gp)
    tokens_p = lm.generate(prompt, conds, max_gen_len=12, use_sampling=False, two_step_cfg=True)
    mg.save('lm_rope_two_step_prepend', cfg, lm.state_dict(), prepend_src=ct['description'][0], null_prepend_src=nt['description'][0],
            greedy_tokens=tokens, cond_step_logits=torch.stack([r[:, :, -1] for r in rec[0::2]], dim=2),
            uncond_step_logits=torch.stack([r[:, :, -1] for r in rec[1::2]], dim=2), prompt=prompt, greedy_tokens_prompt=tokens_p)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'two_step':
        make_two_step_prepend_rope()
        sys.exit(0)
    make('lm_rope', dict(positional_embedding='rope', xpos=True, past_context=6, layer_scale=0.3, positional_scale=0.8))
    make('lm_sin_rope', dict(positional_embedding='sin_rope'))
    make_two_step_prepend_rope()
