"""Checkpoints in the reference's release format, WRITTEN BY THE REFERENCE (audiocraft/utils/export.py:22-79, imported
unmodified through oracle/refstubs.py), for the loader tests:

  ckpt_ref/text/state_dict.bin               export_lm of the `lm_text` golden model (its greedy tokens are in lm_text.npz);
                                             xp.cfg composed from the reference's own YAML files (config/model/lm/*.yaml,
                                             config/conditioner/text2music.yaml, config/solver/musicgen/default.yaml) with the
                                             tiny dimensions, plus the `conditioners.args` entry released checkpoints carry
  ckpt_ref/text/compression_state_dict.bin   export_encodec of the `codec_noncausal` golden model
  ckpt_ref/melody/state_dict.bin             export_lm of the `lm_melody` golden model: chroma2music conditioners incl.
                                             `chroma_stem.cache_path`, `attribute_dropout.args`, and a persistent third-party
                                             buffer under the chroma conditioner (torchaudio's Spectrogram window)
  ckpt_ref/mha/state_dict.bin                a model built with custom=False, memory_efficient=False: its attention is an
                                             nn.MultiheadAttention and the keys read `self_attn.mha.*` (transformer.py:224-231)
  ckpt_ref/stub/compression_state_dict.bin   export_pretrained_compression_model: {'pretrained': name}

Run in the build container only:   python tests/golden/make_ckpt_golden.py
"""
import os
import sys
import tempfile

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, '..'))
import make_golden as mg  # noqa: E402  (installs the import stubs)
from conftest import load_golden  # noqa: E402
from audiocraft.utils import export  # noqa: E402

REFCFG = os.path.join(mg.refstubs.REF_ROOT, 'config')
OUT = os.path.join(HERE, 'ckpt_ref')


def _yaml(rel):
    with open(os.path.join(REFCFG, rel)) as f:
        return yaml.safe_load(f)


def _merge(a, b):
    for k, v in b.items():
        if k == 'defaults':
            continue
        if isinstance(v, dict) and isinstance(a.get(k), dict):
            _merge(a[k], v)
        else:
            a[k] = v
    return a


def lm_xp_cfg(g, conditioner_yaml):
    """What Hydra composes for a MusicGen run, restricted to the groups the loaders read."""
    cfg = {}
    _merge(cfg, {k: v for k, v in _yaml('solver/musicgen/default.yaml').items() if k in ('dataset', 'tokens', 'interleave_stereo_codebooks')})
    _merge(cfg, _yaml('model/lm/default.yaml'))
    _merge(cfg, _yaml('model/lm/musicgen_lm.yaml'))
    _merge(cfg, _yaml(conditioner_yaml))
    cfg['transformer_lm'].update(dim=g['dim'], num_heads=g['num_heads'], num_layers=g['num_layers'], card=g['card'],
                                 n_q=g['n_q'], hidden_scale=g['hidden_scale'], cross_attention=g['cross_attention'])
    cfg['codebooks_pattern']['delay']['delays'] = g['delays']
    cfg['classifier_free_guidance']['inference_coef'] = g['cfg_coef']
    cfg['conditioners']['args'] = {'merge_text_conditions_p': 0.25, 'drop_desc_p': 0.5}   # as in released checkpoints
    cfg.update(sample_rate=32000, channels=1, compression_model_checkpoint='//pretrained/facebook/encodec_32khz')
    return cfg


def codec_xp_cfg(g):
    cfg = {}
    _merge(cfg, _yaml('model/encodec/default.yaml'))
    _merge(cfg, _yaml('model/encodec/encodec_large_nq4_s640.yaml'))
    cfg.update(sample_rate=g['sample_rate'], channels=g['channels'])
    for k in ('dimension', 'n_filters', 'n_residual_layers', 'ratios', 'kernel_size', 'residual_kernel_size',
              'last_kernel_size', 'dilation_base', 'pad_mode', 'true_skip', 'compress', 'lstm', 'norm'):
        cfg['seanet'][k] = g[k]
    cfg['seanet']['activation_params'] = {'alpha': g['elu_alpha']}
    cfg['seanet']['decoder']['trim_right_ratio'] = g['trim_right_ratio']
    cfg['rvq'].update(n_q=g['n_q'], bins=g['bins'])
    cfg['encodec'].update(causal=g['causal'], renormalize=g['renormalize'])
    return cfg


def _export_lm(sd, cfg, sub):
    with tempfile.TemporaryDirectory() as tmp:
        ck = os.path.join(tmp, 'checkpoint.th')
        torch.save({'best_state': {'model': sd}, 'fsdp_best_state': {}, 'xp.cfg': cfg}, ck)
        out = export.export_lm(ck, os.path.join(OUT, sub, 'state_dict.bin'))
    print('wrote', out, os.path.getsize(out) // 1024, 'KiB')


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    # ---- text LM + its codec
    g, sd, _ = load_golden('lm_text')
    _export_lm(sd, lm_xp_cfg(g, 'conditioner/text2music.yaml'), 'text')
    gc, csd, _ = load_golden('codec_noncausal')
    with tempfile.TemporaryDirectory() as tmp:
        ck = os.path.join(tmp, 'checkpoint.th')
        torch.save({'best_state': {'model': csd}, 'xp.cfg': codec_xp_cfg(gc)}, ck)
        out = export.export_encodec(ck, os.path.join(OUT, 'text', 'compression_state_dict.bin'))
    print('wrote', out, os.path.getsize(out) // 1024, 'KiB')
    # ---- melody LM: chroma2music conditioners; a released melody checkpoint also carries the (persistent) window buffer
    # of the third-party torchaudio Spectrogram inside the chroma conditioner
    g, sd, _ = load_golden('lm_melody')
    cfg = lm_xp_cfg(g, 'conditioner/chroma2music.yaml')
    cfg['conditioners']['self_wav']['chroma_stem']['cache_path'] = '/checkpoint/someone/chroma_cache'
    cfg['dataset']['segment_duration'] = 30
    sd = dict(sd)
    sd['condition_provider.conditioners.self_wav.chroma.spec.window'] = torch.hann_window(16384)
    _export_lm(sd, cfg, 'melody')
    # ---- nn.MultiheadAttention key layout
    gm = dict(dim=32, num_heads=4, num_layers=2, hidden_scale=4, n_q=4, card=32, cross_attention=True,
              delays=[0, 1, 2, 3], cfg_coef=3.0, seed=7, cond_dim=8, Lc=5)
    torch.manual_seed(1000 + gm['seed'])
    cp = mg.ConditioningProvider({'description': mg.SynthText(gm['cond_dim'], gm['dim'], gm['Lc'])})
    torch.manual_seed(gm['seed'])
    lm = mg.LMModel(mg.DelayedPatternProvider(4, delays=gm['delays']), cp,
                    mg.ConditionFuser({'cross': ['description'], 'prepend': [], 'sum': [], 'input_interpolate': []}),
                    n_q=4, card=32, dim=32, num_heads=4, hidden_scale=4, norm='layer_norm', norm_first=True,
                    bias_proj=False, weight_init='gaussian', depthwise_init='current', zero_bias_init=True, cfg_coef=3.0,
                    num_layers=2, dropout=0., activation='gelu', bias_ff=False, bias_attn=False, causal=True,
                    custom=False, memory_efficient=False, attention_as_float32=False, cross_attention=True,
                    positional_embedding='sin').eval()
    assert any('.mha.' in k for k in lm.state_dict())
    conds = [mg.ConditioningAttributes(text={'description': f'p{i}'}) for i in range(3)]
    tokens, _, ct = mg.run_lm(lm, conds, None, 12, use_sampling=False)
    cfg = lm_xp_cfg(gm, 'conditioner/text2music.yaml')
    cfg['transformer_lm'].update(custom=False, memory_efficient=False)
    _export_lm(lm.state_dict(), cfg, 'mha')
    np.savez_compressed(os.path.join(OUT, 'mha', 'expected.npz'), greedy_tokens=tokens.numpy(),
                        cross_src=ct['description'][0].detach().numpy())
    # ---- {'pretrained': name} stub
    export.export_pretrained_compression_model('facebook/encodec_32khz', os.path.join(OUT, 'stub', 'compression_state_dict.bin'))
    print('wrote stub')
