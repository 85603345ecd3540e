"""Model builders: config -> module tree.

Mirrors the parts of `audiocraft.models.builders` used for inference (reference
audiocraft/models/builders.py:70-175, :257-335): `get_compression_model`, `get_lm_model`,
`get_conditioner_provider`, `get_condition_fuser`, `get_codebooks_pattern_provider` and the debug
models.  Configs are plain nested dicts (the YAML embedded in released checkpoints parses to the same
structure; OmegaConf is not needed at inference time).

`ARCHITECTURES` holds the shapes pinned by the reference configs so that a
random-init model of the right architecture can be built when no checkpoint is on disk.
"""
import typing as tp

import torch

from ..modules.conditioners import (ChromaStemConditioner, ConditionFuser, ConditioningProvider,
                                    SyntheticChromaEmbedder, SyntheticTextEmbedder, T5Conditioner)
from ..modules.seanet import SEANetDecoder, SEANetEncoder
from ..quantization.vq import ResidualVectorQuantizer
from .encodec import EncodecModel
from .lm import LMModel

# config/model/lm/model_scale/{small,medium,large}.yaml + config/model/lm/musicgen_lm.yaml
LM_SCALES = {
    'small': dict(dim=1024, num_heads=16, num_layers=24),
    'medium': dict(dim=1536, num_heads=24, num_layers=48),
    'large': dict(dim=2048, num_heads=32, num_layers=48),
}

# config/model/encodec/encodec_large_nq4_s640.yaml over default.yaml ("facebook/encodec_32khz")
ENCODEC_32KHZ = dict(
    seanet=dict(channels=1, dimension=128, n_filters=64, n_residual_layers=1, ratios=[8, 5, 4, 4], activation='ELU',
                activation_params={'alpha': 1.0}, norm='weight_norm', norm_params={}, kernel_size=7,
                last_kernel_size=7, residual_kernel_size=3, dilation_base=2, causal=False, pad_mode='constant',
                true_skip=True, compress=2, lstm=2, disable_norm_outer_blocks=0),
    rvq=dict(n_q=4, bins=2048), sample_rate=32000, frame_rate=50, channels=1, causal=False, renormalize=False)

# config/model/encodec/encodec_large_nq4_s320.yaml over default.yaml (AudioGen's 16 kHz codec: hop 320 -> 50 fps)
ENCODEC_16KHZ = dict(ENCODEC_32KHZ, seanet=dict(ENCODEC_32KHZ['seanet'], ratios=[8, 5, 4, 2]), sample_rate=16000)

# config/model/encodec/encodec_base_causal.yaml ("encodec_24khz" geometry, BASELINE.json config #1)
ENCODEC_24KHZ = dict(
    seanet=dict(channels=1, dimension=128, n_filters=32, n_residual_layers=1, ratios=[8, 5, 4, 2], activation='ELU',
                activation_params={'alpha': 1.0}, norm='weight_norm', norm_params={}, kernel_size=7,
                last_kernel_size=7, residual_kernel_size=3, dilation_base=2, causal=True, pad_mode='constant',
                true_skip=True, compress=2, lstm=2, disable_norm_outer_blocks=0),
    rvq=dict(n_q=32, bins=1024), sample_rate=24000, frame_rate=75, channels=1, causal=True, renormalize=False)


def get_compression_model(cfg: dict, device='cuda') -> EncodecModel:
    """cfg: {'seanet': SEANet kwargs, 'rvq': {'n_q', 'bins'}, 'sample_rate', 'frame_rate', 'channels', ...}."""
    sk = dict(cfg['seanet'])
    encoder = SEANetEncoder(**sk, device=device)
    decoder = SEANetDecoder(**sk, trim_right_ratio=cfg.get('trim_right_ratio', 1.0), device=device)
    quantizer = ResidualVectorQuantizer(dimension=sk['dimension'], device=device, **cfg['rvq'])
    return EncodecModel(encoder, decoder, quantizer, frame_rate=cfg['frame_rate'], sample_rate=cfg['sample_rate'],
                        channels=cfg['channels'], causal=cfg.get('causal', False),
                        renormalize=cfg.get('renormalize', False)).to(device)


def get_codebooks_pattern_provider(n_q: int, cfg: tp.Optional[dict] = None):
    """reference builders.py:240-254: `cfg['modeling']` names the provider, `cfg[<that name>]` holds its arguments."""
    from ..modules import codebooks_patterns as cp
    providers = {'parallel': cp.ParallelPatternProvider, 'delay': cp.DelayedPatternProvider, 'unroll': cp.UnrolledPatternProvider,
                 'coarse_first': cp.CoarseFirstPattern, 'musiclm': cp.MusicLMPattern}
    cfg = cfg or {'modeling': 'delay', 'delay': {'delays': list(range(n_q))}}
    name = cfg.get('modeling', 'delay')
    if name not in providers:
        raise KeyError(f"unknown codebooks pattern '{name}' (one of {sorted(providers)})")
    return providers[name](n_q, **dict(cfg.get(name) or {}))


def get_lm_model(cfg: dict, device='cuda', weight_dtype=torch.bfloat16, kv_dtype=None) -> LMModel:
    """cfg: {'dim','num_heads','num_layers','n_q','card','hidden_scale', 'cfg_coef',
             'conditioners': {name: {'kind': 't5'|'chroma', ...}}, 'fuser': {'cross': [...], 'prepend': [...]}}."""
    dim = cfg['dim']
    conds = {}
    for name, c in cfg.get('conditioners', {}).items():
        kind = c['kind']
        if kind == 't5':
            embedder = c.get('embedder')
            if embedder == 'synthetic':
                embedder = SyntheticTextEmbedder(c.get('dim', 768), c.get('length', 16), c.get('seed', 0),
                                                 c.get('lengths'))
            conds[name] = T5Conditioner(c.get('name', 't5-base'), dim, device=device, embedder=embedder,
                                        dim=c.get('dim'))
        elif kind == 'chroma':
            cc = ChromaStemConditioner(dim, c.get('sample_rate', 32000), c.get('n_chroma', 12),
                                       c.get('radix2_exp', 14), c.get('duration', 30.), device=device,
                                       argmax=c.get('argmax', True))
            embedder = c.get('embedder')
            if embedder == 'synthetic':
                embedder = SyntheticChromaEmbedder(c.get('n_frames', cc.chroma_len), c.get('n_chroma', 12),
                                                   c.get('seed', 3))
            cc.embedder = embedder
            conds[name] = cc
        else:
            raise ValueError(f"unknown conditioner kind {kind}")
    provider = ConditioningProvider(conds, device=device)
    fuse = {'cross': [], 'prepend': [], 'sum': [], 'input_interpolate': []}
    fuse.update(cfg.get('fuser', {}))
    fuser = ConditionFuser(fuse, cross_attention_pos_emb=cfg.get('cross_attention_pos_emb', False),
                           cross_attention_pos_emb_scale=cfg.get('cross_attention_pos_emb_scale', 1.0))
    n_q = cfg.get('n_q', 4)
    lm = LMModel(get_codebooks_pattern_provider(n_q, cfg.get('codebooks_pattern')), provider, fuser, n_q=n_q,
                 card=cfg.get('card', 2048), dim=dim, num_heads=cfg['num_heads'],
                 hidden_scale=cfg.get('hidden_scale', 4), norm='layer_norm', norm_first=cfg.get('norm_first', True),
                 bias_proj=cfg.get('bias_proj', False),
                 weight_init=cfg.get('weight_init', 'gaussian'), depthwise_init=cfg.get('depthwise_init', 'current'),
                 zero_bias_init=True, cfg_coef=cfg.get('cfg_coef', 3.0), num_layers=cfg['num_layers'],
                 cross_attention=bool(fuse['cross']), bias_ff=cfg.get('bias_ff', False), bias_attn=cfg.get('bias_attn', False),
                 positional_embedding=cfg.get('positional_embedding', 'sin'), max_period=cfg.get('max_period', 10000.),
                 positional_scale=cfg.get('positional_scale', 1.0), xpos=cfg.get('xpos', False),
                 past_context=cfg.get('past_context'), layer_scale=cfg.get('layer_scale'),
                 kv_repeat=cfg.get('kv_repeat', 1), qk_layer_norm=cfg.get('qk_layer_norm', False),
                 qk_layer_norm_cross=cfg.get('qk_layer_norm_cross', False),
                 weight_dtype=weight_dtype, kv_dtype=kv_dtype, device=device)
    return lm.to(device)


def musicgen_lm_cfg(scale: str = 'small', melody: bool = False, synthetic: bool = True, text_len: int = 16,
                    stereo: bool = False, synthetic_chroma: bool = False) -> dict:
    """Architecture of facebook/musicgen-{small,medium,large,melody}[ -stereo ].  Stereo:
    8 codebooks = left / right interleaved per RVQ level, each pair sharing its delay."""
    cfg = dict(LM_SCALES[scale], n_q=8 if stereo else 4, card=2048, hidden_scale=4, cfg_coef=3.0)
    if stereo:
        cfg['codebooks_pattern'] = {'modeling': 'delay', 'delay': {'delays': [0, 0, 1, 1, 2, 2, 3, 3]}}
    emb = 'synthetic' if synthetic else None
    cfg['conditioners'] = {'description': {'kind': 't5', 'name': 't5-base', 'embedder': emb, 'length': text_len}}
    if melody:  # config/conditioner/chroma2music.yaml: prepend [self_wav, description], no cross-attention
        # the chroma front-end runs on the device (acmi_chroma) unless a synthetic one is asked for; Demucs is pluggable
        cfg['conditioners']['self_wav'] = {'kind': 'chroma', 'embedder': 'synthetic' if synthetic_chroma else None,
                                           'n_chroma': 12, 'radix2_exp': 14,
                                           'duration': 30., 'sample_rate': 32000}
        cfg['fuser'] = {'prepend': ['self_wav', 'description']}
    else:       # config/conditioner/text2music.yaml
        cfg['fuser'] = {'cross': ['description']}
    return cfg


def audiogen_lm_cfg(scale: str = 'medium', synthetic: bool = True, text_len: int = 16) -> dict:
    """Architecture of facebook/audiogen-medium: the MusicGen LM (config/model/lm/audiogen_lm.yaml) with
    T5-large text conditioning through cross-attention (config/conditioner/text2sound.yaml)."""
    cfg = dict(LM_SCALES[scale], n_q=4, card=2048, hidden_scale=4, cfg_coef=3.0)
    cfg['conditioners'] = {'description': {'kind': 't5', 'name': 't5-large', 'dim': 1024,
                                           'embedder': 'synthetic' if synthetic else None, 'length': text_len}}
    cfg['fuser'] = {'cross': ['description']}
    return cfg


def get_diffusion_model(cfg: dict):
    """reference builders.py:291-295"""
    from .unet import DiffusionUnet
    return DiffusionUnet(chin=cfg['channels'], num_steps=cfg['schedule']['num_steps'], **cfg['diffusion_unet'])


def get_processor(cfg: dict, sample_rate: int = 24000):
    """reference builders.py:298-306"""
    from ..modules.diffusion_schedule import MultiBandProcessor, SampleProcessor
    sample_processor = SampleProcessor()
    if cfg['use']:
        kw = {k: v for k, v in cfg.items() if k not in ('use', 'name')}
        if cfg['name'] == "multi_band_processor":
            sample_processor = MultiBandProcessor(sample_rate=sample_rate, **kw)
    return sample_processor


def get_debug_compression_model(device='cuda', sample_rate: int = 32000) -> EncodecModel:
    """reference builders.py:257-288: n_filters 4, ratios [10, 8, 16] (or [10, 8, 8] at 16 kHz), RVQ 4 x 400."""
    assert sample_rate in [16000, 32000]
    ratios = {16000: [10, 8, 8], 32000: [10, 8, 16]}[sample_rate]
    frame_rate = 25
    cfg = dict(seanet=dict(channels=1, dimension=32, n_filters=4, ratios=ratios), rvq=dict(n_q=4, bins=400),
               sample_rate=sample_rate, frame_rate=frame_rate, channels=1)
    return get_compression_model(cfg, device)


def get_debug_lm_model(device='cuda', weight_dtype=torch.float32) -> LMModel:
    """reference builders.py:309-335: dim 16, 2 layers, 4 heads, n_q 4, card 400, text cross-attention."""
    cfg = dict(dim=16, num_heads=4, num_layers=2, n_q=4, card=400, hidden_scale=4, cfg_coef=3.0,
               conditioners={'description': {'kind': 't5', 'embedder': 'synthetic', 'dim': 16, 'length': 4}},
               fuser={'cross': ['description']})
    return get_lm_model(cfg, device, weight_dtype)
