"""Checkpoint loading in the reference's release format (reference audiocraft/models/loaders.py:40-126,
audiocraft/utils/export.py:22-79): a torch-pickled dict {'best_state': state_dict, 'xp.cfg': yaml,
'version', 'exported'} in `state_dict.bin` (LM) / `compression_state_dict.bin` (EnCodec).

Files are looked up on disk only (a file, a directory, or `$AUDIOCRAFT_CACHE_DIR/<name>/`): this
environment has no network, so the HuggingFace-hub / URL branches of the reference are not reproduced.
The embedded YAML is parsed with PyYAML into plain dicts (simple `${a.b}` interpolations resolved);
OmegaConf is not required.
"""
import os
import re
import typing as tp

import torch
import yaml

from . import builders


def get_audiocraft_cache_dir() -> tp.Optional[str]:
    return os.environ.get('AUDIOCRAFT_CACHE_DIR', None)


def _hub_cache_lookup(repo_id: str, filename: str, cache_dir: tp.Optional[str]) -> tp.Optional[str]:
    if repo_id.count('/') != 1 or repo_id.startswith(('/', '.')):
        return None
    try:
        from huggingface_hub import try_to_load_from_cache
        hit = try_to_load_from_cache(repo_id=repo_id, filename=filename, cache_dir=cache_dir)
    except Exception:   # no huggingface_hub, malformed id: not a hub checkpoint
        return None
    return hit if isinstance(hit, str) and os.path.isfile(hit) else None


def _find(file_or_id: str, filename: str) -> str:
    if os.path.isfile(file_or_id):
        return file_or_id
    if os.path.isdir(file_or_id):
        return os.path.join(file_or_id, filename)
    cache = get_audiocraft_cache_dir()
    if cache is not None:
        cand = os.path.join(cache, file_or_id.replace('/', '--'), filename)
        if os.path.isfile(cand):
            return cand
        cand = os.path.join(cache, file_or_id, filename)
        if os.path.isfile(cand):
            return cand
    # the reference resolves names through huggingface_hub.hf_hub_download (loaders.py:63-70): a checkpoint fetched by it
    # earlier sits in the hub's cache layout (<cache>/models--org--name/snapshots/<rev>/<filename>), under
    # $AUDIOCRAFT_CACHE_DIR if set, else HuggingFace's default cache -- looked up without touching the network
    hit = _hub_cache_lookup(file_or_id, filename, cache)
    if hit is not None:
        return hit
    raise FileNotFoundError(
        f"checkpoint '{file_or_id}' ({filename}) not found on disk; downloading is not possible here. "
        "Pass a directory containing the exported files, or set AUDIOCRAFT_CACHE_DIR. "
        "For benchmarking without weights use MusicGen.get_random_init(name).")


def _get_state_dict(file_or_id: str, filename: str, device='cpu'):
    return torch.load(_find(file_or_id, filename), map_location=device, weights_only=False)


def _resolve(cfg):
    """Resolve `${a.b.c}` interpolations against the root mapping (enough for exported xp.cfg files)."""
    pat = re.compile(r'\$\{([^}]+)\}')

    def lookup(path):
        node = cfg
        for part in path.split('.'):
            node = node[part]
        return node

    def walk(node):
        if isinstance(node, dict):
            return {k: walk(v) for k, v in node.items()}
        if isinstance(node, list):
            return [walk(v) for v in node]
        if isinstance(node, str):
            m = pat.fullmatch(node)
            if m:
                return walk(lookup(m.group(1)))
            return pat.sub(lambda mm: str(walk(lookup(mm.group(1)))), node)
        return node
    return walk(cfg)


def parse_cfg(text_or_dict) -> dict:
    cfg = yaml.safe_load(text_or_dict) if isinstance(text_or_dict, str) else dict(text_or_dict)
    return _resolve(cfg)


TRANSFORMER_OPTIONS = ('positional_embedding', 'xpos', 'past_context', 'layer_scale', 'positional_scale', 'max_period',
                       'bias_ff', 'bias_attn', 'bias_proj',   # biases: true in config/model/lm/default.yaml, false in the releases
                       'kv_repeat', 'qk_layer_norm', 'qk_layer_norm_cross',   # config/model/lm/default.yaml:43-46
                       'norm_first')                                          # default.yaml:21 (false there, true in every release)


def lm_cfg_from_xp(cfg: dict) -> dict:
    """`xp.cfg` of a MusicGen checkpoint -> builders.get_lm_model cfg (reference builders.py:136-175)."""
    t = dict(cfg['transformer_lm'])
    out = dict(dim=t['dim'], num_heads=t['num_heads'], num_layers=t['num_layers'],
               hidden_scale=t.get('hidden_scale', 4), n_q=t.get('n_q', 4), card=t.get('card', 2048),
               codebooks_pattern=cfg.get('codebooks_pattern'),
               cfg_coef=cfg.get('classifier_free_guidance', {}).get('inference_coef', 3.0))
    for k in TRANSFORMER_OPTIONS:   # transformer options no release sets away from config/model/lm/default.yaml:25-33
        if t.get(k) is not None:
            out[k] = t[k]
    conds = {}
    for name, c in (cfg.get('conditioners') or {}).items():
        # `conditioners.args` (merge_text_conditions_p, drop_desc_p) is not a conditioner: the reference pops it
        # (builders.py get_conditioner_provider: dict_cfg.pop('args', {})) and load_lm_model deletes its entries
        # (loaders.py:118-120); null entries are conditioners switched off by the experiment config
        if name == 'args' or c is None:
            continue
        model = c['model']
        if model == 't5':
            conds[name] = {'kind': 't5', 'name': c['t5']['name']}
        elif model == 'chroma_stem':
            cs = c['chroma_stem']
            # cache_path / eval_wavs are training-time conveniences (the reference deletes cache_path on load,
            # loaders.py:117); the front-end parameters are what matters here
            conds[name] = {'kind': 'chroma', 'n_chroma': cs['n_chroma'], 'radix2_exp': cs['radix2_exp'],
                           'argmax': cs.get('argmax', False), 'match_len_on_eval': True,
                           'sample_rate': cfg.get('sample_rate', 32000),
                           'duration': cfg.get('dataset', {}).get('segment_duration', 30.)}
        else:
            raise NotImplementedError(f"conditioner model '{model}' is not on the MusicGen path")
    out['conditioners'] = conds
    fuser = dict(cfg.get('fuser') or {})
    out['fuser'] = {k: list(v) for k, v in fuser.items() if k in ('cross', 'prepend', 'sum', 'input_interpolate')}
    for k in ('cross_attention_pos_emb', 'cross_attention_pos_emb_scale'):   # reference builders.py get_condition_fuser
        if fuser.get(k) is not None:
            out[k] = fuser[k]
    return out


def load_lm_model(file_or_id: str, device='cuda', weight_dtype=None):
    """reference loaders.py:111-126"""
    pkg = _get_state_dict(file_or_id, 'state_dict.bin')
    cfg = parse_cfg(pkg['xp.cfg'])
    if weight_dtype is None:
        weight_dtype = torch.bfloat16
    state = _drop_third_party_buffers(_remap_mha_keys(pkg['best_state']))
    lm_cfg = lm_cfg_from_xp(cfg)
    for name, c in lm_cfg['conditioners'].items():   # the checkpoint is the truth for a conditioner's input width
        w = state.get(f'condition_provider.conditioners.{name}.output_proj.weight')
        if w is not None and c['kind'] == 't5':
            c['dim'] = int(w.shape[1])
    lm = builders.get_lm_model(lm_cfg, device, weight_dtype)
    lm.load_state_dict(state)
    lm.cfg = cfg
    return lm


def _remap_mha_keys(state: dict) -> dict:
    """Checkpoints of models built with custom=False, memory_efficient=False hold their attention inside an
    nn.MultiheadAttention (`...self_attn.mha.in_proj_weight`, transformer.py:211-214); the reference renames between
    the two layouts on load (transformer.py:224-231).  Same tensors, same layout: strip the `mha.` level."""
    out = {}
    for k, v in state.items():
        for att in ('.self_attn.mha.', '.cross_attention.mha.'):
            if att in k:
                k = k.replace(att, att[:-4])
        out[k] = v
    return out


def _drop_third_party_buffers(state: dict) -> dict:
    """The module tree mirrors only `output_proj` of each conditioner.  Released checkpoints may also carry buffers of
    the third-party models the reference conditioners embed (e.g. `ChromaStemConditioner.chroma.spec.window`, a
    persistent torchaudio buffer, or a fine-tuned T5): those stay behind the `embedder` boundary and are dropped
    here, by name, so that the strict load below still catches every key that IS mirrored."""
    pat = re.compile(r'^condition_provider\.conditioners\.[^.]+\.(?!output_proj\.)')
    return {k: v for k, v in state.items() if not pat.match(k)}


def compression_cfg_from_xp(cfg: dict) -> dict:
    """`xp.cfg` of an EnCodec checkpoint -> builders.get_compression_model cfg (builders.py:70-91)."""
    sk = dict(cfg['seanet'])
    enc_over = sk.pop('encoder', {}) or {}
    sk.pop('decoder', None)
    sk.update(enc_over)
    for k in ('final_activation', 'final_activation_params', 'trim_right_ratio'):
        sk.pop(k, None)
    q = cfg['rvq']
    return dict(seanet=sk, rvq=dict(n_q=q['n_q'], bins=q['bins']), sample_rate=cfg['sample_rate'],
                frame_rate=cfg['encodec']['frame_rate'] if 'frame_rate' in cfg.get('encodec', {}) else
                cfg['sample_rate'] // int(torch.tensor(sk['ratios']).prod()),
                channels=cfg['channels'], causal=cfg['encodec'].get('causal', False),
                renormalize=cfg['encodec'].get('renormalize', False))


def load_compression_model(file_or_id: str, device='cuda'):
    """reference loaders.py:78-91.  `{'pretrained': name}` stubs redirect to the named codec
    (only the 32 kHz MusicGen codec, whose weights must also be on disk)."""
    pkg = _get_state_dict(file_or_id, 'compression_state_dict.bin')
    if 'pretrained' in pkg:   # written by export_pretrained_compression_model (utils/export.py:37-55)
        from .encodec import CompressionModel
        return CompressionModel.get_pretrained(pkg['pretrained'], device)
    cfg = parse_cfg(pkg['xp.cfg'])
    model = builders.get_compression_model(compression_cfg_from_xp(cfg), device)
    model.load_state_dict(pkg['best_state'])
    model.cfg = cfg
    return model


def load_mbd_ckpt(file_or_id: str, filename: tp.Optional[str] = None):
    """reference loaders.py:175-178"""
    return _get_state_dict(file_or_id, filename or 'mbd.pt')


def _pick_mbd_cfg(cfg, is_config, to_container) -> dict:
    """The saved `cfg` of a released MultiBandDiffusion band is an OmegaConf node of the WHOLE training xp.  Only the
    sub-trees the reference reads (loaders.py:188-199, builders.py:291-306: channels, schedule, diffusion_unet, processor)
    are resolved: a full resolve would also evaluate unrelated interpolations (`${oc.env:USER}` in dora.dir, hydra / dora
    resolvers, `???` nodes) and fail on nodes nobody uses."""
    picked = {}
    for k in ('channels', 'schedule', 'diffusion_unet', 'processor', 'sample_rate'):
        if k in cfg:
            node = cfg[k]
            picked[k] = to_container(node) if is_config(node) else node
    return picked


def load_diffusion_models(file_or_id: str, device='cuda', filename: tp.Optional[str] = None):
    """reference loaders.py:181-203: one (DiffusionUnet, sample processor, cfg) per band from a MultiBandDiffusion package
    {'sample_rate', 'n_bands', i: {'cfg', 'model_state', 'processor_state'}}.  `cfg` may be an OmegaConf node (released
    files; OmegaConf is needed to unpickle those), a dict or YAML text; it is returned as a plain dict."""
    pkg = load_mbd_ckpt(file_or_id, filename=filename)
    models, processors, cfgs = [], [], []
    sample_rate = pkg['sample_rate']
    for i in range(pkg['n_bands']):
        cfg = pkg[i]['cfg']
        if not isinstance(cfg, (dict, str)):
            from omegaconf import OmegaConf   # only reachable when the pickle itself needed it
            cfg = _pick_mbd_cfg(cfg, OmegaConf.is_config, lambda node: OmegaConf.to_container(node, resolve=True))
        cfg = parse_cfg(cfg)
        model = builders.get_diffusion_model(cfg)
        model.load_state_dict(pkg[i]['model_state'])
        model.to(device)
        processor = builders.get_processor(cfg=cfg['processor'], sample_rate=sample_rate)
        processor.load_state_dict(pkg[i]['processor_state'])
        processor.to(device)
        models.append(model)
        processors.append(processor)
        cfgs.append(cfg)
    return models, processors, cfgs


def export_lm(lm, path: str, xp_cfg: dict):
    """Write an LM checkpoint in the reference export format (utils/export.py:58-79) -- used by tests."""
    torch.save({'best_state': {k: v.detach().cpu() for k, v in lm.state_dict().items()},
                'xp.cfg': yaml.safe_dump(xp_cfg), 'version': '1.4.0a2', 'exported': True}, path)
