"""-m gpu: the product classes (EncodecModel / LMModel on HIP kernels) against
(a) the committed golden vectors of the unmodified reference and (b) the CPU oracle at larger seeded sizes."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import load_golden  # noqa: E402
from oracle import codec as ocodec  # noqa: E402
from oracle import lm as olm  # noqa: E402
from parity_utils import assert_codes_near_tie  # noqa: E402
from test_oracle_golden import codec_cfg, lm_cfg  # noqa: E402


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def build_codec(cfg, sd):
    from audiocraft_amd.models import builders
    sk = dict(channels=cfg['channels'], dimension=cfg['dimension'], n_filters=cfg['n_filters'],
              n_residual_layers=cfg['n_residual_layers'], ratios=cfg['ratios'], activation='ELU',
              activation_params={'alpha': cfg['elu_alpha']}, norm=cfg['norm'], norm_params={},
              kernel_size=cfg['kernel_size'], residual_kernel_size=cfg['residual_kernel_size'],
              last_kernel_size=cfg['last_kernel_size'], dilation_base=cfg['dilation_base'], causal=cfg['causal'],
              pad_mode=cfg['pad_mode'], true_skip=cfg['true_skip'], compress=cfg['compress'], lstm=cfg['lstm'],
              disable_norm_outer_blocks=0)
    m = builders.get_compression_model(dict(seanet=sk, rvq=dict(n_q=cfg['n_q'], bins=cfg['bins']),
                                            sample_rate=cfg['sample_rate'], frame_rate=cfg['frame_rate'],
                                            channels=cfg['channels'], causal=cfg['causal'],
                                            renormalize=cfg['renormalize'],
                                            trim_right_ratio=cfg['trim_right_ratio']), 'cuda')
    m.load_state_dict({k: v for k, v in sd.items()}, strict=True)
    return m


@pytest.mark.parametrize('name', ['codec_noncausal', 'codec_causal', 'codec_renorm'])
def test_encodec_vs_reference_golden(name):
    cfg, sd, a = load_golden(name)
    m = build_codec(cfg, sd)
    wav = a['wav'].cuda()
    x, _ = m.preprocess(wav)
    lat = m.encoder(x).cpu()
    assert lat.shape == a['latents'].shape
    assert torch.allclose(lat, a['latents'], atol=3e-5, rtol=1e-4), (lat - a['latents']).abs().max()
    # hard gate: RVQ on the reference's own latents is bit exact
    codes = m.quantizer.encode(a['latents'].cuda()).cpu()
    assert torch.equal(codes, a['codes'])
    # end to end encode: every index that differs from the reference's must be a near tie that the fp32 round-off of the
    # latents can flip, per RVQ level (parity_utils.assert_codes_near_tie)
    codes2, scale = m.encode(wav)
    assert_codes_near_tie(codes2, a['codes'], lat, a['latents'], ocodec.codebooks_from_state(sd, cfg['n_q']),
                          what=f'golden {name}')
    assert torch.equal(m.decode_latent(a['codes'].cuda()).cpu(), a['quantized_latents'])
    dec = m.decode(a['codes'].cuda(), None if 'scale' not in a else a['scale'].cuda()).cpu()
    assert dec.shape == a['decoded'].shape
    err = (dec - a['decoded']).abs().max().item()
    assert err < 1e-4, f"waveform max abs err {err}"   # BASELINE.md section 4 gate
    out = m(wav)
    assert out.x.shape == wav.shape and out.codes.shape == a['codes'].shape


def test_encodec_32khz_geometry_vs_oracle():
    """facebook/encodec_32khz architecture, seeded random weights, 1.3 s of audio, vs the CPU oracle."""
    from audiocraft_amd.models import builders
    torch.manual_seed(0)
    m = builders.get_compression_model(builders.ENCODEC_32KHZ, 'cuda')
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    c = ocodec.CodecConfig(channels=1, dimension=128, n_filters=64, n_residual_layers=1, ratios=[8, 5, 4, 4],
                           causal=False, pad_mode='constant', lstm=2, norm='weight_norm', n_q=4, bins=2048,
                           sample_rate=32000, frame_rate=50)
    wav = 0.3 * torch.randn(2, 1, 41611, generator=torch.Generator().manual_seed(1))
    lat_ref = ocodec.seanet_encoder(sd, c, wav, fast_lstm=True)
    lat = m.encoder(wav.cuda()).cpu()
    assert lat.shape == lat_ref.shape == (2, 128, math.ceil(41611 / 640))
    assert rel(lat, lat_ref) < 2e-5
    codes_ref = ocodec.rvq_encode(lat_ref, ocodec.codebooks_from_state(sd, 4))
    assert torch.equal(m.quantizer.encode(lat_ref.cuda()).cpu(), codes_ref)   # identical latents: bit exact
    codes, _ = m.encode(wav.cuda())
    assert_codes_near_tie(codes, codes_ref, lat, lat_ref, ocodec.codebooks_from_state(sd, 4), what='EnCodec-32k geometry')
    dec_ref = ocodec.encodec_decode(sd, c, codes_ref, fast_lstm=True)
    dec = m.decode(codes_ref.cuda()).cpu()
    assert dec.shape == dec_ref.shape
    assert (dec - dec_ref).abs().max().item() < 1e-4


def test_encodec_24khz_geometry_config1():
    """BASELINE.json configs[0] / SURVEY.md section 8(d) #1: EnCodec-24 kHz geometry (causal SEANet n_filters 32,
    ratios [8,5,4,2], RVQ 32 x 1024 x 128), seeded random weights, input 0.1 * randn(1, 1, 240000) seed 1
    -> codes [1, 32, 750].  Gates: RVQ indices bit exact on identical latents (all 32 stages), latents and
    waveform within fp32 tolerance of the CPU oracle."""
    from audiocraft_amd.models import builders
    torch.manual_seed(0)
    m = builders.get_compression_model(builders.ENCODEC_24KHZ, 'cuda')
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    c = ocodec.CodecConfig(channels=1, dimension=128, n_filters=32, n_residual_layers=1, ratios=[8, 5, 4, 2],
                           causal=True, pad_mode='constant', lstm=2, norm='weight_norm', n_q=32, bins=1024,
                           sample_rate=24000, frame_rate=75)
    wav = 0.1 * torch.randn(1, 1, 240000, generator=torch.Generator().manual_seed(1))
    lat_ref = ocodec.seanet_encoder(sd, c, wav, fast_lstm=True)
    lat = m.encoder(wav.cuda()).cpu()
    assert lat.shape == lat_ref.shape == (1, 128, 750)
    assert rel(lat, lat_ref) < 2e-5
    codes_ref = ocodec.rvq_encode(lat_ref, ocodec.codebooks_from_state(sd, 32))
    assert codes_ref.shape == (1, 32, 750) and int(codes_ref.max()) < 1024
    assert torch.equal(m.quantizer.encode(lat_ref.cuda()).cpu(), codes_ref)   # identical latents: bit exact
    codes, scale = m.encode(wav.cuda())
    assert scale is None and codes.shape == (1, 32, 750)
    # all 32 levels: a differing index must be a near tie of the oracle's decision (latents differ by fp32 round-off only)
    assert_codes_near_tie(codes, codes_ref, lat, lat_ref, ocodec.codebooks_from_state(sd, 32), what='configs[0] randn input')
    dec_ref = ocodec.encodec_decode(sd, c, codes_ref, fast_lstm=True)
    dec = m.decode(codes_ref.cuda()).cpu()
    assert dec.shape == dec_ref.shape == (1, 1, 240000)
    assert (dec - dec_ref).abs().max().item() < 1e-4


def test_encodec_24khz_geometry_on_reference_asset():
    """configs[0] on real audio (SURVEY.md section 8c iii): the 3 s of the reference's assets/epic.wav (32 kHz mono f32,
    committed as tests/golden/epic_wav.npz by make_epic_fixture.py; taken as raw samples at the model rate -- content, not
    pitch, matters for a codes gate) through the EnCodec-24 kHz geometry.  Codes [1, 32, 300]: bit exact on identical
    latents, and end to end every differing index is a near tie (per RVQ level); waveform <= 1e-4."""
    import os
    import numpy as np
    from audiocraft_amd.models import builders
    from conftest import GOLDEN
    wav = torch.from_numpy(np.load(os.path.join(GOLDEN, 'epic_wav.npz'))['wav']).view(1, 1, -1)
    assert wav.shape[-1] == 96000 and 0.5 < float(wav.abs().max()) < 1.0
    torch.manual_seed(0)
    m = builders.get_compression_model(builders.ENCODEC_24KHZ, 'cuda')
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    c = ocodec.CodecConfig(channels=1, dimension=128, n_filters=32, n_residual_layers=1, ratios=[8, 5, 4, 2],
                           causal=True, pad_mode='constant', lstm=2, norm='weight_norm', n_q=32, bins=1024,
                           sample_rate=24000, frame_rate=75)
    lat_ref = ocodec.seanet_encoder(sd, c, wav, fast_lstm=True)
    lat = m.encoder(wav.cuda()).cpu()
    assert lat.shape == lat_ref.shape == (1, 128, 300)
    assert rel(lat, lat_ref) < 2e-5
    cb = ocodec.codebooks_from_state(sd, 32)
    codes_ref = ocodec.rvq_encode(lat_ref, cb)
    assert torch.equal(m.quantizer.encode(lat_ref.cuda()).cpu(), codes_ref)   # identical latents: bit exact, 32 levels
    codes, _ = m.encode(wav.cuda())
    assert_codes_near_tie(codes, codes_ref, lat, lat_ref, cb, what='configs[0] epic.wav')
    dec_ref = ocodec.encodec_decode(sd, c, codes_ref, fast_lstm=True)
    dec = m.decode(codes_ref.cuda()).cpu()
    assert (dec - dec_ref).abs().max().item() < 1e-4


def test_encodec_random_lengths_roundtrip_shapes():
    """reference tests/models/test_encodec_model.py:37-46"""
    from audiocraft_amd.models import builders
    m = builders.get_debug_compression_model('cuda')
    g = torch.Generator().manual_seed(0)
    for _ in range(6):
        length = int(torch.randint(1, 10000, (1,), generator=g))
        x = torch.randn(2, 1, length).cuda()
        res = m(x)
        assert res.x.shape == x.shape
        codes, scale = m.encode(x)
        assert codes.shape[:2] == (2, 4) and scale is None
        assert m.decode(codes).shape[-1] >= length


def test_encodec_graph_replay_matches_eager():
    """Short inputs run the SEANet passes as hipGraph replays from the second call with a shape on (EncodecModel._seanet):
    first (eager), second (capture + replay), third (replay) call and a call with other VALUES in the same shape all give the
    eager result bit for bit; a different shape takes a slot of its own; load_state_dict drops the captured graphs."""
    from audiocraft_amd.models import builders
    torch.manual_seed(3)
    m = builders.get_compression_model(builders.ENCODEC_24KHZ, 'cuda')
    eager = builders.get_compression_model(builders.ENCODEC_24KHZ, 'cuda')
    eager.load_state_dict(m.state_dict())
    eager.GRAPH_MAX_SAMPLES = 0
    g = torch.Generator().manual_seed(4)
    a, b = (0.2 * torch.randn(2, 1, 24000, generator=g)).cuda(), (0.2 * torch.randn(2, 1, 24000, generator=g)).cuda()
    c = (0.2 * torch.randn(1, 1, 7777, generator=g)).cuda()
    for x in (a, a, a, b, c, c, b):
        codes, _ = m.encode(x)
        codes_e, _ = eager.encode(x)
        assert torch.equal(codes, codes_e)
        assert torch.equal(m.decode(codes), eager.decode(codes_e))
    kinds = sorted(k[0] for k, v in m._graphs.items() if v is not False)
    assert kinds == ['dec', 'dec', 'enc', 'enc'], m._graphs.keys()
    m.load_state_dict(eager.state_dict())
    assert m._graphs == {}
    big = torch.zeros(8, 1, 240000).cuda()      # 1.9 M samples: above GRAPH_MAX_SAMPLES, always eager
    m.encode(big), m.encode(big)
    assert m._graphs == {}


# ------------------------------------------------------------------------------------------ LM

def build_lm(cfg, sd, weight_dtype=torch.float32):
    from audiocraft_amd.models import builders
    conds = {'description': {'kind': 't5', 'embedder': 'synthetic', 'dim': cfg['cond_dim'], 'length': cfg['Lc']}}
    fuser = {'cross': ['description']} if cfg['cross_attention'] else {'prepend': ['self_wav', 'description']}
    if not cfg['cross_attention'] and 'P' not in cfg:
        fuser = {'prepend': ['description']}            # text prepended, no melody condition (lm_two_step_prepend)
    elif not cfg['cross_attention']:
        conds['self_wav'] = {'kind': 'chroma', 'embedder': 'synthetic', 'n_frames': cfg['P']}
    lm = builders.get_lm_model(dict(dim=cfg['dim'], num_heads=cfg['num_heads'], num_layers=cfg['num_layers'],
                                    n_q=cfg['n_q'], card=cfg['card'], hidden_scale=cfg['hidden_scale'],
                                    cfg_coef=cfg['cfg_coef'], conditioners=conds, fuser=fuser,
                                    codebooks_pattern={'modeling': 'delay', 'delay': {'delays': cfg['delays']}},
                                    **{k: cfg[k] for k in ('positional_embedding', 'xpos', 'past_context', 'layer_scale',
                                                           'positional_scale') if k in cfg}),
                               'cuda', weight_dtype)
    missing = lm.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys, missing.unexpected_keys
    assert all('self_wav' in k for k in missing.missing_keys), missing.missing_keys
    return lm


@pytest.fixture(params=['big', 'chunk'])
def prefill_mode(request, monkeypatch):
    """Both prefill paths against the same reference goldens: 'big' = ONE MFMA-tiled forward over the prompt / prefix
    (acmi_prefill.hip; forced from 2 positions on), 'chunk' = the decode kernels on 8 positions per call."""
    monkeypatch.setenv('ACMI_PREFILL', request.param)
    return request.param


def test_lm_text_vs_reference_golden(prefill_mode):
    cfg, sd, a = load_golden('lm_text')
    lm = build_lm(cfg, sd)
    ones = torch.ones(a['cross_src'].shape[:2], dtype=torch.int64)
    ct = {'description': (a['cross_src'].cuda(), ones.cuda())}
    # teacher forced streaming logits == reference batch forward (streaming == batch invariant)
    logits = lm.forward_steps(a['tf_sequence'].cuda(), ct).cpu()
    assert logits.shape == a['tf_logits'].shape
    r = rel(logits, a['tf_logits'])
    assert r < 1e-4, f"logits rel-L2 {r}"     # BASELINE.md section 4 gate (fp32 mode)
    # greedy generation: identical tokens, per-step CFG logits within tolerance
    toks, lg = lm.generate(None, [], num_samples=3, max_gen_len=12, use_sampling=False, condition_tensors=ct,
                           return_logits=True, check=True)
    assert torch.equal(toks.cpu(), a['greedy_tokens'])
    assert rel(lg.cpu(), olm.cfg_mix(a['greedy_step_logits'], cfg['cfg_coef'])) < 1e-4
    # continuation from a 3 step prompt
    toks = lm.generate(a['prompt'].cuda(), [], max_gen_len=10, use_sampling=False, condition_tensors=ct, check=True)
    assert torch.equal(toks.cpu(), a['cont_tokens'])
    toks = lm.generate(a['prompt'].cuda(), [], max_gen_len=10, use_sampling=False, condition_tensors=ct,
                       remove_prompts=True)
    assert torch.equal(toks.cpu(), a['cont_tokens_removed'])
    # graph replay and eager launches agree bit for bit
    t1 = lm.generate(None, [], num_samples=3, max_gen_len=12, use_sampling=False, condition_tensors=ct, use_graph=False)
    assert torch.equal(t1.cpu(), a['greedy_tokens'])


@pytest.mark.parametrize('name', ['lm_rope', 'lm_sin_rope'])
def test_lm_rope_past_context_layer_scale_vs_reference_golden(name, prefill_mode):
    """Rotary positions (+ xPos decay) applied by the QKV launch, the bounded receptive field of the attention kernel
    and LayerScale folded into the branch matrices, against goldens from the unmodified reference (custom attention:
    tests/golden/make_rope_golden.py); lm_rope's 9-step prompt exceeds past_context = 6, which exercises the windowed
    multi-position prefill and the reference's lagging rotary positions after a long first call (acmi_lm_state.rope_shift)."""
    cfg, sd, a = load_golden(name)
    lm = build_lm(cfg, sd)
    assert lm.positional_embedding == cfg['positional_embedding'] and lm.past_context == cfg.get('past_context')
    ones = torch.ones(a['cross_src'].shape[:2], dtype=torch.int64)
    ct = {'description': (a['cross_src'].cuda(), ones.cuda())}
    logits = lm.forward_steps(a['tf_sequence'].cuda(), ct).cpu()
    r = rel(logits, a['tf_logits'])
    assert r < 1e-4, f"teacher-forced logits rel-L2 {r}"
    toks, lg = lm.generate(None, [], num_samples=3, max_gen_len=14, use_sampling=False, condition_tensors=ct,
                           return_logits=True, check=True)
    assert torch.equal(toks.cpu(), a['greedy_tokens'])
    assert rel(lg.cpu(), olm.cfg_mix(a['greedy_step_logits'], cfg['cfg_coef'])) < 1e-4
    toks = lm.generate(a['prompt'].cuda(), [], max_gen_len=14, use_sampling=False, condition_tensors=ct, check=True)
    assert torch.equal(toks.cpu(), a['cont_tokens'])
    # the streaming protocol: one multi-step first call, then single steps == the reference's first-call logits
    with lm.streaming():
        first = lm(torch.cat([a['prompt'], a['prompt']]).cuda()[..., :1], [], ct)
        assert first.shape[:3] == (6, cfg['n_q'], 1)
    # bf16 weights / caches run the same path
    lm16 = build_lm(cfg, sd, torch.bfloat16)
    l16 = lm16.forward_steps(a['tf_sequence'].cuda(), ct).cpu()
    assert rel(l16, a['tf_logits']) < 3e-2


def test_lm_melody_vs_reference_golden(prefill_mode):
    cfg, sd, a = load_golden('lm_melody')
    lm = build_lm(cfg, sd)
    P, Lc = cfg['P'], cfg['Lc']
    pre = a['prepend_src'].cuda()
    ct = {'description': (pre[:, P:], torch.ones(pre.shape[0], Lc, dtype=torch.int64).cuda()),
          'self_wav': (pre[:, :P], torch.ones(pre.shape[0], P, dtype=torch.int64).cuda())}
    toks, lg = lm.generate(None, [], num_samples=2, max_gen_len=9, use_sampling=False, condition_tensors=ct,
                           return_logits=True, check=True)
    assert torch.equal(toks.cpu(), a['greedy_tokens'])
    assert rel(lg.cpu(), olm.cfg_mix(a['greedy_step_logits'], cfg['cfg_coef'])) < 1e-4


@pytest.mark.parametrize('B', [3, 12, 20])   # CFG rows 6 / 24 / 40: 1, 2 and 4 (3 valid) 16-row blocks per GEMM
@pytest.mark.parametrize('wdt,tol', [(torch.float32, 1e-4), (torch.bfloat16, 3e-2)])
def test_lm_midsize_vs_oracle(wdt, tol, B):
    """d=256, 4 layers, card 2048, cross attention: teacher-forced logits + greedy tokens vs the oracle."""
    from audiocraft_amd.models import builders
    torch.manual_seed(0)
    cfg = dict(dim=256, num_heads=4, num_layers=4, n_q=4, card=2048, hidden_scale=4, cfg_coef=3.0,
               conditioners={'description': {'kind': 't5', 'embedder': 'synthetic', 'dim': 64, 'length': 6}},
               fuser={'cross': ['description']})
    lm = builders.get_lm_model(cfg, 'cuda', wdt)
    with torch.no_grad():
        for k, p in lm.named_parameters():
            if 'norm' in k:
                p.add_(0.1 * torch.randn_like(p))
    sd = {k: v.detach().float().cpu() for k, v in lm.state_dict().items()}
    if wdt == torch.bfloat16:   # the oracle sees the same bf16-rounded matrices; activations stay f32 there
        sd = {k: (v.bfloat16().float() if v.dim() == 2 and 'output_proj' not in k else v) for k, v in sd.items()}
    oc = olm.LMConfig(dim=256, num_heads=4, num_layers=4, n_q=4, card=2048, cross_attention=True)
    g = torch.Generator().manual_seed(5)
    cross = torch.randn(2 * B, 6, 256, generator=g)
    cross[B:] = 0
    ct = {'description': (cross.cuda(), torch.ones(2 * B, 6, dtype=torch.int64).cuda())}
    seq = torch.randint(0, 2049, (2 * B, 4, 20 if B == 3 else 8), generator=g)
    ref = olm.lm_forward(sd, oc, seq, cross)
    got = lm.forward_steps(seq.cuda(), ct).cpu()
    r = rel(got, ref)
    assert r < tol, f"teacher-forced logits rel-L2 {r} (tol {tol})"
    if wdt == torch.float32 and B == 3:
        toks = lm.generate(None, [], num_samples=B, max_gen_len=16, use_sampling=False, condition_tensors=ct)
        ref_t = olm.generate(sd, oc, None, B, cross, max_gen_len=16, use_sampling=False)
        assert torch.equal(toks.cpu(), ref_t)


@pytest.mark.parametrize('wdt,tol', [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize('with_bias', [False, True])
def test_score_folded_cross_attention_matches_the_separate_launches(wdt, tol, with_bias, monkeypatch):
    """The decode step's cross-attention in its score-folded form (acmi_lm_state.xs_rows: scores out of the QKV / out-projection
    launches, ONE launch for LayerNorm hook + softmax + p U + residual; modules/cross_fold.py) against the separate launches
    (ACMI_CROSS_FOLD=0) and against the oracle: per-step CFG logits of a greedy generate with a prompt, tokens."""
    from audiocraft_amd.models import builders
    torch.manual_seed(3)
    B = 3
    cfg = dict(dim=256, num_heads=8, num_layers=4, n_q=4, card=2048, hidden_scale=4, cfg_coef=3.0, bias_attn=with_bias,
               conditioners={'description': {'kind': 't5', 'embedder': 'synthetic', 'dim': 64, 'length': 7}},
               fuser={'cross': ['description']})
    lm = builders.get_lm_model(cfg, 'cuda', wdt)
    with torch.no_grad():
        for k, p in lm.named_parameters():
            if 'norm' in k:
                p.add_(0.1 * torch.randn_like(p))
            if k.endswith('out_proj.bias'):      # (biases on the k / v in-projection would make the null rows live: not folded)
                p.add_(0.05 * torch.randn_like(p))
            if k.endswith('in_proj_bias'):
                p.zero_()
                p[:256].add_(0.05 * torch.randn(256, device=p.device))
    g = torch.Generator().manual_seed(8)
    cross = torch.randn(2 * B, 7, 256, generator=g)
    cross[B:] = 0
    ct = {'description': (cross.cuda(), torch.ones(2 * B, 7, dtype=torch.int64).cuda())}
    prompt = torch.randint(0, 2048, (B, 4, 5), generator=g).cuda()
    kw = dict(max_gen_len=30, use_sampling=False, condition_tensors=ct, return_logits=True)
    monkeypatch.setenv('ACMI_CROSS_FOLD', '1')       # opt-in: the separate launches are the default
    toks_f, lg_f = lm.generate(prompt, [], **kw)
    assert lm._run.get('xs') is not None and lm._run['xs']['key'] == (B, 7), "the folded path did not run"
    monkeypatch.setenv('ACMI_CROSS_FOLD', '0')
    lm._run = None
    toks_s, lg_s = lm.generate(prompt, [], **kw)
    assert lm._run.get('xs') is None
    # free-running greedy generates: once a token differs (a near tie under bf16 rounding) the streams are different problems,
    # so the logits are compared up to and including the first step at which the two runs pick different tokens
    # (gen_sequence steps: the delay pattern makes step t of codebook k token t - k; the per-step logits are what matters)
    same = (lg_f.argmax(-1) == lg_s.argmax(-1)).all(dim=1).all(dim=0)          # [steps]
    n_same = int(same.long().cumprod(0).sum())
    upto = min(n_same + 1, lg_f.shape[2])
    r = rel(lg_f[:, :, :upto].cpu(), lg_s[:, :, :upto].cpu())
    assert r < tol, f"folded vs separate launches: per-step logits rel-L2 {r} over the first {upto} of {lg_f.shape[2]} steps"
    # (bf16: two roundings of the same algebra part at the first near tie -- 3 steps in on the biased model; the logits gate above
    # covers the step of the parting itself)
    assert n_same >= (lg_f.shape[2] if wdt == torch.float32 else 2), f"the runs part after {n_same} steps"
    if wdt == torch.float32:
        assert torch.equal(toks_f, toks_s)
        sd = {k: v.detach().float().cpu() for k, v in lm.state_dict().items()}
        oc = olm.LMConfig(dim=256, num_heads=8, num_layers=4, n_q=4, card=2048, cross_attention=True)
        ref_t = olm.generate(sd, oc, prompt.cpu(), B, cross, max_gen_len=30, use_sampling=False)
        assert torch.equal(toks_f.cpu(), ref_t)


def test_sampled_tokens_reproduced_by_philox_replay():
    """The headline configuration SAMPLES (top-k 250, temperature 1): token-level parity of that path.  The device's draw is
    a counter-based exponential race (sample_kernel: Philox4x32-10 on (vocabulary index, sample * K + codebook, stream position),
    key = seed) -- the algorithm of torch.multinomial on a stream a host can replay.  Two stages, like the greedy parity of
    configs[1]: (1) the device generates 80 frames x 3 samples with sampling; (2) the ORACLE runs teacher-forced on the device's
    sequence (reference arithmetic: lm.py:391-399 CFG mix, :402-418 softmax / top-k incl. ties, utils.py:108-122), and
    oracle/sampler.py replays the draw of every (sample, codebook, position) on the oracle's probabilities: the tokens must be
    the device's.  A decision whose two best candidates lie within 1e-4 of each other (or whose support boundary is a near
    tie) may legitimately differ (f32 rounding of p, log, the division) and is reported, not failed -- none is expected."""
    import numpy as np
    from audiocraft_amd.models import builders
    from oracle import patterns as opat
    from oracle.sampler import race
    torch.manual_seed(0)
    B, T, K, card, top_k, seed = 3, 80, 4, 2048, 250, 0x5eed1234abcd
    cfg = dict(dim=256, num_heads=4, num_layers=4, n_q=K, card=card, hidden_scale=4, cfg_coef=3.0,
               conditioners={'description': {'kind': 't5', 'embedder': 'synthetic', 'dim': 64, 'length': 6}},
               fuser={'cross': ['description']})
    lm = builders.get_lm_model(cfg, 'cuda', torch.float32)
    with torch.no_grad():
        for k, p in lm.named_parameters():
            if 'norm' in k:
                p.add_(0.1 * torch.randn_like(p))
            if 'linears' in k:          # sharper logits: a top-250 filter that actually cuts mass
                p.mul_(4.0)
    sd = {k: v.detach().float().cpu() for k, v in lm.state_dict().items()}
    oc = olm.LMConfig(dim=256, num_heads=4, num_layers=4, n_q=K, card=card, cross_attention=True)
    g = torch.Generator().manual_seed(11)
    cross = torch.randn(2 * B, 6, 256, generator=g)
    cross[B:] = 0
    ct = {'description': (cross.cuda(), torch.ones(2 * B, 6, dtype=torch.int64).cuda())}
    toks = lm.generate(None, [], num_samples=B, max_gen_len=T, use_sampling=True, temp=1.0, top_k=top_k, seed=seed,
                       condition_tensors=ct, check=True).cpu()
    assert toks.shape == (B, K, T) and toks.min() >= 0 and toks.max() < card
    again = lm.generate(None, [], num_samples=B, max_gen_len=T, use_sampling=True, temp=1.0, top_k=top_k, seed=seed,
                        condition_tensors=ct).cpu()
    assert torch.equal(toks, again)                      # same seed, same tokens
    # stage 2: teacher-forced oracle on the device's sequence
    seq, mask = opat.build_pattern_sequence(toks, card)  # [B, K, S] with the special token where the pattern has no code
    S = seq.shape[-1]
    logits = olm.lm_forward(sd, oc, torch.cat([seq, seq], dim=0), cross)     # [2B, K, S, card]; step s predicts position s + 1
    mixed = olm.cfg_mix(logits, 3.0)
    probs = torch.softmax(mixed / 1.0, dim=-1).numpy()
    checked = near = 0
    wrong = []
    for offset in range(1, S):
        gpos = offset - 1                                # stream position of the step that fills sequence position `offset`
        for b in range(B):
            for k in range(K):
                if not bool(mask[k, offset]):
                    continue
                tok, margin, boundary = race(probs[b, k, offset - 1], top_k, b * K + k, gpos, seed)
                checked += 1
                if tok != int(seq[b, k, offset]):
                    if margin < 1e-4 or boundary:
                        near += 1
                    else:
                        wrong.append((b, k, offset, tok, int(seq[b, k, offset]), margin))
    print(f"philox replay: {checked} sampled decisions over {S - 1} positions x {B} samples, {near} near ties, {len(wrong)} wrong")
    assert checked >= 200 * K and not wrong, wrong[:5]
    assert near <= 2
    # the samples are not degenerate: a different seed gives other tokens, and the draws are not the arg-max
    other = lm.generate(None, [], num_samples=B, max_gen_len=T, use_sampling=True, temp=1.0, top_k=top_k, seed=seed + 1,
                        condition_tensors=ct).cpu()
    assert not torch.equal(toks, other)
    greedy_frac = float((torch.from_numpy(probs[:B, :, :S - 1].argmax(-1)) == seq[:, :, 1:]).float().mean())
    assert greedy_frac < 0.9, greedy_frac


def test_lm_sampling_generate_is_well_formed():
    from audiocraft_amd.models import builders
    lm = builders.get_debug_lm_model('cuda')
    from audiocraft_amd.modules.conditioners import ConditioningAttributes
    conds = [ConditioningAttributes(text={'description': 'a b c'}), ConditioningAttributes(text={'description': None})]
    calls = []
    toks = lm.generate(None, conds, max_gen_len=30, use_sampling=True, top_k=50, seed=1, check=True,
                       callback=lambda i, n: calls.append((i, n)))
    assert toks.shape == (2, 4, 30) and toks.min() >= 0 and toks.max() < 400
    assert calls[0] == (1, 33) and calls[-1] == (33, 33)
    toks2 = lm.generate(None, conds, max_gen_len=30, use_sampling=True, top_k=50, seed=1)
    assert torch.equal(toks, toks2)
    toks3 = lm.generate(None, conds, max_gen_len=30, use_sampling=True, top_k=50, seed=2)
    assert not torch.equal(toks, toks3)


_FUSED_AB = r"""
import sys, torch
from audiocraft_amd import _C
from audiocraft_amd.models import builders
torch.manual_seed(0)
cross, B, out = sys.argv[1] == '1', int(sys.argv[2]), sys.argv[3]
cfgd = dict(dim=1536, num_heads=24, num_layers=2, n_q=4, card=512, hidden_scale=4, cfg_coef=3.0)
if cross:
    cfgd.update(conditioners={'description': {'kind': 't5', 'embedder': 'synthetic', 'dim': 48, 'length': 5}}, fuser={'cross': ['description']})
lm = builders.get_lm_model(cfgd, 'cuda', torch.bfloat16)
kw = dict(num_samples=B, max_gen_len=330, use_sampling=True, top_k=50, seed=77, return_logits=True)
if cross:
    src = torch.randn(2 * B, 5, 1536, generator=torch.Generator().manual_seed(3))
    src[B:] = 0
    kw['condition_tensors'] = {'description': (src.cuda(), torch.ones(2 * B, 5, dtype=torch.int64).cuda())}
prompt = torch.randint(0, 512, (B, 4, 70), generator=torch.Generator().manual_seed(4)).cuda()
n0 = _C.qkv_attn_launches()
t_a, l_a = lm.generate(None, [], **kw)
t_b, l_b = lm.generate(prompt, [], **dict(kw, max_gen_len=150))
err = int(lm._run['hand_err'][0]) if 'hand_err' in lm._run else 0
armed = bool((lm._run['qkv_hand'] == lm.HAND_SENTINEL).all()) if 'qkv_hand' in lm._run else True
torch.save({'t_a': t_a.cpu(), 'l_a': l_a.cpu(), 't_b': t_b.cpu(), 'l_b': l_b.cpu(), 'fused': _C.qkv_attn_launches() - n0, 'err': err, 'armed': armed}, out)
"""


@pytest.mark.parametrize('cross,B', [(True, 8), (True, 3), (False, 16), (False, 5), (True, 16), (False, 27)])
def test_fused_qkv_attention_launch_is_bit_identical(cross, B, tmp_path):
    """acmi_lm_state.qkv_hand (0.2.0): the decode step's QKV GEMM and the self-attention that consumes it as ONE launch with a
    per-(row, head) sentinel hand-off (csrc/acmi_attn_fused.h; reference op sequence transformer.py:362-399, 412-414).  The
    attention's arithmetic is the separate kernel's in the same order, and with the SAME split of K over the GEMM's waves
    (ACMI_LIN_WIDE=0: 16-feature workgroups of 4 waves, what the fused launch runs) tokens AND logits are bit-identical --
    sampled, 330 frames (contexts crossing the 64-position chunks and the second round that is staged in LDS), MusicGen-medium's
    width and head count, with and without cross-attention (the paired and the plain out-projection layouts), 16 / 6 / 16 / 5
    rows in one 16-row block and 32 / 27 rows in two, and behind a prompt (prefill, then decode).  One subprocess per mode: the switches are read once per process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for fused in ('0', '1'):
        out = str(tmp_path / f'ab{fused}.pt')
        env = dict(os.environ, ACMI_QKV_ATTN=fused, ACMI_QKV_ATTN_ROWS='32', ACMI_LIN_WIDE='0', PYTHONPATH=root)
        r = subprocess.run([sys.executable, '-c', _FUSED_AB, '1' if cross else '0', str(B), out], env=env, cwd=root, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[fused] = torch.load(out)
    assert res['0']['fused'] == 0 and res['1']['fused'] > 0, (res['0']['fused'], res['1']['fused'])   # the one-launch form did run
    assert res['1']['err'] == 0 and res['1']['armed']
    for key in ('t_a', 'l_a', 't_b', 'l_b'):
        assert torch.equal(res['0'][key], res['1'][key]), key


def test_two_host_threads_generate_serialised():
    """Concurrency is a DEFINED behaviour (audiocraft_amd/_C.py, device_lock): two host threads, each with its own LM replica
    and its own HIP stream, call generate / codec decode at the same time; the process-wide lock serialises them around
    hipGraph capture + replay (global capture mode: one thread's capture would otherwise make the other's synchronisations
    illegal), nothing raises, and every thread's tokens and audio equal what it computes alone.  A third thread hammers the
    codec meanwhile (another graph-capturing entry)."""
    import threading
    from audiocraft_amd import _C
    from audiocraft_amd.models import builders
    from audiocraft_amd.modules.conditioners import ConditioningAttributes
    lms = [builders.get_debug_lm_model('cuda') for _ in range(2)]
    codec = builders.get_debug_compression_model('cuda')
    conds = [ConditioningAttributes(text={'description': 'a b c'}), ConditioningAttributes(text={'description': 'd e'})]
    kw = dict(max_gen_len=60, use_sampling=True, top_k=50)
    alone = [lm.generate(None, conds, seed=11 + i, **kw).cpu() for i, lm in enumerate(lms)]
    K = codec.num_codebooks
    codes = torch.randint(0, codec.cardinality, (2, K, 50), device='cuda', generator=torch.Generator('cuda').manual_seed(3))
    codec.decode(codes)                       # first sight of the shape (eager); the next call captures
    wav_alone = codec.decode(codes).cpu()
    streams = [torch.cuda.Stream() for _ in range(3)]
    got, wavs, errors = [None, None], [], []
    start = threading.Barrier(3)

    def work(i):
        try:
            start.wait()
            with torch.cuda.stream(streams[i]):
                for _ in range(4):   # several generates each: captures and replays of the two threads interleave
                    t = lms[i].generate(None, conds, seed=11 + i, **kw)
                streams[i].synchronize()
            got[i] = t.cpu()
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    def work_codec():
        try:
            start.wait()
            with torch.cuda.stream(streams[2]):
                for _ in range(12):
                    codec.__dict__.pop('_graphs', None)     # force eager + capture + replay again and again
                    codec.decode(codes)
                    codec.decode(codes)
                    wavs.append(codec.decode(codes).cpu())
        except Exception as e:   # noqa: BLE001
            errors.append(e)
    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)] + [threading.Thread(target=work_codec)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "a thread is stuck behind the device lock"
    assert not errors, errors
    assert torch.equal(got[0], alone[0]) and torch.equal(got[1], alone[1])
    assert len(wavs) == 12 and all(torch.equal(w, wav_alone) for w in wavs)
    assert _C.device_lock.acquire(blocking=False)    # released by everyone
    _C.device_lock.release()


def test_compute_predictions_matches_oracle():
    """LMModel.compute_predictions (lm.py:270-321): logits re-aligned with the codes + validity mask."""
    cfg, sd, a = load_golden('lm_text')
    lm = build_lm(cfg, sd)
    ones = torch.ones(a['cross_src'].shape[:2], dtype=torch.int64)
    ct = {'description': (a['cross_src'].cuda(), ones.cuda())}
    g = torch.Generator().manual_seed(9)
    codes = torch.randint(0, cfg['card'], (6, 4, 7), generator=g)
    out = lm.compute_predictions(codes.cuda(), [], ct)
    assert out.logits.shape == (6, 4, 7, cfg['card']) and out.mask.shape == (6, 4, 7)
    seq, mask = __import__('oracle.patterns', fromlist=['x']).build_pattern_sequence(codes, cfg['card'])
    ref = olm.lm_forward(sd, lm_cfg(cfg), seq[..., :8], a['cross_src'])   # valid steps only: T + 1
    for q in range(4):       # position t of codebook q is predicted by the model output at step t + q (delay
        for t in range(7):   # pattern); the output of the last valid step has no target and is dropped
            if t + q < 7:
                assert out.mask[0, q, t]
                assert torch.allclose(out.logits[:, q, t].cpu(), ref[:, q, t + q], atol=1e-4, rtol=1e-3)
            else:
                assert not out.mask[0, q, t] and torch.isnan(out.logits[:, q, t]).all()


def test_musicgen_small_architecture_greedy_parity():
    """BASELINE.json configs[1] exactly as SURVEY.md section 8(d) states it: MusicGen-small architecture (d = 1024, 24
    layers, 16 heads, cross-attention), 1 prompt x 10 s => T = 500, **503 autoregressive steps**, greedy, cfg_coef 3,
    synthetic text condition of 12 rows, fp32 mode.  Gates: tokens identical to `oracle.lm.generate` (free running),
    CFG-mixed logits of all 503 steps rel-L2 <= 1e-4.

    A free-running greedy comparison over 2012 arg-max decisions is only well posed where no decision is a near tie, so the
    test is two-stage: (1) the oracle, teacher-forced with the DEVICE's tokens (one batch forward: batch == streaming),
    gives logits for every step -- rel-L2 gate, and every device token must be the oracle's arg-max unless the oracle's
    top-2 gap at that decision is below twice the measured logit error of that step (a near tie, reported); (2) the
    oracle's own free-running generate must give identical tokens up to the first such near tie (all 500 frames when
    there is none)."""
    from audiocraft_amd.models import builders
    from oracle import patterns as opat
    torch.manual_seed(0)
    lm = builders.get_lm_model(builders.musicgen_lm_cfg('small', text_len=12), 'cuda', torch.float32)
    sd = {k: v.detach().float().cpu() for k, v in lm.state_dict().items()}
    oc = olm.LMConfig(dim=1024, num_heads=16, num_layers=24, n_q=4, card=2048, cross_attention=True)
    g = torch.Generator().manual_seed(2)
    cross = torch.randn(2, 12, 1024, generator=g)
    cross[1:] = 0
    ct = {'description': (cross.cuda(), torch.ones(2, 12, dtype=torch.int64).cuda())}
    T = 500
    toks, lg = lm.generate(None, [], num_samples=1, max_gen_len=T, use_sampling=False, condition_tensors=ct,
                           return_logits=True, check=True)
    toks, lg = toks.cpu(), lg.cpu()                     # [1, 4, 500], [1, 4, 503, 2048]
    assert lg.shape[2] == T + 3
    # (1) teacher-forced oracle on the device's tokens
    seq, mask = opat.build_pattern_sequence(toks, 2048)                # [1, 4, 504]; step s is predicted by output s - 1
    S = seq.shape[-1]
    pair = torch.cat([seq, seq], dim=0)[..., :S - 1]
    ref = olm.cfg_mix(olm.lm_forward(sd, oc, pair, cross), 3.0)        # [1, 4, 503, 2048]
    r = rel(lg, ref)
    print(f"[parity] configs[1] small fp32, 503 steps: CFG logits rel-L2 {r:.3e}")
    assert r < 1e-4, f"logits rel-L2 {r}"
    err = (lg - ref).abs().amax(dim=-1)[0]                               # [4, 503] max abs error per decision
    top2 = ref[0].topk(2, dim=-1).values                                 # [4, 503, 2]
    gap = top2[..., 0] - top2[..., 1]
    dev_arg = lg[0].argmax(dim=-1)
    ref_arg = ref[0].argmax(dim=-1)
    valid = mask[:, 1:S]                                                 # decisions that are written back (not special)
    differ = (dev_arg != ref_arg) & valid
    near = gap <= 2.0 * err
    assert not (differ & ~near).any(), "a device token is not the oracle's arg-max and the decision is not a near tie"
    nv = (near & valid).any(dim=0)                                       # [503]
    first_tie = int(nv.float().argmax()) if bool(nv.any()) else None
    print(f"[parity] configs[1]: {int(valid.sum())} arg-max decisions, {int(differ.sum())} differ (all near ties), "
          f"min top-2 gap {float(gap[valid].min()):.3e}, max logit err {float(err.max()):.3e}, first near tie at step {first_tie}")
    # (2) free-running oracle
    ref_t = olm.generate(sd, oc, None, 1, cross, max_gen_len=T, use_sampling=False)
    if first_tie is None:
        assert torch.equal(toks, ref_t), "greedy tokens differ from the oracle's free-running generate"
    else:
        upto = max(first_tie - 3, 0)          # frames fully decided before the near-tie step (max delay 3)
        assert torch.equal(toks[..., :upto], ref_t[..., :upto])


# ------------------------------------------------------------------------------------------ CFG modes 2 and 3

def test_lm_two_step_cfg_vs_reference_golden():
    """two_step_cfg (reference lm.py:377-387, 498-505): conditional / unconditional passes with their own condition length
    (5 vs 1) -- on the device the two row groups of one step with per-row cross-attention lengths; the mix uses the
    MODEL's cfg_coef (the golden run passed cfg_coef=7 on purpose)."""
    cfg, sd, a = load_golden('lm_two_step')
    lm = build_lm(cfg, sd)
    ct = {'description': (a['cross_src'].cuda(), torch.ones(a['cross_src'].shape[:2], dtype=torch.int64).cuda())}
    nt = {'description': (a['null_cross_src'].cuda(), torch.ones(a['null_cross_src'].shape[:2], dtype=torch.int64).cuda())}
    toks, lg = lm.generate(None, [], num_samples=3, max_gen_len=10, use_sampling=False, condition_tensors=(ct, nt),
                           cfg_coef=7.0, return_logits=True, check=True)
    assert torch.equal(toks.cpu(), a['greedy_tokens'])
    ref = a['uncond_step_logits'] + (a['cond_step_logits'] - a['uncond_step_logits']) * cfg['cfg_coef']
    assert rel(lg.cpu(), ref) < 1e-4
    # a longer null source than the conditional one is just as valid (lengths are per row, the buffer is the max)
    nt2 = {'description': (torch.zeros(3, 8, cfg['dim']).cuda(), torch.ones(3, 8, dtype=torch.int64).cuda())}
    toks2 = lm.generate(None, [], num_samples=3, max_gen_len=10, use_sampling=False, condition_tensors=(ct, nt2))
    assert torch.equal(toks2.cpu(), a['greedy_tokens'])   # an all-zero source contributes exactly 0 at any length


def test_lm_two_step_cfg_unequal_prepend_vs_reference_golden():
    """two_step_cfg on a prepend fuser (reference lm.py:378-390: the two passes keep their own streaming states): the
    conditional stream starts with 5 prepended rows, the unconditional one with 1.  On the device both are row groups of ONE
    step, the shorter stream left-padded (acmi_lm_state.row_off): own positions for the sinusoidal embedding, own first key
    for the self-attention.  Greedy tokens and step logits against the unmodified reference; also with a 4-token prompt."""
    cfg, sd, a = load_golden('lm_two_step_prepend')
    lm = build_lm(cfg, sd)
    ones = lambda t: torch.ones(t.shape[:2], dtype=torch.int64).cuda()  # noqa: E731
    ct = {'description': (a['prepend_src'].cuda(), ones(a['prepend_src']))}
    nt = {'description': (a['null_prepend_src'].cuda(), ones(a['null_prepend_src']))}
    toks, lg = lm.generate(None, [], num_samples=3, max_gen_len=11, use_sampling=False, condition_tensors=(ct, nt),
                           cfg_coef=7.0, return_logits=True, check=True)
    assert torch.equal(toks.cpu(), a['greedy_tokens'])
    ref = a['uncond_step_logits'] + (a['cond_step_logits'] - a['uncond_step_logits']) * cfg['cfg_coef']
    assert rel(lg.cpu(), ref) < 1e-4
    toks_p = lm.generate(a['prompt'].cuda(), [], max_gen_len=12, use_sampling=False, condition_tensors=(ct, nt))
    assert torch.equal(toks_p.cpu(), a['greedy_tokens_prompt'])
    # the roles swapped (the longer prefix in the unconditional pass) and equal lengths take the same path
    toks_eq = lm.generate(None, [], num_samples=3, max_gen_len=11, use_sampling=False, condition_tensors=(ct, ct))
    both = {'description': (torch.cat([a['prepend_src'], a['prepend_src']]).cuda(), ones(torch.cat([a['prepend_src'], a['prepend_src']])))}
    toks_one = lm.generate(None, [], num_samples=3, max_gen_len=11, use_sampling=False, condition_tensors=both,
                           cfg_coef=cfg['cfg_coef'])
    assert torch.equal(toks_eq, toks_one)


def test_lm_two_step_cfg_unequal_prepend_on_a_rotary_model_vs_reference_golden():
    """The same two left-padded streams on a ROTARY model (round 5; raised NotImplementedError before): the rotary position
    of a row is its OWN position (stream position - row_off), as in the reference, where each pass has its own streaming
    state (lm.py:378-390, transformer.py:300-313).  Golden from the unmodified reference (`make_rope_golden.py two_step`)."""
    cfg, sd, a = load_golden('lm_rope_two_step_prepend')
    lm = build_lm(cfg, sd)
    ones = lambda t: torch.ones(t.shape[:2], dtype=torch.int64).cuda()  # noqa: E731
    ct = {'description': (a['prepend_src'].cuda(), ones(a['prepend_src']))}
    nt = {'description': (a['null_prepend_src'].cuda(), ones(a['null_prepend_src']))}
    toks, lg = lm.generate(None, [], num_samples=3, max_gen_len=11, use_sampling=False, condition_tensors=(ct, nt),
                           return_logits=True, check=True)
    assert torch.equal(toks.cpu(), a['greedy_tokens'])
    ref = a['uncond_step_logits'] + (a['cond_step_logits'] - a['uncond_step_logits']) * cfg['cfg_coef']
    assert rel(lg.cpu(), ref) < 1e-4
    toks_p = lm.generate(a['prompt'].cuda(), [], max_gen_len=12, use_sampling=False, condition_tensors=(ct, nt))
    assert torch.equal(toks_p.cpu(), a['greedy_tokens_prompt'])
    # with a bounded context the rotary lag would differ per row group: still refused, with a message that says so
    lm_pc = build_lm(dict(cfg, past_context=6), sd)
    with pytest.raises(NotImplementedError, match='past_context'):
        lm_pc.generate(None, [], num_samples=3, max_gen_len=11, use_sampling=False, condition_tensors=(ct, nt))


def test_lm_double_cfg_vs_reference_golden(prefill_mode):
    """cfg_coef_beta (MusicGen-Style double CFG, lm.py:362-376): 3B rows [text + wav; wav; null]."""
    cfg, sd, a = load_golden('lm_double_cfg')
    lm = build_lm(cfg, sd)
    P, Lc = cfg['P'], cfg['Lc']
    pre = a['prepend_src'].cuda()
    ones = lambda n: torch.ones(pre.shape[0], n, dtype=torch.int64).cuda()  # noqa: E731
    ct = {'description': (pre[:, P:], ones(Lc)), 'self_wav': (pre[:, :P], ones(P))}
    toks, lg = lm.generate(None, [], num_samples=2, max_gen_len=9, use_sampling=False, condition_tensors=ct,
                           cfg_coef_beta=cfg['cfg_coef_beta'], return_logits=True, check=True)
    assert torch.equal(toks.cpu(), a['greedy_tokens'])
    assert rel(lg.cpu(), olm.double_cfg_mix(a['step_logits'], cfg['cfg_coef'], cfg['cfg_coef_beta'])) < 1e-4


def test_cfg_modes_from_attributes_midsize():
    """The three modes through `conditions` (attributes -> provider -> fuser) on a mid-size model vs the oracle."""
    from audiocraft_amd.models import builders
    from audiocraft_amd.modules.conditioners import ConditioningAttributes, WavCondition
    torch.manual_seed(0)
    cfg = dict(dim=128, num_heads=4, num_layers=3, n_q=4, card=512, hidden_scale=4, cfg_coef=2.5,
               conditioners={'description': {'kind': 't5', 'embedder': 'synthetic', 'dim': 32, 'length': 5},
                             'self_wav': {'kind': 'chroma', 'embedder': 'synthetic', 'n_frames': 9}},
               fuser={'prepend': ['self_wav', 'description']})
    lm = builders.get_lm_model(cfg, 'cuda', torch.float32)
    cw = lm.condition_provider.conditioners['self_wav']
    cw.chroma_len = 9

    def chroma_of(x):   # content dependent (the stock synthetic embedder draws by row index): same wav -> same frames
        cls = (x.wav.reshape(x.wav.shape[0], -1)[:, :9 * 7:7].abs() * 1000).long() % 12
        e = torch.nn.functional.one_hot(cls, 12).float()
        return torch.where((x.length.cpu() == 0).view(-1, 1, 1), torch.zeros_like(e), e.cpu())
    cw.embedder = chroma_of
    sd = {k: v.detach().float().cpu() for k, v in lm.state_dict().items()}
    oc = olm.LMConfig(dim=128, num_heads=4, num_layers=3, n_q=4, card=512, cross_attention=False, cfg_coef=2.5)
    conds = []
    for i in range(2):
        c = ConditioningAttributes(text={'description': f'x{i}'})
        c.wav['self_wav'] = WavCondition(torch.randn(1, 1, 640), torch.tensor([640]), [32000], [None], [0.])
        conds.append(c)
    # double CFG: [text + wav; wav; null]
    ct3 = lm._cfg_condition_tensors(conds, cfg_coef_beta=4.0)
    pre3, _ = lm.fuser.fuse(ct3)
    assert pre3.shape[0] == 6
    toks = lm.generate(None, conds, max_gen_len=12, use_sampling=False, cfg_coef_beta=4.0, check=True)
    ref = olm.generate(sd, oc, None, 2, None, pre3.float().cpu(), max_gen_len=12, use_sampling=False, cfg_coef_beta=4.0)
    assert torch.equal(toks.cpu(), ref)
    # the middle group really is "wav kept, text dropped"
    d = ct3['description'][0]
    assert (d[2:4] == 0).all() and (d[4:] == 0).all() and not (d[:2] == 0).all()
    w = ct3['self_wav'][0]
    assert torch.equal(w[:2], w[2:4]) and not torch.equal(w[:2], w[4:])
    # two-step with equal prepend lengths == one-step (same rows, same conditions; the reference's two orders of
    # evaluation differ only in batching)
    t2 = lm.generate(None, conds, max_gen_len=12, use_sampling=False, two_step_cfg=True, check=True)
    t1 = lm.generate(None, conds, max_gen_len=12, use_sampling=False, check=True)
    assert torch.equal(t1, t2)


# ------------------------------------------------------------------------------------------ streaming protocol (a10)

def test_streaming_protocol_matches_batch_and_rewinds():
    """StreamingModule protocol on LMModel (reference streaming.py:20-119, tests/modules/test_transformer.py:133-161):
    chunked streaming forwards == one forward over the whole sequence; get / set_streaming_state rewinds a stream;
    the key names are the reference's; leaving the context resets the state."""
    cfg, sd, a = load_golden('lm_text')
    lm = build_lm(cfg, sd)
    ones = torch.ones(a['cross_src'].shape[:2], dtype=torch.int64)
    ct = {'description': (a['cross_src'].cuda(), ones.cuda())}
    seq = a['tf_sequence'].cuda()                       # [6, 4, 9]
    full = lm(seq, [], ct)
    assert rel(full.cpu(), a['tf_logits']) < 1e-4        # the reference's batch forward
    assert lm.get_streaming_state() == {}
    with lm.streaming():
        parts = [lm(seq[..., 0:1], [], ct), lm(seq[..., 1:4], [], ct)]
        state = lm.get_streaming_state()
        keys = set(state.keys())
        assert {'transformer.offsets', 'fuser.offsets', 'transformer.layers.0.self_attn.past_keys',
                'transformer.layers.1.self_attn.past_values', 'transformer.layers.0.self_attn.offset'} <= keys
        H, hd = cfg['num_heads'], cfg['dim'] // cfg['num_heads']
        assert state['transformer.layers.0.self_attn.past_keys'].shape == (6, H, 4, hd)   # time_dim 2, like the torch backend
        assert int(state['transformer.offsets'][0]) == 4 and state['transformer.offsets'].shape == (6,)
        tail_a = lm(seq[..., 4:9], [], ct)
        assert int(lm.get_streaming_state()['transformer.offsets'][0]) == 9
        lm.set_streaming_state(state)                   # rewind to step 4 and replay the tail
        tail_b = lm(seq[..., 4:9], [], ct)
        assert torch.equal(tail_a, tail_b)
        streamed = torch.cat(parts + [tail_a], dim=2)
    assert torch.equal(streamed, full)                   # streaming == batch (same kernels, same order: bit exact)
    assert lm.get_streaming_state() == {}                # reset on exit
    # a state restored from COPIES (not views of the live caches) works too
    with lm.streaming():
        lm(seq[..., 0:4], [], ct)
        saved = {k: v.clone() for k, v in lm.get_streaming_state().items()}
        lm(seq[..., 4:6], [], ct)
        lm.set_streaming_state(saved)
        tail_c = lm(seq[..., 4:9], [], ct)
    assert torch.equal(tail_c, tail_a)


# ------------------------------------------------------------------------------------------ reference-written checkpoints (f1)

def test_models_loaded_from_reference_written_checkpoints():
    """loaders.load_lm_model / load_compression_model on files written by the reference's own utils/export.py
    (tests/golden/ckpt_ref, make_ckpt_golden.py) -> greedy tokens / waveform identical to the reference's."""
    import os
    import numpy as np
    from audiocraft_amd.models import loaders
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ckpt_ref')
    # text LM (xp.cfg with conditioners.args) + its codec
    cfg, _, a = load_golden('lm_text')
    lm = loaders.load_lm_model(os.path.join(root, 'text'), device='cuda', weight_dtype=torch.float32)
    ct = {'description': (a['cross_src'].cuda(), torch.ones(a['cross_src'].shape[:2], dtype=torch.int64).cuda())}
    toks = lm.generate(None, [], num_samples=3, max_gen_len=12, use_sampling=False, condition_tensors=ct, check=True)
    assert torch.equal(toks.cpu(), a['greedy_tokens'])
    codec = loaders.load_compression_model(os.path.join(root, 'text'), device='cuda')
    _, _, c = load_golden('codec_noncausal')
    assert (codec.decode(c['codes'].cuda()).cpu() - c['decoded']).abs().max().item() < 1e-4
    # melody LM (chroma2music conditioners, third-party buffer in the checkpoint)
    cfg, _, a = load_golden('lm_melody')
    lm = loaders.load_lm_model(os.path.join(root, 'melody'), device='cuda', weight_dtype=torch.float32)
    P, Lc = cfg['P'], cfg['Lc']
    pre = a['prepend_src'].cuda()
    ct = {'description': (pre[:, P:], torch.ones(pre.shape[0], Lc, dtype=torch.int64).cuda()),
          'self_wav': (pre[:, :P], torch.ones(pre.shape[0], P, dtype=torch.int64).cuda())}
    toks = lm.generate(None, [], num_samples=2, max_gen_len=9, use_sampling=False, condition_tensors=ct, check=True)
    assert torch.equal(toks.cpu(), a['greedy_tokens'])
    # nn.MultiheadAttention key layout
    exp = np.load(os.path.join(root, 'mha', 'expected.npz'))
    lm = loaders.load_lm_model(os.path.join(root, 'mha'), device='cuda', weight_dtype=torch.float32)
    cs = torch.from_numpy(exp['cross_src']).cuda()
    ct = {'description': (cs, torch.ones(cs.shape[:2], dtype=torch.int64).cuda())}
    toks = lm.generate(None, [], num_samples=3, max_gen_len=12, use_sampling=False, condition_tensors=ct, check=True)
    assert torch.equal(toks.cpu(), torch.from_numpy(exp['greedy_tokens']))


def test_t5_conditioner_real_huggingface_path(tmp_path):
    """T5Conditioner with `embedder=None`: the HuggingFace tokenizer + T5EncoderModel path (reference conditioners.py:
    422-515), exercised for real on a tiny random-init T5 saved to disk (no released weights exist offline): batch
    padding to the longest prompt, attention mask, empty strings -> zero mask -> exactly zero rows after output_proj."""
    import sentencepiece as spm
    from transformers import T5Config, T5EncoderModel, T5Tokenizer
    from audiocraft_amd.modules.conditioners import ConditioningAttributes, ConditioningProvider, T5Conditioner
    corpus = tmp_path / 'corpus.txt'
    corpus.write_text('\n'.join(['happy rock with electric guitar', 'sad slow piano ballad', 'energetic drum and bass',
                                 'lofi hip hop beat to relax', 'orchestral epic trailer music'] * 20))
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=str(tmp_path / 'spiece'), vocab_size=40,
                                   pad_id=0, eos_id=1, unk_id=2, bos_id=-1, hard_vocab_limit=False)
    sp = spm.SentencePieceProcessor(model_file=str(tmp_path / 'spiece.model'))
    t5dir = tmp_path / 't5-tiny'
    tok = T5Tokenizer([(sp.id_to_piece(i), sp.get_score(i)) for i in range(sp.get_piece_size())], extra_ids=0)
    tok.save_pretrained(str(t5dir))
    torch.manual_seed(0)
    enc = T5EncoderModel(T5Config(vocab_size=64, d_model=32, d_kv=8, d_ff=64, num_layers=2, num_heads=4)).eval()
    enc.save_pretrained(str(t5dir))
    cond = T5Conditioner(str(t5dir), output_dim=48, device='cuda', dim=32)
    provider = ConditioningProvider({'description': cond}, device='cuda')
    texts = ['sad slow piano ballad', 'rock', None]
    attrs = [ConditioningAttributes(text={'description': t}) for t in texts]
    emb, mask = provider(provider.tokenize(attrs))['description']
    # reference semantics, computed directly with HuggingFace + torch
    ins = tok(['sad slow piano ballad', 'rock', ''], return_tensors='pt', padding=True)
    ref_mask = ins['attention_mask'].clone()
    ref_mask[2] = 0
    with torch.no_grad():
        hid = enc(**ins).last_hidden_state
    W, bvec = cond.output_proj.weight.detach().cpu(), cond.output_proj.bias.detach().cpu()
    ref = (hid @ W.T + bvec) * ref_mask.unsqueeze(-1)
    assert torch.equal(mask.cpu(), ref_mask) and mask.shape[1] == ins['input_ids'].shape[1]
    assert torch.allclose(emb.cpu(), ref, atol=2e-5, rtol=1e-4), (emb.cpu() - ref).abs().max()
    assert (emb[2] == 0).all() and (emb[1, int(ref_mask[1].sum()):] == 0).all()   # null prompt and padding: exact zeros


# ------------------------------------------------------------------------------------------ prefill as one forward

@pytest.mark.parametrize('wdt,tol', [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize('variant', ['text', 'prefix', 'rope_window'])
def test_prefill_one_forward_matches_oracle_and_chunk_path(wdt, tol, variant, monkeypatch):
    """The MFMA-tiled prefill (one forward over prompt + prefix: tiled GEMMs with the QKV scatter epilogue, causal prefill
    attention over the K cache and the time-minor V) at a mid-size geometry (d 256, 4 heads of 64, 4 layers), 3 samples
    (6 CFG rows), a 37-token prompt (npos_pad 48: pad rows, a ragged last query block): the first decode steps'
    CFG logits against the oracle teacher-forced with the device's tokens, and against the 8-positions-per-call path.
    'prefix': no cross-attention, 21 prepended rows; 'rope_window': rotary positions + past_context 20 (< prompt:
    the lagging rotary offsets of the reference's long first call)."""
    from audiocraft_amd.models import builders
    from oracle import patterns as opat
    torch.manual_seed(0)
    cross = variant != 'prefix'
    extra = dict(positional_embedding='rope', past_context=20, xpos=True) if variant == 'rope_window' else {}
    conds = {'description': {'kind': 't5', 'embedder': 'synthetic', 'dim': 64, 'length': 6}}
    if not cross:
        conds['self_wav'] = {'kind': 'chroma', 'embedder': 'synthetic', 'n_frames': 15}
    cfg = dict(dim=256, num_heads=4, num_layers=4, n_q=4, card=2048, hidden_scale=4, cfg_coef=3.0, conditioners=conds,
               fuser={'cross': ['description']} if cross else {'prepend': ['self_wav', 'description']}, **extra)
    lm = builders.get_lm_model(cfg, 'cuda', wdt)
    with torch.no_grad():
        for k, prm in lm.named_parameters():
            if 'norm' in k:
                prm.add_(0.1 * torch.randn_like(prm))
    sd = {k: v.detach().float().cpu() for k, v in lm.state_dict().items()}
    if wdt == torch.bfloat16:
        sd = {k: (v.bfloat16().float() if v.dim() == 2 and 'output_proj' not in k else v) for k, v in sd.items()}
    oc = olm.LMConfig(dim=256, num_heads=4, num_layers=4, n_q=4, card=2048, cross_attention=cross, **extra)
    B, T0, T = 3, 37, 44
    g = torch.Generator().manual_seed(7)
    if cross:
        src = torch.randn(2 * B, 6, 256, generator=g)
        src[B:] = 0
        ct = {'description': (src.cuda(), torch.ones(2 * B, 6, dtype=torch.int64).cuda())}
        cross_src, prepend = src, None
    else:
        pre = torch.randn(2 * B, 21, 256, generator=g)
        ones = lambda n: torch.ones(2 * B, n, dtype=torch.int64).cuda()   # noqa: E731
        # dict order as the reference's ConditioningProvider yields it: every 'prepend' condition goes IN FRONT of what has
        # been fused so far (conditioners.py:1730-1741) => rows [self_wav; description; tokens]
        ct = {'description': (pre[:, 15:].cuda(), ones(6)), 'self_wav': (pre[:, :15].cuda(), ones(15))}
        cross_src, prepend = None, pre
    prompt = torch.randint(0, 2048, (B, 4, T0), generator=g)
    out = {}
    for mode in ('big', 'chunk'):
        monkeypatch.setenv('ACMI_PREFILL', mode)
        toks, lg = lm.generate(prompt.cuda(), [], max_gen_len=T, use_sampling=False, condition_tensors=ct,
                               return_logits=True, check=True)
        out[mode] = (toks.cpu(), lg.cpu())
    toks, lg = out['big']
    assert torch.equal(toks[..., :T0], prompt)
    # the first decode step's logits depend on nothing but the prefill: the two paths agree to rounding
    r_paths = rel(lg[:, :, 0], out['chunk'][1][:, :, 0])
    if variant == 'rope_window':
        # A first call longer than past_context makes the reference's later rotary positions lag (transformer.py:294-313),
        # which only its STREAMING evaluation shows: compare with the oracle's own generate (f32: free running, identical
        # tokens); in bf16 the two device paths are compared with each other (the chunk path is pinned to the reference
        # golden `lm_rope`, same situation)
        if wdt == torch.float32:
            ref_t, ref = olm.generate(sd, oc, prompt, B, cross_src, max_gen_len=T, use_sampling=False, return_logits=True)
            assert torch.equal(toks, ref_t)
            r = rel(lg, ref)
        else:
            r = rel(lg[:, :, 0], out['chunk'][1][:, :, 0])
        r_chunk = float('nan')
    else:
        # oracle, teacher-forced with the device's tokens (batch forward == the reference's multi-step first call + steps)
        seq, _ = opat.build_pattern_sequence(toks, 2048)
        S = seq.shape[-1]
        steps = lg.shape[2]
        pair = torch.cat([seq, seq], dim=0)[..., :S - 1]
        ref = olm.cfg_mix(olm.lm_forward(sd, oc, pair, cross_src, prepend), 3.0)[:, :, S - 1 - steps:]
        r = rel(lg, ref)
        r_chunk = rel(out['chunk'][1], ref) if torch.equal(out['chunk'][0], toks) else float('nan')
    print(f"[parity] prefill one-forward, {variant}, {wdt}: CFG logits vs oracle rel-L2 {r:.3e} (chunk path {r_chunk:.3e}), "
          f"first step big vs chunk {r_paths:.3e}")
    assert r < tol and r_paths < 2 * tol, (r, r_paths)
    if wdt == torch.float32:
        assert torch.equal(toks, out['chunk'][0])


@pytest.mark.parametrize('wdt,tol', [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
def test_lm_with_projection_biases_vs_oracle(wdt, tol, prefill_mode):
    """The reference's DEFAULT transformer configuration has biases everywhere (config/model/lm/default.yaml: bias_ff,
    bias_attn, bias_proj true; the released MusicGen checkpoints turn them off): in_proj / out_proj of self- and
    cross-attention (incl. the k / v biases of the cross-attention source, so that a null condition no longer means
    K = V = 0 and no row may be skipped), linear1 / linear2, the heads; with LayerScale on top.  Prompt + CFG generate
    through both prefill paths vs the oracle."""
    from audiocraft_amd.models import builders
    from oracle import patterns as opat
    torch.manual_seed(0)
    cfg = dict(dim=256, num_heads=4, num_layers=3, n_q=4, card=1024, hidden_scale=4, cfg_coef=3.0, bias_ff=True, bias_attn=True,
               bias_proj=True, layer_scale=0.5,
               conditioners={'description': {'kind': 't5', 'embedder': 'synthetic', 'dim': 64, 'length': 6}},
               fuser={'cross': ['description']})
    lm = builders.get_lm_model(cfg, 'cuda', wdt)
    with torch.no_grad():
        for k, prm in lm.named_parameters():
            if 'norm' in k or k.endswith('bias'):
                prm.add_(0.1 * torch.randn_like(prm))
    sd = {k: v.detach().float().cpu() for k, v in lm.state_dict().items()}
    assert 'transformer.layers.0.self_attn.out_proj.bias' in sd and 'transformer.layers.0.cross_attention.in_proj_bias' in sd
    assert 'transformer.layers.0.linear2.bias' in sd and 'linears.0.bias' in sd
    if wdt == torch.bfloat16:
        sd = {k: (v.bfloat16().float() if v.dim() == 2 and 'output_proj' not in k else v) for k, v in sd.items()}
    oc = olm.LMConfig(dim=256, num_heads=4, num_layers=3, n_q=4, card=1024, cross_attention=True)
    B, T0, T = 3, 9, 16
    g = torch.Generator().manual_seed(11)
    src = torch.randn(2 * B, 6, 256, generator=g)
    src[B:] = 0
    ct = {'description': (src.cuda(), torch.ones(2 * B, 6, dtype=torch.int64).cuda())}
    prompt = torch.randint(0, 1024, (B, 4, T0), generator=g)
    toks, lg = lm.generate(prompt.cuda(), [], max_gen_len=T, use_sampling=False, condition_tensors=ct,
                           return_logits=True, check=True)
    toks, lg = toks.cpu(), lg.cpu()
    seq, _ = opat.build_pattern_sequence(toks, 1024)
    S, steps = seq.shape[-1], lg.shape[2]
    ref = olm.cfg_mix(olm.lm_forward(sd, oc, torch.cat([seq, seq], 0)[..., :S - 1], src), 3.0)[:, :, S - 1 - steps:]
    r = rel(lg, ref)
    print(f"[parity] LM with every bias + LayerScale, {wdt}, prefill {prefill_mode}: CFG logits rel-L2 {r:.3e}")
    assert r < tol, r
    if wdt == torch.float32:
        ref_t = olm.generate(sd, oc, prompt, B, src, max_gen_len=T, use_sampling=False)
        assert torch.equal(toks, ref_t)


def test_release_master_weights_keeps_generating_and_state_dict():
    """Opt-in `release_master_weights()`: the big f32 matrices are freed once packed; generation is unchanged, state_dict()
    gives the matrices back from the packs (f32 packs: to rounding of the LayerNorm fold; bf16 packs: the bf16-rounded
    weights), load_state_dict restores full masters, moving a released model raises."""
    from audiocraft_amd.models import builders
    torch.manual_seed(0)
    cfg = dict(dim=128, num_heads=4, num_layers=2, n_q=4, card=256, hidden_scale=4, cfg_coef=3.0, layer_scale=0.7,
               conditioners={'description': {'kind': 't5', 'embedder': 'synthetic', 'dim': 32, 'length': 5}},
               fuser={'cross': ['description']})
    for wdt, tol in ((torch.float32, 1e-6), (torch.bfloat16, 6e-3)):
        lm = builders.get_lm_model(cfg, 'cuda', wdt)
        with torch.no_grad():
            for k, prm in lm.named_parameters():
                if 'norm' in k:
                    prm.add_(0.1 * torch.randn_like(prm))
        sd0 = {k: v.detach().clone() for k, v in lm.state_dict().items()}
        src = torch.randn(4, 5, 128, generator=torch.Generator().manual_seed(1))
        src[2:] = 0
        ct = {'description': (src.cuda(), torch.ones(4, 5, dtype=torch.int64).cuda())}
        t0 = lm.generate(None, [], num_samples=2, max_gen_len=10, use_sampling=False, condition_tensors=ct)
        before = torch.cuda.memory_allocated()
        lm.release_master_weights()
        assert torch.cuda.memory_allocated() < before
        t1 = lm.generate(None, [], num_samples=2, max_gen_len=10, use_sampling=False, condition_tensors=ct)
        assert torch.equal(t0, t1)
        sd1 = lm.state_dict()
        assert sd1.keys() == sd0.keys()
        for k, v in sd0.items():
            assert sd1[k].shape == v.shape, k
            err = rel(sd1[k].float().cpu(), v.float().cpu()) if v.numel() else 0.0
            assert err < tol, (k, err)
        with pytest.raises(RuntimeError, match='released'):
            lm.to('cuda', torch.float32) if False else lm._invalidate()
        lm.load_state_dict(sd0)
        assert all(p.numel() > 0 for p in lm.parameters())
        t2 = lm.generate(None, [], num_samples=2, max_gen_len=10, use_sampling=False, condition_tensors=ct)
        assert torch.equal(t0, t2)
