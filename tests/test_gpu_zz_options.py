"""-m gpu: the transformer / fuser options of the reference that no MusicGen release switches on -- kv_repeat,
qk_layer_norm (+ cross), the fuser's 'sum' / 'input_interpolate' methods and cross_attention_pos_emb
(config/model/lm/default.yaml:43-46; transformer.py:196-222, 358-400; conditioners.py:1733-1757) -- on the HIP path, against
goldens of the unmodified reference (tests/golden/make_options_golden.py) and the CPU oracle at a larger size.

(File name: sorts after the suites of the released configurations, which the driver's `-x` run therefore finishes first.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import load_golden  # noqa: E402
from oracle import lm as olm  # noqa: E402
from test_host_cpu import options_lm_cfg  # noqa: E402
from test_oracle_golden import lm_cfg, options_inputs  # noqa: E402


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def build(cfg, sd, wdt=torch.float32):
    from audiocraft_amd.models import builders
    lm = builders.get_lm_model(options_lm_cfg(cfg), 'cuda', wdt)
    lm.load_state_dict({k: v for k, v in sd.items()
                        if not (k.startswith('condition_provider.conditioners.') and '.description.' not in k)})
    return lm


def condition_tensors(name, a):
    ones = lambda t: torch.ones(t.shape[:2], dtype=torch.int64).cuda()  # noqa: E731
    ct = {'description': (a['cond_description'].cuda(), ones(a['cond_description']))}
    if name == 'lm_fuser_sum':      # the provider's dict order: description, genre, curve
        ct['genre'] = (a['cond_genre'].cuda(), ones(a['cond_genre']))
        ct['curve'] = (a['cond_curve'].cuda(), ones(a['cond_curve']))
    return ct


@pytest.mark.parametrize('M,d', [(1, 32), (6, 256), (48, 1536), (5, 2048), (3, 1000)])
def test_layer_norm_rows_vs_torch(M, d):
    """acmi_layer_norm_rows (the kernel behind qk_layer_norm) against torch's fp32 layer_norm: affine, plain, in place."""
    from audiocraft_amd import _C
    g = torch.Generator().manual_seed(M * 7 + d)
    x = (3.0 * torch.randn(M, d, generator=g) + 5.0).cuda()
    gamma, beta = (1.0 + 0.2 * torch.randn(d, generator=g)).cuda(), (0.3 * torch.randn(d, generator=g)).cuda()
    ref = torch.nn.functional.layer_norm(x.double(), (d,), gamma.double(), beta.double(), 1e-5)
    got = _C.layer_norm_rows(x, gamma, beta, 1e-5)
    assert (got.double() - ref).abs().max().item() < 2e-5
    ref0 = torch.nn.functional.layer_norm(x.double(), (d,), None, None, 1e-5)
    assert (_C.layer_norm_rows(x, None, None, 1e-5).double() - ref0).abs().max().item() < 2e-5
    y = x.clone()
    _C.layer_norm_rows(y, gamma, beta, 1e-5, out=y)
    assert torch.equal(y, got)
    with pytest.raises(_C.AcmiError):
        _C.layer_norm_rows(torch.zeros(2, 4096, device='cuda'), None, None)


@pytest.fixture(params=['big', 'chunk'])
def prefill_mode(request, monkeypatch):
    monkeypatch.setenv('ACMI_PREFILL', request.param)
    return request.param


@pytest.mark.parametrize('name', ['lm_kv_repeat', 'lm_qk_ln', 'lm_fuser_sum', 'lm_post_norm'])
def test_lm_options_vs_reference_golden(name, prefill_mode):
    """(lm_post_norm: norm_first=False, the reference's constructor default -- transformer.py:567-573, the cross-attention's
    query from the layer input, no out_norm -- through the decode step, the chunked and the one-forward prefill.)
    Teacher-forced logits, greedy tokens + per-step CFG logits without and with a 4-step prompt (the first call then spans
    several positions: the multi-position prefill, and the length an interpolated condition is resampled to), graph replay ==
    eager launches, bf16 packs."""
    cfg, sd, a = load_golden(name)
    lm = build(cfg, sd)
    ct = condition_tensors(name, a)
    logits = lm.forward_steps(a['tf_sequence'].cuda(), ct).cpu()
    r = rel(logits, a['tf_logits'])
    assert r < 1e-4, f"teacher-forced logits rel-L2 {r}"
    toks, lg = lm.generate(None, [], num_samples=3, max_gen_len=12, use_sampling=False, condition_tensors=ct,
                           return_logits=True, check=True)
    assert torch.equal(toks.cpu(), a['greedy_tokens'])
    assert rel(lg.cpu(), olm.cfg_mix(a['greedy_step_logits'], cfg['cfg_coef'])) < 1e-4
    toks, lg = lm.generate(a['prompt'].cuda(), [], max_gen_len=11, use_sampling=False, condition_tensors=ct, check=True,
                           return_logits=True)
    assert torch.equal(toks.cpu(), a['cont_tokens'])
    first = olm.cfg_mix(a['cont_first_logits'][:, :, -1:], cfg['cfg_coef'])      # the reference's first call, last step
    assert rel(lg[:, :, :1].cpu(), first) < 1e-4
    t1 = lm.generate(None, [], num_samples=3, max_gen_len=12, use_sampling=False, condition_tensors=ct, use_graph=False)
    assert torch.equal(t1.cpu(), a['greedy_tokens'])
    lm16 = build(cfg, sd, torch.bfloat16)
    assert rel(lm16.forward_steps(a['tf_sequence'].cuda(), ct).cpu(), a['tf_logits']) < 3e-2


@pytest.mark.parametrize('name', ['lm_kv_repeat', 'lm_fuser_sum', 'lm_post_norm'])
def test_lm_options_streaming_calls_vs_oracle(name):
    """The StreamingModule protocol on these options: a first call of 4 steps, then single steps, against the oracle run on the
    same split (every call resamples an 'input_interpolate' condition to ITS length; a kv_repeat stream lists the stored
    H / kv_repeat heads and can be rewound from that state)."""
    cfg, sd, a = load_golden(name)
    c = lm_cfg(cfg)
    lm = build(cfg, sd)
    ct = condition_tensors(name, a)
    cross, ops = options_inputs(name, cfg, a)
    seq = a['tf_sequence']
    st = olm.LMState(c.num_layers)
    ref = [olm.lm_forward(sd, c, seq[..., :4], cross, None, st, ops)]
    ref += [olm.lm_forward(sd, c, seq[..., i:i + 1], cross, None, st, ops) for i in range(4, 7)]
    with lm.streaming():
        got = [lm(seq[..., :4].cuda(), [], ct).cpu()]
        state5 = None
        for i in range(4, 7):
            got.append(lm(seq[..., i:i + 1].cuda(), [], ct).cpu())
            if i == 4:
                state5 = {k: v.clone() for k, v in lm.get_streaming_state().items()}
        assert rel(torch.cat(got, dim=2), torch.cat(ref, dim=2)) < 1e-4
        k0 = state5['transformer.layers.0.self_attn.past_keys']
        assert k0.shape == (seq.shape[0], c.num_heads // c.kv_repeat, 5, c.dim // c.num_heads)
        assert rel(k0.cpu(), st.past_k[0][:, :, :5]) < 1e-5
        lm.set_streaming_state(state5)            # rewind to 5 steps and replay step 5
        again = lm(seq[..., 5:6].cuda(), [], ct).cpu()
        assert rel(again, ref[2]) < 1e-4


@pytest.mark.parametrize('wdt,tol', [(torch.float32, 1e-4), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize('B', [3, 12])       # CFG rows 6 / 24: one and two 16-row blocks per GEMM
def test_lm_options_midsize_vs_oracle(wdt, tol, B):
    """d = 256, 4 layers, card 2048: qk_layer_norm + qk_layer_norm_cross + attention biases, a 'sum' and an 'input_interpolate'
    condition, and (second model) kv_repeat = 4 with rotary positions, against the oracle."""
    from audiocraft_amd.models import builders
    for extra in (dict(qk_layer_norm=True, qk_layer_norm_cross=True, bias_attn=True,
                       fuser={'cross': ['description'], 'sum': ['genre'], 'input_interpolate': ['curve']}),
                  dict(kv_repeat=4, positional_embedding='sin_rope', bias_attn=True, fuser={'cross': ['description']}),
                  dict(norm_first=False, bias_attn=True, bias_ff=True, fuser={'cross': ['description']})):
        torch.manual_seed(1)
        cfg = dict(dim=256, num_heads=8, num_layers=4, n_q=4, card=2048, hidden_scale=4, cfg_coef=3.0,
                   conditioners={'description': {'kind': 't5', 'embedder': 'synthetic', 'dim': 64, 'length': 6}}, **extra)
        lm = builders.get_lm_model(cfg, 'cuda', wdt)
        with torch.no_grad():
            for k, p in lm.named_parameters():
                if 'norm' in k:
                    p.add_(0.1 * torch.randn_like(p))
                if k.endswith('in_proj_bias') or k.endswith('out_proj.bias') or k.endswith('linear1.bias') or k.endswith('linear2.bias'):
                    p.add_(0.05 * torch.randn_like(p))
        sd = {k: v.detach().float().cpu() for k, v in lm.state_dict().items()}
        if wdt == torch.bfloat16:   # the oracle sees the same bf16-rounded matrices; activations stay f32 there
            sd = {k: (v.bfloat16().float() if v.dim() == 2 and 'output_proj' not in k else v) for k, v in sd.items()}
        oc = olm.LMConfig(dim=256, num_heads=8, num_layers=4, n_q=4, card=2048, cross_attention=True,
                          kv_repeat=extra.get('kv_repeat', 1), qk_layer_norm=extra.get('qk_layer_norm', False),
                          qk_layer_norm_cross=extra.get('qk_layer_norm_cross', False),
                          positional_embedding=extra.get('positional_embedding', 'sin'), norm_first=extra.get('norm_first', True))
        g = torch.Generator().manual_seed(5)
        cross = torch.randn(2 * B, 6, 256, generator=g)
        cross[B:] = 0
        ones = lambda n: torch.ones(2 * B, n, dtype=torch.int64).cuda()  # noqa: E731
        ct = {'description': (cross.cuda(), ones(6))}
        ops = []
        if 'sum' in extra['fuser']:
            genre, curve = 0.5 * torch.randn(2 * B, 1, 256, generator=g), 0.5 * torch.randn(2 * B, 7, 256, generator=g)
            genre[B:], curve[B:] = 0, 0
            ct.update({'genre': (genre.cuda(), ones(1)), 'curve': (curve.cuda(), ones(7))})
            ops = [('sum', genre), ('input_interpolate', curve)]
        seq = torch.randint(0, 2049, (2 * B, 4, 12), generator=g)
        ref = olm.lm_forward(sd, oc, seq, cross, input_ops=ops)
        got = lm.forward_steps(seq.cuda(), ct).cpu()
        r = rel(got, ref)
        assert r < tol, f"{sorted(extra)}: teacher-forced logits rel-L2 {r} (tol {tol})"
        if wdt == torch.float32 and B == 3:
            prompt = torch.randint(0, 2048, (B, 4, 5), generator=g)
            toks = lm.generate(prompt.cuda(), [], max_gen_len=14, use_sampling=False, condition_tensors=ct)
            ref_t = olm.generate(sd, oc, prompt, B, cross, max_gen_len=14, use_sampling=False, input_ops=ops)
            assert torch.equal(toks.cpu(), ref_t)


def test_kv_repeat_state_dict_survives_release_of_the_masters():
    """release_master_weights rebuilds in_proj_weight from the packs: with kv_repeat the packs hold every shared head once per
    query head, the state dict must come back in the reference's narrow [d + 2 kv_dim, d] layout."""
    cfg, sd, a = load_golden('lm_kv_repeat')
    lm = build(cfg, sd)
    ct = condition_tensors('lm_kv_repeat', a)
    before = lm.forward_steps(a['tf_sequence'].cuda(), ct).cpu()
    lm.release_master_weights()
    back = lm.state_dict()
    k = 'transformer.layers.1.self_attn.in_proj_weight'
    assert back[k].shape == sd[k].shape and torch.allclose(back[k].cpu(), sd[k], atol=1e-6, rtol=1e-5)
    assert torch.equal(lm.forward_steps(a['tf_sequence'].cuda(), ct).cpu(), before)
    lm2 = build(cfg, {k: v.cpu() for k, v in back.items()})
    assert rel(lm2.forward_steps(a['tf_sequence'].cuda(), ct).cpu(), a['tf_logits']) < 1e-4


def test_lm_fuser_sum_in_two_step_and_double_cfg_vs_oracle():
    """The 'sum' / 'input_interpolate' table in the other two CFG modes (lm.py:362-390): two_step_cfg encodes the conditional
    and the null conditions separately (two row groups, each with its own table rows), double CFG runs three row groups."""
    cfg, sd, a = load_golden('lm_fuser_sum')
    c = lm_cfg(cfg)
    lm = build(cfg, sd)
    B = 3
    full = condition_tensors('lm_fuser_sum', a)
    half = lambda lo: {k: (e[lo:lo + B].contiguous(), m[lo:lo + B].contiguous()) for k, (e, m) in full.items()}  # noqa: E731
    cond, null = half(0), half(B)
    cross, ops = options_inputs('lm_fuser_sum', cfg, a)
    part = lambda lo: [(op, t[lo:lo + B]) for op, t in ops]  # noqa: E731
    prompt = a['prompt']
    for p in (None, prompt):
        toks = lm.generate(None if p is None else p.cuda(), [], num_samples=B, max_gen_len=11, use_sampling=False,
                           condition_tensors=(cond, null), check=True)
        ref = olm.generate(sd, c, p, B, cross[:B], max_gen_len=11, use_sampling=False, null_cross_src=cross[B:],
                           input_ops=part(0), null_input_ops=part(B))
        assert torch.equal(toks.cpu(), ref)
    # double CFG: rows [text + wav; wav; null] -- here the middle group carries the conditional rows once more
    tri = {k: (torch.cat([e[:B], e[:B], e[B:]]).contiguous(), torch.cat([m[:B], m[:B], m[B:]]).contiguous())
           for k, (e, m) in full.items()}
    toks, lg = lm.generate(None, [], num_samples=B, max_gen_len=11, use_sampling=False, condition_tensors=tri,
                           cfg_coef_beta=2.5, return_logits=True, check=True)
    cat3 = lambda t: torch.cat([t[:B], t[:B], t[B:]])  # noqa: E731
    ref, rlg = olm.generate(sd, c, None, B, cat3(cross), max_gen_len=11, use_sampling=False, cfg_coef_beta=2.5,
                            input_ops=[(op, cat3(t)) for op, t in ops], return_logits=True)
    assert torch.equal(toks.cpu(), ref) and rel(lg.cpu(), rlg) < 1e-4


@pytest.mark.parametrize('name', ['codec_noncausal', 'codec_causal'])
def test_quantizer_forward_is_the_reference_eval_forward(name):
    """ResidualVectorQuantizer.forward (vq.py:76-85) in eval mode on the reference's own latents: its quantized latents and
    codes (goldens of the unmodified reference), the bandwidth, a zero penalty."""
    import math
    from test_gpu_models import build_codec
    cfg, sd, a = load_golden(name)
    m = build_codec(cfg, sd)
    q = m.quantizer(a['latents'].cuda(), cfg['frame_rate'])
    assert torch.equal(q.codes.cpu(), a['codes']) and torch.equal(q.x.cpu(), a['quantized_latents'])
    assert abs(float(q.bandwidth) - cfg['n_q'] * math.log2(cfg['bins']) * cfg['frame_rate'] / 1000) < 1e-6
    assert float(q.penalty) == 0.0


def test_hf_encodec_wrapper_vs_transformers_golden():
    """HFEncodecCompressionModel on the device against what `transformers` itself computed (tests/golden/
    make_hf_encodec_golden.py: a causal, reflect-padded, conv-shortcut EnCodec): latents, codes at every target bandwidth
    (end to end: any differing index must be a provable near tie), decoded audio."""
    import types
    from audiocraft_amd.models.encodec import HFEncodecCompressionModel
    from parity_utils import assert_codes_near_tie
    from test_oracle_golden import hf_native, ocodec
    cfg, sd, a = load_golden('hf_encodec_small')
    hf = types.SimpleNamespace(config=types.SimpleNamespace(**cfg), state_dict=lambda: sd, parameters=lambda: iter(()))
    m = HFEncodecCompressionModel(hf, 'cuda')
    _, conv, c = hf_native(cfg, sd)
    books = ocodec.codebooks_from_state(conv, c.n_q)
    wav = a['wav'].cuda()
    lat = m.model.encoder(wav).cpu()
    assert torch.allclose(lat, a['latents'], atol=3e-5, rtol=1e-4)
    for bw, k in zip(cfg['target_bandwidths'], (1, 2, 4)):
        m.set_num_codebooks(k)
        codes, scale = m.encode(wav)
        assert scale is None and codes.shape == a[f'codes_bw{bw}'].shape
        assert_codes_near_tie(codes, a[f'codes_bw{bw}'], lat, a['latents'], books[:k], what=f'HF EnCodec golden, {k} codebooks')
        dec = m.decode(a[f'codes_bw{bw}'].cuda(), None).cpu()
        assert dec.shape == a[f'decoded_bw{bw}'].shape
        assert (dec - a[f'decoded_bw{bw}']).abs().max().item() < 1e-4
    assert torch.equal(m.decode_latent(a['codes_bw2.4'].cuda()).cpu(), a['quantized'])


def test_hf_encodec_24khz_configuration_vs_transformers_run_here():
    """The configuration of facebook/encodec_24khz itself (EncodecConfig's defaults: 32 x 1024 codebooks, ratios 8 5 4 2, 75 fps)
    with seeded random weights: `transformers` runs on the host, the wrapper on the device, on the same 0.4 s of audio."""
    transformers = pytest.importorskip('transformers')
    from audiocraft_amd.models.encodec import HFEncodecCompressionModel
    from parity_utils import assert_codes_near_tie
    torch.manual_seed(3)
    hf = transformers.EncodecModel(transformers.EncodecConfig()).eval()
    with torch.no_grad():
        for layer in hf.quantizer.layers:       # HF initialises the codebooks with zeros
            layer.codebook.embed.copy_(torch.randn_like(layer.codebook.embed) * 0.3)
    m = HFEncodecCompressionModel(hf, 'cuda')
    assert m.possible_num_codebooks == [2, 4, 8, 16, 32] and m.frame_rate == 75.0 and m.sample_rate == 24000
    wav = 0.3 * torch.randn(2, 1, 9611, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        lat_ref = hf.encoder(wav)
        enc = hf.encode(wav, None, 24.0)
        codes_ref = enc[0][0]
        dec_ref = hf.decode(codes_ref[None], [None])[0]
        enc6 = hf.encode(wav, None, 6.0)[0][0]
    lat = m.model.encoder(wav.cuda()).cpu()
    assert torch.allclose(lat, lat_ref, atol=3e-5, rtol=1e-4)
    codes, _ = m.encode(wav.cuda())
    books = torch.stack([layer.codebook.embed for layer in hf.quantizer.layers])
    assert_codes_near_tie(codes, codes_ref, lat, lat_ref, books, what='HF EnCodec-24k configuration, 32 codebooks')
    m.set_num_codebooks(8)     # 6 kb/s
    codes8, _ = m.encode(wav.cuda())
    assert codes8.shape == enc6.shape == (2, 8, 31)
    dec = m.decode(codes_ref.cuda(), None).cpu()
    assert dec.shape == dec_ref.shape and (dec - dec_ref).abs().max().item() < 1e-4


def test_lm_other_codebook_patterns_vs_reference_golden():
    """LM generation through the NON-delay codebook patterns of the reference's builder (`parallel`, `unroll`, `coarse_first`,
    `musiclm`, `delay` with flatten_first / empty_initial: codebooks_patterns.py:359-548, builders.py:233-254) on the device,
    against `lm_patterns.npz` (greedy tokens + per-step logits recorded from the unmodified reference, reproduced by the oracle
    in tests/test_oracle_golden.py).  The kernels only see the [K, S] validity mask and the token sequence."""
    from audiocraft_amd.models import builders
    cfg, sd, a = load_golden('lm_patterns')
    cross = a['cond_description'].cuda()
    ct = {'description': (cross, torch.ones(cross.shape[:2], dtype=torch.int64).cuda())}
    for i, (name, kw) in enumerate(cfg['patterns']):
        lm = builders.get_lm_model(dict(dim=cfg['dim'], num_heads=cfg['num_heads'], num_layers=cfg['num_layers'], n_q=cfg['n_q'],
                                        card=cfg['card'], hidden_scale=cfg['hidden_scale'], cfg_coef=cfg['cfg_coef'],
                                        conditioners={'description': {'kind': 't5', 'embedder': 'synthetic', 'dim': cfg['cond_dim'],
                                                                      'length': cfg['Lc']}},
                                        fuser={'cross': ['description']}, codebooks_pattern={'modeling': name, name: kw}),
                                   'cuda', torch.float32)
        lm.load_state_dict(sd)
        toks, lg = lm.generate(None, [], num_samples=3, max_gen_len=7, use_sampling=False, condition_tensors=ct,
                               return_logits=True, check=True)
        assert torch.equal(toks.cpu(), a[f'tokens_{i}']), name
        assert rel(lg.cpu(), olm.cfg_mix(a[f'step_logits_{i}'], cfg['cfg_coef'])) < 1e-4, name
        if f'cont_tokens_{i}' in a:
            toks = lm.generate(a['prompt'].cuda(), [], max_gen_len=7, use_sampling=False, condition_tensors=ct, check=True)
            assert torch.equal(toks.cpu(), a[f'cont_tokens_{i}']), name


def test_lm_fuser_sum_after_prepend_vs_reference_golden():
    """A 'sum' and an 'input_interpolate' condition AFTER a 'prepend' one in the provider's order (round 5; raised
    NotImplementedError before): the reference adds them to the prepended rows of the first call as well
    (conditioners.py:1730-1748); golden from the unmodified reference, greedy tokens + step logits, with and without a prompt."""
    cfg, sd, a = load_golden('lm_fuser_prepend_sum')
    lm = build(cfg, sd)
    ones = lambda t: torch.ones(t.shape[:2], dtype=torch.int64).cuda()  # noqa: E731
    ct = {k: (a['cond_' + k].cuda(), ones(a['cond_' + k])) for k in ('description', 'genre', 'curve')}   # the provider's order
    toks, lg = lm.generate(None, [], num_samples=3, max_gen_len=12, use_sampling=False, condition_tensors=ct,
                           return_logits=True, check=True)
    assert torch.equal(toks.cpu(), a['greedy_tokens'])
    assert rel(lg.cpu(), olm.cfg_mix(a['greedy_step_logits'], cfg['cfg_coef'])) < 1e-4
    toks = lm.generate(a['prompt'].cuda(), [], max_gen_len=11, use_sampling=False, condition_tensors=ct, check=True)
    assert torch.equal(toks.cpu(), a['cont_tokens'])
    # the teacher-forced forward (one call of S steps: the reference golden) and the streaming protocol (a first call of 4 steps,
    # whose length shapes what is added to the prepended rows, then single steps: against the oracle on the same split)
    c = lm_cfg(cfg)
    seq = a['tf_sequence']
    assert rel(lm.forward_steps(seq.cuda(), ct).cpu(), a['tf_logits']) < 1e-4
    ops = [('prepend', a['cond_description']), ('sum', a['cond_genre']), ('input_interpolate', a['cond_curve'])]
    st = olm.LMState(c.num_layers)
    ref = [olm.lm_forward(sd, c, seq[..., :4], None, None, st, ops)]
    ref += [olm.lm_forward(sd, c, seq[..., i:i + 1], None, None, st, ops) for i in range(4, 7)]
    with lm.streaming():
        got = [lm(seq[..., :4].cuda(), [], ct).cpu()] + [lm(seq[..., i:i + 1].cuda(), [], ct).cpu() for i in range(4, 7)]
    assert rel(torch.cat(got, dim=2), torch.cat(ref, dim=2)) < 1e-4
    # the other order (sum / interpolate first) is a different model output: the order is honoured, not normalised
    ct2 = {k: ct[k] for k in ('genre', 'curve', 'description')}
    toks2 = lm.generate(None, [], num_samples=3, max_gen_len=12, use_sampling=False, condition_tensors=ct2)
    assert not torch.equal(toks2.cpu(), a['greedy_tokens'])
