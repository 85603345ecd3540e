"""TEST INFRASTRUCTURE ONLY -- re-check the oracle against the UNMODIFIED reference at sizes larger than the committed
golden fixtures.  Runs in the build container only (needs /root/reference; nothing here travels to the GPU box):

    python -m oracle.validate_against_reference            # all checks, ~1 min on 8 cores
    python -m oracle.validate_against_reference lm codec   # a subset

Checks (each prints a line and the script exits non-zero on the first failure):
  lm       mid-size LMModel (d 256, 4 layers, 8 heads, cross-attention, card 2048, seeded random weights with perturbed
           LayerNorm parameters): reference `LMModel.forward` (batch) and `LMModel.generate` (greedy, CFG; plain, two_step_cfg)
           vs oracle.lm.lm_forward / generate  -> logits rel-L2 <= 1e-5, tokens identical
  bias     the reference's default transformer configuration: every projection bias on (+ LayerScale)
  melody   prepend-conditioned LMModel (no cross-attention) incl. double CFG (cfg_coef_beta)
  stereo   8 codebooks with delays [0,0,1,1,2,2,3,3]
  codec    EncodecModel at the 32 kHz geometry with n_filters 16 (all layers, LSTM, RVQ 4 x 2048) on 0.7 s of audio:
           latents, codes (bit exact on the reference's own latents), decoded waveform
  epic     configs[0]: the reference EncodecModel at the full EnCodec-24 kHz geometry on assets/epic.wav (32 RVQ levels)
  mbd      MultiBandDiffusion: the reference DiffusionUnet at the released geometry (hidden 48, depth 4, growth 4, kernel 8,
           stride 4, all-layer embeddings, 128-d condition) on 0.5 s of 32 kHz audio, and with the BiLSTM bottleneck (depth 3);
           the full DDPM reverse process (NoiseSchedule.generate, 6 steps) with the reference's draws replayed
  chroma   oracle.chroma against the reference ChromaExtractor arithmetic is NOT possible here: torchaudio / librosa are
           third-party and absent -- see oracle/chroma.py for how that row is pinned instead.
"""
import os
import sys

import torch

if __package__ in (None, ''):   # `python oracle/validate_against_reference.py`: make the relative imports below resolve
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    __package__ = 'oracle'
    import oracle  # noqa: F401

from . import refstubs  # noqa: F401,E402  (installs the import stubs; afterwards the reference imports)

if not refstubs.available():   # pragma: no cover
    print("validate_against_reference: /root/reference not present, nothing to do")
    sys.exit(0)

from audiocraft.models.encodec import EncodecModel  # noqa: E402
from audiocraft.models.lm import LMModel  # noqa: E402
from audiocraft.modules.codebooks_patterns import DelayedPatternProvider  # noqa: E402
from audiocraft.modules.conditioners import (  # noqa: E402
    ClassifierFreeGuidanceDropout, ConditionFuser, ConditioningAttributes, ConditioningProvider, TextConditioner,
    WaveformConditioner, WavCondition)
from audiocraft.modules.seanet import SEANetDecoder, SEANetEncoder  # noqa: E402
from audiocraft.quantization.vq import ResidualVectorQuantizer  # noqa: E402

from . import codec as ocodec  # noqa: E402
from . import lm as olm  # noqa: E402


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


class _Text(TextConditioner):
    """Seeded stand-in for the T5 encoder output (third party), then the real output_proj + mask multiply; an all-null
    batch is one position long (what T5 gives for empty strings)."""
    def __init__(self, dim, output_dim, L):
        super().__init__(dim, output_dim)
        self.L = L

    def tokenize(self, x):
        return x

    def forward(self, x):
        g = torch.Generator().manual_seed(99)
        L = 1 if all(xi is None for xi in x) else self.L
        mask = torch.tensor([[1] * L if xi is not None else [0] * L for xi in x])
        e = torch.randn(len(x), L, self.dim, generator=g)
        return self.output_proj(e) * mask.unsqueeze(-1), mask


class _Chroma(WaveformConditioner):
    def __init__(self, output_dim, P):
        super().__init__(12, output_dim, 'cpu')
        self.P = P
        self._use_masking = False

    def _downsampling_factor(self):
        return 1

    def _get_wav_embedding(self, x):
        g = torch.Generator().manual_seed(98)
        cls = torch.randint(0, 12, (x.wav.shape[0], self.P), generator=g)
        e = torch.nn.functional.one_hot(cls, 12).float()
        return torch.where((x.length == 0).view(-1, 1, 1), torch.zeros_like(e), e)


def _build_lm(dim, heads, layers, n_q, card, delays, conditioners, fuse, seed, **extra):
    torch.manual_seed(seed)
    if extra:   # rotary positions / past_context: the custom attention (see tests/golden/make_rope_golden.py for why)
        extra = dict(dict(custom=True, memory_efficient=False), **extra)
    lm = LMModel(DelayedPatternProvider(n_q, delays=delays), ConditioningProvider(conditioners), ConditionFuser(fuse),
                 n_q=n_q, card=card, dim=dim, num_heads=heads, hidden_scale=4, norm='layer_norm', norm_first=True,
                 bias_proj=False, weight_init='gaussian', depthwise_init='current', zero_bias_init=True, cfg_coef=3.0,
                 num_layers=layers, dropout=0., activation='gelu', bias_ff=False, bias_attn=False, causal=True,
                 attention_as_float32=False, cross_attention=bool(fuse['cross']),
                 **dict(dict(custom=False, memory_efficient=True, positional_embedding='sin'), **extra)).eval()
    with torch.no_grad():
        for k, p in lm.named_parameters():
            if '.norm' in k or k.startswith('out_norm'):
                p.add_(0.1 * torch.randn_like(p))
    return lm


def _record(lm, fn):
    rec = []
    h = lm.register_forward_hook(lambda mod, inp, out: rec.append(out.detach().clone()))
    out = fn()
    h.remove()
    return out, rec


def check_lm():
    torch.manual_seed(1)
    fuse = {'cross': ['description'], 'prepend': [], 'sum': [], 'input_interpolate': []}
    lm = _build_lm(256, 8, 4, 4, 2048, [0, 1, 2, 3], {'description': _Text(64, 256, 7)}, fuse, seed=11)
    sd = {k: v.detach() for k, v in lm.state_dict().items()}
    oc = olm.LMConfig(dim=256, num_heads=8, num_layers=4, n_q=4, card=2048, cross_attention=True)
    conds = [ConditioningAttributes(text={'description': f't{i}'}) for i in range(3)]
    null = ClassifierFreeGuidanceDropout(p=1.0)(conds)
    ct = lm.condition_provider(lm.condition_provider.tokenize(conds + null))
    seq = torch.randint(0, 2049, (6, 4, 40), generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        ref = lm(seq, [], ct)
    got = olm.lm_forward(sd, oc, seq, ct['description'][0])
    r = rel(got, ref)
    assert r < 1e-5, r
    toks, rec = _record(lm, lambda: lm.generate(None, conds, max_gen_len=24, use_sampling=False))
    otoks, ologits = olm.generate(sd, oc, None, 3, ct['description'][0], max_gen_len=24, use_sampling=False,
                                  return_logits=True)
    assert torch.equal(toks, otoks)
    r2 = rel(ologits, olm.cfg_mix(torch.stack([x[:, :, -1] for x in rec], dim=2), 3.0))
    assert r2 < 1e-5, r2
    # two_step_cfg: separate passes, own condition lengths (7 vs 1), model's cfg_coef
    c1 = lm.condition_provider(lm.condition_provider.tokenize(conds))
    n1 = lm.condition_provider(lm.condition_provider.tokenize(null))
    toks2 = lm.generate(None, conds, max_gen_len=16, use_sampling=False, two_step_cfg=True, cfg_coef=9.0)
    otoks2 = olm.generate(sd, oc, None, 3, c1['description'][0], max_gen_len=16, use_sampling=False, cfg_coef=9.0,
                          null_cross_src=n1['description'][0])
    assert torch.equal(toks2, otoks2)
    print(f"lm      ok: batch forward rel-L2 {r:.1e}, greedy tokens identical (plain + two_step_cfg), step logits rel-L2 {r2:.1e}")


def check_bias():
    """Every bias of the reference's default transformer configuration (bias_ff / bias_attn / bias_proj true) + LayerScale:
    the oracle's handling of in_proj / out_proj / cross k, v / linear1, 2 / head biases against the reference."""
    torch.manual_seed(31)
    dim, B = 128, 3
    lm = LMModel(DelayedPatternProvider(4, delays=[0, 1, 2, 3]), ConditioningProvider({'description': _Text(32, dim, 5)}),
                 ConditionFuser({'cross': ['description'], 'prepend': [], 'sum': [], 'input_interpolate': []}),
                 n_q=4, card=256, dim=dim, num_heads=4, hidden_scale=4, norm='layer_norm', norm_first=True, bias_proj=True,
                 weight_init='gaussian', depthwise_init='current', zero_bias_init=False, cfg_coef=3.0, num_layers=3,
                 dropout=0., activation='gelu', bias_ff=True, bias_attn=True, causal=True, custom=False, memory_efficient=True,
                 attention_as_float32=False, cross_attention=True, positional_embedding='sin', layer_scale=0.5).eval()
    with torch.no_grad():
        for k, p in lm.named_parameters():
            if '.norm' in k or k.startswith('out_norm') or k.endswith('bias'):
                p.add_(0.1 * torch.randn_like(p))
    sd = {k: v.detach() for k, v in lm.state_dict().items()}
    assert 'transformer.layers.0.cross_attention.in_proj_bias' in sd and 'transformer.layers.0.linear2.bias' in sd
    oc = olm.LMConfig(dim=dim, num_heads=4, num_layers=3, n_q=4, card=256, cross_attention=True)
    conds = [ConditioningAttributes(text={'description': f'c{i}'}) for i in range(B)]
    null = ClassifierFreeGuidanceDropout(p=1.0)(conds)
    ct = lm.condition_provider(lm.condition_provider.tokenize(conds + null))
    seq = torch.randint(0, 257, (2 * B, 4, 11))
    with torch.no_grad():
        ref = lm.forward(seq, [], ct)
    got = olm.lm_forward(sd, oc, seq, ct['description'][0])
    assert rel(got, ref) < 1e-5, rel(got, ref)
    toks = lm.generate(None, conds, max_gen_len=12, use_sampling=False)
    otoks = olm.generate(sd, oc, None, B, ct['description'][0], max_gen_len=12, use_sampling=False)
    assert torch.equal(toks, otoks)
    print(f"bias    ok: all projection biases + LayerScale: batch forward rel-L2 {rel(got, ref):.1e}, greedy tokens identical")


def check_rope():
    """Rotary positions + xPos + past_context + LayerScale at d 256 / 8 heads / 4 layers, window 20 against 60 positions
    and a 30-step prompt (longer than the window: the reference's lagging rotary positions after a long first call)."""
    fuse = {'cross': ['description'], 'prepend': [], 'sum': [], 'input_interpolate': []}
    for extra in (dict(positional_embedding='rope', xpos=True, past_context=20, layer_scale=0.2, positional_scale=0.9),
                  dict(positional_embedding='sin_rope', past_context=33),
                  dict(positional_embedding='rope')):
        lm = _build_lm(256, 8, 4, 4, 2048, [0, 1, 2, 3], {'description': _Text(64, 256, 7)}, fuse, seed=21, **extra)
        sd = {k: v.detach() for k, v in lm.state_dict().items()}
        oc = olm.LMConfig(dim=256, num_heads=8, num_layers=4, n_q=4, card=2048, cross_attention=True,
                          **{k: v for k, v in extra.items() if k != 'layer_scale'})
        conds = [ConditioningAttributes(text={'description': f't{i}'}) for i in range(2)]
        null = ClassifierFreeGuidanceDropout(p=1.0)(conds)
        ct = lm.condition_provider(lm.condition_provider.tokenize(conds + null))
        seq = torch.randint(0, 2049, (4, 4, 60), generator=torch.Generator().manual_seed(3))
        with torch.no_grad():
            ref = lm(seq, [], ct)
        r = rel(olm.lm_forward(sd, oc, seq, ct['description'][0]), ref)
        assert r < 1e-5, (extra, r)
        toks = lm.generate(None, conds, max_gen_len=40, use_sampling=False)
        assert torch.equal(toks, olm.generate(sd, oc, None, 2, ct['description'][0], max_gen_len=40, use_sampling=False))
        prompt = torch.randint(0, 2048, (2, 4, 30), generator=torch.Generator().manual_seed(4))
        toks = lm.generate(prompt, conds, max_gen_len=44, use_sampling=False)
        assert torch.equal(toks, olm.generate(sd, oc, prompt, 2, ct['description'][0], max_gen_len=44, use_sampling=False)), extra
    print(f"rope    ok: full forward rel-L2 {r:.1e}, greedy + continuation tokens identical (rope+xpos+window+LayerScale, sin_rope+window, rope)")


class _Frames(TextConditioner):
    """A text-keyed condition of `frames` frames through the real output_proj (zeros for null conditions)."""
    def __init__(self, dim, output_dim, frames, seed):
        super().__init__(dim, output_dim)
        self.frames, self.seed = frames, seed

    def tokenize(self, x):
        return x

    def forward(self, x):
        g = torch.Generator().manual_seed(self.seed)
        mask = torch.tensor([[1] * self.frames if xi is not None else [0] * self.frames for xi in x])
        e = torch.randn(len(x), self.frames, self.dim, generator=g)
        return self.output_proj(e) * mask.unsqueeze(-1), mask


def check_options():
    """kv_repeat, qk_layer_norm (+ cross), fuser 'sum' / 'input_interpolate' + cross_attention_pos_emb at d 256 / 8 heads /
    4 layers against the imported reference (its custom attention): full forward, greedy tokens, continuation from a 9-step
    prompt (the interpolated condition is resampled to that first call's length)."""
    cross_only = {'cross': ['description'], 'prepend': [], 'sum': [], 'input_interpolate': []}
    for extra in (dict(kv_repeat=4), dict(kv_repeat=2, positional_embedding='sin_rope'),
                  dict(qk_layer_norm=True, qk_layer_norm_cross=True)):
        lm = _build_lm(256, 8, 4, 4, 2048, [0, 1, 2, 3], {'description': _Text(64, 256, 7)}, cross_only, seed=41, **extra)
        with torch.no_grad():
            for k, p in lm.named_parameters():
                if '_layer_norm.' in k:
                    p.add_(0.1 * torch.randn_like(p))
        sd = {k: v.detach() for k, v in lm.state_dict().items()}
        oc = olm.LMConfig(dim=256, num_heads=8, num_layers=4, n_q=4, card=2048, cross_attention=True, **extra)
        conds = [ConditioningAttributes(text={'description': f't{i}'}) for i in range(2)]
        null = ClassifierFreeGuidanceDropout(p=1.0)(conds)
        ct = lm.condition_provider(lm.condition_provider.tokenize(conds + null))
        seq = torch.randint(0, 2049, (4, 4, 30), generator=torch.Generator().manual_seed(5))
        with torch.no_grad():
            ref = lm(seq, [], ct)
        r = rel(olm.lm_forward(sd, oc, seq, ct['description'][0]), ref)
        assert r < 1e-5, (extra, r)
        toks = lm.generate(None, conds, max_gen_len=24, use_sampling=False)
        assert torch.equal(toks, olm.generate(sd, oc, None, 2, ct['description'][0], max_gen_len=24, use_sampling=False)), extra
    # the fuser's additive methods
    fuse = {'cross': ['description'], 'prepend': [], 'sum': ['genre'], 'input_interpolate': ['curve']}
    torch.manual_seed(43)
    cds = {'description': _Text(64, 256, 7), 'genre': _Frames(64, 256, 1, 77), 'curve': _Frames(64, 256, 11, 78)}
    torch.manual_seed(42)
    lm = LMModel(DelayedPatternProvider(4, delays=[0, 1, 2, 3]), ConditioningProvider(cds),
                 ConditionFuser(fuse, cross_attention_pos_emb=True, cross_attention_pos_emb_scale=0.6),
                 n_q=4, card=2048, dim=256, num_heads=8, hidden_scale=4, norm='layer_norm', norm_first=True, bias_proj=False,
                 weight_init='gaussian', depthwise_init='current', zero_bias_init=True, cfg_coef=3.0, num_layers=4, dropout=0.,
                 activation='gelu', bias_ff=False, bias_attn=False, causal=True, attention_as_float32=False,
                 cross_attention=True, custom=False, memory_efficient=True, positional_embedding='sin').eval()
    sd = {k: v.detach() for k, v in lm.state_dict().items()}
    oc = olm.LMConfig(dim=256, num_heads=8, num_layers=4, n_q=4, card=2048, cross_attention=True)
    conds = [ConditioningAttributes(text={'description': f't{i}', 'genre': f'g{i}', 'curve': f'c{i}'}) for i in range(2)]
    null = ClassifierFreeGuidanceDropout(p=1.0)(conds)
    ct = lm.condition_provider(lm.condition_provider.tokenize(conds + null))
    cross = olm.cross_pos_emb(ct['description'][0], 0.6)
    ops = [('sum', ct['genre'][0]), ('input_interpolate', ct['curve'][0])]
    seq = torch.randint(0, 2049, (4, 4, 30), generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        ref = lm(seq, [], ct)
    r2 = rel(olm.lm_forward(sd, oc, seq, cross, input_ops=ops), ref)
    assert r2 < 1e-5, r2
    toks = lm.generate(None, conds, max_gen_len=24, use_sampling=False)
    assert torch.equal(toks, olm.generate(sd, oc, None, 2, cross, max_gen_len=24, use_sampling=False, input_ops=ops))
    prompt = torch.randint(0, 2048, (2, 4, 9), generator=torch.Generator().manual_seed(7))
    toks = lm.generate(prompt, conds, max_gen_len=24, use_sampling=False)
    assert torch.equal(toks, olm.generate(sd, oc, prompt, 2, cross, max_gen_len=24, use_sampling=False, input_ops=ops))
    print(f"options ok: kv_repeat 4 / 2 + sin_rope / qk_layer_norm (+ cross): forward rel-L2 {r:.1e}, greedy tokens identical; "
          f"fuser sum + input_interpolate + cross pos emb: forward rel-L2 {r2:.1e}, greedy + continuation tokens identical")


def check_hf_encodec():
    """HuggingFace EnCodec at EncodecConfig's defaults (= facebook/encodec_24khz: causal, reflect padding, conv shortcuts, 32 x
    1024 codebooks) with seeded random weights, run by `transformers` itself, vs the oracle on the re-keyed state dict (what
    HFEncodecCompressionModel loads): the third party on this path that IS installed here."""
    import transformers
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
    from test_oracle_golden import hf_native
    torch.manual_seed(3)
    hf = transformers.EncodecModel(transformers.EncodecConfig()).eval()
    with torch.no_grad():
        for layer in hf.quantizer.layers:
            layer.codebook.embed.copy_(torch.randn_like(layer.codebook.embed) * 0.3)
    wav = 0.3 * torch.randn(2, 1, 24000, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        lat_ref = hf.encoder(wav)
        codes_ref = hf.encode(wav, None, 24.0)[0][0]
        dec_ref = hf.decode(codes_ref[None], [None])[0]
    cfg = {k: getattr(hf.config, k) for k in ('audio_channels', 'sampling_rate', 'target_bandwidths', 'hidden_size', 'codebook_dim',
                                              'num_filters', 'num_residual_layers', 'upsampling_ratios', 'codebook_size',
                                              'kernel_size', 'last_kernel_size', 'residual_kernel_size', 'dilation_growth_rate',
                                              'use_causal_conv', 'pad_mode', 'compress', 'num_lstm_layers', 'trim_right_ratio',
                                              'use_conv_shortcut', 'norm_type', 'normalize', 'chunk_length_s', 'num_quantizers')}
    _, conv, c = hf_native(cfg, {k: v.detach() for k, v in hf.state_dict().items()})
    lat = ocodec.seanet_encoder(conv, c, wav)
    r = rel(lat, lat_ref)
    assert r < 1e-5, r
    codes = ocodec.rvq_encode(lat_ref, ocodec.codebooks_from_state(conv, c.n_q))
    assert torch.equal(codes, codes_ref)
    dec = ocodec.encodec_decode(conv, c, codes_ref, None)
    e = (dec - dec_ref).abs().max().item()
    assert e < 2e-5, e
    print(f"hf      ok: transformers {transformers.__version__} EncodecModel (24 kHz configuration) vs the oracle on re-keyed weights: "
          f"latents rel-L2 {r:.1e}, {tuple(codes.shape)} codes bit exact, waveform max abs {e:.1e}")


def check_melody():
    fuse = {'cross': [], 'prepend': ['self_wav', 'description'], 'sum': [], 'input_interpolate': []}
    torch.manual_seed(3)
    lm = _build_lm(128, 4, 3, 4, 512, [0, 1, 2, 3], {'description': _Text(32, 128, 5), 'self_wav': _Chroma(128, 20)},
                   fuse, seed=12)
    sd = {k: v.detach() for k, v in lm.state_dict().items()}
    oc = olm.LMConfig(dim=128, num_heads=4, num_layers=3, n_q=4, card=512, cross_attention=False)
    conds = []
    for i in range(2):
        c = ConditioningAttributes(text={'description': f'm{i}'})
        c.wav['self_wav'] = WavCondition(torch.randn(1, 1, 64), torch.tensor([64]), [1200], [None], [0.])
        conds.append(c)
    null = ClassifierFreeGuidanceDropout(p=1.0)(conds)
    ct = lm.condition_provider(lm.condition_provider.tokenize(conds + null))
    prepend = torch.cat([ct['self_wav'][0], ct['description'][0]], dim=1)
    toks = lm.generate(None, conds, max_gen_len=14, use_sampling=False)
    otoks = olm.generate(sd, oc, None, 2, None, prepend, max_gen_len=14, use_sampling=False)
    assert torch.equal(toks, otoks)
    from audiocraft.models.lm import _drop_description_condition
    ct3 = lm.condition_provider(lm.condition_provider.tokenize(conds + _drop_description_condition(conds) + null))
    prepend3 = torch.cat([ct3['self_wav'][0], ct3['description'][0]], dim=1)
    toks3 = lm.generate(None, conds, max_gen_len=14, use_sampling=False, cfg_coef_beta=4.0)
    otoks3 = olm.generate(sd, oc, None, 2, None, prepend3, max_gen_len=14, use_sampling=False, cfg_coef_beta=4.0)
    assert torch.equal(toks3, otoks3)
    print("melody  ok: greedy tokens identical (prepend path; double CFG)")


def check_stereo():
    fuse = {'cross': ['description'], 'prepend': [], 'sum': [], 'input_interpolate': []}
    delays = [0, 0, 1, 1, 2, 2, 3, 3]
    torch.manual_seed(4)
    lm = _build_lm(128, 4, 2, 8, 256, delays, {'description': _Text(32, 128, 4)}, fuse, seed=13)
    sd = {k: v.detach() for k, v in lm.state_dict().items()}
    oc = olm.LMConfig(dim=128, num_heads=4, num_layers=2, n_q=8, card=256, cross_attention=True, delays=delays)
    conds = [ConditioningAttributes(text={'description': f's{i}'}) for i in range(2)]
    null = ClassifierFreeGuidanceDropout(p=1.0)(conds)
    ct = lm.condition_provider(lm.condition_provider.tokenize(conds + null))
    prompt = torch.randint(0, 256, (2, 8, 4), generator=torch.Generator().manual_seed(5))
    for pr in (None, prompt):
        toks = lm.generate(pr, conds, max_gen_len=15, use_sampling=False)
        otoks = olm.generate(sd, oc, pr, 2, ct['description'][0], max_gen_len=15, use_sampling=False)
        assert torch.equal(toks, otoks)
    print("stereo  ok: greedy tokens identical with delays [0,0,1,1,2,2,3,3] (no prompt, 4-step prompt)")


def check_codec():
    torch.manual_seed(6)
    kw = dict(channels=1, dimension=128, n_filters=16, n_residual_layers=1, ratios=[8, 5, 4, 4], activation='ELU',
              activation_params={'alpha': 1.}, norm='weight_norm', norm_params={}, kernel_size=7, residual_kernel_size=3,
              last_kernel_size=7, dilation_base=2, causal=False, pad_mode='constant', true_skip=True, compress=2, lstm=2,
              disable_norm_outer_blocks=0)
    m = EncodecModel(SEANetEncoder(**kw), SEANetDecoder(**kw, trim_right_ratio=1.0),
                     ResidualVectorQuantizer(dimension=128, n_q=4, bins=2048, kmeans_init=False),
                     frame_rate=50, sample_rate=32000, channels=1).eval()
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    c = ocodec.CodecConfig(channels=1, dimension=128, n_filters=16, n_residual_layers=1, ratios=[8, 5, 4, 4],
                           causal=False, pad_mode='constant', lstm=2, norm='weight_norm', n_q=4, bins=2048,
                           sample_rate=32000, frame_rate=50)
    wav = 0.3 * torch.randn(2, 1, 22400)
    with torch.no_grad():
        lat = m.encoder(wav)
        codes, _ = m.encode(wav)
        dec = m.decode(codes)
    olat = ocodec.seanet_encoder(sd, c, wav)
    assert rel(olat, lat) < 1e-5
    assert torch.equal(ocodec.rvq_encode(lat, ocodec.codebooks_from_state(sd, 4)), codes)
    odec = ocodec.encodec_decode(sd, c, codes)
    assert (odec - dec).abs().max().item() < 2e-5
    print(f"codec   ok: latents rel-L2 {rel(olat, lat):.1e}, codes bit exact, waveform max abs {(odec - dec).abs().max().item():.1e}")


def check_epic():
    """BASELINE.json configs[0] on real audio: the reference's assets/epic.wav (read with scipy; SURVEY.md section 8c iii)
    through the REFERENCE EncodecModel at the full EnCodec-24 kHz geometry (causal, n_filters 32, ratios [8,5,4,2], RVQ
    32 x 1024) vs the oracle: latents, all 32 levels of codes on the reference's own latents (bit exact), and end to end."""
    from scipy.io import wavfile
    sr, x = wavfile.read('/root/reference/assets/epic.wav')
    wav = torch.from_numpy(x).view(1, 1, -1)
    torch.manual_seed(7)
    kw = dict(channels=1, dimension=128, n_filters=32, n_residual_layers=1, ratios=[8, 5, 4, 2], activation='ELU',
              activation_params={'alpha': 1.}, norm='weight_norm', norm_params={}, kernel_size=7, residual_kernel_size=3,
              last_kernel_size=7, dilation_base=2, causal=True, pad_mode='constant', true_skip=True, compress=2, lstm=2,
              disable_norm_outer_blocks=0)
    m = EncodecModel(SEANetEncoder(**kw), SEANetDecoder(**kw, trim_right_ratio=1.0),
                     ResidualVectorQuantizer(dimension=128, n_q=32, bins=1024, kmeans_init=False),
                     frame_rate=75, sample_rate=24000, channels=1, causal=True).eval()
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    c = ocodec.CodecConfig(channels=1, dimension=128, n_filters=32, n_residual_layers=1, ratios=[8, 5, 4, 2],
                           causal=True, pad_mode='constant', lstm=2, norm='weight_norm', n_q=32, bins=1024,
                           sample_rate=24000, frame_rate=75)
    with torch.no_grad():
        lat = m.encoder(wav)
        codes, _ = m.encode(wav)
        dec = m.decode(codes)
    olat = ocodec.seanet_encoder(sd, c, wav)
    assert codes.shape == (1, 32, 300) and rel(olat, lat) < 1e-5
    cb = ocodec.codebooks_from_state(sd, 32)
    assert torch.equal(ocodec.rvq_encode(lat, cb), codes)
    ocodes = ocodec.rvq_encode(olat, cb)
    odec = ocodec.encodec_decode(sd, c, codes)
    assert (odec - dec).abs().max().item() < 2e-5
    print(f"epic    ok: EnCodec-24k geometry on assets/epic.wav: latents rel-L2 {rel(olat, lat):.1e}, 32 x 300 codes bit exact "
          f"on the reference's latents (end to end {float((ocodes == codes).float().mean()):.4f} equal), waveform max abs "
          f"{(odec - dec).abs().max().item():.1e}")


def check_mbd():
    """The reference's DiffusionUnet / NoiseSchedule (julius only supplies SplitBands, not used here) vs oracle.mbd."""
    from audiocraft.models.unet import DiffusionUnet
    from audiocraft.modules.diffusion_schedule import NoiseSchedule
    from . import mbd as ombd
    for bilstm, depth, growth, T in ((False, 4, 4., 16000), (True, 3, 2., 8000)):
        torch.manual_seed(8)
        kw = dict(hidden=48, depth=depth, growth=growth, max_channels=10_000, emb_all_layers=True, bilstm=bilstm, codec_dim=128,
                  kernel=8, stride=4, norm_groups=4, res_blocks=1)
        m = DiffusionUnet(chin=1, num_steps=1000, **kw).eval()
        with torch.no_grad():
            for k, p in m.named_parameters():
                if 'norm' in k:
                    p.add_(0.2 * torch.randn_like(p))
        sd = {k: v.detach() for k, v in m.state_dict().items()}
        uc = ombd.UnetConfig(chin=1, num_steps=1000, **kw)
        x, cond, step = torch.randn(2, 1, T), torch.randn(2, 128, T // 640), torch.tensor([3, 977])
        with torch.no_grad():
            ref = m(x, step, condition=cond).sample
        got = ombd.unet_forward(sd, uc, x, step, cond)
        assert rel(got, ref) < 1e-5, rel(got, ref)
        print(f"mbd     ok: DiffusionUnet hidden 48 depth {depth} growth {growth:g} bilstm {bilstm} on {T} samples: rel-L2 {rel(got, ref):.1e}")
    ns = NoiseSchedule(beta_t0=1e-4, beta_t1=0.2, num_steps=6, variance='beta_tilde', clip=4., rescale=0.9, noise_scale=0.95, device='cpu')
    g = torch.Generator().manual_seed(9)
    init, cond = torch.randn(1, 1, 4000, generator=g), torch.randn(1, 128, 6, generator=g)
    m6 = DiffusionUnet(chin=1, num_steps=6, **dict(kw, bilstm=False, depth=2)).eval()
    sd6, uc6 = {k: v.detach() for k, v in m6.state_dict().items()}, ombd.UnetConfig(chin=1, num_steps=6, **dict(kw, bilstm=False, depth=2))
    draws, real = [], torch.randn_like

    def recorded(t, *a, **k):
        draws.append(torch.randn(t.shape, generator=g))
        return draws[-1]
    torch.randn_like = recorded
    try:
        ref = ns.generate(m6, initial=init, condition=cond)
    finally:
        torch.randn_like = real
    betas, cur = ns.betas, init
    alpha_bar = (1 - betas).prod()
    for i, step in enumerate(range(6)[::-1]):
        est = ombd.unet_forward(sd6, uc6, cur, step, cond)
        alpha = 1 - betas[step]
        prev = (cur - (1 - alpha) / (1 - alpha_bar).sqrt() * est) / alpha.sqrt()
        pab = (1 - betas[:step]).prod()
        if step > 0:
            prev = prev + ((1 - pab) / (1 - alpha_bar) * (1 - alpha)) ** 0.5 * draws[i] * 0.95
        cur, alpha_bar = prev.clamp(-4., 4.), pab
    assert (cur * 0.9 - ref).abs().max().item() < 2e-5
    print(f"mbd     ok: NoiseSchedule.generate (6 steps, beta_tilde, clip, rescale, noise_scale) max abs {(cur * 0.9 - ref).abs().max().item():.1e}")


CHECKS = {'epic': check_epic, 'mbd': check_mbd, 'lm': check_lm, 'bias': check_bias, 'rope': check_rope, 'options': check_options, 'hf': check_hf_encodec, 'melody': check_melody, 'stereo': check_stereo, 'codec': check_codec}

if __name__ == '__main__':
    torch.set_num_threads(8)
    names = sys.argv[1:] or list(CHECKS)
    for n in names:
        CHECKS[n]()
    print("all checks passed")
