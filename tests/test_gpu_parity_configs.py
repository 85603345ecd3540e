"""-m gpu: parity with the CPU oracle AT THE GEOMETRIES THE NUMBERS ARE QUOTED ON (BASELINE.json configs):

  configs[2]  MusicGen-medium (d 1536 / 48 layers / 24 heads), bf16 weights + bf16 KV, 8 prompts -> 16 CFG rows:
              teacher-forced logits over 32 positions, and a late-context slice (1400-token prompt through the
              8-positions-per-call prefill, then decode steps at t ~ 1400);
  configs[3]  MusicGen-large (d 2048: 128 LayerNorm-statistics partials, the a_np <= 128 edge), one GPU's shard;
  configs[4]  MusicGen-melody (no cross-attention, 235 chroma + 16 text prepended rows through the prefill);
  f2          stereo 8-codebook delays [0,0,1,1,2,2,3,3] and the 16 kHz EnCodec geometry;
  a13         ConditioningProvider -> output_proj -> mask, end to end against the reference golden.

The oracle sees the same bf16-rounded matrices (its activations stay f32); tolerance rel-L2 <= 3e-2 is the
bf16 gate of BASELINE.md section 4 / SURVEY.md section 8(d).  Where the device run is free-running (generate),
the oracle is teacher-forced with the device's own tokens, so an argmax flip cannot hide or fake a mismatch.
Oracle cost is bounded by checking one CFG pair of the batch (rows are independent; the device still runs all).
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import load_golden  # noqa: E402
from oracle import codec as ocodec  # noqa: E402
from oracle import lm as olm  # noqa: E402
from oracle import patterns as opat  # noqa: E402
from parity_utils import assert_codes_near_tie  # noqa: E402

BF16_TOL = 3e-2


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def _oracle_sd(lm, bf16: bool):
    """Reference-format state dict on the CPU; matrices rounded to bf16 like the packed device copies."""
    sd = {}
    for k, v in lm.state_dict().items():
        v = v.detach().float().cpu()
        if bf16 and v.dim() == 2 and 'output_proj' not in k:
            v = v.bfloat16().float()
        sd[k] = v
    return sd


def _perturb_norms(lm, scale=0.1):
    with torch.no_grad():
        for k, p in lm.named_parameters():
            if 'norm' in k:
                p.add_(scale * torch.randn_like(p))


def _build(scale, melody=False, text_len=16, wdt=torch.bfloat16):
    from audiocraft_amd.models import builders
    torch.manual_seed(0)
    lm = builders.get_lm_model(builders.musicgen_lm_cfg(scale, melody=melody, text_len=text_len), 'cuda', wdt)
    _perturb_norms(lm)
    return lm


def _cross(Beff, Lc, d, seed):
    g = torch.Generator().manual_seed(seed)
    cross = torch.randn(Beff, Lc, d, generator=g)
    cross[Beff // 2:] = 0      # null conditions: all-zero source (conditioners.py:492-506, 514)
    return cross


def _oracle_cfg_logits(sd, oc, seq_pair, cross_pair, prepend_pair, coef):
    """Non-streaming (batch == streaming) oracle forward of one [cond; uncond] pair -> CFG-mixed logits [1, K, S, card]."""
    lg = olm.lm_forward(sd, oc, seq_pair, cross_pair, prepend_pair)
    return olm.cfg_mix(lg, coef)


# ------------------------------------------------------------------------------------------ configs[2]

def test_medium_bf16_cfg16_teacher_forced():
    """d 1536 / L 48 / H 24, cross-attention, bf16 weights + bf16 KV, 16 rows, 32 positions: the folded
    hi / lo LayerNorm through 48 layers and the stacked 8192-row head with out_norm folded, at size."""
    lm = _build('medium')
    sd = _oracle_sd(lm, True)
    oc = olm.LMConfig(dim=1536, num_heads=24, num_layers=48, n_q=4, card=2048, cross_attention=True)
    Beff, S = 16, 32
    cross = _cross(Beff, 16, 1536, 11)
    ct = {'description': (cross.cuda(), torch.ones(Beff, 16, dtype=torch.int64).cuda())}
    seq = torch.randint(0, 2049, (Beff, 4, S), generator=torch.Generator().manual_seed(12))
    got = lm.forward_steps(seq.cuda(), ct).cpu()
    ref = olm.lm_forward(sd, oc, seq, cross)
    assert got.shape == ref.shape == (Beff, 4, S, 2048)
    r = rel(got, ref)
    print(f"[parity] medium bf16 16 rows x 32 positions: logits rel-L2 {r:.3e}")
    assert r < BF16_TOL, f"teacher-forced logits rel-L2 {r}"
    # no drift with depth of context: the last 8 positions alone meet the same gate
    r_last = rel(got[:, :, -8:], ref[:, :, -8:])
    assert r_last < BF16_TOL, f"last positions rel-L2 {r_last}"
    # per-row worst case (a single bad row must not hide in the norm of 16)
    worst = max(rel(got[b], ref[b]) for b in range(Beff))
    assert worst < 2 * BF16_TOL, f"worst row rel-L2 {worst}"


def test_medium_bf16_sampled_decisions_accounted_for():
    """The headline MODE in one piece: MusicGen-medium architecture, bf16 weights + bf16 KV, CFG, top-k 250, temperature 1,
    SAMPLED, 204 frames x 2 samples (>= 1600 decisions).  Every sampled token is accounted for in three steps
    (reference arithmetic: lm.py:391-399 CFG mix, :402-418 softmax / top-k / multinomial, utils.py:108-122):
      (1) sampler given ITS inputs: oracle/sampler.py's race replayed on the device's own CFG-mixed logits (softmax on the host)
          must give the device's token -- every one (near ties at 1e-4 are reported);
      (2) the inputs: device logits vs the oracle's f32 logits, teacher-forced on the device's sequence: rel-L2 under the bf16
          gate over all steps, and the per-step centred max error `delta` is measured;
      (3) the decisions: the race replayed on the ORACLE's probabilities.  Where its token differs from the device's, the
          oracle's log race margin between the two candidates must be below 2 delta of that step (the most a logit error of
          delta can move a ratio p_i / p_j), or the loser must sit within 2 delta of the top-k threshold -- i.e. every
          differing decision is inside the measured bf16 logit error.  The differing fraction is printed and bounded."""
    import numpy as np
    from oracle.sampler import race, uniforms
    lm = _build('medium')
    with torch.no_grad():
        for k, p in lm.named_parameters():
            if 'linears' in k:          # sharper logits: a top-250 filter that actually cuts mass
                p.mul_(4.0)
    sd = _oracle_sd(lm, True)
    oc = olm.LMConfig(dim=1536, num_heads=24, num_layers=48, n_q=4, card=2048, cross_attention=True)
    B, K, T, card, top_k, seed, coef = 2, 4, 204, 2048, 250, 0x0badc0de5eed, 3.0
    cross = _cross(2 * B, 16, 1536, 41)
    ct = {'description': (cross.cuda(), torch.ones(2 * B, 16, dtype=torch.int64).cuda())}
    toks, lg = lm.generate(None, [], num_samples=B, max_gen_len=T, use_sampling=True, temp=1.0, top_k=top_k, seed=seed,
                           condition_tensors=ct, return_logits=True, check=True, cfg_coef=coef)
    toks, lg = toks.cpu(), lg.float().cpu()               # [B, K, T], [B, K, steps, card] (CFG-mixed, what the sampler saw)
    del lm
    torch.cuda.empty_cache()
    seq, mask = opat.build_pattern_sequence(toks, card)    # [B, K, S]
    S = seq.shape[-1]
    steps = lg.shape[2]
    first = S - steps                                      # sequence position the first step fills
    assert first == 1 and steps >= 200
    ref = olm.cfg_mix(olm.lm_forward(sd, oc, torch.cat([seq, seq], dim=0)[..., :S - 1], cross), coef)   # [B, K, S - 1, card]
    assert ref.shape == lg.shape
    r_all = rel(lg, ref)
    print(f"[parity] medium bf16 sampled, {steps} steps x {B} samples: CFG logits rel-L2 {r_all:.3e}")
    assert r_all < BF16_TOL, r_all
    lg64, ref64 = lg.double().numpy(), ref.double().numpy()

    def softmax(x):
        e = np.exp(x - x.max())
        return e / e.sum()
    checked = sampler_wrong = sampler_near = differ = unexplained = 0
    deltas, worst = [], []
    for offset in range(1, S):
        gpos = offset - 1
        for b in range(B):
            for k in range(K):
                if not bool(mask[k, offset]):
                    continue
                checked += 1
                dev_tok = int(seq[b, k, offset])
                ld, lo = lg64[b, k, offset - 1], ref64[b, k, offset - 1]
                # (1) the sampler on its own inputs (f32 softmax like the device's)
                t1, m1, bd1 = race(softmax(ld.astype(np.float32).astype(np.float64)), top_k, b * K + k, gpos, seed)
                if t1 != dev_tok:
                    if m1 < 1e-4 or bd1:
                        sampler_near += 1
                    else:
                        sampler_wrong += 1
                # (2) centred logit error of this step over the union of the two top-k supports
                sup = (lo >= np.partition(lo, card - top_k)[card - top_k]) | (ld >= np.partition(ld, card - top_k)[card - top_k])
                e = (ld - lo)[sup]
                delta = float(np.abs(e - e.mean()).max())
                deltas.append(delta)
                # (3) the oracle's decision with the device's random stream
                t3, _, _ = race(softmax(lo), top_k, b * K + k, gpos, seed)
                if t3 != dev_tok:
                    differ += 1
                    q = -np.log(uniforms(card, b * K + k, gpos, seed))
                    thr = np.partition(lo, card - top_k)[card - top_k]
                    # log of the oracle's race ratio between its winner and the device's token, or the device token's distance
                    # below the oracle's top-k threshold when the oracle does not have it in the support at all
                    gap = (lo[t3] - np.log(q[t3])) - (lo[dev_tok] - np.log(q[dev_tok]))
                    below = max(0.0, float(thr - lo[dev_tok]))
                    # the oracle's winner may in turn be outside the DEVICE's support
                    thr_d = np.partition(ld, card - top_k)[card - top_k]
                    below_d = max(0.0, float(thr_d - ld[t3]))
                    ok = (below <= 2 * delta + 1e-6) if below > 0 else ((gap <= 2 * delta + 1e-6) or (0 < below_d <= 2 * delta + 1e-6))
                    if not ok:
                        unexplained += 1
                        worst.append((b, k, offset, dev_tok, t3, float(gap), below, below_d, delta))
    deltas = np.array(deltas)
    print(f"[parity] medium bf16 sampled: {checked} decisions; sampler on its own logits: {sampler_wrong} wrong, {sampler_near} near "
          f"ties; centred logit error per step: median {np.median(deltas):.3e}, max {deltas.max():.3e}; oracle replay differs on "
          f"{differ} decisions ({100.0 * differ / checked:.2f} %), {unexplained} of them outside 2 x the step's logit error")
    assert checked >= 1600
    assert sampler_wrong == 0 and sampler_near <= 3, (sampler_wrong, sampler_near)
    assert unexplained == 0, worst[:5]
    assert differ <= 0.10 * checked, differ      # bf16 through 48 layers moves a few race outcomes, not the distribution


@pytest.mark.parametrize('wdt,tol', [(torch.float32, 1e-4), (torch.bfloat16, BF16_TOL)])
def test_rotary_window_small_geometry_vs_oracle(wdt, tol):
    """Rotary positions + xPos, past_context and LayerScale at the MusicGen-small geometry (d 1024 / 16 heads of 64; 6
    layers keep the oracle cheap): 96 teacher-forced positions against a window of 40 -- the attention kernel's start
    offset crosses several of its 64-position chunks --, rows of 5, f32 and bf16 (bf16 cache: k is rounded twice, by the
    GEMM's store and by the rotary launch)."""
    from audiocraft_amd.models import builders
    torch.manual_seed(0)
    cfg = dict(builders.musicgen_lm_cfg('small', text_len=12), num_layers=6, positional_embedding='rope', xpos=True,
               past_context=40, layer_scale=0.25, positional_scale=0.9)
    lm = builders.get_lm_model(cfg, 'cuda', wdt)
    _perturb_norms(lm)
    with torch.no_grad():
        for k, p in lm.named_parameters():
            if 'layer_scale' in k:
                p.mul_(1.0 + 0.5 * torch.rand_like(p))
    sd = _oracle_sd(lm, wdt == torch.bfloat16)
    assert any('layer_scale_cross' in k for k in sd) and 'transformer.rope.frequencies' in sd
    oc = olm.LMConfig(dim=1024, num_heads=16, num_layers=6, n_q=4, card=2048, cross_attention=True,
                      positional_embedding='rope', xpos=True, past_context=40, positional_scale=0.9)
    Beff, S = 5, 96
    cross = torch.randn(Beff, 12, 1024, generator=torch.Generator().manual_seed(21))
    ct = {'description': (cross.cuda(), torch.ones(Beff, 12, dtype=torch.int64).cuda())}
    seq = torch.randint(0, 2049, (Beff, 4, S), generator=torch.Generator().manual_seed(22))
    got = lm.forward_steps(seq.cuda(), ct).cpu()
    ref = olm.lm_forward(sd, oc, seq, cross)
    r, r_last = rel(got, ref), rel(got[:, :, -8:], ref[:, :, -8:])
    print(f"[parity] rotary + window, small geometry, {wdt}: logits rel-L2 {r:.3e} (last 8 positions {r_last:.3e})")
    assert r < tol and r_last < tol, (r, r_last)


def test_medium_bf16_late_context_prefill_then_decode():
    """1400-token prompt for 8 samples (16 CFG rows) through the 8-positions-per-call prefill (128 rows per GEMM
    launch: 4 row blocks x 2 row groups, per-row positions in the QKV scatter and the attention), then 11 decode
    positions with bf16 KV at t ~ 1400 inside the model.  Oracle: batch forward of sample 0's [cond; uncond] pair
    over the whole pattern sequence, teacher-forced with the device's tokens."""
    lm = _build('medium')
    sd = _oracle_sd(lm, True)
    oc = olm.LMConfig(dim=1536, num_heads=24, num_layers=48, n_q=4, card=2048, cross_attention=True)
    B, T0, T = 8, 1400, 1408
    cross = _cross(2 * B, 16, 1536, 21)
    ct = {'description': (cross.cuda(), torch.ones(2 * B, 16, dtype=torch.int64).cuda())}
    prompt = torch.randint(0, 2048, (B, 4, T0), generator=torch.Generator().manual_seed(22))
    toks, lg = lm.generate(prompt.cuda(), [], max_gen_len=T, use_sampling=False, condition_tensors=ct,
                           return_logits=True, check=True)
    toks, lg = toks.cpu(), lg.cpu()            # [B, 4, T], [B, 4, steps, card]
    assert torch.equal(toks[..., :T0], prompt)
    steps = lg.shape[2]
    # the full pattern sequence of sample 0 as the device filled it
    seq, _ = opat.build_pattern_sequence(toks[:1], 2048)          # [1, 4, T + 4]
    S = seq.shape[-1]
    first = S - steps                                            # first sequence step the device sampled
    pair = torch.cat([seq, seq], dim=0)[..., :S - 1]             # inputs: steps 0 .. S-2 predict steps 1 .. S-1
    ref = _oracle_cfg_logits(sd, oc, pair, cross[[0, B]], None, 3.0)   # [1, 4, S-1, card]
    ref_steps = ref[:, :, first - 1:]                            # outputs that produced steps first .. S-1
    assert ref_steps.shape[2] == steps
    r = rel(lg[:1], ref_steps)
    print(f"[parity] medium bf16 late context (t ~ 1400, after 8-per-call prefill): CFG logits rel-L2 {r:.3e}")
    assert r < BF16_TOL, f"late-context CFG logits rel-L2 {r}"
    r0 = rel(lg[:1, :, :1], ref_steps[:, :, :1])                 # the very first decode position after the prefill
    assert r0 < BF16_TOL, f"first decode position after prefill rel-L2 {r0}"


# ------------------------------------------------------------------------------------------ configs[3]

def test_large_bf16_cfg16_teacher_forced():
    """d 2048 / L 48 / H 32: 128 statistics partials per row (np <= 128 edge of the folded LayerNorm), 8 positions."""
    lm = _build('large')
    sd = _oracle_sd(lm, True)
    oc = olm.LMConfig(dim=2048, num_heads=32, num_layers=48, n_q=4, card=2048, cross_attention=True)
    Beff, S = 16, 8
    cross = _cross(Beff, 16, 2048, 31)
    ct = {'description': (cross.cuda(), torch.ones(Beff, 16, dtype=torch.int64).cuda())}
    seq = torch.randint(0, 2049, (Beff, 4, S), generator=torch.Generator().manual_seed(32))
    got = lm.forward_steps(seq.cuda(), ct).cpu()
    del lm
    torch.cuda.empty_cache()
    ref = olm.lm_forward(sd, oc, seq, cross)
    r = rel(got, ref)
    print(f"[parity] large bf16 16 rows x 8 positions: logits rel-L2 {r:.3e}")
    assert r < BF16_TOL, f"teacher-forced logits rel-L2 {r}"


def test_large_bf16_late_context_prefill_then_decode():
    """MusicGen-large (d 2048 / 48 layers / 32 heads) at late context: a 1000-token prompt for 4 samples (8 CFG rows)
    through the prefill, then 11 decode positions at t ~ 1000 with bf16 weights + bf16 KV.  Oracle: batch forward of
    sample 0's [cond; uncond] pair over the whole pattern sequence, teacher-forced with the device's tokens."""
    lm = _build('large')
    sd = _oracle_sd(lm, True)
    oc = olm.LMConfig(dim=2048, num_heads=32, num_layers=48, n_q=4, card=2048, cross_attention=True)
    B, T0, T = 4, 1000, 1008
    cross = _cross(2 * B, 16, 2048, 33)
    ct = {'description': (cross.cuda(), torch.ones(2 * B, 16, dtype=torch.int64).cuda())}
    prompt = torch.randint(0, 2048, (B, 4, T0), generator=torch.Generator().manual_seed(34))
    toks, lg = lm.generate(prompt.cuda(), [], max_gen_len=T, use_sampling=False, condition_tensors=ct,
                           return_logits=True, check=True)
    toks, lg = toks.cpu(), lg.cpu()
    del lm
    torch.cuda.empty_cache()
    assert torch.equal(toks[..., :T0], prompt)
    steps = lg.shape[2]
    assert steps >= 8
    seq, _ = opat.build_pattern_sequence(toks[:1], 2048)
    S = seq.shape[-1]
    first = S - steps
    pair = torch.cat([seq, seq], dim=0)[..., :S - 1]
    ref = _oracle_cfg_logits(sd, oc, pair, cross[[0, B]], None, 3.0)
    ref_steps = ref[:, :, first - 1:]
    assert ref_steps.shape[2] == steps
    r = rel(lg[:1], ref_steps)
    print(f"[parity] large bf16 late context (t ~ 1000, after the prefill): CFG logits rel-L2 {r:.3e}")
    assert r < BF16_TOL, f"late-context CFG logits rel-L2 {r}"


# ------------------------------------------------------------------------------------------ configs[4]

def test_melody_medium_bf16_real_prefix():
    """MusicGen-melody architecture: no cross-attention, prefix = 235 chroma rows + 16 text rows prepended in the
    order [self_wav; description; tokens] (conditioners.py:1730-1741), 251 rows through the 8-per-call prefill
    (31 full calls + a 3-position tail), 16 samples -> 32 CFG rows."""
    from audiocraft_amd.modules.conditioners import ConditioningAttributes, WavCondition
    lm = _build('medium', melody=True)
    sd = _oracle_sd(lm, True)
    oc = olm.LMConfig(dim=1536, num_heads=24, num_layers=48, n_q=4, card=2048, cross_attention=False)
    B, T = 16, 10
    conds = []
    for i in range(B):
        c = ConditioningAttributes(text={'description': f'melody {i}'})
        c.wav['self_wav'] = WavCondition(torch.randn(1, 1, 32000), torch.tensor([32000]), [32000], [None], [0.])
        conds.append(c)
    cfg_conditions = lm._cfg_condition_tensors(conds)
    prepend, cross_src = lm.fuser.fuse(cfg_conditions)
    assert cross_src is None and prepend.shape == (2 * B, 235 + 16, 1536)
    # null rows: zero chroma / zero text embeddings x output_proj = bias only ... x mask (0 for text) -> see oracle input
    toks, lg = lm.generate(None, [], num_samples=B, max_gen_len=T, use_sampling=False, condition_tensors=cfg_conditions,
                           return_logits=True, check=True)
    toks, lg = toks.cpu(), lg.cpu()
    seq, _ = opat.build_pattern_sequence(toks[:1], 2048)
    S = seq.shape[-1]
    pair = torch.cat([seq, seq], dim=0)[..., :S - 1]
    pre = prepend.float().cpu()[[0, B]]
    ref = _oracle_cfg_logits(sd, oc, pair, None, pre, 3.0)        # logits cropped to the S-1 token steps
    assert lg.shape[2] == S - 1
    r = rel(lg[:1], ref)
    print(f"[parity] melody-medium bf16, 251-row prefix: CFG logits rel-L2 {r:.3e}")
    assert r < BF16_TOL, f"melody CFG logits rel-L2 {r}"


# ------------------------------------------------------------------------------------------ f2: stereo, 16 kHz codec

@pytest.mark.parametrize('wdt,tol', [(torch.float32, 1e-4), (torch.bfloat16, BF16_TOL)])
def test_stereo_delays_vs_oracle(wdt, tol):
    """8 codebooks with the stereo delays [0,0,1,1,2,2,3,3] (builders.py:338-351 of the reference): greedy tokens
    identical to the oracle in fp32 mode, logits within tolerance in both modes."""
    from audiocraft_amd.models import builders
    torch.manual_seed(0)
    delays = [0, 0, 1, 1, 2, 2, 3, 3]
    cfg = dict(dim=256, num_heads=4, num_layers=4, n_q=8, card=2048, hidden_scale=4, cfg_coef=3.0,
               conditioners={'description': {'kind': 't5', 'embedder': 'synthetic', 'dim': 64, 'length': 6}},
               fuser={'cross': ['description']},
               codebooks_pattern={'modeling': 'delay', 'delay': {'delays': delays}})
    lm = builders.get_lm_model(cfg, 'cuda', wdt)
    _perturb_norms(lm)
    sd = _oracle_sd(lm, wdt == torch.bfloat16)
    oc = olm.LMConfig(dim=256, num_heads=4, num_layers=4, n_q=8, card=2048, cross_attention=True, delays=delays)
    B, T = 3, 14
    cross = _cross(2 * B, 6, 256, 41)
    ct = {'description': (cross.cuda(), torch.ones(2 * B, 6, dtype=torch.int64).cuda())}
    toks, lg = lm.generate(None, [], num_samples=B, max_gen_len=T, use_sampling=False, condition_tensors=ct,
                           return_logits=True, check=True)
    assert toks.shape == (B, 8, T)
    if wdt == torch.float32:
        ref_t, ref_l = olm.generate(sd, oc, None, B, cross, max_gen_len=T, use_sampling=False, return_logits=True)
        assert torch.equal(toks.cpu(), ref_t)
        assert rel(lg.cpu(), ref_l) < tol
    else:   # teacher-force the oracle with the device's tokens
        seq, _ = opat.build_pattern_sequence(toks.cpu(), 2048, delays)
        S = seq.shape[-1]
        ref = olm.cfg_mix(olm.lm_forward(sd, oc, torch.cat([seq, seq], 0)[..., :S - 1], cross), 3.0)
        assert rel(lg.cpu(), ref) < tol
    # continuation from a stereo prompt (first forward sees several steps, shared delays)
    prompt = torch.randint(0, 2048, (B, 8, 5), generator=torch.Generator().manual_seed(42))
    if wdt == torch.float32:
        t2 = lm.generate(prompt.cuda(), [], max_gen_len=T, use_sampling=False, condition_tensors=ct, check=True)
        ref2 = olm.generate(sd, oc, prompt, B, cross, max_gen_len=T, use_sampling=False)
        assert torch.equal(t2.cpu(), ref2)


def test_encodec_16khz_geometry_vs_oracle():
    """AudioGen's codec (config/model/encodec/encodec_large_nq4_s320.yaml): n_filters 64, ratios [8,5,4,2] -> hop 320,
    50 fps at 16 kHz, RVQ 4 x 2048; seeded random weights vs the CPU oracle (same gates as the 32 kHz test)."""
    from audiocraft_amd.models import builders
    torch.manual_seed(0)
    m = builders.get_compression_model(builders.ENCODEC_16KHZ, 'cuda')
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    c = ocodec.CodecConfig(channels=1, dimension=128, n_filters=64, n_residual_layers=1, ratios=[8, 5, 4, 2],
                           causal=False, pad_mode='constant', lstm=2, norm='weight_norm', n_q=4, bins=2048,
                           sample_rate=16000, frame_rate=50)
    wav = 0.3 * torch.randn(2, 1, 20011, generator=torch.Generator().manual_seed(1))
    lat_ref = ocodec.seanet_encoder(sd, c, wav, fast_lstm=True)
    lat = m.encoder(wav.cuda()).cpu()
    assert lat.shape == lat_ref.shape == (2, 128, math.ceil(20011 / 320))
    assert rel(lat, lat_ref) < 2e-5
    codes_ref = ocodec.rvq_encode(lat_ref, ocodec.codebooks_from_state(sd, 4))
    assert torch.equal(m.quantizer.encode(lat_ref.cuda()).cpu(), codes_ref)   # identical latents: bit exact
    codes, _ = m.encode(wav.cuda())
    assert_codes_near_tie(codes, codes_ref, lat, lat_ref, ocodec.codebooks_from_state(sd, 4), what='EnCodec-16k geometry')
    dec_ref = ocodec.encodec_decode(sd, c, codes_ref, fast_lstm=True)
    dec = m.decode(codes_ref.cuda()).cpu()
    assert dec.shape == dec_ref.shape
    assert (dec - dec_ref).abs().max().item() < 1e-4


def test_stereo_codec_interleave_vs_oracle():
    """InterleaveStereoCompressionModel (encodec.py:397-506 of the reference): codes of left / right interleaved per
    RVQ level; decode == the oracle's mono decode of each de-interleaved half."""
    from audiocraft_amd.models import builders
    from audiocraft_amd.models.encodec import InterleaveStereoCompressionModel
    torch.manual_seed(0)
    ccfg = dict(builders.ENCODEC_32KHZ)
    ccfg['seanet'] = dict(ccfg['seanet'], n_filters=16)
    mono = builders.get_compression_model(ccfg, 'cuda')
    st = InterleaveStereoCompressionModel(mono)
    sd = {k: v.detach().cpu() for k, v in mono.state_dict().items()}
    c = ocodec.CodecConfig(channels=1, dimension=128, n_filters=16, n_residual_layers=1, ratios=[8, 5, 4, 4],
                           causal=False, pad_mode='constant', lstm=2, norm='weight_norm', n_q=4, bins=2048,
                           sample_rate=32000, frame_rate=50)
    codes = torch.randint(0, 2048, (2, 8, 40), generator=torch.Generator().manual_seed(5))
    dec = st.decode(codes.cuda()).cpu()
    left = ocodec.encodec_decode(sd, c, codes[:, 0::2], fast_lstm=True)
    right = ocodec.encodec_decode(sd, c, codes[:, 1::2], fast_lstm=True)
    ref = torch.cat([left, right], dim=1)
    assert dec.shape == ref.shape == (2, 2, 40 * 640)
    assert (dec - ref).abs().max().item() < 1e-4
    wav = 0.3 * torch.randn(2, 2, 6400, generator=torch.Generator().manual_seed(6))
    got, _ = st.encode(wav.cuda())
    lat_l = ocodec.seanet_encoder(sd, c, wav[:, :1], fast_lstm=True)
    lat_r = ocodec.seanet_encoder(sd, c, wav[:, 1:], fast_lstm=True)
    cb = ocodec.codebooks_from_state(sd, 4)
    ref_codes = torch.stack([ocodec.rvq_encode(lat_l, cb), ocodec.rvq_encode(lat_r, cb)], dim=2).reshape(2, 8, -1)
    assert got.shape == ref_codes.shape
    # per channel: a differing index must be a near tie of the oracle's decision (latents differ by fp32 round-off only)
    for ch, (lat_o, sl) in enumerate(((lat_l, slice(0, None, 2)), (lat_r, slice(1, None, 2)))):
        lat_d = mono.encoder(wav[:, ch:ch + 1].cuda()).cpu()
        assert_codes_near_tie(got[:, sl], ref_codes[:, sl], lat_d, lat_o, cb, what=f'stereo wrapper channel {ch}')


# ------------------------------------------------------------------------------------------ a13: provider end to end

def test_conditioning_provider_vs_reference_golden():
    """ConditioningAttributes -> ClassifierFreeGuidanceDropout(p=1) -> tokenize -> T5Conditioner.forward ->
    output_proj (acmi_linear) -> * mask, against `cross_src` / `cross_mask` recorded from the unmodified reference
    (tests/golden/make_golden.py: SynthText = seeded randn(seed 1234) in place of the T5 encoder, ragged lengths)."""
    from audiocraft_amd.models import builders
    from audiocraft_amd.modules.conditioners import ConditioningAttributes
    cfg, sd, a = load_golden('lm_text')
    L, dim = cfg['Lc'], cfg['cond_dim']

    def synth_text(texts):   # the golden generator's SynthText.forward up to (not including) output_proj
        g = torch.Generator().manual_seed(1234)
        B = len(texts)
        mask = torch.ones(B, L, dtype=torch.int64)
        for b in range(B):
            if texts[b] != "" and b % 2 == 1:
                mask[b, L - 2:] = 0
        return torch.randn(B, L, dim, generator=g), mask

    conds_cfg = {'description': {'kind': 't5', 'embedder': synth_text, 'dim': dim}}
    lm = builders.get_lm_model(dict(dim=cfg['dim'], num_heads=cfg['num_heads'], num_layers=cfg['num_layers'],
                                    n_q=cfg['n_q'], card=cfg['card'], hidden_scale=cfg['hidden_scale'],
                                    cfg_coef=cfg['cfg_coef'], conditioners=conds_cfg, fuser={'cross': ['description']}),
                               'cuda', torch.float32)
    lm.load_state_dict(sd, strict=True)
    conds = [ConditioningAttributes(text={'description': f'p{i}'}) for i in range(3)]
    ct = lm._cfg_condition_tensors(conds)
    emb, mask = ct['description']
    assert torch.equal(mask.cpu().to(a['cross_mask'].dtype), a['cross_mask'])
    assert emb.shape == a['cross_src'].shape
    assert torch.allclose(emb.cpu(), a['cross_src'], atol=2e-6, rtol=1e-5), (emb.cpu() - a['cross_src']).abs().max()
    assert (emb[3:] == 0).all()                       # null conditions: exactly zero rows
    # and the whole path from attributes: greedy tokens identical to the reference
    toks = lm.generate(None, conds, max_gen_len=12, use_sampling=False, check=True)
    assert torch.equal(toks.cpu(), a['greedy_tokens'])
