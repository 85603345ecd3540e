"""Generate the golden fixtures under tests/golden/ by running the UNMODIFIED reference
(/root/reference, imported through oracle/refstubs.py) on seeded tiny models.

Run in the build container only:   python tests/golden/make_golden.py
The resulting *.npz files are committed; neither the tests nor bench.py ever need /root/reference.

Every fixture stores: the reference state dict ("sd/<key>"), the constructor config ("cfg" as
json), the seeded inputs and the reference outputs.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
from oracle import refstubs  # noqa: E402,F401  (installs import stubs, then the reference is importable)

from audiocraft.models.encodec import EncodecModel  # noqa: E402
from audiocraft.models.lm import LMModel  # noqa: E402
from audiocraft.modules.codebooks_patterns import DelayedPatternProvider  # noqa: E402
from audiocraft.modules.conditioners import (  # noqa: E402
    ConditioningProvider, ConditionFuser, ConditioningAttributes, TextConditioner, WaveformConditioner,
    WavCondition, ClassifierFreeGuidanceDropout)
from audiocraft.modules.seanet import SEANetEncoder, SEANetDecoder  # noqa: E402
from audiocraft.quantization.vq import ResidualVectorQuantizer  # noqa: E402
from audiocraft.utils import utils as ref_utils  # noqa: E402
from audiocraft.modules.transformer import create_sin_embedding  # noqa: E402


def save(name, cfg, sd, **arrays):
    out = {'cfg': np.array(json.dumps(cfg))}
    for k, v in sd.items():
        out['sd/' + k] = v.detach().cpu().numpy()
    for k, v in arrays.items():
        out[k] = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print(f'wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB')


# --------------------------------------------------------------------------------- codec

def make_codec(name, cfg, wav_shape, seed):
    torch.manual_seed(seed)
    kw = dict(channels=cfg['channels'], dimension=cfg['dimension'], n_filters=cfg['n_filters'],
              n_residual_layers=cfg['n_residual_layers'], ratios=cfg['ratios'], activation='ELU',
              activation_params={'alpha': cfg['elu_alpha']}, norm=cfg['norm'], norm_params={},
              kernel_size=cfg['kernel_size'], residual_kernel_size=cfg['residual_kernel_size'],
              last_kernel_size=cfg['last_kernel_size'], dilation_base=cfg['dilation_base'],
              causal=cfg['causal'], pad_mode=cfg['pad_mode'], true_skip=cfg['true_skip'],
              compress=cfg['compress'], lstm=cfg['lstm'], disable_norm_outer_blocks=0)
    enc = SEANetEncoder(**kw)
    dec = SEANetDecoder(**kw, trim_right_ratio=cfg['trim_right_ratio'])
    q = ResidualVectorQuantizer(dimension=cfg['dimension'], n_q=cfg['n_q'], bins=cfg['bins'], kmeans_init=False)
    m = EncodecModel(enc, dec, q, frame_rate=cfg['frame_rate'], sample_rate=cfg['sample_rate'],
                     channels=cfg['channels'], causal=cfg['causal'], renormalize=cfg['renormalize']).eval()
    # give weight_g a non-trivial value so that the weight-norm fold is actually exercised
    with torch.no_grad():
        for k, p in m.named_parameters():
            if k.endswith('weight_g'):
                p.mul_(1.0 + 0.25 * torch.rand_like(p))
            if k.endswith('.bias'):
                p.add_(0.05 * torch.randn_like(p))
    wav = 0.5 * torch.randn(*wav_shape)
    with torch.no_grad():
        latents = m.encoder(m.preprocess(wav)[0])
        codes, scale = m.encode(wav)
        decoded = m.decode(codes, scale)
        qlat = m.decode_latent(codes)
        fwd = m(wav).x
    arrays = dict(wav=wav, latents=latents, codes=codes, decoded=decoded, quantized_latents=qlat, forward=fwd)
    if scale is not None:
        arrays['scale'] = scale
    save(name, cfg, m.state_dict(), **arrays)


BASE_CODEC = dict(channels=1, dimension=16, n_filters=4, n_residual_layers=1, ratios=[4, 3, 2], kernel_size=7,
                  last_kernel_size=7, residual_kernel_size=3, dilation_base=2, causal=False, pad_mode='constant',
                  true_skip=True, compress=2, lstm=2, norm='weight_norm', elu_alpha=1.0, trim_right_ratio=1.0,
                  n_q=4, bins=32, sample_rate=1200, frame_rate=50, renormalize=False)


# --------------------------------------------------------------------------------- LM

class SynthText(TextConditioner):
    """Stand-in for T5Conditioner (conditioners.py:422-515): seeded `randn` instead of the T5
    encoder output, then the real `output_proj` and mask multiply (conditioners.py:509-515)."""
    def __init__(self, dim, output_dim, L):
        super().__init__(dim, output_dim)
        self.L = L

    def tokenize(self, x):
        return x

    def forward(self, x):
        g = torch.Generator().manual_seed(1234)
        B = len(x)
        mask = torch.tensor([[1] * self.L if xi is not None else [0] * self.L for xi in x])
        # ragged lengths: odd rows are 2 tokens shorter (padding rows are zero but NOT masked in attention)
        for b in range(B):
            if x[b] is not None and b % 2 == 1:
                mask[b, self.L - 2:] = 0
        e = torch.randn(B, self.L, self.dim, generator=g)
        e = self.output_proj(e) * mask.unsqueeze(-1)
        return e, mask


class SynthChroma(WaveformConditioner):
    """Stand-in for ChromaStemConditioner (conditioners.py:571-759): seeded one-hot chroma frames,
    zeros for null wavs, then the real WaveformConditioner.forward (output_proj, mask)."""
    def __init__(self, output_dim, P):
        super().__init__(12, output_dim, 'cpu')
        self.P = P
        self._use_masking = False  # musicgen.py:90-92

    def _downsampling_factor(self):
        return 1

    def _get_wav_embedding(self, x):
        g = torch.Generator().manual_seed(4321)
        B = x.wav.shape[0]
        cls = torch.randint(0, 12, (B, self.P), generator=g)
        e = torch.nn.functional.one_hot(cls, 12).float()
        null = (x.length == 0).view(-1, 1, 1)
        return torch.where(null, torch.zeros_like(e), e)


def build_lm(cfg, conditioners, fuse):
    torch.manual_seed(cfg['seed'])
    cp = ConditioningProvider(conditioners)
    fuser = ConditionFuser(fuse)
    lm = LMModel(DelayedPatternProvider(cfg['n_q'], delays=cfg['delays']), cp, fuser, n_q=cfg['n_q'],
                 card=cfg['card'], dim=cfg['dim'], num_heads=cfg['num_heads'], hidden_scale=cfg['hidden_scale'],
                 norm='layer_norm', norm_first=True, bias_proj=False, weight_init='gaussian',
                 depthwise_init='current', zero_bias_init=True, cfg_coef=cfg['cfg_coef'],
                 num_layers=cfg['num_layers'], dropout=0., activation='gelu', bias_ff=False, bias_attn=False,
                 causal=True, custom=False, memory_efficient=True, attention_as_float32=False,
                 cross_attention=cfg['cross_attention'], positional_embedding='sin').eval()
    with torch.no_grad():  # make LayerNorm affine parameters non-trivial
        for k, p in lm.named_parameters():
            if '.norm' in k or k.startswith('out_norm'):
                p.add_(0.1 * torch.randn_like(p))
    return lm


def run_lm(lm, conditions, prompt, max_gen_len, **gen_kw):
    """Run reference generate, recording every forward's logits ([cond; uncond] rows, all steps)."""
    rec = []
    h = lm.register_forward_hook(lambda mod, inp, out: rec.append(out.detach().clone()))
    tokens = lm.generate(prompt, conditions, max_gen_len=max_gen_len, **gen_kw)
    h.remove()
    # condition tensors exactly as generate() builds them (lm.py:497-509)
    null = ClassifierFreeGuidanceDropout(p=1.0)(conditions)
    ct = lm.condition_provider(lm.condition_provider.tokenize(conditions + null))
    return tokens, rec, ct


def make_lm_text():
    cfg = dict(dim=32, num_heads=4, num_layers=2, hidden_scale=4, n_q=4, card=32, cross_attention=True,
               delays=[0, 1, 2, 3], cfg_coef=3.0, seed=0, cond_dim=8, Lc=5)
    lm = build_lm(cfg, {'description': SynthText(cfg['cond_dim'], cfg['dim'], cfg['Lc'])},
                  {'cross': ['description'], 'prepend': [], 'sum': [], 'input_interpolate': []})
    conds = [ConditioningAttributes(text={'description': f'p{i}'}) for i in range(3)]
    arrays = {}
    # (1) greedy, no prompt
    tokens, rec, ct = run_lm(lm, conds, None, 12, use_sampling=False)
    arrays['cross_src'] = ct['description'][0]
    arrays['cross_mask'] = ct['description'][1]
    arrays['greedy_tokens'] = tokens
    arrays['greedy_step_logits'] = torch.stack([r[:, :, -1] for r in rec], dim=2)  # [2B, K, steps, card]
    # (2) greedy continuation from a 3-step prompt (first forward sees S>1 tokens)
    g = torch.Generator().manual_seed(5)
    prompt = torch.randint(0, cfg['card'], (3, cfg['n_q'], 3), generator=g)
    tokens, rec, _ = run_lm(lm, conds, prompt, 10, use_sampling=False)
    arrays['prompt'] = prompt
    arrays['cont_tokens'] = tokens
    arrays['cont_first_logits'] = rec[0]  # [2B, K, S0, card]
    tokens_rp, _, _ = run_lm(lm, conds, prompt, 10, use_sampling=False, remove_prompts=True)
    arrays['cont_tokens_removed'] = tokens_rp
    # (3) non-streaming teacher-forced forward over a full pattern sequence (batch == streaming check)
    seq = torch.randint(0, cfg['card'] + 1, (6, cfg['n_q'], 9), generator=g)
    with torch.no_grad():
        arrays['tf_sequence'] = seq
        arrays['tf_logits'] = lm(seq, [], ct)
    # (4) sampling helpers on fixed probabilities (deterministic halves)
    probs = torch.softmax(2.0 * torch.randn(3, 4, 32, generator=g), dim=-1)
    pk = probs.clone()
    kth = torch.topk(pk, 5, dim=-1)[0][..., [-1]]
    pk *= (pk >= kth).float()
    pk.div_(pk.sum(dim=-1, keepdim=True))
    arrays['probs'] = probs
    arrays['probs_top5'] = pk
    arrays['sin_emb'] = create_sin_embedding(torch.arange(7).view(1, -1, 1) + 3, cfg['dim'])
    save('lm_text', cfg, lm.state_dict(), **arrays)


def make_lm_melody():
    cfg = dict(dim=32, num_heads=4, num_layers=2, hidden_scale=4, n_q=4, card=32, cross_attention=False,
               delays=[0, 1, 2, 3], cfg_coef=3.0, seed=1, cond_dim=8, Lc=3, P=6)
    lm = build_lm(cfg, {'description': SynthText(cfg['cond_dim'], cfg['dim'], cfg['Lc']),
                        'self_wav': SynthChroma(cfg['dim'], cfg['P'])},
                  {'cross': [], 'prepend': ['self_wav', 'description'], 'sum': [], 'input_interpolate': []})
    conds = []
    for i in range(2):
        c = ConditioningAttributes(text={'description': f'm{i}'})
        c.wav['self_wav'] = WavCondition(torch.randn(1, 1, 64), torch.tensor([64]), [1200], [None], [0.])
        conds.append(c)
    tokens, rec, ct = run_lm(lm, conds, None, 9, use_sampling=False)
    # fuser order (conditioners.py:1730-1741): dict order is text then wav, each prepend goes in
    # front, so the final prefix is [self_wav ; description ; tokens]
    prepend = torch.cat([ct['self_wav'][0], ct['description'][0]], dim=1)
    save('lm_melody', cfg, lm.state_dict(), prepend_src=prepend, greedy_tokens=tokens,
         greedy_step_logits=torch.stack([r[:, :, -1] for r in rec], dim=2))


if __name__ == '__main__':
    make_codec('codec_noncausal', BASE_CODEC, (2, 1, 517), seed=0)
    make_codec('codec_causal', dict(BASE_CODEC, causal=True, pad_mode='reflect', n_residual_layers=2,
                                    true_skip=False, lstm=1, norm='none', ratios=[4, 2, 2], n_q=8, bins=16,
                                    frame_rate=75, renormalize=False, trim_right_ratio=1.0), (1, 1, 403), seed=1)
    make_codec('codec_renorm', dict(BASE_CODEC, lstm=0, renormalize=True, ratios=[5, 2], channels=2,
                                    kernel_size=5, last_kernel_size=3), (2, 2, 250), seed=2)
    make_lm_text()
    make_lm_melody()
