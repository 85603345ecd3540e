"""Golden fixtures for the two non-default classifier-free-guidance modes of the reference's LMModel.generate
(SURVEY.md section 8f / DESIGN.md section 7: "next" rows; the oracle is pinned to them ahead of the HIP path):

  lm_two_step.npz    two_step_cfg=True (lm.py:377-386, 498-505): conditional and unconditional forwards run
                     separately, each with its own padded condition length and its own streaming state; the
                     mix uses the MODEL's cfg_coef (lm.py:386), not the argument.
  lm_two_step_prepend.npz  two_step_cfg=True on a PREPEND fuser (the melody models' fusing): the conditional pass prepends
                     5 condition positions, the unconditional pass 1 (an all-null text batch), so the two streams sit at
                     different transformer positions for the same token (lm.py:378-390: two streaming states).
  lm_double_cfg.npz  cfg_coef_beta (MusicGen-Style double CFG, lm.py:362-376, 490-496): rows
                     [text + wav; wav only; null], logits = u + coef * (w + beta * (c - w) - u).

Run in the build container only:   python tests/golden/make_cfg_modes_golden.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (installs the import stubs, defines the tiny-model builders)
from make_golden import ConditioningAttributes, WavCondition, TextConditioner  # noqa: E402


class SynthTextRagged(TextConditioner):
    """Like make_golden.SynthText, but an all-null batch is ONE position long -- what T5 gives for empty
    strings -- so that the two passes of two_step_cfg see different condition lengths."""
    def __init__(self, dim, output_dim, L):
        super().__init__(dim, output_dim)
        self.L = L

    def tokenize(self, x):
        return x

    def forward(self, x):
        g = torch.Generator().manual_seed(4321)
        B = len(x)
        L = 1 if all(xi is None for xi in x) else self.L
        mask = torch.tensor([[1] * L if xi is not None else [0] * L for xi in x])
        e = torch.randn(B, L, self.dim, generator=g)
        return self.output_proj(e) * mask.unsqueeze(-1), mask


def make_two_step():
    cfg = dict(dim=32, num_heads=4, num_layers=2, hidden_scale=4, n_q=4, card=32, cross_attention=True,
               delays=[0, 1, 2, 3], cfg_coef=3.0, seed=3, cond_dim=8, Lc=5)
    torch.manual_seed(1000 + cfg['seed'])   # the conditioners' output_proj is initialised HERE, before build_lm seeds
    lm = mg.build_lm(cfg, {'description': SynthTextRagged(cfg['cond_dim'], cfg['dim'], cfg['Lc'])},
                     {'cross': ['description'], 'prepend': [], 'sum': [], 'input_interpolate': []})
    conds = [ConditioningAttributes(text={'description': f'p{i}'}) for i in range(3)]
    rec = []
    h = lm.register_forward_hook(lambda mod, inp, out: rec.append(out.detach().clone()))
    # cfg_coef=7 is passed on purpose: the two-step branch ignores it and uses lm.cfg_coef (3.0)
    tokens = lm.generate(None, conds, max_gen_len=10, use_sampling=False, two_step_cfg=True, cfg_coef=7.0)
    h.remove()
    null = mg.ClassifierFreeGuidanceDropout(p=1.0)(conds)
    ct = lm.condition_provider(lm.condition_provider.tokenize(conds))
    nt = lm.condition_provider(lm.condition_provider.tokenize(null))
    assert ct['description'][0].shape[1] == 5 and nt['description'][0].shape[1] == 1
    mg.save('lm_two_step', cfg, lm.state_dict(), cross_src=ct['description'][0], null_cross_src=nt['description'][0],
            greedy_tokens=tokens, cond_step_logits=torch.stack([r[:, :, -1] for r in rec[0::2]], dim=2),
            uncond_step_logits=torch.stack([r[:, :, -1] for r in rec[1::2]], dim=2))


def make_two_step_prepend():
    cfg = dict(dim=32, num_heads=4, num_layers=2, hidden_scale=4, n_q=4, card=32, cross_attention=False,
               delays=[0, 1, 2, 3], cfg_coef=3.0, seed=6, cond_dim=8, Lc=5)
    torch.manual_seed(1000 + cfg['seed'])
    lm = mg.build_lm(cfg, {'description': SynthTextRagged(cfg['cond_dim'], cfg['dim'], cfg['Lc'])},
                     {'cross': [], 'prepend': ['description'], 'sum': [], 'input_interpolate': []})
    conds = [ConditioningAttributes(text={'description': f'q{i}'}) for i in range(3)]
    rec = []
    h = lm.register_forward_hook(lambda mod, inp, out: rec.append(out.detach().clone()))
    tokens = lm.generate(None, conds, max_gen_len=11, use_sampling=False, two_step_cfg=True, cfg_coef=7.0)
    h.remove()
    null = mg.ClassifierFreeGuidanceDropout(p=1.0)(conds)
    ct = lm.condition_provider(lm.condition_provider.tokenize(conds))
    nt = lm.condition_provider(lm.condition_provider.tokenize(null))
    assert ct['description'][0].shape[1] == 5 and nt['description'][0].shape[1] == 1
    # a second case: a 4-token prompt (the first call of both streams then covers prepend + prompt)
    gp = torch.Generator().manual_seed(9)
    prompt = torch.randint(0, cfg['card'], (3, cfg['n_q'], 4), generator=gp)
    tokens_p = lm.generate(prompt, conds, max_gen_len=12, use_sampling=False, two_step_cfg=True)
    mg.save('lm_two_step_prepend', cfg, lm.state_dict(), prepend_src=ct['description'][0], null_prepend_src=nt['description'][0],
            greedy_tokens=tokens, cond_step_logits=torch.stack([r[:, :, -1] for r in rec[0::2]], dim=2),
            uncond_step_logits=torch.stack([r[:, :, -1] for r in rec[1::2]], dim=2), prompt=prompt, greedy_tokens_prompt=tokens_p)


def make_double_cfg():
    cfg = dict(dim=32, num_heads=4, num_layers=2, hidden_scale=4, n_q=4, card=32, cross_attention=False,
               delays=[0, 1, 2, 3], cfg_coef=3.0, seed=4, cond_dim=8, Lc=3, P=6, cfg_coef_beta=5.0)
    torch.manual_seed(1000 + cfg['seed'])
    lm = mg.build_lm(cfg, {'description': mg.SynthText(cfg['cond_dim'], cfg['dim'], cfg['Lc']),
                           'self_wav': mg.SynthChroma(cfg['dim'], cfg['P'])},
                     {'cross': [], 'prepend': ['self_wav', 'description'], 'sum': [], 'input_interpolate': []})
    conds = []
    gw = torch.Generator().manual_seed(77)
    for i in range(2):
        c = ConditioningAttributes(text={'description': f'm{i}'})
        c.wav['self_wav'] = WavCondition(torch.randn(1, 1, 64, generator=gw), torch.tensor([64]), [1200], [None], [0.])
        conds.append(c)
    rec = []
    h = lm.register_forward_hook(lambda mod, inp, out: rec.append(out.detach().clone()))
    tokens = lm.generate(None, conds, max_gen_len=9, use_sampling=False, cfg_coef_beta=cfg['cfg_coef_beta'])
    h.remove()
    # the 3B-row condition batch exactly as generate() builds it (lm.py:490-496)
    from audiocraft.models.lm import _drop_description_condition
    null = mg.ClassifierFreeGuidanceDropout(p=1.0)(conds)
    allc = conds + _drop_description_condition(conds) + null
    ct = lm.condition_provider(lm.condition_provider.tokenize(allc))
    prepend = torch.cat([ct['self_wav'][0], ct['description'][0]], dim=1)   # [3B, P + Lc, d]
    mg.save('lm_double_cfg', cfg, lm.state_dict(), prepend_src=prepend, greedy_tokens=tokens,
            step_logits=torch.stack([r[:, :, -1] for r in rec], dim=2))     # [3B, K, steps, card]


if __name__ == '__main__':
    make_two_step()
    make_two_step_prepend()
    make_double_cfg()
