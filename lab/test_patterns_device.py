"""NOT part of the collected suite (lives under lab/): LM generation through the non-delay codebook patterns on the device,
against the reference golden `tests/golden/lm_patterns.npz` (already reproduced by the oracle on CPU:
tests/test_oracle_golden.py::test_lm_other_codebook_patterns_oracle_matches_reference).  Written after round 4's GPU budget was
spent; run it once on an MI355X and move it into tests/test_gpu_zz_options.py when green:

    python -m pytest lab/test_patterns_device.py -q
"""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
from conftest import load_golden  # noqa: E402
from oracle import lm as olm  # noqa: E402

pytestmark = pytest.mark.gpu


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs an MI355X")
def test_lm_other_codebook_patterns_vs_reference_golden():
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
