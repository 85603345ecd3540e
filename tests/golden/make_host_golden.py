"""Golden fixtures for the HOST logic of the path, recorded from the unmodified reference:

1. tests/golden/windowing.json -- the windowed generation (durations above max_duration).

Runs the UNMODIFIED reference `MusicGen` (debug model, through oracle/refstubs.py) with `lm.generate` replaced by
a deterministic recorder, so that only `BaseGenModel._generate_tokens` / `MusicGen._generate_tokens` (reference
genmodel.py:193-260, musicgen.py:251-337) are exercised: which (prompt length, max_gen_len) every window asks
for, which melody excerpt it carries, and how the windows are stitched.

2. tests/golden/stereo.npz -- `InterleaveStereoCompressionModel` (reference encodec.py:397-506) over a deterministic
stand-in mono codec: interleaved codes and decoded audio, per-codebook and per-timestep variants.

Run in the build container only:   python tests/golden/make_host_golden.py
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
from oracle import refstubs  # noqa: E402,F401

from audiocraft.models.encodec import CompressionModel, InterleaveStereoCompressionModel  # noqa: E402
from audiocraft.models.musicgen import MusicGen  # noqa: E402


def fake_lm_generate(calls):
    def generate(prompt, attributes, callback=None, max_gen_len=256, **kw):
        B, K = len(attributes), 4
        T0 = 0 if prompt is None else prompt.shape[-1]
        mel = []
        for a in attributes:
            w = a.wav['self_wav']
            mel.append([int(w.length[0]), [round(float(v), 6) for v in w.wav.flatten()[:3]],
                        round(float(w.wav.double().sum()), 4)])
        calls.append({'prompt_len': T0, 'max_gen_len': int(max_gen_len), 'melody': mel})
        out = torch.empty(B, K, max_gen_len, dtype=torch.long)
        if prompt is not None:
            out[..., :T0] = prompt
        out[..., T0:] = (torch.arange(T0, max_gen_len) * 7 + len(calls) * 13) % 400
        return out
    return generate


def case(duration, max_duration, stride, prompt_len, melody_len):
    torch.manual_seed(0)
    mg = MusicGen.get_pretrained('debug', device='cpu')
    mg.max_duration = max_duration
    mg.set_generation_params(duration=duration, extend_stride=stride)
    calls = []
    mg.lm.generate = fake_lm_generate(calls)
    B = 2
    melody = None
    if melody_len:
        melody = [torch.arange(melody_len, dtype=torch.float32)[None] * (i + 1) * 1e-3 for i in range(B)]
    attributes, _ = mg._prepare_tokens_and_attributes(['a', 'b'], None)
    if melody is not None:  # what musicgen.py:225-236 attaches (the debug LM itself has no melody conditioner)
        from audiocraft.modules.conditioners import WavCondition
        for attr, m in zip(attributes, melody):
            attr.wav['self_wav'] = WavCondition(m[None], torch.tensor([m.shape[-1]]), sample_rate=[mg.sample_rate],
                                                path=[None])
    prompt = None
    if prompt_len:
        prompt = (torch.arange(prompt_len)[None, None] + torch.arange(4)[None, :, None]).repeat(B, 1, 1) % 400
    tokens = mg._generate_tokens(attributes, prompt)
    return {'duration': duration, 'max_duration': max_duration, 'extend_stride': stride, 'prompt_len': prompt_len,
            'melody_len': melody_len, 'frame_rate': mg.frame_rate, 'sample_rate': mg.sample_rate,
            'calls': calls, 'tokens_shape': list(tokens.shape), 'tokens_row': tokens[0, 0].tolist()}


class StandInMono(CompressionModel):
    """Deterministic mono 'codec' (the same arithmetic as the stand-in of tests/test_host_cpu.py)."""
    channels, frame_rate, sample_rate, cardinality, num_codebooks, total_codebooks = 1, 50, 32000, 2048, 4, 4

    def __init__(self):
        torch.nn.Module.__init__(self)

    def encode(self, x):
        T = x.shape[-1] // 4
        return (x[:, 0, :T * 4:4].abs() * 1000).long()[:, None].repeat(1, 4, 1) + torch.arange(4).view(1, 4, 1), None

    def decode(self, codes, scale=None):
        return codes.float().sum(1, keepdim=True).repeat_interleave(4, dim=-1)

    def set_num_codebooks(self, n): pass
    def forward(self, x): pass
    def decode_latent(self, c): pass


def stereo_golden():
    import numpy as np
    x = torch.randn(3, 2, 40, generator=torch.Generator().manual_seed(4))
    out = {'x': x.numpy()}
    for tag, per_timestep in (('k', False), ('t', True)):
        st = InterleaveStereoCompressionModel(StandInMono(), per_timestep=per_timestep)
        codes, scale = st.encode(x)
        assert scale is None
        out[f'codes_{tag}'] = codes.numpy()
        out[f'wav_{tag}'] = st.decode(codes).numpy()
        out[f'meta_{tag}'] = np.array([st.num_codebooks, st.frame_rate, st.channels, st.total_codebooks])
    path = os.path.join(HERE, 'stereo.npz')
    np.savez_compressed(path, **out)
    print(f'wrote {path}')


if __name__ == '__main__':
    stereo_golden()
    cases = [case(4., 3., 2., 0, 0), case(7.5, 3., 1., 0, 0), case(5., 3., 2., 25, 0), case(2., 3., 2., 10, 0),
             case(4., 3., 2., 0, 40000), case(6., 2., 1., 30, 100000)]
    path = os.path.join(HERE, 'windowing.json')
    json.dump(cases, open(path, 'w'))
    print(f'wrote {path}: {len(cases)} cases, {sum(len(c["calls"]) for c in cases)} windows')
