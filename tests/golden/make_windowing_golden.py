"""Golden trace of the reference's windowed generation (durations above max_duration), tests/golden/windowing.json.

Runs the UNMODIFIED reference `MusicGen` (debug model, through oracle/refstubs.py) with `lm.generate` replaced by
a deterministic recorder, so that only `BaseGenModel._generate_tokens` / `MusicGen._generate_tokens` (reference
genmodel.py:193-260, musicgen.py:251-337) are exercised: which (prompt length, max_gen_len) every window asks
for, which melody excerpt it carries, and how the windows are stitched.

Run in the build container only:   python tests/golden/make_windowing_golden.py
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
from oracle import refstubs  # noqa: E402,F401

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


if __name__ == '__main__':
    cases = [case(4., 3., 2., 0, 0), case(7.5, 3., 1., 0, 0), case(5., 3., 2., 25, 0), case(2., 3., 2., 10, 0),
             case(4., 3., 2., 0, 40000), case(6., 2., 1., 30, 100000)]
    path = os.path.join(HERE, 'windowing.json')
    json.dump(cases, open(path, 'w'))
    print(f'wrote {path}: {len(cases)} cases, {sum(len(c["calls"]) for c in cases)} windows')
