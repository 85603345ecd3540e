"""Golden fixture for the output stage's level control, recorded from the unmodified reference:

tests/golden/audio_norm.npz -- `audiocraft.data.audio_utils.normalize_audio` (reference audio_utils.py:104-152) on seeded
clips for every strategy that needs no third-party package ('peak', 'clip', 'rms', 'none'; 'loudness' calls torchaudio,
which is absent here -- its meter is pinned separately, tests/test_host_cpu.py), with normalize = True and False, on a
quiet, a hot (would clip) and a two-channel clip, plus `i16_pcm` / `f32_pcm`.

Run in the build container only:   python tests/golden/make_audio_norm_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
from oracle import refstubs  # noqa: E402

refstubs.install()
from audiocraft.data.audio_utils import f32_pcm, i16_pcm, normalize_audio  # noqa: E402


def main():
    g = torch.Generator().manual_seed(20)
    clips = {'quiet': 0.02 * torch.randn(1, 600, generator=g),
             'hot': 0.9 * torch.randn(1, 600, generator=g),
             'stereo': torch.stack([0.3 * torch.randn(500, generator=g), 0.05 * torch.randn(500, generator=g)])}
    out = {}
    for name, wav in clips.items():
        out[f'in_{name}'] = wav.numpy()
        for strategy in ('peak', 'clip', 'rms'):
            for normalize in (True, False):
                for head in ((1.0, 18.0), (3.0, 9.0)):
                    y = normalize_audio(wav.clone(), normalize=normalize, strategy=strategy, peak_clip_headroom_db=head[0],
                                        rms_headroom_db=head[1])
                    out[f'{name}|{strategy}|{int(normalize)}|{head[0]}|{head[1]}'] = y.numpy()
    small = 0.5 * torch.tanh(clips['hot'])
    out['in_none'] = small.numpy()
    out['none'] = normalize_audio(small.clone(), strategy='none').numpy()
    pcm_in = torch.tensor([[0.0, 0.25, -1.0, 0.99997, 1.0, -0.5]])
    out['pcm_in'] = pcm_in.numpy()
    out['pcm_i16'] = i16_pcm(pcm_in).numpy()
    out['pcm_i16_noplus'] = i16_pcm(pcm_in[:, :4]).numpy()
    out['pcm_f32_from_i16'] = f32_pcm(torch.tensor([[0, 16384, -32768, 32767]], dtype=torch.int16)).numpy()
    np.savez_compressed(os.path.join(HERE, 'audio_norm.npz'), **out)
    print(f"wrote audio_norm.npz: {len(out)} arrays")


if __name__ == '__main__':
    main()
