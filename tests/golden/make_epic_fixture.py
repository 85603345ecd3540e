"""Fixture generator (run in the build container, needs /root/reference): the reference's assets/epic.wav (32 kHz mono
float32, 3 s) as an .npz for the configs[0] parity test on real audio (SURVEY.md section 8c iii: the .mp3 assets cannot be
decoded offline; the .wav assets are read with scipy.io.wavfile and treated as raw samples at the model rate).

    python tests/golden/make_epic_fixture.py
"""
import os

import numpy as np
from scipy.io import wavfile

SRC = '/root/reference/assets/epic.wav'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'epic_wav.npz')

if __name__ == '__main__':
    sr, x = wavfile.read(SRC)
    assert sr == 32000 and x.dtype == np.float32 and x.shape == (96000,), (sr, x.dtype, x.shape)
    np.savez_compressed(OUT, wav=x, sample_rate=np.int32(sr))
    print(f"wrote {OUT}: {x.shape[0]} samples, |x|_max {np.abs(x).max():.4f}, {os.path.getsize(OUT)} bytes")
