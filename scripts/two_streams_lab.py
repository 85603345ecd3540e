"""dev tool (lab): does a SECOND independent dependency chain in the same process fill the HBM-idle launch edges of the decode
step?  Two MusicGen-medium replicas, each generating B / 2 prompts on its own HIP stream from its own host thread (independent
hipGraph replays, no cross-stream dependency anywhere), against one replica generating B prompts.

    python scripts/two_streams_lab.py [batch=8] [duration_s=10]
"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from audiocraft_amd.models.musicgen import MusicGen

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
D = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
models = [MusicGen.get_random_init('facebook/musicgen-medium', 'cuda', torch.bfloat16, text_len=16, seed=0) for _ in range(2)]
for m in models:
    m.set_generation_params(use_sampling=True, top_k=250, duration=D)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
# the product captures in the default (global) mode, in which one thread's capture makes the other thread's synchronisations
# illegal; the lab switches to thread-local capture (not robust: a two-thread test of it crashed once in ten runs)
_begin = torch.cuda.CUDAGraph.capture_begin
torch.cuda.CUDAGraph.capture_begin = lambda self, *a, **kw: _begin(self, *a, **dict(kw, capture_error_mode='thread_local'))


def gen(i, n):
    with torch.cuda.stream(streams[i]):
        wav = models[i].generate([f"synthetic prompt {j}" for j in range(n)])
        streams[i].synchronize()
    return wav


def timed(fn, reps=2):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        best = t if best is None else min(best, t)
    return best


def both():
    th = [threading.Thread(target=gen, args=(i, B // 2)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()


gen(0, B); gen(0, B // 2); gen(1, B // 2)     # warm-up: graph capture for both geometries
t_one = timed(lambda: gen(0, B))
t_half = timed(lambda: gen(0, B // 2))
t_seq = timed(lambda: (gen(0, B // 2), gen(1, B // 2)))
t_two = timed(both)
print(f"one chain, {B} prompts x {D:.0f} s: {t_one:.3f} s (RTF {B * D / t_one:.1f})")
print(f"one chain, {B // 2} prompts: {t_half:.3f} s; the two halves one after the other: {t_seq:.3f} s")
print(f"two chains of {B // 2} prompts on two streams (two host threads): {t_two:.3f} s (RTF {B * D / t_two:.1f}) "
      f"= {t_two / t_one:.2f} x the single chain", flush=True)
