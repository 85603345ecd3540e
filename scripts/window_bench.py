"""> 30 s generation (reference musicgen.py:290-337 / genmodel.py:233-262): 60 s of audio for 8 prompts with MusicGen-medium
= one 30 s window + two 18 s extensions, each extension prompted with the last 600 tokens of the previous window.  Prints
the wall time, the real-time factor and the share of the prefill (dev / documentation tool).
    python scripts/window_bench.py [--duration 60] [--batch 8]           ACMI_PREFILL=chunk for the round-2 path
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from audiocraft_amd.models.musicgen import MusicGen  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--duration', type=float, default=60.0)
ap.add_argument('--batch', type=int, default=8)
args = ap.parse_args()
model = MusicGen.get_random_init('facebook/musicgen-medium', 'cuda', torch.bfloat16)
model.set_generation_params(use_sampling=True, top_k=250, duration=args.duration)
lm = model.lm
pf = []
inner = lm._prefill


def timed_prefill(desc, state, n, *a, **k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = inner(desc, state, n, *a, **k)
    torch.cuda.synchronize()
    pf.append((n, time.perf_counter() - t0))
    return out


lm._prefill = timed_prefill
descs = [f'prompt {i}' for i in range(args.batch)]
model.generate(descs)            # warm-up (allocations, graph captures)
torch.cuda.synchronize()
pf.clear()
t0 = time.perf_counter()
wav = model.generate(descs)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps(dict(duration=args.duration, batch=args.batch, wall_s=round(dt, 3), rtf=round(args.batch * args.duration / dt, 2),
                      prefills=[(n, round(s * 1e3, 1)) for n, s in pf], prefill_total_ms=round(sum(s for _, s in pf) * 1e3, 1),
                      path=os.environ.get('ACMI_PREFILL', 'default'), wav=list(wav.shape))))
