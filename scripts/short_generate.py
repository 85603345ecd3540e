"""dev tool / child process of bench.py's in-situ kernel statistics: a 2 s warm-up generate, then ONE generate of the bench
workload's model, so that `rocprofv3 --kernel-trace --stats -- python scripts/short_generate.py ...` yields the per-kernel durations
INSIDE the real decode graph (every launch between its real neighbours).

    python scripts/short_generate.py [model] [batch] [duration_s] [greedy 0|1] [melody_seconds]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from audiocraft_amd.models.musicgen import MusicGen

name = sys.argv[1] if len(sys.argv) > 1 else 'facebook/musicgen-medium'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
duration = float(sys.argv[3]) if len(sys.argv) > 3 else 8.0
greedy = len(sys.argv) > 4 and sys.argv[4] == '1'
melody_seconds = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
model = MusicGen.get_random_init(name, 'cuda', torch.bfloat16, text_len=16, seed=0)
descs = [f"synthetic prompt {i}" for i in range(batch)]
mel = torch.randn(batch, 1, int(32000 * melody_seconds), generator=torch.Generator().manual_seed(5)) if melody_seconds > 0 else None


def once():
    if mel is not None:
        return model.generate_with_chroma(descs, mel, 32000)
    return model.generate(descs)


model.set_generation_params(use_sampling=not greedy, top_k=250, duration=2.0)
once()   # warm-up: clocks, allocator, first graph capture
model.set_generation_params(use_sampling=not greedy, top_k=250, duration=duration)
wav = once()
torch.cuda.synchronize()
print(f"generated {tuple(wav.shape)}", flush=True)
