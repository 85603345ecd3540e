"""dev tool / child process of bench.py's in-situ kernel statistics: ONE short generate of the bench workload's model
(MusicGen-medium bf16, 8 prompts, CFG, top-k 250) so that `rocprofv3 --kernel-trace --stats -- python scripts/short_generate.py`
yields the per-kernel durations INSIDE the real decode graph (every launch between its real neighbours).

    python scripts/short_generate.py [model] [batch] [duration_s]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from audiocraft_amd.models.musicgen import MusicGen

name = sys.argv[1] if len(sys.argv) > 1 else 'facebook/musicgen-medium'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
duration = float(sys.argv[3]) if len(sys.argv) > 3 else 8.0
model = MusicGen.get_random_init(name, 'cuda', torch.bfloat16, text_len=16, seed=0)
model.set_generation_params(use_sampling=True, top_k=250, duration=2.0)
model.generate([f"synthetic prompt {i}" for i in range(batch)])   # warm-up: clocks, allocator, first graph capture
model.set_generation_params(use_sampling=True, top_k=250, duration=duration)
wav = model.generate([f"synthetic prompt {i}" for i in range(batch)])
torch.cuda.synchronize()
print(f"generated {tuple(wav.shape)}", flush=True)
