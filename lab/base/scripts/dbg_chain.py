"""dev tool: average GEMM launch time over one decode position's chain (env knobs such as ACMI_LIN_NW applied);
also the command the rocprofv3 --pmc passes of profiles/ run."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from audiocraft_amd.models.musicgen import MusicGen

model = MusicGen.get_random_init('facebook/musicgen-medium', 'cuda', torch.bfloat16)
model.lm._pack()
r = bench.measure_lin_kernel(model, 16)
print(f"ACMI_LIN_NW={os.environ.get('ACMI_LIN_NW', '-')}: {r['avg_us']:.2f} us/launch, "
      f"{r['bytes_per_launch'] / 1e6:.2f} MB/launch, {r['launches_per_position']} launches", flush=True)
