"""dev tool: average GEMM launch time over one decode position's chain (env knobs such as ACMI_LIN_NW applied);
also the command the rocprofv3 --pmc passes of bench.py / profiles/ run.

    python scripts/dbg_chain.py [model = facebook/musicgen-medium] [CFG rows = 16]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from audiocraft_amd.models.musicgen import MusicGen

name = sys.argv[1] if len(sys.argv) > 1 else 'facebook/musicgen-medium'
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 16
model = MusicGen.get_random_init(name, 'cuda', torch.bfloat16)
model.lm._pack()
r = bench.measure_lin_kernel(model, rows)
print(f"{name}, {rows} rows, ACMI_LIN_NW={os.environ.get('ACMI_LIN_NW', '-')}: {r['avg_us']:.2f} us/launch, "
      f"{r['bytes_per_launch'] / 1e6:.2f} MB/launch, {r['launches_per_position']} launches", flush=True)
