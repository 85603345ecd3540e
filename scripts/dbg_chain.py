"""dev tool: average lin_kernel launch time over one decode position's chain (env knobs applied)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from audiocraft_amd import _C
from audiocraft_amd.models.musicgen import MusicGen
model = MusicGen.get_random_init('facebook/musicgen-medium', 'cuda', torch.bfloat16)
model.lm._pack()
orig = _C.linear
def nopf(*a, **k):
    k.pop('prefetch', None)
    return orig(*a, **k)
_C.linear = nopf
r = bench.measure_lin_kernel(model, 16)
print(f"ACMI_LIN_NW={os.environ.get('ACMI_LIN_NW', '-')}: {r['avg_us']:.2f} us/launch", flush=True)
