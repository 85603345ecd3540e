"""dev tool / child process of bench.py's PMC passes for the fused QKV + self-attention launch: the LM of the named model with
FEWER layers (the per-launch traffic does not depend on the depth), `batch` prompts, a 2 s warm-up generate and one `duration` s
generate of TOKENS only (no codec), so that `rocprofv3 --pmc FETCH_SIZE -- python scripts/fused_chain.py ...` yields the HBM
traffic of the qkv_attn_kernel dispatches.

    python scripts/fused_chain.py [model] [batch] [duration_s] [layers]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from audiocraft_amd import _C
from audiocraft_amd.models import builders

name = sys.argv[1] if len(sys.argv) > 1 else 'facebook/musicgen-medium'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
duration = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
layers = int(sys.argv[4]) if len(sys.argv) > 4 else 4
torch.manual_seed(0)
cfg = dict(builders.musicgen_lm_cfg(name.split('-')[-1], text_len=16), num_layers=layers)
lm = builders.get_lm_model(cfg, 'cuda', torch.bfloat16)
d = cfg['dim']
src = torch.randn(2 * batch, 16, d, generator=torch.Generator().manual_seed(1))
src[batch:] = 0
ct = {'description': (src.cuda(), torch.ones(2 * batch, 16, dtype=torch.int64).cuda())}
for dur in (2.0, duration):
    lm.generate(None, [], num_samples=batch, max_gen_len=int(dur * 50), use_sampling=True, top_k=250, seed=5, condition_tensors=ct)
torch.cuda.synchronize()
print(f"{name} x {layers} layers, {batch} prompts: {_C.qkv_attn_launches()} fused launches", flush=True)
