"""dev tool: the self-attention kernel at the headline shape (16 rows x 24 heads x 64, bf16 cache) at a few context
lengths, cycling over distinct caches (cold, like in a real decode position); also the command the rocprofv3 --pmc
passes of profiles/archive/r02_attn_pmc_* run."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from audiocraft_amd import _C

M, H, hd, Tcap = 16, 24, 64, 1504
x = torch.randn(M, H * hd, device='cuda')
att = _C.tiled_activation_buffer(M, H * hd, torch.bfloat16, 'cuda')
ks = [torch.randn(M, H, Tcap, hd, device='cuda').bfloat16() for _ in range(12)]
vs = [torch.randn_like(k) for k in ks]
for ln in (256, 750, 1500):
    for rep in range(3):
        for k, v in zip(ks, vs):
            _C.attn_decode(x, k, v, att, ln, out_tiled=True)
torch.cuda.synchronize()
print('done: 36 launches per length (256, 750, 1500); algorithmic bytes per launch =',
      [2 * M * H * ln * hd * 2 for ln in (256, 750, 1500)])
