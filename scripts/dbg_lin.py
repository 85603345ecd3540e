import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiocraft_amd import _C
from scripts.microbench import graph_time
M, d = 16, 1536
torch.manual_seed(0)
ws = [_C.TiledWeight(torch.randn(3 * d, d, device='cuda'), torch.bfloat16) for _ in range(48)]
x = torch.randn(M, d, device='cuda'); o3 = torch.empty(M, 3 * d, device='cuda')
bias = torch.zeros(3 * d, device='cuda')
us = graph_time([(lambda w=w: _C.linear(x, w, o3, bias=bias, standardize=True)) for w in ws])
print(f"ACMI_DBG={os.environ.get('ACMI_DBG','0')}: ln_qkv {us:.2f} us/launch", flush=True)
att = _C.tile_matrix(torch.randn(M, d, device='cuda'), torch.bfloat16)
us = graph_time([(lambda w=w: _C.linear(att, w, o3, a_tiled=True, M=M)) for w in ws])
print(f"   tiled-A same shape {us:.2f} us/launch", flush=True)
