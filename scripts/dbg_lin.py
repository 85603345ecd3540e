"""dev tool: ablation of the tiled GEMM kernel (ACMI_DBG: 0 full, 4 no A loads, 5 no reduce/epilogue, 6 no W loads,
7 no loads at all) on the QKV / FFN2 / out-proj shapes of MusicGen-medium, cold weights (48 layers cycled)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiocraft_amd import _C
from scripts.microbench import graph_time
M, d = 16, 1536
torch.manual_seed(0)
dt = torch.bfloat16
for name, N, K in (('qkv', 3 * d, d), ('out', d, d), ('ffn2', d, 4 * d)):
    ws = [_C.TiledWeight(torch.randn(N, K, device='cuda'), dt) for _ in range(48)]
    a = _C.tile_matrix(torch.randn(M, K, device='cuda'), dt)
    o = torch.empty(M, N, device='cuda')
    us = graph_time([(lambda w=w: _C.linear(a, w, o, a_tiled=True, M=M)) for w in ws])
    print(f"ACMI_DBG={os.environ.get('ACMI_DBG','0')} {name:5s} {us:6.2f} us/launch", flush=True)
    del ws
