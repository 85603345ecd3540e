"""Self-attention decode kernel alone at the configs[2] geometry (16 CFG rows x 24 heads x 64, bf16 cache), per context
length: one hipGraph of 48 launches cycling through 48 layers' caches (cold K / V like in a real decode position; 7.2 GB
of cache), replayed; prints us per launch and the KV bandwidth.  Dev tool for same-box A/B of kernel variants
(ACMI_ATTN_DB=0|1, ACMI_LIB=<variant .so>).
    python scripts/attn_bench.py [--rows 16] [--heads 24] [--layers 48]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from audiocraft_amd import _C  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--rows', type=int, default=16)
ap.add_argument('--heads', type=int, default=24)
ap.add_argument('--layers', type=int, default=48)
ap.add_argument('--tcap', type=int, default=1504)
ap.add_argument('--contexts', default='1,32,64,128,256,512,750,1024,1500', help='comma-separated context lengths')
args = ap.parse_args()
B, H, hd, L, Tcap = args.rows, args.heads, 64, args.layers, args.tcap
k = torch.randn(L, B, H, Tcap, hd, device='cuda').bfloat16()
v = torch.randn(L, B, H, Tcap, hd, device='cuda').bfloat16()
q = torch.randn(B, H * hd, device='cuda')
out = _C.tiled_activation_buffer(B, H * hd, torch.bfloat16, 'cuda')
res = {}
for t in [int(c) for c in args.contexts.split(',')]:
    def launch_all():
        for li in range(L):
            _C.attn_decode(q, k[li], v[li], out, t, out_tiled=True)
    launch_all()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        g.capture_begin()
        try:
            launch_all()
        finally:
            g.capture_end()
        g.replay()
        side.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(5):
            g.replay()
        e1.record(side)
        side.synchronize()
    torch.cuda.current_stream().wait_stream(side)
    us = e0.elapsed_time(e1) * 1e3 / (5 * L)
    res[t] = round(us, 2)
    print(f"t={t:5d}: {us:7.2f} us / launch   KV {2 * B * H * t * hd * 2 / us / 1e6:7.2f} TB/s", flush=True)
print(json.dumps({'variant': os.environ.get('ACMI_LIB', 'default') + ' DB=' + os.environ.get('ACMI_ATTN_DB', '1'), 'rows': B,
                  'heads': H, 'us_per_launch': res}))
