"""The prefill's causal attention kernel alone (acmi_attn_prefill through the C-ABI) at the geometry of a MusicGen-medium
window prefill: 16 CFG rows x 24 heads x 64, 600 positions, bf16 caches (dev / documentation tool).

    python scripts/attn_prefill_bench.py [--rows 16] [--heads 24] [--npos 600] [--reps 20]
ACMI_PFA_QB=1|2 (query blocks per wave): same-box A/B, one process each.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from audiocraft_amd import _C  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=16)
    ap.add_argument('--heads', type=int, default=24)
    ap.add_argument('--npos', type=int, default=600)
    ap.add_argument('--reps', type=int, default=20)
    args = ap.parse_args()
    B, H, hd, npos = args.rows, args.heads, 64, args.npos
    npp, tcap = -(-npos // 16) * 16, -(-npos // 32) * 32
    g = torch.Generator().manual_seed(0)
    k = torch.randn(B, H, 2048, hd, generator=g).cuda().bfloat16()
    vt = torch.randn(B, H, hd, tcap, generator=g).cuda().bfloat16()
    q = torch.randn(B * npp, H * hd, generator=g).cuda()
    out = _C.tiled_activation_buffer(B * npp, H * hd, torch.bfloat16, 'cuda')
    pos = torch.zeros(1, dtype=torch.int32, device='cuda')
    run = lambda: _C.attn_prefill(q, k, vt, out, npos, npp, pos)   # noqa: E731
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        run()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / args.reps
    # fragment bytes a wave fetches: per 32 keys 4 KB of K + 4 KB of V^T, shared by QB query blocks
    qb = 2 if os.environ.get('ACMI_PFA_QB', '2') == '2' and npos > 64 else 1
    waves = B * H * -(-npos // (16 * qb))
    blocks = sum(-(-min(npos, (w + 1) * 16 * qb) // 32) for w in range(-(-npos // (16 * qb)))) * B * H
    print(json.dumps({'rows': B, 'heads': H, 'hd': hd, 'npos': npos, 'us': round(us, 1), 'waves': waves,
                      'fragment_GB': round(blocks * 8192 / 1e9, 3), 'fragment_TBps': round(blocks * 8192 / us / 1e6, 2),
                      'qb': os.environ.get('ACMI_PFA_QB', '2')}), flush=True)


if __name__ == '__main__':
    main()
