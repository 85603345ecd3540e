"""Calibration of bench.py's `cpu_baseline` (container only: needs /root/reference).

On the GPU box the reference tree does not exist, so the bench line's CPU leg times the ORACLE (`kind: "port"`).  This script
times, on ONE host with ONE thread count and the SAME weights / inputs, the unmodified reference (oracle/ref_baseline.py:
`LMModel._sample_next_token` inside `lm.streaming()`, its torch.cat KV cache included) and the oracle port (oracle/lm.py) on
the same positions: `--early` positions at the start of the stream and `--late` positions at context `--context`, at the
headline batch (8 samples = 16 CFG rows, MusicGen-medium geometry, top-k 250).  The ratio port / reference is what makes
"port" a stated stand-in for the reference's speed (profiles/r05_cpu_calibration.json; quoted in the bench line's
`cpu_baseline.sample`).

    python scripts/cpu_calibration.py [--threads 8] [--early 16] [--late 14] [--context 1400] [--out profiles/r05_cpu_calibration.json]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import lm as olm
from oracle import ref_baseline

GEOM = {'small': (1024, 16, 24), 'medium': (1536, 24, 48), 'large': (2048, 32, 48)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='medium')
    ap.add_argument('--threads', type=int, default=os.cpu_count() or 8)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--early', type=int, default=16)
    ap.add_argument('--late', type=int, default=14)
    ap.add_argument('--context', type=int, default=1400)
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    assert ref_baseline.available(), "needs the reference tree (/root/reference): run in the build container"
    torch.set_num_threads(args.threads)
    d, H, L = GEOM[args.model]
    B, K, card, Lc, top_k = args.batch, 4, 2048, 16, 250
    torch.manual_seed(0)
    t0 = time.perf_counter()
    ref = ref_baseline.build_reference_lm(None, d, H, L, K, card, True)
    sd = {k: v.detach().float() for k, v in ref.state_dict().items()}
    print(f"reference LMModel ({args.model}: d={d}, {L} layers) built in {time.perf_counter() - t0:.0f} s; {args.threads} threads", flush=True)
    g = torch.Generator().manual_seed(0)
    cross = torch.randn(2 * B, Lc, d, generator=g)
    cross[B:] = 0
    # ---- the unmodified reference
    r_early, r_late = ref_baseline.time_reference_positions(ref, B, cross, top_k, args.early, args.late, args.context)
    print(f"reference: {r_early * 1e3:.0f} ms / position early, {r_late * 1e3:.0f} ms at context {args.context}", flush=True)
    del ref
    # ---- the oracle port, exactly as bench.py's cpu_baseline times it
    oc = olm.LMConfig(dim=d, num_heads=H, num_layers=L, n_q=K, card=card, cross_attention=True)
    with torch.no_grad():
        olm.generate(sd, oc, None, B, cross, max_gen_len=1500, top_k=top_k, max_steps=1)
        t0 = time.perf_counter()
        olm.generate(sd, oc, None, B, cross, max_gen_len=1500, top_k=top_k, max_steps=args.early)
        p_early = (time.perf_counter() - t0) / args.early
        st = olm.LMState(L)
        hd = d // H
        for li in range(L):
            st.past_k[li] = torch.randn(2 * B, H, args.context, hd, generator=g)
            st.past_v[li] = torch.randn(2 * B, H, args.context, hd, generator=g)
        st.offset, st.first_step = args.context, False
        tok = torch.randint(0, card, (2 * B, K, 1), generator=g)
        olm.lm_forward(sd, oc, tok, cross, None, st)
        t0 = time.perf_counter()
        for _ in range(args.late):
            logits = olm.lm_forward(sd, oc, tok, cross, None, st)
            olm.sample_next_token(olm.cfg_mix(logits, 3.0)[:, :, -1], True, 1.0, top_k, 0.0)
        p_late = (time.perf_counter() - t0) / args.late
    print(f"port     : {p_early * 1e3:.0f} ms / position early, {p_late * 1e3:.0f} ms at context {args.context}", flush=True)

    def integrate(te, tl):   # bench.py's protocol: linear in the context between the two samples, 1503 positions
        ce, cl = (args.early - 1) / 2.0 + 1, args.context + 1 + (args.late - 1) / 2.0
        slope = (tl - te) / (cl - ce)
        return sum(te + slope * (t - ce) for t in range(1503))
    res = {'model': args.model, 'batch': B, 'cfg_rows': 2 * B, 'threads': args.threads, 'logical_cores': os.cpu_count(),
           'early_positions': args.early, 'late_positions': args.late, 'late_context': args.context,
           'reference_ms_per_position': {'early': r_early * 1e3, 'late': r_late * 1e3},
           'port_ms_per_position': {'early': p_early * 1e3, 'late': p_late * 1e3},
           'lm_seconds_for_1503_positions': {'reference': integrate(r_early, r_late), 'port': integrate(p_early, p_late)},
           'host': os.uname().nodename}
    res['port_over_reference'] = res['lm_seconds_for_1503_positions']['port'] / res['lm_seconds_for_1503_positions']['reference']
    print(json.dumps(res, indent=1), flush=True)
    if args.out:
        with open(args.out, 'w') as f:
            json.dump(res, f, indent=1)


if __name__ == '__main__':
    main()
