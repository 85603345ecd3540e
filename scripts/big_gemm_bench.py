"""The prefill's MFMA-tiled GEMM alone (acmi_linear_big through the C-ABI), at the shapes of a MusicGen-medium prefill of
600 positions x 16 CFG rows (dev / documentation tool; numbers go to profiles/ and DESIGN.md).

    python scripts/big_gemm_bench.py [--rows 9600] [--reps 20]
ACMI_BIG_TILE=0 / 1 forces the 128 x 128 / 256 x 256 tile (A/B across two processes: the switch is read once).
--trace: with ACMI_LIB pointing at a build made with -DACMI_BIG_TRACE (audiocraft_amd.build.build(out=..., defines=[
'ACMI_BIG_TRACE'])), also prints where the waves of the last launch spent their cycles (acmi_prefill.hip: g_big_trace).
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from audiocraft_amd import _C  # noqa: E402

PEAK_BF16 = 2.5e15   # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=9600)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--trace', action='store_true')
    ap.add_argument('--zeros', action='store_true', help='all-zero operands: the same instruction stream at minimal switching power')
    args = ap.parse_args()
    C = _C
    M = args.rows
    d, F = 1536, 6144
    cases = [('qkv_like_f32', 3 * d, d, 'f32'), ('out_proj_resid', d, d, 'resid'), ('ffn1_tiled_gelu', F, d, 'tiled'),
             ('ffn2_resid', d, F, 'resid')]
    g = torch.Generator().manual_seed(0)
    for name, N, K, mode in cases:
        a = torch.randn(M, K, generator=g).cuda()
        w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
        bias = torch.randn(N, generator=g).cuda()
        if args.zeros:
            a.zero_(); w.zero_()
        at, wt = C.tile_matrix(a, torch.bfloat16), C.TiledWeight(w, torch.bfloat16)
        if mode == 'tiled':
            out = C.tiled_activation_buffer(M, N, torch.bfloat16, 'cuda')
            run = lambda: C.linear_big(at, wt, out, M, bias=bias, act=1)   # noqa: E731
        else:
            out = torch.zeros(M, N, device='cuda')
            run = lambda: C.linear_big(at, wt, out, M, bias=None if mode == 'resid' else bias, accumulate=mode == 'resid')   # noqa: E731
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            run()
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.reps
        fl = 2.0 * M * N * K
        print(json.dumps({'case': name, 'M': M, 'N': N, 'K': K, 'us': round(us, 1), 'tflops_per_s': round(fl / us / 1e6, 1),
                          'frac_of_bf16_peak': round(fl / (us * 1e-6) / PEAK_BF16, 3),
                          'tile': os.environ.get('ACMI_BIG_TILE', 'auto'), 'sched': os.environ.get('ACMI_BIG_SCHED', '0'), 'zeros': args.zeros}), flush=True)
        if args.trace:
            import ctypes
            import numpy as np
            tiles = min(2048, -(-M // 256) * -(-N // 256))
            buf = np.zeros(tiles * 8 * 8, dtype=np.uint64)
            torch.cuda.synchronize()
            rc = C.lib.acmi_big_trace_read(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(buf.size))
            assert rc == 0
            t = buf.reshape(tiles, 8, 8).astype(np.float64)
            t = t[t[:, :, 7].sum(axis=1) > 0]             # workgroups that ran the main loop
            names = ['prologue', 'dma_issue', 'read_mfma', 'vmcnt_wait', 'barrier_wait', 'drain', 'epilogue', 'steady']
            nst = K // 32 - 3
            out = {n: round(float(t[:, :, i].mean()), 0) for i, n in enumerate(names)}
            out.update({'per_stage_' + n: round(float(t[:, :, i].mean()) / nst, 1) for i, n in enumerate(names) if 1 <= i <= 4})
            out['per_stage_steady'] = round(float(t[:, :, 7].mean()) / nst, 1)
            out['workgroups'] = int(t.shape[0])
            # waves 0-3 request the activation, 4-7 the weight: their waits can differ
            for i in (1, 2, 3, 4):
                out[names[i] + '_by_wave'] = [round(float(t[:, w, i].mean()) / nst, 1) for w in range(8)]
            out['epilogue_by_wave'] = [round(float(t[:, w, 6].mean()), 0) for w in range(8)]
            out['steady_p10_p50_p90_over_workgroups'] = [round(float(v) / nst, 1) for v in np.percentile(t[:, 0, 7], [10, 50, 90])]
            print(json.dumps({'trace_cycles': out, 'case': name}), flush=True)


if __name__ == '__main__':
    main()
