"""In-kernel timeline of the decode step's GEMM launches (dev tool; writes profiles/archive/r04_lin_timeline*.csv).

Needs the trace build of the library (every wave of lin_tiled_kernel / lin_pair_kernel stamps s_memrealtime at nine
phases, acmi_lm_internal.h):

    python -c "from audiocraft_amd import build as b; b.build(out=b.OUT.replace('libacmi.so', 'libacmi_trace.so'), defines=['ACMI_TRACE'])"
    ACMI_LIB=audiocraft_amd/csrc/libacmi_trace.so python scripts/lin_timeline.py --out profiles/archive/r04_lin_timeline.csv

One generate of the headline workload (MusicGen-medium, bf16, 8 prompts, CFG rows 16, top-k 250) is run for --frames
positions; the hipGraph of one position is captured right after the trace buffer is armed, so every GEMM launch of the graph
owns a region of the buffer and each replay overwrites it: what is read back is the LAST position's timeline, at context
--frames.  Per launch kind (QKV + cross-q part, paired out-proj, cross-out, FFN1, FFN2 half-tile, heads) the script reports,
in microseconds relative to the launch's first wave entry, the median over waves and the tail (slowest wave) of every
phase, the slowest workgroup's own phases, and the gap between the last retired store of a launch and the first wave of
the next GEMM launch (kernel boundary + whatever runs in between: the attention kernels are not instrumented)."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from audiocraft_amd import _C  # noqa: E402
from audiocraft_amd.models.musicgen import MusicGen  # noqa: E402

NST = 10
PHASES = ['entry', 'issued', 'first_w', 'all_w', 'all_loads', 'mfma_lds', 'barrier', 'stores_issued', 'stores_retired']


def kind_name(kind, N, K, d):
    if kind & 8:
        return 'pair_outproj_crossq'
    if kind & 2:
        return 'qkv_xq'
    if kind & 4:
        return 'ffn2_half'
    if kind & 1:
        return 'ffn1' if N == 4 * d else ('heads' if N != d else 'ln_lin')
    if kind & 16:
        return 'cross_out' if K == d else 'ffn2'
    return 'plain'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='facebook/musicgen-medium')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--frames', type=int, default=300)
    ap.add_argument('--clock-mhz', type=float, default=100.0, help='s_memrealtime rate (hipDeviceAttributeWallClockRate)')
    ap.add_argument('--out', default='profiles/archive/r04_lin_timeline.csv')
    ap.add_argument('--raw', default='', help='also dump the raw stamps (npz)')
    args = ap.parse_args()
    if not hasattr(_C.lib, 'acmi_trace_config'):
        raise SystemExit('this library has no trace support: build with defines=[ACMI_TRACE] and point ACMI_LIB at it')
    _C.lib.acmi_trace_config.argtypes = [C.c_void_p, C.c_longlong]
    _C.lib.acmi_trace_info.argtypes = [C.c_int, C.POINTER(C.c_int)]

    model = MusicGen.get_random_init(args.model, 'cuda', torch.bfloat16)
    lm = model.lm
    d = lm.dim
    M = 2 * args.batch
    cap_words = 64 << 20 >> 3
    buf = torch.zeros(cap_words, dtype=torch.int64, device='cuda')
    orig_capture = lm._capture
    g = torch.Generator().manual_seed(0)
    cross = torch.randn(M, 16, d, generator=g).cuda()
    cross[args.batch:] = 0
    ct = {'description': (cross, torch.ones(M, 16, dtype=torch.int64).cuda())}
    recs = []

    def capture2(desc, state):
        # arm -> capture one position (its GEMM launches take regions 0, 1, ... in launch order) -> describe -> disarm
        _C.lib.acmi_trace_config(buf.data_ptr(), cap_words)
        gph = orig_capture(desc, state)
        for i in range(_C.lib.acmi_trace_count()):
            out = (C.c_int * 8)()
            _C.lib.acmi_trace_info(i, out)
            recs.append(tuple(out))
        _C.lib.acmi_trace_config(None, 0)
        return gph

    lm._capture = capture2
    lm.generate(None, [], num_samples=args.batch, max_gen_len=args.frames, condition_tensors=ct, top_k=250, use_graph=True)
    torch.cuda.synchronize()
    host = buf.cpu().numpy().view(np.uint64)
    tick_us = 1.0 / args.clock_mhz
    launches = []
    for (lo, hi, kind, wgs, waves, N, K, Mr) in recs:
        off = (hi << 32) | (lo & 0xffffffff)
        a = host[off:off + wgs * waves * NST].reshape(wgs, waves, NST)
        launches.append(dict(kind=kind_name(kind, N, K, d), feat=kind >> 8, wgs=wgs, waves=waves, N=N, K=K, st=a))
    if args.raw:
        np.savez_compressed(args.raw, **{f'l{i:03d}_{L["kind"]}': L['st'] for i, L in enumerate(launches)})

    rows = {}
    for i, L in enumerate(launches):
        st = L['st'][..., :9].astype(np.int64)
        live = st[..., 0] > 0
        if not live.any():
            continue
        t0 = st[..., 0][live].min()
        rel = (st - t0) * tick_us                                  # [wg, wave, 9] us since the first wave entry
        rel = np.where(live[..., None], rel, np.nan)
        end = np.nanmax(rel[..., 8])
        gap = np.nan
        if i + 1 < len(launches):
            nx = launches[i + 1]['st'][..., 0].astype(np.int64)
            nx = nx[nx > 0]
            if nx.size:
                gap = (nx.min() - t0) * tick_us - end
        wg_end = np.nanmax(rel[..., 8], axis=1)                    # per workgroup: its last wave's retired stores
        slow = int(np.nanargmax(wg_end))
        xcc = (L['st'][..., 9] >> np.uint64(32)).astype(np.int64) & 0xf
        e = rows.setdefault(L['kind'], dict(n=0, wgs=L['wgs'], waves=L['waves'], N=L['N'], K=L['K'], feat=L['feat'], med=[], tail=[],
                                            slow=[], span=[], gap=[], entry_spread=[], slow_xcc=[]))
        e['n'] += 1
        e['med'].append(np.nanmedian(rel.reshape(-1, 9), axis=0))
        e['tail'].append(np.nanmax(rel.reshape(-1, 9), axis=0))
        e['slow'].append(np.nanmax(rel[slow], axis=0))             # the slowest workgroup: its last wave per phase
        e['span'].append(end)
        e['gap'].append(gap)
        e['entry_spread'].append(np.nanmax(rel[..., 0]))
        e['slow_xcc'].append(int(xcc[slow, 0]))

    os.makedirs(os.path.dirname(args.out) or '.', exist_ok=True)
    with open(args.out, 'w') as f:
        f.write('# in-kernel timeline of the GEMM launches of ONE decode position (the last of a %d-frame generate), %s, B=%d; '
                'us since the first wave entry of the launch; median over the launches of a kind (one per layer)\n'
                % (args.frames, args.model, args.batch))
        f.write('kind,launches,workgroups,waves,features_per_wg,N,K,row,' + ','.join(PHASES) + ',span_us,gap_to_next_gemm_us\n')
        for k, e in rows.items():
            for name, arr in (('median_wave', e['med']), ('slowest_wave', e['tail']), ('slowest_workgroup', e['slow'])):
                v = np.nanmedian(np.stack(arr), axis=0)
                f.write(f"{k},{e['n']},{e['wgs']},{e['waves']},{e['feat']},{e['N']},{e['K']},{name}," +
                        ','.join(f'{x:.2f}' for x in v) +
                        f",{np.nanmedian(e['span']):.2f},{np.nanmedian(e['gap']):.2f}\n")
    print(open(args.out).read())
    for k, e in rows.items():
        print(f"{k:22s} launches {e['n']:3d}  span {np.nanmedian(e['span']):.2f} us  last wave enters at {np.nanmedian(e['entry_spread']):.2f} us  "
              f"gap to next GEMM {np.nanmedian(e['gap']):.2f} us  slowest workgroup's XCD histogram {np.bincount(e['slow_xcc'], minlength=8).tolist()}")


if __name__ == '__main__':
    main()
