"""dev tool: the decode layer's  cross-out -> FFN1 -> FFN2  sub-chain as three launches (what acmi_lm_step runs) against ONE
persistent launch (acmi_ffn_engine), on the real shapes with every layer's own (cold) weights.

    python scripts/engine_lab.py [--model medium|small|large] [--layers 48] [--rows 16] [--reps 200] [--modes 0,1,2] [--waves 4,8]
                                 [--trace out.csv]

For every variant: (1) parity, layer by layer from identical inputs, against the three-launch chain and (layer 0) against an
f64 restatement from the logical weights; (2) determinism / staleness: every replay of the chained graph must reproduce the
first one bit for bit; (3) us per layer from hipGraph replays; (4) optionally the in-kernel timeline (s_memrealtime stamps).
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from audiocraft_amd import _C

GEOM = {'small': (1024, 4096), 'medium': (1536, 6144), 'large': (2048, 8192)}


def bf(t):
    return t.to(torch.bfloat16)


class Layer:
    def __init__(self, d, F, g, dev):
        s = 0.02
        self.w0 = bf(torch.randn(d, d, generator=g) * s).to(dev)
        w1 = torch.randn(F, d, generator=g) * s
        gamma = 1.0 + 0.1 * torch.randn(d, generator=g)
        beta = 0.1 * torch.randn(d, generator=g)
        self.w1f = bf(w1 * gamma[None, :]).to(dev)              # W1 diag(gamma), rounded
        self.b1 = (w1 @ beta + 0.02 * torch.randn(F, generator=g)).to(dev)
        self.cs1 = self.w1f.double().sum(dim=1).float()
        self.w2 = bf(torch.randn(d, F, generator=g) * s).to(dev)
        self.b0 = (0.02 * torch.randn(d, generator=g)).to(dev)
        self.b2 = (0.02 * torch.randn(d, generator=g)).to(dev)
        self.t_w0 = _C.TiledWeight(self.w0, torch.bfloat16)              # 16-feature form (the launch chain)
        self.t_w0h = _C.TiledWeight(self.w0, torch.bfloat16, half=True)  # half-tile order (the engine)
        self.t_w1 = _C.TiledWeight(self.w1f, torch.bfloat16)
        self.t_w2h = _C.TiledWeight(self.w2, torch.bfloat16, half=True)
        self.att = _C.tile_matrix(bf(torch.randn(16, d, generator=g)).to(dev), torch.bfloat16)   # cross-attention output


class Bufs:
    def __init__(self, M, d, F, dev):
        self.x = torch.zeros(M, d, device=dev)
        self.xt = [_C.tiled_activation_buffer(16, d, torch.bfloat16, dev) for _ in range(2)]
        self.hidden = _C.tiled_activation_buffer(16, F, torch.bfloat16, dev)
        self.shift = torch.zeros(16, device=dev)


def launch_chain(L, b, M, d, F, eps):
    """the three launches of acmi_lm_step for this sub-chain (gemm_produce_x / gemm_ln_x in LN mode 4)"""
    d0 = _C.linear_desc(L.att, L.t_w0, b.x, M, _C.A_TILED, _C.OUT_F32, bias=L.b0, residual=b.x, xt_hi=b.xt[0], xt_shift=b.shift)
    _C.linear_launch(d0)
    d1 = _C.linear_desc(b.xt[0], L.t_w1, b.hidden, M, _C.A_TILED, _C.OUT_TILED, bias=L.b1, act=1, eps=eps, colsum=L.cs1,
                        a_shift=b.shift)
    _C.linear_launch(d1)
    d2 = _C.linear_desc(b.hidden, L.t_w2h, b.x, M, _C.A_TILED, _C.OUT_F32, bias=L.b2, residual=b.x, xt_hi=b.xt[1], xt_shift=b.shift)
    _C.linear_launch(d2)
    return (d0, d1, d2)


def engine_desc(L, b, M, d, F, eps, flags, flags_next, err, acq, waves, trace=None, chunk=0, epi=0, sleep=0):
    e = _C.FfnEngineDesc()
    e.w0, e.w1, e.w2 = L.t_w0h.data_ptr(), L.t_w1.data_ptr(), L.t_w2h.data_ptr()
    e.b0, e.b1, e.cs1, e.b2 = _C.ptr(L.b0), _C.ptr(L.b1), _C.ptr(L.cs1), _C.ptr(L.b2)
    e.a0, e.x = _C.ptr(L.att), _C.ptr(b.x)
    e.xt_mid, e.xt_out, e.xt_rbs, e.hidden = _C.ptr(b.xt[0]), _C.ptr(b.xt[1]), 0, _C.ptr(b.hidden)
    e.shift, e.flags, e.flags_next, e.err = _C.ptr(b.shift), flags.data_ptr(), flags_next.data_ptr(), _C.ptr(err)
    e.M, e.d, e.ffn, e.eps, e.acq_mode, e.waves = M, d, F, eps, acq, waves
    e.dma_chunk, e.dma_epi, e.poll_sleep = chunk, epi, sleep
    e.trace = None if trace is None else trace.data_ptr()
    return e


def reference_layer0(L, x1, shift, M, d, F, eps):
    """f64 restatement from the logical (rounded) weights, with the kernels' bf16 roundings of the activations"""
    att = _C.untile_matrix(L.att, 16, d)[:M].double().cpu()
    x2 = x1.double().cpu() + att @ L.w0.double().cpu().T + L.b0.double().cpu()
    frag = (x2 - shift[:M, None].double().cpu()).float().to(torch.bfloat16).double()     # bf16(x2 - shift)
    mean_s = frag.mean(dim=1, keepdim=True)
    var = (frag * frag).mean(dim=1, keepdim=True) - mean_s * mean_s
    h = (frag - mean_s) / torch.sqrt(var.clamp_min(0) + eps) @ L.w1f.double().cpu().T + L.b1.double().cpu()
    h = torch.nn.functional.gelu(h).float().to(torch.bfloat16).double()
    x3 = x2 + h @ L.w2.double().cpu().T + L.b2.double().cpu()
    return x2, h, x3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='medium')
    ap.add_argument('--layers', type=int, default=48)
    ap.add_argument('--rows', type=int, default=16)
    ap.add_argument('--reps', type=int, default=200)
    ap.add_argument('--modes', default='0,1,2')
    ap.add_argument('--waves', default='4,8')
    ap.add_argument('--trace', default=None)
    ap.add_argument('--chunks', default='4')
    ap.add_argument('--epi', default='0')
    ap.add_argument('--sleep', default='0')
    ap.add_argument('--check-reps', type=int, default=50)
    args = ap.parse_args()
    dev = torch.device('cuda')
    d, F = GEOM[args.model]
    M, NL, eps = args.rows, args.layers, 1e-5
    nwg = d // 8
    g = torch.Generator().manual_seed(1234)
    layers = [Layer(d, F, g, dev) for _ in range(NL)]
    x_init = (torch.randn(M, d, generator=g) * 1.5 + 0.7 * torch.randn(M, 1, generator=g)).to(dev)
    shift = torch.zeros(16)
    shift[:M] = x_init.mean(dim=1).cpu()
    wbytes = (d * d + 2 * d * F) * 2
    print(f"model {args.model}: d={d} ffn={F} rows={M} layers={NL}; {wbytes / 1e6:.1f} MB of weights per layer-sub-chain; "
          f"acmi {_C.version()}", flush=True)

    # ------------------------------------------------------------ the launch chain: graph over all layers, timing
    bc = Bufs(M, d, F, dev)
    bc.shift.copy_(shift)
    keep = []

    def run_chain():
        bc.x.copy_(x_init)
        for L in layers:
            keep.append(launch_chain(L, bc, M, d, F, eps))

    def graph_of(fn):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn()          # warm-up (lazy attribute calls etc.)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            fn()
        return gr

    def time_graph(gr, reps):
        gr.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            gr.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    gc = graph_of(run_chain)
    gc.replay()
    torch.cuda.synchronize()
    x_chain_final = bc.x.clone()
    t_chain = time_graph(gc, args.reps)
    print(f"chain  : {t_chain / NL * 1e6:7.2f} us per layer (3 launches), {wbytes / (t_chain / NL) / 1e12:.2f} TB/s", flush=True)

    # per-layer reference outputs of the chain from identical inputs (layer-by-layer parity)
    per_layer = []
    xin = x_init.clone()
    for L in layers[:min(NL, 8)]:
        bc.x.copy_(xin)
        launch_chain(L, bc, M, d, F, eps)
        torch.cuda.synchronize()
        per_layer.append((xin.clone(), bc.x.clone(), bc.hidden.clone(), bc.xt[0].clone(), bc.xt[1].clone()))
        xin = bc.x.clone()
    x2r, hr, x3r = reference_layer0(layers[0], x_init, shift, M, d, F, eps)
    e_chain = (per_layer[0][1].double().cpu() - x3r).abs().max().item()
    print(f"chain  : layer 0 vs f64 restatement: x3 max abs err {e_chain:.3e} (|x3| max {x3r.abs().max():.2f})", flush=True)

    # ------------------------------------------------------------ the engine
    results = {}
    import itertools
    ints = lambda v: [int(q) for q in v.split(',')]   # noqa: E731
    for waves, acq, chunk, epi, sleep in itertools.product(ints(args.waves), ints(args.modes), ints(args.chunks), ints(args.epi), ints(args.sleep)):
        if True:
            tag = f"engine w{waves} acq{acq} chunk{chunk} epi{epi} sleep{sleep}"
            be = Bufs(M, d, F, dev)
            be.shift.copy_(shift)
            flags = torch.zeros(3, _C.FFN_ENGINE_FLAG_BYTES, dtype=torch.uint8, device=dev)
            err = torch.zeros(1, dtype=torch.int32, device=dev)
            set_of = lambda li: 2 if (li == NL - 1 and NL % 2 == 1) else li & 1   # noqa: E731
            descs = []

            def run_engine(trace=None):
                be.x.copy_(x_init)
                for li, L in enumerate(layers):
                    tr = None if trace is None else trace[li]
                    e = engine_desc(L, be, M, d, F, eps, flags[set_of(li)], flags[set_of((li + 1) % NL)], err, acq, waves, tr, chunk, epi, sleep)
                    descs.append(e)
                    _C.ffn_engine(e)
            try:
                # parity layer by layer (single launches; flags re-zeroed by hand)
                worst = [0.0, 0.0, 0.0]
                nflip = 0
                for li, (xi, xo, hid, xt0, xt1) in enumerate(per_layer):
                    flags.zero_()
                    be.x.copy_(xi)
                    e = engine_desc(layers[li], be, M, d, F, eps, flags[0], flags[1], err, acq, waves, None, chunk, epi, sleep)
                    _C.ffn_engine(e)
                    torch.cuda.synchronize()
                    worst[0] = max(worst[0], (be.x - xo).abs().max().item())
                    dh = (be.hidden.float() - hid.float()).abs()
                    worst[1] = max(worst[1], dh.max().item())
                    nflip += int((dh > 0).sum().item())
                    worst[2] = max(worst[2], (be.xt[1].float() - xt1.float()).abs().max().item(),
                                   (be.xt[0].float() - xt0.float()).abs().max().item())
                    if li == 0:
                        e0 = (be.x.double().cpu() - x3r).abs().max().item()
                        eh = (_C.untile_matrix(be.hidden, 16, F)[:M].double().cpu() - hr).abs().max().item()
                        print(f"{tag}: layer 0 vs f64 restatement: x3 max abs err {e0:.3e}, h max abs err {eh:.3e}", flush=True)
                print(f"{tag}: vs the launch chain over {len(per_layer)} layers: x3 max abs diff {worst[0]:.3e}, hidden max abs diff "
                      f"{worst[1]:.3e} ({nflip} elements differ), fragments max abs diff {worst[2]:.3e}; err word {int(err.item())}",
                      flush=True)
                if int(err.item()) != 0 or not (worst[0] < 0.05):
                    print(f"{tag}: NOT timed (err word / parity)", flush=True)
                    continue
                flags.zero_()
                ge = graph_of(run_engine)
                flags.zero_()
                ge.replay()
                torch.cuda.synchronize()
                first = be.x.clone()
                drift = (first - x_chain_final).abs().max().item()
                bad = 0
                for _ in range(args.check_reps):
                    ge.replay()
                    torch.cuda.synchronize()
                    bad += int(not torch.equal(be.x, first))
                t_e = time_graph(ge, args.reps)
                results[tag] = t_e
                print(f"{tag}: {t_e / NL * 1e6:7.2f} us per layer (1 launch) = {t_e / t_chain:.3f} x the chain, "
                      f"{wbytes / (t_e / NL) / 1e12:.2f} TB/s; after {NL} layers max abs diff to the chain {drift:.3e}; "
                      f"{bad}/{args.check_reps} replays differ from the first; err word {int(err.item())}", flush=True)
                if args.trace:
                    trace = torch.zeros(NL, nwg * 2 * 16, dtype=torch.int64, device=dev)
                    flags.zero_()
                    gt = graph_of(lambda: run_engine(trace))
                    for _ in range(5):
                        gt.replay()
                    torch.cuda.synchronize()
                    dump_trace(trace.cpu(), NL, nwg, f"{args.trace}.w{waves}_acq{acq}_chunk{chunk}_epi{epi}_sleep{sleep}.csv")
            except _C.AcmiError as ex:
                print(f"{tag}: {ex}", flush=True)


def dump_trace(tr, NL, nwg, path):
    """per layer: stamps relative to the launch's earliest stamp 0; rows = (layer, stamp): min / median / max over workgroups"""
    import numpy as np
    t = tr.numpy().reshape(NL, nwg, 2, 16).astype(np.int64)
    names_c = ['entry', 'B1 passed (partials of op0)', 'flag 1 stored', 'all flags 1 seen', 'B3b passed (h staged)',
               'flag 2 stored', 'all flags 2 seen', 'B5 passed (partials of op2)', 'x3 stored']
    names_w = ['entry', 'act0 back', 'op0 MFMAs done', 'edge 1 seen, rest of op1 requested', 'act1 back', 'op1 MFMAs done',
               'edge 2 seen, rest of op2 requested', 'act2 back (first chunk)', 'op2 MFMAs done']
    with open(path, 'w') as f:
        f.write('wave,stamp,name,min_us,median_us,max_us\n')
        for which, names in ((0, names_c), (1, names_w)):
            rel = []
            for li in range(2, NL):       # skip the first launches of the graph
                t0 = t[li, :, :, 0].min()
                rel.append((t[li, :, which, :9] - t0) / 100.0)
            rel = np.stack(rel)           # [layers, nwg, 9]
            for i, nm in enumerate(names):
                v = rel[:, :, i]
                f.write(f"{'control' if which == 0 else 'compute0'},{i},{nm},{np.median(v.min(axis=1)):.2f},{np.median(v):.2f},"
                        f"{np.median(v.max(axis=1)):.2f}\n")
        # skew: per XCD (control wave's XCC_ID in stamp 15) and per workgroup index, 'B1 passed' and 'flag 2 stored'
        xcc = (t[2, :, 0, 15] >> 32) & 0xf
        for i, nm in ((1, 'B1 passed'), (5, 'flag 2 stored')):
            v = np.stack([(t[li, :, 0, i] - t[li, :, :, 0].min()) / 100.0 for li in range(2, NL)])   # [layers, nwg]
            per_x = [float(np.median(v[:, xcc == x])) for x in range(8)]
            per_j = np.median(v, axis=0)
            f.write(f"skew,{i},{nm}: median by XCD," + ' '.join(f'{q:.2f}' for q in per_x) + ",-,-\n")
            f.write(f"skew,{i},{nm}: per-workgroup medians (min / median / max over workgroups) and same-workgroup spread over layers (median of max - min),"
                    f"{per_j.min():.2f},{np.median(per_j):.2f},{per_j.max():.2f} / {np.median(v.max(axis=0) - v.min(axis=0)):.2f}\n")
        cnt = t[2:, :, :, 9:11]
        f.write(f"control,9,polls of edge 1 / edge 2 (median),{np.median(cnt[:, :, 0, 0]):.1f},{np.median(cnt[:, :, 0, 1]):.1f},-\n")
        f.write(f"compute0,9,stream fragments requested when edge 1 / edge 2 resolved (median),{np.median(cnt[:, :, 1, 0]):.1f},"
                f"{np.median(cnt[:, :, 1, 1]):.1f},-\n")
        # launch-to-launch: entry of layer l+1 minus x3 stored of layer l (slowest workgroup)
        gaps = [(t[li + 1, :, :, 0].min() - t[li, :, 0, 8].max()) / 100.0 for li in range(2, NL - 1)]
        spans = [(t[li, :, 0, 8].max() - t[li, :, :, 0].min()) / 100.0 for li in range(2, NL)]
        f.write(f"all,-,span (first entry -> last x3 store),{min(spans):.2f},{np.median(spans):.2f},{max(spans):.2f}\n")
        f.write(f"all,-,gap to the next launch's first entry,{min(gaps):.2f},{np.median(gaps):.2f},{max(gaps):.2f}\n")
    print(open(path).read(), flush=True)


if __name__ == '__main__':
    main()
