"""Per-kernel timing of the hot path at the BASELINE.json configs[2] shapes (dev tool).
GPU-side times: each op is captured into a hipGraph of REP launches cycling over all layers' weights
(cold, like in a real decode position) and replayed, so python/ctypes overhead is excluded."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from audiocraft_amd import _C  # noqa: E402
from audiocraft_amd.models.musicgen import MusicGen  # noqa: E402


def graph_time(fn_list, reps=5):
    """fn_list: callables launched back to back in one graph; returns us per callable."""
    for f in fn_list[:2]:
        f()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        g.capture_begin()
        try:
            for f in fn_list:
                f()
        finally:
            g.capture_end()
        g.replay()
        side.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(reps):
            g.replay()
        e1.record(side)
        side.synchronize()
    torch.cuda.current_stream().wait_stream(side)
    return e0.elapsed_time(e1) * 1e3 / (reps * len(fn_list))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='facebook/musicgen-medium')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--gen', type=int, default=0, help='also time a generate of this many frames')
    ap.add_argument('--codec', action='store_true')
    args = ap.parse_args()
    t0 = time.time()
    model = MusicGen.get_random_init(args.model, 'cuda', torch.bfloat16)
    torch.cuda.synchronize()
    print(f'model build {time.time() - t0:.1f}s', flush=True)
    lm = model.lm
    pk = lm._pack()
    M, d, ffn, wd = 2 * args.batch, lm.dim, lm.ffn_dim, lm.weight_dtype
    L = pk['per_layer']
    x = torch.randn(M, d, device='cuda')
    att = _C.tile_matrix(torch.randn(M, d, device='cuda'), wd)
    hid = _C.tile_matrix(torch.randn(M, ffn, device='cuda'), wd)
    o3 = torch.empty(M, 3 * d, device='cuda')
    o1 = torch.empty(M, d, device='cuda')
    oh = _C.tiled_activation_buffer(M, ffn, wd, 'cuda')
    lg = torch.empty(M, lm.n_q * lm.card, device='cuda')
    ops = [
        ('ln_qkv  ', lambda e: _C.linear(att, e['w_qkv'], o3, bias=e['b_qkv'], a_tiled=True, M=M), 3 * d * d * 2),
        ('out_proj', lambda e: _C.linear(att, e['w_out'], o1, a_tiled=True, M=M, residual=o1), d * d * 2),
        ('ln_ffn1 ', lambda e: _C.linear(att, e['w_ff1'], oh, bias=e['b_ff1'], a_tiled=True, M=M, act=1,
                                         out_mode=_C.OUT_TILED), d * ffn * 2),
        ('ffn2    ', lambda e: _C.linear(hid, e['w_ff2'], o1, a_tiled=True, M=M, residual=o1), d * ffn * 2),
    ]
    for name, fn, nbytes in ops:
        us = graph_time([(lambda e=e: fn(e)) for e in L])
        print(f'{name} {us:8.2f} us/launch  {nbytes / us / 1e3:8.1f} GB/s  (cold weights, incl. launch gap)', flush=True)
    us = graph_time([lambda: _C.linear(att, pk['w_head'], lg, bias=pk['b_head'], a_tiled=True, M=M)] * 8)
    print(f'head     {us:8.2f} us/launch  {lm.n_q * lm.card * d * 2 / us / 1e3:8.1f} GB/s  (warm)', flush=True)
    xn = _C.tiled_activation_buffer(M, d, wd, 'cuda')
    us = graph_time([lambda: _C.ln_tile(x, xn)] * 16)
    print(f'ln_tile  {us:8.2f} us/launch', flush=True)
    H, hd = lm.num_heads, d // lm.num_heads
    for ln in (64, 512, 1500):
        ks = [torch.randn(M, H, 1504, hd, device='cuda').bfloat16() for _ in range(8)]
        vs = [torch.randn_like(k) for k in ks]
        us = graph_time([(lambda k=k, v=v: _C.attn_decode(x, k, v, att, ln, out_tiled=True)) for k, v in zip(ks, vs)])
        print(f'attn len={ln:5d} {us:8.2f} us  {2 * M * H * ln * hd * 2 / us / 1e3:8.1f} GB/s', flush=True)
        del ks, vs
    if args.gen:
        g = torch.Generator().manual_seed(0)
        cross = torch.randn(M, 16, d, generator=g).cuda()
        cross[args.batch:] = 0
        ct = {'description': (cross, torch.ones(M, 16, dtype=torch.int64).cuda())}
        for use_graph in (False, True):
            t0 = time.time()
            lm.generate(None, [], num_samples=args.batch, max_gen_len=args.gen, condition_tensors=ct, top_k=250,
                        use_graph=use_graph)
            torch.cuda.synchronize()
            dt = time.time() - t0
            print(f'generate {args.gen} frames graph={use_graph}: {dt:.2f}s  {dt / (args.gen + 3) * 1e3:.3f} ms/position',
                  flush=True)
    if args.codec:
        codes = torch.randint(0, 2048, (args.batch, 4, 1500), device='cuda')
        for _ in range(2):
            t0 = time.time()
            wav = model.compression_model.decode(codes)
            torch.cuda.synchronize()
            print(f'encodec decode {tuple(wav.shape)}: {time.time() - t0:.3f}s', flush=True)
        t0 = time.time()
        c2, _ = model.compression_model.encode(wav)
        torch.cuda.synchronize()
        print(f'encodec encode -> {tuple(c2.shape)}: {time.time() - t0:.3f}s', flush=True)


if __name__ == '__main__':
    main()
