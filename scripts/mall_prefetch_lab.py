"""Lab (dev tool): a CONCURRENT Infinity-Cache prefetcher for the decode GEMM chain -- the experiment DESIGN.md section 5.5
bounded from above (weights MALL-resident: -0.85 us per big launch, -2.9 us per layer) and left unbuilt.

The GEMM chain of one decode position (bench.measure_lin_kernel's launches) captured into a hipGraph with a second branch:
while launch k runs, a touch kernel on a side stream (lab/prefetch_lab.hip: `wgs` workgroups of 256 threads) reads the weights
of launch k + 1, so that they sit in the 256 MB memory-side cache when that launch requests them.  Fork / join per launch:
event after launch k - 1 -> side stream -> touch(w[k + 1]); the chain never waits for a touch (best effort), the graph joins
at the end.  Prints us per GEMM launch for the plain chain and for each prefetcher geometry.

    python scripts/mall_prefetch_lab.py [--wgs 16,32,64] [--frac 1.0]
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from audiocraft_amd import _C  # noqa: E402
from audiocraft_amd.models.musicgen import MusicGen  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--wgs', default='16,32,64')
ap.add_argument('--frac', type=float, default=1.0)
ap.add_argument('--reps', type=int, default=5)
ap.add_argument('--paced', default='', help='persistent time-paced prefetcher: comma list of wgs:lead, wgs:lead:percent, e.g. 32:3:100,64:3:50')
args = ap.parse_args()
lab = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'lab', 'libprefetch_lab.so'))
lab.lab_touch.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
lab.lab_paced_touch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
sink = torch.zeros(4, dtype=torch.int32, device='cuda')

model = MusicGen.get_random_init('facebook/musicgen-medium', 'cuda', torch.bfloat16)
model.lm._pack()

orig_desc, orig_launch, orig_pair = _C.linear_desc, _C.linear_launch, _C.linear_pair
state = {'mode': 'record', 'seq': [], 'k': 0, 'wgs': 0, 'side': None}


def weights_of(d):
    return getattr(d, '_probe_w', [])


def hook(ws):
    """called right before launch k is enqueued on the main stream"""
    if state['mode'] == 'record':
        state['seq'].append(ws)
        return
    k = state['k']
    state['k'] += 1
    if state['mode'] == 'paced' and k == 0:
        main, side = torch.cuda.current_stream(), state['side']
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        lab.lab_paced_touch(state['table'].data_ptr(), len(state['seq_flat']), state['period'], state['lead'], state['wgs'],
                            state['pct'], sink.data_ptr(), side.cuda_stream)
    if state['mode'] == 'prefetch' and k + 1 < len(state['seq']):
        main, side = torch.cuda.current_stream(), state['side']
        ev = torch.cuda.Event()
        ev.record(main)                 # launch k - 1 (everything enqueued so far) has finished
        side.wait_event(ev)
        for w in state['seq'][k + 1]:   # the NEXT launch's weights, read while launch k runs
            nbytes = int(w.data.numel() * w.data.element_size() * args.frac) // 4096 * 4096
            lab.lab_touch(w.data.data_ptr(), nbytes, state['wgs'], sink.data_ptr(), side.cuda_stream)


def linear_desc(a, w, *a2, **kw):
    d = orig_desc(a, w, *a2, **kw)
    d._probe_w = [w]
    return d


def linear_launch(d):
    hook(weights_of(d))
    return orig_launch(d)


def linear_pair(p0, p1):
    hook(weights_of(p0) + weights_of(p1))
    return orig_pair(p0, p1)


def linear_ex(a, w, out, M, a_mode, out_mode, **kw):
    hook([w])
    orig_launch(orig_desc(a, w, out, M, a_mode, out_mode, **kw))
    return out


_C.linear_desc, _C.linear_launch, _C.linear_pair, _C.linear_ex = linear_desc, linear_launch, linear_pair, linear_ex


def run(mode, wgs=0):
    """bench.measure_lin_kernel with the hooks in `mode`; its capture runs on a side stream of its own, ours forks from it"""
    state.update(mode=mode, k=0, wgs=wgs, side=torch.cuda.Stream())
    orig_cap_end = torch.cuda.CUDAGraph.capture_end

    def capture_end(self):              # join the prefetch branch before the capture closes
        torch.cuda.current_stream().wait_stream(state['side'])
        return orig_cap_end(self)
    torch.cuda.CUDAGraph.capture_end = capture_end
    try:
        # measure_lin_kernel calls one_position() once eagerly (warm) and once inside the capture: restart the launch index
        orig_begin = torch.cuda.CUDAGraph.capture_begin

        def capture_begin(self, *a, **kw):
            state['k'] = 0
            return orig_begin(self, *a, **kw)
        torch.cuda.CUDAGraph.capture_begin = capture_begin
        r = bench.measure_lin_kernel(model, 16, reps=args.reps)
    finally:
        torch.cuda.CUDAGraph.capture_end = orig_cap_end
        torch.cuda.CUDAGraph.capture_begin = orig_begin
    torch.cuda.synchronize()
    return r


run('record')
n = len(state['seq']) // 2    # the eager warm-up pass + ... (record mode sees both passes)
state['seq'] = state['seq'][:n]
print(f"{n} GEMM launches per position", flush=True)
base = run('plain')
print(f"plain chain            : {base['avg_us']:.3f} us per launch", flush=True)
for wgs in [int(v) for v in args.wgs.split(',') if v]:
    r = run('prefetch', wgs)
    print(f"prefetch, {wgs:3d} workgroups: {r['avg_us']:.3f} us per launch ({r['avg_us'] / base['avg_us']:.3f} x)", flush=True)
if args.paced:
    import struct
    flat = [w for ws in state['seq'] for w in ws]          # one entry per weight matrix, in launch order
    per_launch = [len(ws) for ws in state['seq']]
    # pace per ENTRY: entries of one launch share its due time -> expand the period table by repeating (lead counts entries ~ launches)
    raw = b''.join(struct.pack('<QQ', w.data.data_ptr(), w.data.numel() * w.data.element_size()) for w in flat)
    state['table'] = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
    state['seq_flat'] = flat
    total_us = base['avg_us'] * n
    state['period'] = max(1, int(total_us / len(flat) * 100))      # ticks of 10 ns per table entry
    for spec in args.paced.split(','):
        wgs, lead, pct = (int(v) for v in spec.split(':'))
        state['lead'], state['pct'] = lead, pct
        r = run('paced', wgs)
        print(f"paced prefetcher, {wgs:3d} workgroups, {lead} entries ahead, {pct:3d} % of every matrix: {r['avg_us']:.3f} us per launch ({r['avg_us'] / base['avg_us']:.3f} x)", flush=True)
base2 = run('plain')
print(f"plain chain (again)    : {base2['avg_us']:.3f} us per launch", flush=True)
