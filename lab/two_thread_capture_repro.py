"""Lab (dev tool): minimal repro attempt for the round-5 crash "two host threads generating concurrently with hipGraph capture in
thread-local mode dumped core 1 run in 10" (profiles/r05_two_thread_generate_x10.txt).

Mode `torch` uses NO kernel of this package: each of two threads captures a short chain of torch element-wise kernels on its own
stream with capture_error_mode='thread_local' and replays it, again and again, while the other thread does the same.  Mode `acmi`
runs the withdrawn test's body (two debug LMs generating concurrently, thread-local capture, the device lock bypassed).
A crash in `torch` mode pins the fault on the runtime's concurrent capture / replay path; a crash only in `acmi` mode on this package.

    python -X faulthandler lab/two_thread_capture_repro.py torch|acmi [rounds]
"""
import faulthandler
import sys
import threading

import torch

faulthandler.enable(all_threads=True)
mode = sys.argv[1] if len(sys.argv) > 1 else 'torch'
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 30
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
errors = []


def torch_worker(i):
    try:
        x = torch.full((1 << 16,), float(i + 1), device='cuda')
        with torch.cuda.stream(streams[i]):
            for r in range(rounds):
                y = torch.empty_like(x)
                g = torch.cuda.CUDAGraph()
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    g.capture_begin(capture_error_mode='thread_local')
                    try:
                        t = x
                        for _ in range(40):
                            t = t * 1.0001 + 0.5
                        y.copy_(t)
                    finally:
                        g.capture_end()
                torch.cuda.current_stream().wait_stream(side)
                for _ in range(60):
                    g.replay()
                streams[i].synchronize()
                v = float(y[0])              # a synchronising read, like generate()'s .item() calls
                assert v > 0
                del g
    except Exception as e:   # noqa: BLE001
        errors.append(repr(e))


def acmi_workers():
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from audiocraft_amd import _C
    from audiocraft_amd.models import builders
    from audiocraft_amd.modules.conditioners import ConditioningAttributes
    # bypass the lock and capture thread-locally: the round-5 configuration
    _C.device_lock = type('NoLock', (), {'__enter__': lambda s: None, '__exit__': lambda s, *a: False})()
    _begin = torch.cuda.CUDAGraph.capture_begin
    torch.cuda.CUDAGraph.capture_begin = lambda self, *a, **kw: _begin(self, *a, **dict(kw, capture_error_mode='thread_local'))
    lms = [builders.get_debug_lm_model('cuda') for _ in range(2)]
    conds = [ConditioningAttributes(text={'description': 'a b c'}), ConditioningAttributes(text={'description': 'd e'})]

    def work(i):
        try:
            with torch.cuda.stream(streams[i]):
                for _ in range(rounds):
                    lms[i].generate(None, conds, max_gen_len=60, use_sampling=True, top_k=50, seed=11 + i)
                streams[i].synchronize()
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))
    return work


worker = torch_worker if mode == 'torch' else acmi_workers()
threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
for t in threads:
    t.start()
for t in threads:
    t.join()
print(f"mode {mode}: {rounds} rounds per thread, errors: {errors if errors else 'none'}", flush=True)
