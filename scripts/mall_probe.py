"""Lab (dev tool): how much faster is a decode GEMM launch when its weights sit in the Infinity Cache (MALL)?
The GEMM chain of one decode position (bench.measure_lin_kernel) with every launch preceded by a plain read of its weight
matrix (a torch reduction, same stream: the data is then MALL-resident, at most partly L2-resident), against the plain chain.
Run under `rocprofv3 --kernel-trace --stats` and compare the lin_tiled_kernel / lin_pair_kernel averages:
    ACMI_PROBE_TOUCH=1 rocprofv3 --kernel-trace --stats -d out -- python scripts/mall_probe.py
An upper bound for what a concurrent prefetcher could buy; it says nothing about what such a prefetcher costs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from audiocraft_amd import _C  # noqa: E402
from audiocraft_amd.models.musicgen import MusicGen  # noqa: E402

touch_on = os.environ.get('ACMI_PROBE_TOUCH', '0') == '1'
sink = []


def touch(w):
    if touch_on:
        sink.append(torch.sum(w.data, dtype=torch.float32))


orig_ex, orig_desc, orig_launch, orig_pair = _C.linear_ex, _C.linear_desc, _C.linear_launch, _C.linear_pair


def linear_ex(a, w, *args, **kw):
    touch(w)
    return orig_ex(a, w, *args, **kw)


def linear_desc(a, w, *args, **kw):
    d = orig_desc(a, w, *args, **kw)
    d._probe_w = w
    return d


def linear_launch(d):
    touch(d._probe_w)
    return orig_launch(d)


def linear_pair(p0, p1):
    touch(p0._probe_w)
    touch(p1._probe_w)
    return orig_pair(p0, p1)


_C.linear_ex, _C.linear_desc, _C.linear_launch, _C.linear_pair = linear_ex, linear_desc, linear_launch, linear_pair
# linear_ex builds its descriptor through the module-level linear_desc / linear_launch: route it around the wrappers
def _plain_ex(a, w, out, M, a_mode, out_mode, **kw):
    orig_launch(orig_desc(a, w, out, M, a_mode, out_mode, **kw))
    return out
orig_ex = _plain_ex  # noqa: E305

model = MusicGen.get_random_init('facebook/musicgen-medium', 'cuda', torch.bfloat16)
model.lm._pack()
r = bench.measure_lin_kernel(model, 16)
print(f"touch={int(touch_on)}: {r['avg_us']:.2f} us per GEMM launch incl. whatever ran in between ({r['launches_per_position']} launches)", flush=True)
