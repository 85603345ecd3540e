"""Per-launch breakdown of the EnCodec-32k decode / encode of 8 x 30 s (dev tool): every acmi_conv1d / acmi_lstm_layer call
timed with HIP events (synchronised: ~10 us of launch overhead per row, irrelevant at these sizes), with its shape, flops
and fraction of the f32 MFMA peak.
    python scripts/codec_layers.py [decode|encode]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from audiocraft_amd import _C  # noqa: E402
from audiocraft_amd.models import builders  # noqa: E402
import audiocraft_amd.modules.seanet as seanet  # noqa: E402

rows = []
_conv1d, _lstm = _C.conv1d_tiled, _C.lstm_layer


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1)


def conv1d(d, x, w, bias, residual, y):
    ms = timed(lambda: _conv1d(d, x, w, bias, residual, y))
    tq = d.Tout // max(d.shuffle, 1) if d.shuffle > 1 else d.Tout
    fl = 2.0 * d.B * d.Cout * d.Cin * d.ksize * tq
    rows.append((f"conv Cin {d.Cin:4d} Cout {d.Cout:4d} k {d.ksize:2d} s {d.stride} dil {d.dilation} shuffle {d.shuffle} "
                 f"Tin {d.Tin:7d} Tout {d.Tout:7d}", ms, fl))


def lstm_layer(gates, w_hh, skip, out, work, B, H, T):
    ms = timed(lambda: _lstm(gates, w_hh, skip, out, work, B, H, T))
    rows.append((f"lstm B {B} H {H} T {T}", ms, 2.0 * B * 4 * H * H * T))


_lstm2 = _C.lstm_stack2


def lstm_stack2(gates, w_hh0, w_ih1, w_hh1, bias1, skip, out, B, H, T):
    ms = timed(lambda: _lstm2(gates, w_hh0, w_ih1, w_hh1, bias1, skip, out, B, H, T))
    rows.append((f"lstm x2 (wavefront, incl. layer 1's input projection) B {B} H {H} T {T}", ms, 3 * 2.0 * B * 4 * H * H * T))


what = sys.argv[1] if len(sys.argv) > 1 else 'decode'
torch.manual_seed(0)
m = builders.get_compression_model(builders.ENCODEC_32KHZ, 'cuda')
wav = 0.1 * torch.randn(8, 1, 30 * 32000, device='cuda')
codes, _ = m.encode(wav)
m.decode(codes)
torch.cuda.synchronize()
_C.conv1d_tiled, _C.lstm_layer = conv1d, lstm_layer
seanet._C.conv1d_tiled, seanet._C.lstm_layer = conv1d, lstm_layer
_C.lstm_stack2 = seanet._C.lstm_stack2 = lstm_stack2
(m.decode(codes) if what == 'decode' else m.encode(wav))
torch.cuda.synchronize()
tot = sum(r[1] for r in rows)
for name, ms, fl in rows:
    print(f"{ms:8.3f} ms {100 * ms / tot:5.1f} %  {fl / ms / 1e9:7.2f} TF/s ({fl / ms / 1e9 / 157.3 * 100:4.1f} % f32 MFMA)  {name}")
print(f"{tot:8.3f} ms total ({what}, EnCodec-32k, 8 x 30 s)")
