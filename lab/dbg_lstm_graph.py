"""Debug: the codec's hipGraph path with the XCD-local LSTM form (ACMI_LSTM_WAVE=0): which error word comes back?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiocraft_amd import _C
from audiocraft_amd.models import builders
from audiocraft_amd.modules.seanet import StreamableLSTM

torch.manual_seed(0)
for (B, H, T) in [(1, 512, 750), (8, 1024, 200)]:
    m = StreamableLSTM(H, 2, device='cuda')
    x = torch.randn(B, H, T, device='cuda')
    ref = m.run(x).clone()
    torch.cuda.synchronize()
    checks = []
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    static_in = x.clone()
    with torch.cuda.stream(side):
        _C.defer_lstm_checks(checks)
        g.capture_begin()
        out = m.run(static_in)
        g.capture_end()
        _C.defer_lstm_checks(None)
    torch.cuda.current_stream().wait_stream(side)
    for rep in range(4):
        x2 = torch.randn(B, H, T, device='cuda') if rep % 2 else x
        static_in.copy_(x2)
        g.replay()
        torch.cuda.synchronize()
        errs = [hex(int(w.view(torch.int32)[0])) for w, _ in checks]
        want = m.run(x2)
        torch.cuda.synchronize()
        print(B, H, T, 'replay', rep, 'err words', errs, 'max diff vs eager', float((out - want).abs().max()), flush=True)
