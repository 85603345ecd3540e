"""The LSTM recurrence alone (acmi_lstm_layer / acmi_lstm_layer_ex through the C-ABI) at EnCodec-32k's shape: B rows x H = 1024 x
T steps, one layer (dev / documentation tool; numbers go to profiles/ and DESIGN.md).

    python scripts/lstm_bench.py [--B 8] [--T 1500] [--reps 5]
ACMI_LSTM_XCD = 1 (default: one recurrence per XCD) / 2 (same kernel, memory-side stores and loads) / 0 (all-CU form).
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from audiocraft_amd import _C  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--B', type=int, default=8)
    ap.add_argument('--H', type=int, default=1024)
    ap.add_argument('--T', type=int, default=1500)
    ap.add_argument('--reps', type=int, default=5)
    args = ap.parse_args()
    B, H, T = args.B, args.H, args.T
    g = torch.Generator().manual_seed(0)
    gates = torch.randn(B, 4 * H, T, generator=g).cuda()
    w_hh = torch.empty(4 * H, H).uniform_(-1 / H ** 0.5, 1 / H ** 0.5, generator=g).cuda()
    y = torch.empty(B, H, T, device='cuda')
    for mode in os.environ.get('LSTM_MODES', '1,0,2').split(','):
        os.environ['ACMI_LSTM_XCD'] = mode
        work = torch.empty(_C.lstm_layer_work_floats(B, H, T), device='cuda')

        def run():
            work[:5 * B * H + 4].zero_()
            _C.lstm_layer(gates, w_hh, None, y, work, B, H, T)
        run()
        torch.cuda.synchronize()
        err = int(work[5 * B * H:5 * B * H + 1].view(torch.int32)[0])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            run()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        print(json.dumps({'B': B, 'H': H, 'T': T, 'ACMI_LSTM_XCD': mode, 'ms_per_layer': round(ms, 3), 'us_per_step': round(ms * 1e3 / T, 3),
                          'give_ups': err, 'checksum': float(y.double().abs().sum())}), flush=True)


if __name__ == '__main__':
    main()
