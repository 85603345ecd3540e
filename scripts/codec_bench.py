"""EnCodec throughput (BASELINE.json configs[0] geometry and the 32 kHz MusicGen codec): audio seconds / wall second of
encode + decode on one MI355X, with the two rooflines of SURVEY.md section 8(d): activation bytes (every conv / LSTM
layer reads its input and writes its output once, f32) against HBM peak, and conv flops against the dense f32 MFMA peak
(157.3 TF; the codec computes in exact f32).  Random-init weights, synthetic audio.  Dev / documentation
tool: one JSON line per configuration.

    python scripts/codec_bench.py > profiles/rNN_codec_bench.jsonl
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from audiocraft_amd import _C  # noqa: E402
from audiocraft_amd.models import builders  # noqa: E402

HBM_PEAK, F32_MFMA_PEAK = 8.0e12, 157.3e12
_acc = {'bytes': 0, 'flops': 0}
_conv1d, _lstm = _C.conv1d_tiled, _C.lstm_layer


def conv1d(d, x, w, bias, residual, y):
    _acc['bytes'] += 4 * (x.numel() + y.numel() + (residual.numel() if residual is not None else 0))
    _acc['flops'] += 2 * d.B * d.Cout * d.Cin * d.ksize * (d.Tout // max(d.shuffle, 1) if d.shuffle > 1 else d.Tout)
    return _conv1d(d, x, w, bias, residual, y)


def lstm_layer(gates, w_hh, skip, out, work, B, H, T):
    _acc['bytes'] += 4 * (gates.numel() + out.numel() + (skip.numel() if skip is not None else 0))
    _acc['flops'] += 2 * B * 4 * H * H * T
    return _lstm(gates, w_hh, skip, out, work, B, H, T)


_lstm2 = _C.lstm_stack2


def lstm_stack2(gates, w_hh0, w_ih1, w_hh1, bias1, skip, out, B, H, T):
    _acc['bytes'] += 4 * (gates.numel() + out.numel() + (skip.numel() if skip is not None else 0))
    _acc['flops'] += 3 * 2 * B * 4 * H * H * T
    return _lstm2(gates, w_hh0, w_ih1, w_hh1, bias1, skip, out, B, H, T)


def run(name, cfg, B, seconds):
    torch.manual_seed(0)
    m = builders.get_compression_model(cfg, 'cuda')
    wav = 0.1 * torch.randn(B, cfg['channels'], int(seconds * cfg['sample_rate']), device='cuda')
    codes, _ = m.encode(wav)          # warm-up (weight-norm fold, polyphase weights)
    m.decode(codes)
    torch.cuda.synchronize()
    out = {'config': name, 'batch': B, 'seconds': seconds, 'codes': list(codes.shape)}
    import audiocraft_amd.modules.seanet as seanet
    for what, fn in (('encode', lambda: m.encode(wav)), ('decode', lambda: m.decode(codes))):
        _acc['bytes'] = _acc['flops'] = 0
        _C.conv1d_tiled, _C.lstm_layer = conv1d, lstm_layer
        seanet._C.conv1d_tiled, seanet._C.lstm_layer = conv1d, lstm_layer
        _C.lstm_stack2 = seanet._C.lstm_stack2 = lstm_stack2
        fn()
        _C.conv1d_tiled, _C.lstm_layer = _conv1d, _lstm
        seanet._C.conv1d_tiled, seanet._C.lstm_layer = _conv1d, _lstm
        _C.lstm_stack2 = seanet._C.lstm_stack2 = _lstm2
        torch.cuda.synchronize()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out[what] = {'ms': round(dt * 1e3, 2), 'audio_s_per_wall_s': round(B * seconds / dt, 1),
                     'activation_bytes': _acc['bytes'], 'hbm_frac': round(_acc['bytes'] / dt / HBM_PEAK, 4),
                     'flops': _acc['flops'], 'f32_mfma_frac': round(_acc['flops'] / dt / F32_MFMA_PEAK, 4)}
    total = (out['encode']['ms'] + out['decode']['ms']) * 1e-3
    out['encode_plus_decode_audio_s_per_wall_s'] = round(B * seconds / total, 1)
    print(json.dumps(out), flush=True)
    del m
    torch.cuda.empty_cache()


if __name__ == '__main__':
    run('EnCodec-24k geometry (BASELINE.json configs[0]), 1 x 10 s', builders.ENCODEC_24KHZ, 1, 10.0)
    run('EnCodec-24k geometry, 16 x 10 s', builders.ENCODEC_24KHZ, 16, 10.0)
    run('EnCodec-32k (MusicGen codec), 8 x 30 s', builders.ENCODEC_32KHZ, 8, 30.0)
