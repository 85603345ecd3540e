"""Bench-format JSON line for the codec half of the path (BASELINE.json configs[0]: EnCodec-24k geometry, 1 x 10 s; and the
MusicGen codec EnCodec-32k, 8 x 30 s): encode + decode audio seconds per wall second on one MI355X, with

  roofline      the convolutions (`conv_mfma_kernel`, exact-f32 MFMA): conv flops of one encode + decode pass / the summed durations
                of that kernel family in a rocprofv3 --kernel-trace --stats child pass of the same passes, against the dense f32
                MFMA peak (157.3 TF/s); `mfma_busy` = SQ_VALU_MFMA_BUSY_CYCLES of those dispatches (a separate --pmc child pass)
                over duration x 1024 SIMDs x 2.4 GHz
  roofline_rvq  `rvq_encode_kernel`: algorithmic bytes (latents read once + codes written + the codebooks once) / its duration
  lstm          the recurrence kernels: us per time step and layer
  cpu_baseline  the oracle (port of the reference codec) on the host cores, on a bounded sample (2 s of audio at the same batch)

    python scripts/codec_line.py [24k|32k] [batch] [seconds]      (child mode: ... --child N)
"""
import csv
import glob
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from audiocraft_amd import _C  # noqa: E402
from audiocraft_amd.models import builders  # noqa: E402

F32_MFMA_PEAK_TFS, CLOCK_HZ, N_SIMD = 157.3, 2.4e9, 1024
CFGS = {'24k': ('EnCodec-24k geometry (BASELINE.json configs[0])', builders.ENCODEC_24KHZ),
        '32k': ('EnCodec-32k (the MusicGen codec)', builders.ENCODEC_32KHZ)}


def build(which, B, seconds):
    torch.manual_seed(0)
    cfg = CFGS[which][1]
    m = builders.get_compression_model(cfg, 'cuda')
    wav = 0.1 * torch.randn(B, cfg['channels'], int(seconds * cfg['sample_rate']), device='cuda')
    return cfg, m, wav


def child(which, B, seconds, passes):
    """`passes` encode + decode passes (each call executes the pass's kernels exactly once: eagerly, or as a hipGraph replay)."""
    _, m, wav = build(which, B, seconds)
    for _ in range(passes):
        codes, _ = m.encode(wav)
        m.decode(codes)
    torch.cuda.synchronize()
    print(f"ran {passes} passes", flush=True)


def profile_pass(args, extra, pattern, timeout_s=300):
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    tmp = tempfile.mkdtemp(prefix='acmi_codec_', dir='/tmp')
    try:
        cmd = [exe] + extra + ['--output-format', 'csv', '-d', tmp, '--', sys.executable, os.path.abspath(__file__)] + args
        bench._run_child(cmd, dict(os.environ, TMPDIR='/tmp'), timeout_s)
        files = glob.glob(os.path.join(tmp, '**', pattern), recursive=True)
        with open(files[0]) as f:
            return list(csv.DictReader(f))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def family(name):
    for key in ('conv_mfma_kernel', 'conv_fewout_kernel', 'conv_pw_kernel', 'resblock_kernel', 'conv_pack_kernel', 'rvq_encode_kernel', 'rvq_decode_kernel', 'lstm_'):
        if key in name:
            return key.rstrip('_')
    return None


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else '24k'
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
    if '--child' in sys.argv:
        return child(which, B, seconds, int(sys.argv[sys.argv.index('--child') + 1]))
    passes = 4
    cfg, m, wav = build(which, B, seconds)
    codes, _ = m.encode(wav)
    m.decode(codes)
    torch.cuda.synchronize()
    # flops / bytes of one pass, counted at the C-ABI boundary
    import audiocraft_amd.modules.seanet as seanet
    acc = {'flops': 0.0, 'lstm_steps': 0, 'pw_bytes': 0.0, 'pw_flops': 0.0}
    conv0, lstm0, lstm20 = _C.conv1d_tiled, _C.lstm_layer, _C.lstm_stack2

    pw_on = os.environ.get('ACMI_CONV_PW', '1') != '0'

    def conv1d(d, x, w, bias, residual, y):
        fl = 2.0 * d.B * d.Cout * d.Cin * d.ksize * (d.Tout // max(d.shuffle, 1) if d.shuffle > 1 else d.Tout)
        # (the condition of conv_geometry's pointwise path, acmi_conv.hip: those launches are conv_pw_kernel's, a stream, not the MFMA kernel's)
        if (pw_on and d.ksize == 1 and d.stride == 1 and d.elu_in and d.Tout == d.Tin and d.Tout % 4 == 0 and d.B * d.Tout >= 65536 and
                (d.Cin, d.Cout) in ((32, 64), (64, 128))):
            acc['pw_flops'] += fl
            acc['pw_bytes'] += 4.0 * d.B * d.Tout * (d.Cin + d.Cout * (2 if residual is not None else 1))
        else:
            acc['flops'] += fl
        return conv0(d, x, w, bias, residual, y)

    def lstm_layer(gates, w_hh, skip, out, work, Bq, H, T):
        acc['lstm_steps'] += T
        return lstm0(gates, w_hh, skip, out, work, Bq, H, T)

    def lstm_stack2(gates, w_hh0, w_ih1, w_hh1, bias1, skip, out, Bq, H, T):
        acc['lstm_steps'] += 2 * T
        return lstm20(gates, w_hh0, w_ih1, w_hh1, bias1, skip, out, Bq, H, T)
    m.__dict__.pop('_graphs', None)
    _C.conv1d_tiled = seanet._C.conv1d_tiled = conv1d
    _C.lstm_layer = seanet._C.lstm_layer = lstm_layer
    _C.lstm_stack2 = seanet._C.lstm_stack2 = lstm_stack2
    old_max, type(m).GRAPH_MAX_SAMPLES = type(m).GRAPH_MAX_SAMPLES, 0     # eager: the wrappers see every call
    m.encode(wav)
    m.decode(codes)
    type(m).GRAPH_MAX_SAMPLES = old_max
    _C.conv1d_tiled = seanet._C.conv1d_tiled = conv0
    _C.lstm_layer = seanet._C.lstm_layer = lstm0
    _C.lstm_stack2 = seanet._C.lstm_stack2 = lstm20
    torch.cuda.synchronize()
    # ---- the timed region: encode + decode, inputs resident
    for _ in range(2):
        m.decode(m.encode(wav)[0])
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        t_e = time.perf_counter()
        c2, _ = m.encode(wav)
        torch.cuda.synchronize()
        t_d = time.perf_counter()
        m.decode(c2)
        torch.cuda.synchronize()
        t_x = time.perf_counter()
    dt = (t_x - t0) / reps
    ms_enc, ms_dec = (t_d - t_e) * 1e3, (t_x - t_d) * 1e3
    K, T = codes.shape[1], codes.shape[2]
    D, bins = cfg['seanet']['dimension'], cfg['rvq']['bins']
    out = {
        "metric": "audio seconds encoded + decoded / wall-sec, EnCodec", "value": round(B * seconds / dt, 1),
        "unit": "audio-s / wall-s", "n_gpus": 1, "steps": reps, "warmup": 2, "ms_per_step": round(dt * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (random-init weights, synthetic audio)",
        "config": {"workload": f"{CFGS[which][0]}: encode + decode of {B} x {seconds:g} s, {K} codebooks x {bins}, {T} frames",
                   "global_batch": B, "seq_len": T, "parallelism": "dp1"},
        "ms": {"encode": round(ms_enc, 3), "decode": round(ms_dec, 3)},
    }
    # ---- the kernels inside those passes (child processes under rocprofv3)
    args = [which, str(B), str(seconds), '--child', str(passes)]
    try:
        rows = profile_pass(args, ['--kernel-trace', '--stats'], '*kernel_stats.csv')
        fam = {}
        for r in rows:
            k = family(r['Name'])
            if k:
                c, tot = fam.get(k, (0, 0.0))
                fam[k] = (c + int(r['Calls']), tot + float(r['TotalDurationNs']))
        out["kernel_us_per_pass"] = {k: round(tot / passes / 1e3, 2) for k, (c, tot) in fam.items()}
        conv_s = fam['conv_mfma_kernel'][1] / passes * 1e-9
        tfs = acc['flops'] / conv_s / 1e12
        out["roofline"] = {"kernel": "conv_mfma_kernel (implicit-GEMM Conv1d / ConvTranspose1d on v_mfma_f32_32x32x2_f32)", "bound": "mfma",
                           "achieved": round(tfs, 2), "peak": F32_MFMA_PEAK_TFS, "unit": "TFLOP/s", "frac": round(tfs / F32_MFMA_PEAK_TFS, 4),
                           "flops_per_pass": acc['flops'], "launches_per_pass": fam['conv_mfma_kernel'][0] // passes,
                           "avg_launch_us": round(fam['conv_mfma_kernel'][1] / fam['conv_mfma_kernel'][0] / 1e3, 3),
                           "traffic": None, "note": "all convolutions of one encode + decode pass incl. the LSTM input projections"}
        if 'conv_pw_kernel' in fam and acc['pw_bytes'] > 0:
            pw_s = fam['conv_pw_kernel'][1] / passes * 1e-9
            out["roofline_pw"] = {"kernel": "conv_pw_kernel (the k = 1 channel-doubling conv of the narrow resnet blocks: ELU, bias, skip; VALU fmaf chains)",
                                  "bound": "hbm", "achieved": round(acc['pw_bytes'] / pw_s / 1e9, 1), "peak": bench.HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(acc['pw_bytes'] / pw_s / 1e9 / bench.HBM_PEAK_GBS, 4), "bytes_per_pass": acc['pw_bytes'],
                                  "launches_per_pass": fam['conv_pw_kernel'][0] // passes, "tflops": round(acc['pw_flops'] / pw_s / 1e12, 2)}
        if 'rvq_encode_kernel' in fam:
            rvq_bytes = 4 * B * D * T + 8 * B * K * T + 4 * K * bins * D
            us = fam['rvq_encode_kernel'][1] / fam['rvq_encode_kernel'][0] / 1e3
            out["roofline_rvq"] = {"kernel": "rvq_encode_kernel (nearest-codebook cascade)", "bound": "hbm",
                                   "achieved": round(rvq_bytes / (us * 1e-6) / 1e9, 2), "peak": bench.HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(rvq_bytes / (us * 1e-6) / 1e9 / bench.HBM_PEAK_GBS, 5), "bytes_per_launch": rvq_bytes,
                                   "avg_launch_us": round(us, 2),
                                   "search_gflops": round(2.0 * B * T * K * bins * D / (us * 1e-6) / 1e9, 1),
                                   "note": "the search is compute / L2 bound (2 B T K bins D flops on codebooks that stay in L2), not a stream: "
                                           "the HBM fraction is what the contract asks for, the flop rate is what the kernel is priced on"}
        if 'lstm' in fam and acc['lstm_steps'] > 0:
            out["lstm"] = {"us_per_step_and_layer": round(fam['lstm'][1] / passes / 1e3 / acc['lstm_steps'], 3),
                           "steps_per_pass": acc['lstm_steps'], "note": "recurrence kernels only (the input projections are convolutions)"}
        try:
            prow = profile_pass(args, ['--pmc', 'SQ_VALU_MFMA_BUSY_CYCLES'], '*counter_collection.csv')
            busy = sum(float(r['Counter_Value']) for r in prow if r['Counter_Name'] == 'SQ_VALU_MFMA_BUSY_CYCLES' and 'conv_mfma_kernel' in r['Kernel_Name'])
            out["roofline"]["mfma_busy"] = round(busy / passes / (conv_s * CLOCK_HZ * N_SIMD), 4)
            out["roofline"]["mfma_busy_source"] = "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES (own pass) / (conv_mfma_kernel seconds x 2.4 GHz x 1024 SIMDs)"
        except Exception as e:   # noqa: BLE001
            out["roofline"]["mfma_busy"] = None
            out["roofline"]["mfma_busy_source"] = f"the counter pass failed: {type(e).__name__}"
    except Exception as e:   # noqa: BLE001
        out["roofline"] = {"error": f"the kernel-trace pass failed: {type(e).__name__}: {e}"}
    # ---- CPU baseline: the oracle (port of the reference codec) on a bounded sample
    if '--no-cpu' not in sys.argv:
        from oracle import codec as ocodec
        sk = cfg['seanet']
        oc = ocodec.CodecConfig(channels=sk['channels'], dimension=sk['dimension'], n_filters=sk['n_filters'],
                                n_residual_layers=sk['n_residual_layers'], ratios=list(sk['ratios']), causal=sk['causal'],
                                pad_mode=sk['pad_mode'], lstm=sk['lstm'], norm='weight_norm', n_q=cfg['rvq']['n_q'], bins=bins,
                                sample_rate=cfg['sample_rate'], frame_rate=cfg['frame_rate'])
        sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
        s_sec = min(seconds, 2.0)
        w = wav[:, :, :int(s_sec * cfg['sample_rate'])].cpu()
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        t0 = time.perf_counter()
        lat = ocodec.seanet_encoder(sd, oc, w)
        cc = ocodec.rvq_encode(lat, ocodec.codebooks_from_state(sd, oc.n_q))
        ocodec.encodec_decode(sd, oc, cc)
        t_cpu = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(B * s_sec / t_cpu, 3), "unit": "audio-s / wall-s", "cores": torch.get_num_threads(),
                               "kind": "port", "sample": f"oracle encode + RVQ + decode of {B} x {s_sec:g} s ({t_cpu:.2f} s)"}
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
