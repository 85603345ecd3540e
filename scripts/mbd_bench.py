"""MultiBandDiffusion cost at the released geometry (config/model/score/basic.yaml: hidden 48, depth 4, kernel 8, stride 4,
growth 4; 128-d EnCodec condition at 50 Hz; 32 kHz): one U-Net forward, one reverse process of 20 steps, the 32-band EQ
matching.  Random weights, synthetic inputs.  (The CPU side of the comparison -- the oracle's U-Net forward on the host cores --
is timed by `python -m oracle.time_mbd`: test infrastructure, not imported here.)

    python scripts/mbd_bench.py [--seconds 10] [--batch 1]     -> one JSON line
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=10.)
    ap.add_argument('--batch', type=int, default=1)
    ap.add_argument('--reps', type=int, default=5)
    a = ap.parse_args()
    from audiocraft_amd.models.unet import DiffusionUnet
    from audiocraft_amd.modules.diffusion_schedule import MultiBandProcessor, NoiseSchedule, SplitBands
    from audiocraft_amd import _C
    torch.manual_seed(0)
    kw = dict(chin=1, hidden=48, depth=4, growth=4., max_channels=10_000, num_steps=1000, emb_all_layers=True, bilstm=False,
              codec_dim=128, kernel=8, stride=4, norm_groups=4, res_blocks=1)
    m = DiffusionUnet(**kw).cuda()
    T = int(a.seconds * 32000)
    x = torch.randn(a.batch, 1, T, device='cuda')
    cond = torch.randn(a.batch, 128, T // 640, device='cuda')

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    # flops and activation bytes of one forward, counted at the convolution calls (GEMM rows x columns x K per descriptor)
    acct = {'flops': 0.0, 'bytes': 0.0, 'convs': 0}
    orig = (_C.conv1d_tiled, _C.conv1d_tiled_gn)

    def count(d, xx, wt, y):
        cols = (d.Tout + d.trim_left + d.shuffle - 1) // d.shuffle if d.shuffle > 1 else d.Tout
        acct['flops'] += 2.0 * d.B * d.Cout * d.Cin * d.ksize * cols
        acct['bytes'] += 4.0 * (xx.numel() + y.numel()) + 4.0 * d.Cout * d.Cin * d.ksize
        acct['convs'] += 1

    def c1(d, xx, wt, b, r, y):
        count(d, xx, wt, y)
        return orig[0](d, xx, wt, b, r, y)

    def c2(d, xx, wt, b, r, y, *rest):
        count(d, xx, wt, y)
        return orig[1](d, xx, wt, b, r, y, *rest)

    _C.conv1d_tiled, _C.conv1d_tiled_gn = c1, c2
    m(x, 500, cond)
    _C.conv1d_tiled, _C.conv1d_tiled_gn = orig
    t_fwd = timed(lambda: m(x, 500, cond), a.reps)
    proc = MultiBandProcessor(n_bands=4, sample_rate=32000, num_samples=1)
    proc.load_state_dict({'counts': torch.ones(1), 'sum_x': torch.zeros(4), 'sum_x2': torch.ones(4), 'sum_target_x2': torch.ones(4)})
    sched = NoiseSchedule(beta_t0=1e-5, beta_t1=0.029, beta_exp=7.5, num_steps=1000, sample_processor=proc.cuda())
    t_proc = timed(lambda: sched.generate_subsampled(m, x, condition=cond), 1)
    split = SplitBands(32000, 32)
    ref = torch.randn_like(x)

    def eq():
        lows = split.lows(x)
        st, st2 = split.stats(x, lows), split.stats(ref, split.lows(ref))
        g = (st2[:, 1] / st[:, 1]).sqrt().float().cuda()
        return _C.band_mix(x, lows, g)

    t_eq = timed(eq, 3)
    F32_MFMA_TFLOPS, HBM_GBS = 157.3, 8000.0      # MI355X_MICROARCH.md: f32-input MFMA = the f32 vector rate; HBM3E peak
    out = {'workload': f'MBD U-Net (hidden 48, depth 4, growth 4) B={a.batch} {a.seconds:g} s @ 32 kHz', 'unet_forward_ms': t_fwd * 1e3,
           'unet_forward_gflop': acct['flops'] / 1e9, 'unet_forward_tflops_per_s': acct['flops'] / t_fwd / 1e12,
           'f32_mfma_frac': acct['flops'] / t_fwd / 1e12 / F32_MFMA_TFLOPS, 'convolutions': acct['convs'],
           'conv_activation_and_weight_gb': acct['bytes'] / 1e9, 'hbm_frac_if_streamed_once': acct['bytes'] / t_fwd / 1e9 / HBM_GBS,
           'reverse_process_20_steps_ms': t_proc * 1e3, 're_eq_32_bands_ms': t_eq * 1e3,
           'four_band_decode_rtf': a.batch * a.seconds / (4 * t_proc + t_eq)}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
