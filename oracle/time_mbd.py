"""TEST / BENCH INFRASTRUCTURE ONLY -- times the oracle's DiffusionUnet forward (oracle/mbd.py, CPU) at the released
MultiBandDiffusion geometry, for the GPU / CPU comparison quoted in DESIGN.md section 5.4.  Nothing of the product imports this.

    python -m oracle.time_mbd [--seconds 1]     -> one JSON line
"""
import argparse
import json
import time

import torch

from . import mbd as ombd


def _random_state(uc: ombd.UnetConfig, seed: int = 0) -> dict:
    """Reference-format state dict of the right shapes (PyTorch default initialisation ranges)."""
    g = torch.Generator().manual_seed(seed)
    sd, chin, hidden = {}, uc.chin, uc.hidden

    def u(*shape, fan):
        return (torch.rand(*shape, generator=g) * 2 - 1) / fan ** 0.5

    def res(prefix, ch):
        for n in ('1', '2'):
            sd[f'{prefix}.norm{n}.weight'], sd[f'{prefix}.norm{n}.bias'] = torch.ones(ch), torch.zeros(ch)
            sd[f'{prefix}.conv{n}.weight'], sd[f'{prefix}.conv{n}.bias'] = u(ch, ch, 3, fan=3 * ch), u(ch, fan=3 * ch)

    dec = []
    for d in range(uc.depth):
        sd[f'encoders.{d}.conv.weight'] = u(hidden, chin, uc.kernel, fan=chin * uc.kernel)
        sd[f'encoders.{d}.norm.weight'], sd[f'encoders.{d}.norm.bias'] = torch.ones(hidden), torch.zeros(hidden)
        for r in range(uc.res_blocks):
            res(f'encoders.{d}.res_blocks.{r}', hidden)
        dec.insert(0, (hidden, chin))
        if uc.emb_all_layers and d > 0:
            sd[f'embeddings.{d - 1}.weight'] = torch.randn(uc.num_steps, hidden, generator=g)
        chin, hidden = hidden, min(int(hidden * uc.growth), uc.max_channels)
    for i, (cin, cout) in enumerate(dec):
        for r in range(uc.res_blocks):
            res(f'decoders.{i}.res_blocks.{r}', cin)
        sd[f'decoders.{i}.norm.weight'], sd[f'decoders.{i}.norm.bias'] = torch.ones(cin), torch.zeros(cin)
        sd[f'decoders.{i}.convtr.weight'] = u(cin, cout, uc.kernel, fan=cin * uc.kernel)
    sd['embedding.weight'] = torch.randn(uc.num_steps, uc.hidden, generator=g)
    if uc.codec_dim:
        sd['conv_codec.weight'], sd['conv_codec.bias'] = u(chin, uc.codec_dim, 1, fan=uc.codec_dim), u(chin, fan=uc.codec_dim)
    return sd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=1.)
    a = ap.parse_args()
    uc = ombd.UnetConfig(chin=1, hidden=48, depth=4, growth=4., max_channels=10_000, num_steps=1000, emb_all_layers=True, bilstm=False,
                         codec_dim=128, kernel=8, stride=4, norm_groups=4, res_blocks=1)
    sd = _random_state(uc)
    T = int(a.seconds * 32000)
    x, cond = torch.randn(1, 1, T), torch.randn(1, 128, max(1, T // 640))
    with torch.no_grad():
        ombd.unet_forward(sd, uc, x[..., :3200], 500, cond[..., :5])
        t0 = time.perf_counter()
        ombd.unet_forward(sd, uc, x, 500, cond)
        dt = time.perf_counter() - t0
    print(json.dumps({'workload': f'oracle DiffusionUnet forward (hidden 48, depth 4, growth 4), {a.seconds:g} s @ 32 kHz, CPU',
                      'ms': dt * 1e3, 'threads': torch.get_num_threads()}))


if __name__ == '__main__':
    main()
