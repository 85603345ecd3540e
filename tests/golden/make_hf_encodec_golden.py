"""Golden fixture for the HuggingFace EnCodec wrapper (reference audiocraft/models/encodec.py:323-394: `facebook/encodec_24khz`
reaches the reference through `transformers.EncodecModel`).  Third-party arithmetic, so the fixture comes from the third
party itself: `transformers` (the version installed in the build container, printed into the fixture) runs a seeded,
randomly initialised `EncodecModel` of a small causal geometry (reflect padding, conv shortcuts, 2 LSTM layers, weight norm:
the 24 kHz model's structure) and of the 24 kHz model's own configuration on short inputs; the fixture stores the HF-format
state dict (small geometry only), the inputs and HF's latents / codes / decoded audio.

  hf_encodec_small.npz   config + state dict + outputs (committed: the wrapper's tests need no transformers at run time)

Run in the build container only:   python tests/golden/make_hf_encodec_golden.py
"""
import json
import os

import numpy as np
import torch
import transformers
from transformers import EncodecConfig, EncodecModel

HERE = os.path.dirname(os.path.abspath(__file__))

SMALL = dict(audio_channels=1, sampling_rate=2400, target_bandwidths=[0.6, 1.2, 2.4], hidden_size=16, codebook_dim=16,
             num_filters=4, num_residual_layers=1, upsampling_ratios=[4, 3, 2], codebook_size=64, kernel_size=7,
             last_kernel_size=7, residual_kernel_size=3, dilation_growth_rate=2, use_causal_conv=True, pad_mode='reflect',
             compress=2, num_lstm_layers=2, trim_right_ratio=1.0, use_conv_shortcut=True, norm_type='weight_norm',
             normalize=False, chunk_length_s=None, overlap=None)


def seeded_model(cfg_kwargs, seed):
    torch.manual_seed(seed)
    model = EncodecModel(EncodecConfig(**cfg_kwargs)).eval()
    with torch.no_grad():   # HF initialises the codebooks with zeros (they come from the checkpoint): make them count
        for layer in model.quantizer.layers:
            layer.codebook.embed.copy_(torch.randn_like(layer.codebook.embed) * 0.5)
        for k, p in model.named_parameters():
            if k.endswith('original0'):                     # weight-norm gains: a constant would hide a g / v mix-up
                p.mul_(1.0 + 0.25 * torch.rand_like(p))
            if k.endswith('.bias'):
                p.add_(0.05 * torch.randn_like(p))
    return model


def run(model, wav, bandwidths):
    out = {}
    with torch.no_grad():
        out['latents'] = model.encoder(wav)
        for bw in bandwidths:
            enc = model.encode(wav, None, bw)
            codes = enc[0][0]
            assert enc[1][0] is None
            out[f'codes_bw{bw}'] = codes
            out[f'decoded_bw{bw}'] = model.decode(codes[None], [None])[0]
        out['quantized'] = model.quantizer.decode(codes.transpose(0, 1))
    return out


if __name__ == '__main__':
    model = seeded_model(SMALL, 7)
    wav = 0.5 * torch.randn(2, 1, 517, generator=torch.Generator().manual_seed(8))
    arrays = run(model, wav, SMALL['target_bandwidths'])
    arrays['wav'] = wav
    out = {'cfg': np.array(json.dumps(dict(SMALL, transformers_version=transformers.__version__,
                                           num_quantizers=model.config.num_quantizers)))}
    for k, v in model.state_dict().items():
        out['sd/' + k] = v.detach().cpu().numpy()
    for k, v in arrays.items():
        out[k] = v.detach().cpu().numpy()
    path = os.path.join(HERE, 'hf_encodec_small.npz')
    np.savez_compressed(path, **out)
    print(f'wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB; codes {tuple(arrays["codes_bw2.4"].shape)}; '
          f'num_quantizers {model.config.num_quantizers}; transformers {transformers.__version__}')
