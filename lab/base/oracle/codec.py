"""Oracle (TEST INFRASTRUCTURE) -- EnCodec: SEANet encoder/decoder, LSTM, residual vector quantizer.

Functional fp32 restatement over a reference-format state dict (keys as dumped from
`audiocraft.models.encodec.EncodecModel.state_dict()`).  Citations are to /root/reference.
"""
import math
import typing as tp
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F


@dataclass
class CodecConfig:
    """Constructor arguments of SEANetEncoder/SEANetDecoder + ResidualVectorQuantizer
    (audiocraft/modules/seanet.py:85-91,183-190; audiocraft/quantization/vq.py:35-50)."""
    channels: int = 1
    dimension: int = 128
    n_filters: int = 32
    n_residual_layers: int = 3
    ratios: tp.List[int] = field(default_factory=lambda: [8, 5, 4, 2])
    kernel_size: int = 7
    last_kernel_size: int = 7
    residual_kernel_size: int = 3
    dilation_base: int = 2
    causal: bool = False
    pad_mode: str = 'reflect'
    true_skip: bool = True
    compress: int = 2
    lstm: int = 0
    norm: str = 'none'              # 'none' | 'weight_norm'
    elu_alpha: float = 1.0
    trim_right_ratio: float = 1.0
    n_q: int = 8
    bins: int = 1024
    sample_rate: int = 24000
    frame_rate: int = 75
    renormalize: bool = False


# ----------------------------------------------------------------------------- conv primitives

def fold_weight_norm(sd: dict, prefix: str) -> torch.Tensor:
    """Legacy `torch.nn.utils.weight_norm` (dim=0): w = g * v / ||v||, norm over all dims but 0
    (audiocraft/modules/conv.py:21-30).  Falls back to a plain `.weight`."""
    if prefix + '.weight_g' in sd:
        g, v = sd[prefix + '.weight_g'], sd[prefix + '.weight_v']
        norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
        return v * (g / norm)
    return sd[prefix + '.weight']


def get_extra_padding_for_conv1d(length: int, kernel_size: int, stride: int, padding_total: int) -> int:
    """audiocraft/modules/conv.py:47-53."""
    n_frames = (length - kernel_size + padding_total) / stride + 1
    ideal_length = (math.ceil(n_frames) - 1) * stride + (kernel_size - padding_total)
    return ideal_length - length


def pad1d(x: torch.Tensor, paddings: tp.Tuple[int, int], mode: str = 'constant') -> torch.Tensor:
    """audiocraft/modules/conv.py:71-88 (reflect on short inputs inserts zeros on the right first)."""
    length = x.shape[-1]
    pl, pr = paddings
    if mode == 'reflect':
        max_pad = max(pl, pr)
        extra = 0
        if length <= max_pad:
            extra = max_pad - length + 1
            x = F.pad(x, (0, extra))
        padded = F.pad(x, (pl, pr), 'reflect')
        return padded[..., :padded.shape[-1] - extra]
    return F.pad(x, (pl, pr), 'constant', 0.)


def streamable_conv1d(x, w, b, stride=1, dilation=1, causal=False, pad_mode='reflect'):
    """StreamableConv1d.forward (audiocraft/modules/conv.py:185-201)."""
    k = (w.shape[-1] - 1) * dilation + 1
    padding_total = k - stride
    extra = get_extra_padding_for_conv1d(x.shape[-1], k, stride, padding_total)
    if causal:
        x = pad1d(x, (padding_total, extra), pad_mode)
    else:
        pr = padding_total // 2
        pl = padding_total - pr
        x = pad1d(x, (pl, pr + extra), pad_mode)
    return F.conv1d(x, w, b, stride=stride, dilation=dilation)


def streamable_convtr1d(x, w, b, stride, causal=False, trim_right_ratio=1.0):
    """StreamableConvTranspose1d.forward (audiocraft/modules/conv.py:221-243)."""
    k = w.shape[-1]
    padding_total = k - stride
    y = F.conv_transpose1d(x, w, b, stride=stride)
    if causal:
        pr = math.ceil(padding_total * trim_right_ratio)
        pl = padding_total - pr
    else:
        pr = padding_total // 2
        pl = padding_total - pr
    return y[..., pl: y.shape[-1] - pr]


def elu(x, alpha=1.0):
    return F.elu(x, alpha)


def lstm_stack(x: torch.Tensor, sd: dict, prefix: str, num_layers: int, skip: bool = True,
               fast: bool = False) -> torch.Tensor:
    """StreamableLSTM.forward (audiocraft/modules/lstm.py:19-25): conv layout [B, C, T] ->
    nn.LSTM over time -> + skip.  Gate order i, f, g, o; zero initial state."""
    xs = x.permute(2, 0, 1)  # [T, B, C]
    T, B, C = xs.shape
    if fast:  # same ATen op nn.LSTM dispatches to; used only for the timed cpu_baseline leg
        flat = []
        for layer in range(num_layers):
            flat += [sd[f'{prefix}.weight_ih_l{layer}'], sd[f'{prefix}.weight_hh_l{layer}'],
                     sd[f'{prefix}.bias_ih_l{layer}'], sd[f'{prefix}.bias_hh_l{layer}']]
        H = flat[1].shape[1]
        hx = (torch.zeros(num_layers, B, H), torch.zeros(num_layers, B, H))
        y = torch._VF.lstm(xs, hx, flat, True, num_layers, 0.0, False, False, False)[0]
    else:
        y = xs
        for layer in range(num_layers):
            w_ih, w_hh = sd[f'{prefix}.weight_ih_l{layer}'], sd[f'{prefix}.weight_hh_l{layer}']
            b_ih, b_hh = sd[f'{prefix}.bias_ih_l{layer}'], sd[f'{prefix}.bias_hh_l{layer}']
            H = w_hh.shape[1]
            h = torch.zeros(B, H)
            c = torch.zeros(B, H)
            gi_all = y @ w_ih.t() + b_ih
            outs = []
            for t in range(T):
                gates = gi_all[t] + h @ w_hh.t() + b_hh
                i, f, g, o = gates.chunk(4, dim=1)
                c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
                h = torch.sigmoid(o) * torch.tanh(c)
                outs.append(h)
            y = torch.stack(outs)
    if skip:
        y = y + xs
    return y.permute(1, 2, 0)


# ----------------------------------------------------------------------------- SEANet

def _conv(sd, prefix, x, cfg: CodecConfig, stride=1, dilation=1):
    w = fold_weight_norm(sd, prefix + '.conv.conv')
    b = sd.get(prefix + '.conv.conv.bias')
    return streamable_conv1d(x, w, b, stride, dilation, cfg.causal, cfg.pad_mode)


def _resblock(sd, prefix, x, cfg: CodecConfig, dilation: int):
    """SEANetResnetBlock (audiocraft/modules/seanet.py:33-60): block = [ELU, conv k=res d, ELU, conv k=1]."""
    y = elu(x, cfg.elu_alpha)
    y = _conv(sd, prefix + '.block.1', y, cfg, dilation=dilation)
    y = elu(y, cfg.elu_alpha)
    y = _conv(sd, prefix + '.block.3', y, cfg)
    if cfg.true_skip:
        return x + y
    return _conv(sd, prefix + '.shortcut', x, cfg) + y


def seanet_encoder(sd: dict, cfg: CodecConfig, x: torch.Tensor, fast_lstm: bool = False) -> torch.Tensor:
    """SEANetEncoder (audiocraft/modules/seanet.py:111-153); `sd` keys are `encoder.model.{i}...`."""
    p = 'encoder.model.'
    i = 0
    x = _conv(sd, f'{p}{i}', x, cfg)
    i += 1
    for ratio in reversed(cfg.ratios):
        for j in range(cfg.n_residual_layers):
            x = _resblock(sd, f'{p}{i}', x, cfg, cfg.dilation_base ** j)
            i += 1
        x = elu(x, cfg.elu_alpha)
        i += 1
        x = _conv(sd, f'{p}{i}', x, cfg, stride=ratio)
        i += 1
    if cfg.lstm:
        x = lstm_stack(x, sd, f'{p}{i}.lstm', cfg.lstm, fast=fast_lstm)
        i += 1
    x = elu(x, cfg.elu_alpha)
    i += 1
    return _conv(sd, f'{p}{i}', x, cfg)


def seanet_decoder(sd: dict, cfg: CodecConfig, z: torch.Tensor, fast_lstm: bool = False) -> torch.Tensor:
    """SEANetDecoder (audiocraft/modules/seanet.py:205-258); keys `decoder.model.{i}...`."""
    p = 'decoder.model.'
    i = 0
    x = _conv(sd, f'{p}{i}', z, cfg)
    i += 1
    if cfg.lstm:
        x = lstm_stack(x, sd, f'{p}{i}.lstm', cfg.lstm, fast=fast_lstm)
        i += 1
    for ratio in cfg.ratios:
        x = elu(x, cfg.elu_alpha)
        i += 1
        w = fold_weight_norm(sd, f'{p}{i}.convtr.convtr')
        b = sd.get(f'{p}{i}.convtr.convtr.bias')
        x = streamable_convtr1d(x, w, b, ratio, cfg.causal, cfg.trim_right_ratio)
        i += 1
        for j in range(cfg.n_residual_layers):
            x = _resblock(sd, f'{p}{i}', x, cfg, cfg.dilation_base ** j)
            i += 1
    x = elu(x, cfg.elu_alpha)
    i += 1
    return _conv(sd, f'{p}{i}', x, cfg)


# ----------------------------------------------------------------------------- RVQ

def codebooks_from_state(sd: dict, n_q: int) -> torch.Tensor:
    return torch.stack([sd[f'quantizer.vq.layers.{q}._codebook.embed'] for q in range(n_q)])


def euclidean_quantize(x: torch.Tensor, embed: torch.Tensor) -> torch.Tensor:
    """EuclideanCodebook.quantize (audiocraft/quantization/core_vq.py:164-172), x: [N, D], embed: [bins, D]."""
    e = embed.t()
    dist = -(x.pow(2).sum(1, keepdim=True) - 2 * x @ e + e.pow(2).sum(0, keepdim=True))
    return dist.max(dim=-1).indices


def rvq_encode(latents: torch.Tensor, codebooks: torch.Tensor) -> torch.Tensor:
    """ResidualVectorQuantizer.encode (vq.py:87-96) -> ResidualVectorQuantization.encode
    (core_vq.py:386-396).  latents [B, D, T] fp32, codebooks [K, bins, D] -> codes [B, K, T] int64."""
    B, D, T = latents.shape
    residual = latents.permute(0, 2, 1).reshape(B * T, D)
    out = []
    for q in range(codebooks.shape[0]):
        idx = euclidean_quantize(residual, codebooks[q])
        residual = residual - F.embedding(idx, codebooks[q])
        out.append(idx.view(B, T))
    return torch.stack(out, dim=1)


def rvq_decode(codes: torch.Tensor, codebooks: torch.Tensor) -> torch.Tensor:
    """ResidualVectorQuantizer.decode (vq.py:98-103) -> core_vq.py:398-404, 305-310: sum of
    embeddings, in level order, then b n d -> b d n.  codes [B, K, T] -> [B, D, T]."""
    out = torch.tensor(0.0)
    for q in range(codes.shape[1]):
        out = out + F.embedding(codes[:, q], codebooks[q])
    return out.permute(0, 2, 1)


# ----------------------------------------------------------------------------- EncodecModel

def encodec_encode(sd: dict, cfg: CodecConfig, wav: torch.Tensor, fast_lstm: bool = False):
    """EncodecModel.encode (audiocraft/models/encodec.py:223-238) -> (codes, scale)."""
    assert wav.dim() == 3
    scale = None
    if cfg.renormalize:  # encodec.py:186-196
        mono = wav.mean(dim=1, keepdim=True)
        volume = mono.pow(2).mean(dim=2, keepdim=True).sqrt()
        scale = 1e-8 + volume
        wav = wav / scale
        scale = scale.view(-1, 1)
    emb = seanet_encoder(sd, cfg, wav, fast_lstm)
    return rvq_encode(emb, codebooks_from_state(sd, cfg.n_q)), scale


def encodec_decode(sd: dict, cfg: CodecConfig, codes: torch.Tensor, scale=None, fast_lstm: bool = False):
    """EncodecModel.decode (audiocraft/models/encodec.py:240-259)."""
    emb = rvq_decode(codes, codebooks_from_state(sd, codes.shape[1]))
    out = seanet_decoder(sd, cfg, emb, fast_lstm)
    if scale is not None:
        out = out * scale.view(-1, 1, 1)
    return out
