"""Diffusion U-Net of MultiBandDiffusion on MI355X -- host side.

API and parameter names mirror `audiocraft/models/unet.py:23-214` (`DiffusionUnet`, `EncoderLayer`, `DecoderLayer`, `ResBlock`,
`BLSTM`, `Output`) so the reference's diffusion checkpoints load with `load_state_dict`; the torch modules below are
PARAMETER CONTAINERS only -- every operation of `forward` runs in libacmi:

 * Conv1d / ConvTranspose1d           acmi_conv1d (implicit-GEMM MFMA kernel; the transposed convolutions as their polyphase
                                      decomposition, like the SEANet decoder's)
 * GroupNorm + the ReLU behind it     folded into the following convolution's input pack (acmi_conv1d_gn: norm -> ReLU -> conv of a
                                      ResBlock, the decoder layers' norm -> ReLU -> transposed conv); acmi_group_norm (one launch
                                      pair, ReLU fused) where the normalised tensor itself is needed (the encoder layers' output)
 * residual adds                      fused into the second convolution of a ResBlock (its `residual` operand)
 * step embeddings, skip adds, the interpolated codec condition       acmi_channel_add / acmi_add_cropped / acmi_interp_add
 * BiLSTM bottleneck                  input projections as k = 1 convolutions, recurrences on acmi_lstm_layer (the reverse
                                      direction on the time-flipped sequence)

`transformer=True` (a non-causal StreamingTransformer in the bottleneck, optionally cross-attending to the condition) is not
implemented: no configuration in the reference tree enables it (config/model/score/basic.yaml).
"""
import typing as tp
from dataclasses import dataclass

import torch
from torch import nn

from .. import _C


@dataclass
class Output:
    sample: torch.Tensor


class _Tiles(dict):
    """Per-module cache: conv name -> (stamp of the raw weights, weights in the layout acmi_conv1d stages)."""

    def get_tiled(self, name: str, d, w: torch.Tensor) -> torch.Tensor:
        stamp = (w.data_ptr(), w._version, d.Cout, d.Cin, d.ksize, d.stride, d.dilation, d.shuffle)
        hit = self.get(name)
        if hit is None or hit[0] != stamp:
            hit = self[name] = (stamp, _C.conv1d_tile_weights(d, w))
        return hit[1]


def _launch_conv(d, x, wt, b, residual, y, gn: tp.Optional[tp.Tuple[nn.GroupNorm, int]]):
    """gn = (GroupNorm module, groups): the convolution runs on relu(GroupNorm(x)) -- the normalisation is applied by the
    convolution's own input pack (acmi_conv1d_gn) instead of being materialised by a launch of its own."""
    if gn is None:
        _C.conv1d_tiled(d, x, wt, b, residual, y)
    else:
        norm, groups = gn
        _C.conv1d_tiled_gn(d, x, wt, b, residual, y, norm.weight, norm.bias, groups, norm.eps, True)


def _conv1d(tiles: _Tiles, name: str, x: torch.Tensor, w: torch.Tensor, b: tp.Optional[torch.Tensor], stride: int = 1,
            padding: int = 0, dilation: int = 1, residual: tp.Optional[torch.Tensor] = None, right_pad: int = 0,
            gn: tp.Optional[tp.Tuple[nn.GroupNorm, int]] = None) -> torch.Tensor:
    """nn.Conv1d(padding=padding, zero padding; `right_pad` more zeros on the right: F.pad before the conv) on x [B, Cin, T]."""
    B, Cin, T = x.shape
    Cout, _, k = w.shape
    Tout = (T + right_pad + 2 * padding - dilation * (k - 1) - 1) // stride + 1
    d = _C.ConvDesc()
    d.B, d.Cin, d.Tin, d.Cout, d.Tout = B, Cin, T, Cout, Tout
    d.ksize, d.stride, d.dilation, d.pad_left = k, stride, dilation, padding
    d.pad_mode, d.reflect_len, d.elu_in, d.elu_alpha, d.shuffle, d.trim_left = _C.PAD_ZERO, T, 0, 0.0, 1, 0
    y = torch.empty(B, Cout, Tout, device=x.device, dtype=torch.float32)
    _launch_conv(d, x, tiles.get_tiled(name, d, w), b, residual, y, gn)
    return y


def _polyphase(w: torch.Tensor, stride: int) -> tp.Tuple[torch.Tensor, int]:
    """nn.ConvTranspose1d weight [Cin, Cout, k] -> rows (co, r), taps over x[q + j' - (ntaps - 1)] (modules/seanet.py)."""
    cin, cout, k = w.shape
    ntaps = -(-k // stride)
    wp = torch.nn.functional.pad(w, (0, ntaps * stride - k)).reshape(cin, cout, ntaps, stride)
    return wp.permute(1, 3, 0, 2).flip(-1).reshape(cout * stride, cin, ntaps).contiguous(), ntaps


def _convtr1d(tiles: _Tiles, name: str, x: torch.Tensor, wq: torch.Tensor, ntaps: int, k: int, stride: int, padding: int,
              gn: tp.Optional[tp.Tuple[nn.GroupNorm, int]] = None) -> torch.Tensor:
    """nn.ConvTranspose1d(k, stride, padding, bias=False): (T - 1) s + k - 2 padding output samples."""
    B, Cin, T = x.shape
    cout = wq.shape[0] // stride
    Tout = (T - 1) * stride + k - 2 * padding
    d = _C.ConvDesc()
    d.B, d.Cin, d.Tin, d.Cout, d.Tout = B, Cin, T, cout * stride, Tout
    d.ksize, d.stride, d.dilation, d.pad_left = ntaps, 1, 1, ntaps - 1
    d.pad_mode, d.reflect_len, d.elu_in, d.elu_alpha = _C.PAD_ZERO, T, 0, 0.0
    d.shuffle, d.trim_left = stride, padding
    y = torch.empty(B, cout, Tout, device=x.device, dtype=torch.float32)
    _launch_conv(d, x, tiles.get_tiled(name, d, wq), None, None, y, gn)
    return y


class ResBlock(nn.Module):
    """unet.py:32-53: x + conv2(relu(norm2(conv1(relu(norm1(x))))))"""

    def __init__(self, channels: int, kernel: int = 3, norm_groups: int = 4, dilation: int = 1, dropout: float = 0., device=None):
        super().__init__()
        self._tiles = _Tiles()
        padding = dilation * (kernel - 1) // 2
        self.norm_groups, self.dilation, self.padding = norm_groups, dilation, padding
        self.norm1 = nn.GroupNorm(norm_groups, channels, device=device)
        self.conv1 = nn.Conv1d(channels, channels, kernel, 1, padding, dilation=dilation, device=device)
        self.norm2 = nn.GroupNorm(norm_groups, channels, device=device)
        self.conv2 = nn.Conv1d(channels, channels, kernel, 1, padding, dilation=dilation, device=device)

    def run(self, x: torch.Tensor) -> torch.Tensor:
        # norm -> ReLU -> conv twice: both normalisations ride in their convolution's input pack (round 4; before: a
        # statistics + an apply launch and a read + write of the activation per normalisation)
        h = _conv1d(self._tiles, 'conv1', x, self.conv1.weight, self.conv1.bias, 1, self.padding, self.dilation,
                    gn=(self.norm1, self.norm_groups))
        return _conv1d(self._tiles, 'conv2', h, self.conv2.weight, self.conv2.bias, 1, self.padding, self.dilation, residual=x,
                       gn=(self.norm2, self.norm_groups))


class EncoderLayer(nn.Module):
    """unet.py:80-104"""

    def __init__(self, chin: int, chout: int, kernel: int = 4, stride: int = 2, norm_groups: int = 4, res_blocks: int = 1,
                 dropout: float = 0., device=None):
        super().__init__()
        self._tiles = _Tiles()
        self.kernel, self.stride, self.norm_groups = kernel, stride, norm_groups
        self.conv = nn.Conv1d(chin, chout, kernel, stride, (kernel - stride) // 2, bias=False, device=device)
        self.norm = nn.GroupNorm(norm_groups, chout, device=device)
        self.res_blocks = nn.Sequential(*[ResBlock(chout, norm_groups=norm_groups, dilation=2 ** idx, dropout=dropout, device=device)
                                          for idx in range(res_blocks)])

    def run(self, x: torch.Tensor) -> torch.Tensor:
        T = x.shape[-1]
        pad = (self.stride - (T % self.stride)) % self.stride
        z = _conv1d(self._tiles, 'conv', x, self.conv.weight, None, self.stride, (self.kernel - self.stride) // 2, right_pad=pad)
        z = _C.group_norm(z, self.norm.weight, self.norm.bias, self.norm_groups, self.norm.eps, relu=True, out=z)
        for rb in self.res_blocks:
            z = rb.run(z)
        return z


class DecoderLayer(nn.Module):
    """unet.py:56-77"""

    def __init__(self, chin: int, chout: int, kernel: int = 4, stride: int = 2, norm_groups: int = 4, res_blocks: int = 1,
                 dropout: float = 0., device=None):
        super().__init__()
        self._tiles = _Tiles()
        self.kernel, self.stride, self.norm_groups = kernel, stride, norm_groups
        self.res_blocks = nn.Sequential(*[ResBlock(chin, norm_groups=norm_groups, dilation=2 ** idx, dropout=dropout, device=device)
                                          for idx in range(res_blocks)])
        self.norm = nn.GroupNorm(norm_groups, chin, device=device)
        self.convtr = nn.ConvTranspose1d(chin, chout, kernel, stride, (kernel - stride) // 2, bias=False, device=device)
        self._prep = None

    def run(self, x: torch.Tensor) -> torch.Tensor:
        for rb in self.res_blocks:
            x = rb.run(x)
        stamp = (self.convtr.weight.data_ptr(), self.convtr.weight._version)
        if self._prep is None or self._prep[2] != stamp:
            wq, ntaps = _polyphase(self.convtr.weight.detach().float(), self.stride)
            self._prep = (wq, ntaps, stamp)
        return _convtr1d(self._tiles, 'convtr', x, self._prep[0], self._prep[1], self.kernel, self.stride, (self.kernel - self.stride) // 2,
                         gn=(self.norm, self.norm_groups))


class BLSTM(nn.Module):
    """unet.py:106-120: nn.LSTM(bidirectional, 2 layers, hidden = dim) + Linear(2 dim -> dim)"""

    def __init__(self, dim: int, layers: int = 2, device=None):
        super().__init__()
        self._tiles = _Tiles()
        self.dim, self.layers = dim, layers
        self.lstm = nn.LSTM(bidirectional=True, num_layers=layers, hidden_size=dim, input_size=dim, device=device)
        self.linear = nn.Linear(2 * dim, dim, device=device)

    def _direction(self, x: torch.Tensor, layer: int, suffix: str) -> torch.Tensor:
        """one direction of one layer on x [B, I, T] -> [B, H, T]"""
        B, _, T = x.shape
        H = self.dim
        p = lambda n: getattr(self.lstm, f'{n}_l{layer}{suffix}').detach().float()   # noqa: E731
        gates = _conv1d(self._tiles, f'ih{layer}{suffix}', x, p('weight_ih').unsqueeze(-1).contiguous(), (p('bias_ih') + p('bias_hh')).contiguous())
        out = torch.empty(B, H, T, device=x.device, dtype=torch.float32)
        work = torch.zeros(_C.lstm_work_floats(B, H), device=x.device, dtype=torch.float32)
        _C.lstm_layer(gates, p('weight_hh').contiguous(), None, out, work, B, H, T)
        _C.lstm_check(work[5 * B * H:], 'acmi_lstm_layer')
        return out

    def run(self, x: torch.Tensor) -> torch.Tensor:
        y = x
        for layer in range(self.layers):
            fw = self._direction(y, layer, '')
            bw = self._direction(y.flip(-1).contiguous(), layer, '_reverse').flip(-1)   # time reversal: data movement only
            y = torch.cat([fw, bw], dim=1)
        return _conv1d(self._tiles, 'linear', y, self.linear.weight.detach().float().unsqueeze(-1).contiguous(), self.linear.bias.detach().float())


class DiffusionUnet(nn.Module):
    """unet.py:123-214.  forward(x [B, chin, T], step (int or LongTensor [B]), condition [B, codec_dim, Tc]) -> Output(sample)."""

    def __init__(self, chin: int = 3, hidden: int = 24, depth: int = 3, growth: float = 2., max_channels: int = 10_000,
                 num_steps: int = 1000, emb_all_layers=False, cross_attention: bool = False, bilstm: bool = False,
                 transformer: bool = False, codec_dim: tp.Optional[int] = None, device=None, **kwargs):
        super().__init__()
        self._tiles = _Tiles()
        if transformer or cross_attention:
            raise NotImplementedError("DiffusionUnet(transformer=True): the non-causal transformer bottleneck is not implemented "
                                      "(no configuration of the reference enables it, config/model/score/basic.yaml)")
        kwargs.pop('activation', None)
        self.encoders = nn.ModuleList()
        self.decoders = nn.ModuleList()
        self.embeddings: tp.Optional[nn.ModuleList] = nn.ModuleList() if emb_all_layers else None
        self.embedding = nn.Embedding(num_steps, hidden, device=device)
        for d in range(depth):
            self.encoders.append(EncoderLayer(chin, hidden, device=device, **kwargs))
            self.decoders.insert(0, DecoderLayer(hidden, chin, device=device, **kwargs))
            if emb_all_layers and d > 0:
                self.embeddings.append(nn.Embedding(num_steps, hidden, device=device))
            chin = hidden
            hidden = min(int(chin * growth), max_channels)
        self.bilstm: tp.Optional[BLSTM] = BLSTM(chin, device=device) if bilstm else None
        self.use_transformer = False
        self.cross_attention = False
        self.use_codec = codec_dim is not None
        if self.use_codec:
            self.conv_codec = nn.Conv1d(codec_dim, chin, 1, device=device)
        self.eval()

    @torch.no_grad()
    def forward(self, x: torch.Tensor, step: tp.Union[int, torch.Tensor], condition: tp.Optional[torch.Tensor] = None) -> Output:
        if not x.is_cuda:
            raise RuntimeError("audiocraft_amd.DiffusionUnet runs on an MI355X only (no CPU fallback); move it to 'cuda'")
        x = x.float().contiguous()
        bs = x.size(0)
        if isinstance(step, torch.Tensor):
            steps = step.to(device=x.device, dtype=torch.long).contiguous()
        else:
            steps = torch.full((bs,), int(step), device=x.device, dtype=torch.long)
        skips = []
        z = x
        for idx, encoder in enumerate(self.encoders):
            z = encoder.run(z)
            if idx == 0:
                _C.channel_add(z, self.embedding.weight.detach().float(), steps)
            elif self.embeddings is not None:
                _C.channel_add(z, self.embeddings[idx - 1].weight.detach().float(), steps)
            skips.append(z)
        if self.use_codec:
            assert condition is not None, "Model defined for conditionnal generation"
            ce = _conv1d(self._tiles, 'conv_codec', condition.float().contiguous(), self.conv_codec.weight.detach().float(), self.conv_codec.bias.detach().float())
            assert ce.size(-1) <= 2 * z.size(-1), \
                f"You are downsampling the conditionning with factor >=2 : {ce.size(-1)=} and {z.size(-1)=}"
            # the reference adds IN PLACE (`z += condition_emb`, unet.py:193) to the tensor it has just pushed onto `skips`:
            # the condition also enters through the deepest skip connection (oracle/mbd.py)
            _C.interp_add(z, ce)
        if self.bilstm is not None:
            z = self.bilstm.run(z)
        else:
            z = torch.zeros_like(z)
        for decoder in self.decoders:
            s = skips.pop(-1)
            z = _C.add_cropped(z, s)
            z = decoder.run(z)
        return Output(z[:, :, :x.shape[2]].contiguous())
