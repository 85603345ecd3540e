"""SEANet encoder / decoder on MI355X -- host side.

Module tree and parameter names mirror the reference (`audiocraft/modules/seanet.py:16-258`,
`audiocraft/modules/conv.py:98-243`, `audiocraft/modules/lstm.py:10-25`) so EnCodec checkpoints
load unchanged (`encoder.model.{i}.conv.conv.weight_g`, `...block.1.conv.conv.weight_v`,
`...convtr.convtr.weight_g`, `...lstm.weight_ih_l0`, ...).  Execution is different:

 * weight-norm is folded once per load (the reference recomputes g * v / ||v|| on every call);
 * every nn.ELU is fused into the load of the convolution that follows it, every resnet skip add
   and every bias into the convolution's store;
 * padding (asymmetric / causal / reflect / "extra") and the transposed-conv trim are index math in
   the kernel -- no padded copies, no full-length transposed-conv output that is then cut;
 * ConvTranspose1d runs as its polyphase decomposition through the same implicit-GEMM kernel;
 * the LSTM input projections are hoisted out of the time loop (one GEMM per layer).
"""
import math
import typing as tp

import torch
from torch import nn

from .. import _C


def get_extra_padding_for_conv1d(length: int, kernel_size: int, stride: int, padding_total: int = 0) -> int:
    """Right padding that makes the last window full (reference conv.py:47-53)."""
    n_frames = (length - kernel_size + padding_total) / stride + 1
    ideal_length = (math.ceil(n_frames) - 1) * stride + (kernel_size - padding_total)
    return ideal_length - length


class _ConvW(nn.Module):
    """Parameters of a (weight-normalised) nn.Conv1d / nn.ConvTranspose1d."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, norm: str, transposed: bool,
                 bias: bool = True, device=None):
        super().__init__()
        if norm not in ('none', 'weight_norm'):
            raise NotImplementedError(f"conv normalisation '{norm}' is not used by the MusicGen codecs")
        shape = (in_channels, out_channels, kernel_size) if transposed else (out_channels, in_channels, kernel_size)
        self.transposed = transposed
        self.kernel_size = (kernel_size,)
        bound = 1 / math.sqrt((out_channels if transposed else in_channels) * kernel_size)
        w = torch.empty(*shape, device=device).uniform_(-bound, bound)
        if norm == 'weight_norm':
            self.weight_g = nn.Parameter(w.reshape(shape[0], -1).norm(dim=1).reshape(shape[0], 1, 1))
            self.weight_v = nn.Parameter(w)
        else:
            self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.empty(out_channels, device=device).uniform_(-bound, bound)) if bias else None

    def folded(self) -> torch.Tensor:
        """Legacy weight_norm(dim=0) fold: g * v / ||v|| over all dims but 0 (reference conv.py:21-30)."""
        if hasattr(self, 'weight_g'):
            v = self.weight_v.detach().float()
            n = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, 1, 1)
            return v * (self.weight_g.detach().float() / n)
        return self.weight.detach().float()


class _Norm(nn.Module):
    def __init__(self, attr: str, conv: _ConvW):
        super().__init__()
        setattr(self, attr, conv)


class StreamableConv1d(nn.Module):
    """reference conv.py:165-201"""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1, dilation: int = 1,
                 groups: int = 1, bias: bool = True, causal: bool = False, norm: str = 'none',
                 norm_kwargs: tp.Dict[str, tp.Any] = {}, pad_mode: str = 'reflect', device=None):
        super().__init__()
        assert groups == 1, "grouped convolutions are not on the EnCodec path"
        assert pad_mode in ('constant', 'reflect'), pad_mode
        self.conv = _Norm('conv', _ConvW(in_channels, out_channels, kernel_size, norm, False, bias, device))
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.dilation = kernel_size, stride, dilation
        self.causal, self.pad_mode = causal, pad_mode
        self._prep: tp.Optional[tp.Tuple[torch.Tensor, tp.Optional[torch.Tensor]]] = None

    def _weights(self):
        if self._prep is None:
            c = self.conv.conv
            self._prep = (c.folded().contiguous(), None if c.bias is None else c.bias.detach().float().contiguous())
        return self._prep

    def out_length(self, T: int) -> int:
        k = (self.kernel_size - 1) * self.dilation + 1
        pt = k - self.stride
        return (T + pt + get_extra_padding_for_conv1d(T, k, self.stride, pt) - k) // self.stride + 1

    def run(self, x: torch.Tensor, elu_alpha: tp.Optional[float] = None,
            residual: tp.Optional[torch.Tensor] = None) -> torch.Tensor:
        """y = conv(pad(ELU?(x))) + bias (+ residual); x [B, Cin, T] f32 on device."""
        B, Cin, T = x.shape
        assert Cin == self.in_channels
        w, b = self._weights()[:2]
        k = (self.kernel_size - 1) * self.dilation + 1
        padding_total = k - self.stride
        extra = get_extra_padding_for_conv1d(T, k, self.stride, padding_total)
        if self.causal:
            pl, pr = padding_total, extra
        else:
            pr0 = padding_total // 2
            pl, pr = padding_total - pr0, pr0 + extra
        Tout = (T + pl + pr - k) // self.stride + 1
        d = _C.ConvDesc()
        d.B, d.Cin, d.Tin, d.Cout, d.Tout = B, Cin, T, self.out_channels, Tout
        d.ksize, d.stride, d.dilation, d.pad_left = self.kernel_size, self.stride, self.dilation, pl
        if self.pad_mode == 'reflect':
            d.pad_mode = _C.PAD_REFLECT
            max_pad = max(pl, pr)
            d.reflect_len = T if T > max_pad else max_pad + 1  # short-input rule of pad1d (conv.py:79-86)
        else:
            d.pad_mode, d.reflect_len = _C.PAD_ZERO, T
        d.elu_in, d.elu_alpha = int(elu_alpha is not None), float(elu_alpha or 0.0)
        d.shuffle, d.trim_left = 1, 0
        y = torch.empty(B, self.out_channels, Tout, device=x.device, dtype=torch.float32)
        _C.conv1d_tiled(d, x, self._tiled(d, w), b, residual, y)
        return y

    def _tiled(self, d, w: torch.Tensor) -> torch.Tensor:
        """The weights in the layout acmi_conv1d stages (made once; `invalidate_prepared` drops it with `_prep`)."""
        if len(self._prep) == 2:
            self._prep = self._prep + (_C.conv1d_tile_weights(d, w),)
        return self._prep[2]


class StreamableConvTranspose1d(nn.Module):
    """reference conv.py:204-243"""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1, causal: bool = False,
                 norm: str = 'none', trim_right_ratio: float = 1., norm_kwargs: tp.Dict[str, tp.Any] = {},
                 device=None):
        super().__init__()
        self.convtr = _Norm('convtr', _ConvW(in_channels, out_channels, kernel_size, norm, True, True, device))
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = kernel_size, stride
        self.causal, self.trim_right_ratio = causal, trim_right_ratio
        assert self.causal or self.trim_right_ratio == 1., \
            "`trim_right_ratio` != 1.0 only makes sense for causal convolutions"
        assert 0. <= self.trim_right_ratio <= 1.
        self._prep = None

    def _weights(self):
        """Polyphase rearrangement: rows (co, r), taps j' over x[q + j' - (ntaps-1)]:
        W'[co*s + r, ci, j'] = w[ci, co, r + (ntaps-1-j')*s]  (0 beyond the kernel)."""
        if self._prep is None:
            c = self.convtr.convtr
            w = c.folded()  # [Cin, Cout, k]
            s, k = self.stride, self.kernel_size
            ntaps = -(-k // s)
            wp = torch.nn.functional.pad(w, (0, ntaps * s - k)).reshape(self.in_channels, self.out_channels, ntaps, s)
            wq = wp.permute(1, 3, 0, 2).flip(-1).reshape(self.out_channels * s, self.in_channels, ntaps)
            self._prep = (wq.contiguous(), c.bias.detach().float().contiguous(), ntaps)
        return self._prep

    def run(self, x: torch.Tensor, elu_alpha: tp.Optional[float] = None) -> torch.Tensor:
        B, Cin, T = x.shape
        w, b, ntaps = self._weights()[:3]
        s, k = self.stride, self.kernel_size
        padding_total = k - s
        if self.causal:
            pr = math.ceil(padding_total * self.trim_right_ratio)
        else:
            pr = padding_total // 2
        pl = padding_total - pr
        Tout = (T - 1) * s + k - pl - pr
        d = _C.ConvDesc()
        d.B, d.Cin, d.Tin, d.Cout, d.Tout = B, Cin, T, self.out_channels * s, Tout
        d.ksize, d.stride, d.dilation, d.pad_left = ntaps, 1, 1, ntaps - 1
        d.pad_mode, d.reflect_len = _C.PAD_ZERO, T
        d.elu_in, d.elu_alpha = int(elu_alpha is not None), float(elu_alpha or 0.0)
        d.shuffle, d.trim_left = s, pl
        y = torch.empty(B, self.out_channels, Tout, device=x.device, dtype=torch.float32)
        if len(self._prep) == 3:   # the staged layout of the polyphase weights, once
            self._prep = self._prep + (_C.conv1d_tile_weights(d, w),)
        _C.conv1d_tiled(d, x, self._prep[3], b, None, y)
        return y


class _LSTMParams(nn.Module):
    def __init__(self, dimension: int, num_layers: int, device=None):
        super().__init__()
        bound = 1 / math.sqrt(dimension)
        for layer in range(num_layers):
            for name, shape in (('weight_ih', (4 * dimension, dimension)), ('weight_hh', (4 * dimension, dimension)),
                                ('bias_ih', (4 * dimension,)), ('bias_hh', (4 * dimension,))):
                setattr(self, f'{name}_l{layer}',
                        nn.Parameter(torch.empty(*shape, device=device).uniform_(-bound, bound)))


class StreamableLSTM(nn.Module):
    """reference lstm.py:10-25: nn.LSTM(dim, dim, num_layers) over time + skip, conv layout in/out."""

    def __init__(self, dimension: int, num_layers: int = 2, skip: bool = True, device=None):
        super().__init__()
        self.skip, self.dimension, self.num_layers = skip, dimension, num_layers
        self.lstm = _LSTMParams(dimension, num_layers, device)
        self._prep = None

    def _run_once(self, x: torch.Tensor):
        """-> (y, err word of the per-layer recurrence kernels or None)"""
        B, C, T = x.shape
        H = self.dimension
        if self._prep is None:
            self._prep = []
            for layer in range(self.num_layers):
                p = self.lstm
                w_ih = getattr(p, f'weight_ih_l{layer}').detach().float().reshape(4 * H, H, 1).contiguous()
                w_hh = getattr(p, f'weight_hh_l{layer}').detach().float().contiguous()
                bias = (getattr(p, f'bias_ih_l{layer}').detach().float()
                        + getattr(p, f'bias_hh_l{layer}').detach().float()).contiguous()
                self._prep.append([w_ih, w_hh, bias, None])
        nwork = _C.lstm_layer_work_floats(B, H, T)    # legacy area (+ the per-step exchange array of the XCD-local form)
        work = torch.empty(nwork, device=x.device, dtype=torch.float32)
        work[:5 * B * H + 4].zero_()
        d = _C.ConvDesc()
        d.B, d.Cin, d.Tin, d.Cout, d.Tout = B, H, T, 4 * H, T
        d.ksize, d.stride, d.dilation, d.pad_left = 1, 1, 1, 0
        d.pad_mode, d.reflect_len, d.elu_in, d.elu_alpha, d.shuffle, d.trim_left = _C.PAD_ZERO, T, 0, 0.0, 1, 0
        if self.num_layers == 2 and T > 0 and _C.lstm_stack2_supported(B, H, T):
            # both layers in one launch, layer 1 a step behind layer 0 (T + 1 dependent steps instead of 2 T)
            p0, p1 = self._prep
            if p0[3] is None:
                p0[3] = _C.conv1d_tile_weights(d, p0[0])
            gates = torch.empty(B, 4 * H, T, device=x.device, dtype=torch.float32)
            _C.conv1d_tiled(d, x, p0[3], p0[2], None, gates)
            out = torch.empty(B, H, T, device=x.device, dtype=torch.float32)
            _C.lstm_stack2(gates, p0[1], p1[0], p1[1], p1[2], x if self.skip else None, out, B, H, T)
            return out, None
        y = x
        for layer, prep in enumerate(self._prep):
            w_ih, w_hh, bias = prep[:3]
            if prep[3] is None:
                prep[3] = _C.conv1d_tile_weights(d, w_ih)
            gates = torch.empty(B, 4 * H, T, device=x.device, dtype=torch.float32)
            _C.conv1d_tiled(d, y, prep[3], bias, None, gates)  # input projection for all T at once
            out = torch.empty(B, H, T, device=x.device, dtype=torch.float32)
            last = layer == self.num_layers - 1
            _C.lstm_layer(gates, w_hh, x if (self.skip and last) else None, out, work, B, H, T)
            y = out
        # the persistent recurrence kernel counts bounded-spin give-ups of its all-gather in the last words of `work`
        # (never seen on an otherwise idle device; a non-zero count means the result is not to be trusted)
        return y, work[5 * B * H:5 * B * H + 4]

    def run(self, x: torch.Tensor) -> torch.Tensor:
        y, err = self._run_once(x)
        if err is None:
            return y
        deferred = getattr(_C._lstm_tls, 'sink', None) is not None   # inside a capture: the owner checks after each replay
        if not deferred and x.is_cuda and torch.cuda.is_current_stream_capturing():
            # somebody else's capture (EncodecModel._seanet runs plain launches there) and no sink to defer to: reading the
            # give-up word would synchronise and invalidate that capture -- refuse with the remedy instead
            raise _C.AcmiError("StreamableLSTM inside a foreign stream capture: the persistent kernel's give-up word cannot be read "
                               "there; wrap the capture in audiocraft_amd._C.defer_lstm_checks(list) and check the list after replays")
        if not deferred and _C._lstm_xcd_enabled and _C.lstm_failed(err):
            # the XCD-local form lost residency / placement (shared or partitioned device): degrade in speed, not in
            # availability -- once, on the all-CU form, which this process keeps from now on
            _C.disable_lstm_xcd("its bounded waits gave up")
            y, err = self._run_once(x)
        _C.lstm_check(err, 'acmi_lstm_layer')
        return y



class _ELU(nn.Module):
    """Placeholder keeping the reference's nn.Sequential numbering; fused into the next convolution."""
    def __init__(self, alpha: float = 1.0):
        super().__init__()
        self.alpha = alpha


class SEANetResnetBlock(nn.Module):
    """reference seanet.py:16-60: x + conv_k1(ELU(conv_k3_dilated(ELU(x))))"""

    def __init__(self, dim: int, kernel_sizes: tp.List[int] = [3, 1], dilations: tp.List[int] = [1, 1],
                 activation: str = 'ELU', activation_params: dict = {'alpha': 1.0}, norm: str = 'none',
                 norm_params: tp.Dict[str, tp.Any] = {}, causal: bool = False, pad_mode: str = 'reflect',
                 compress: int = 2, true_skip: bool = True, device=None):
        super().__init__()
        assert activation == 'ELU', "only ELU (the EnCodec activation) is implemented"
        assert len(kernel_sizes) == len(dilations)
        self.alpha = activation_params.get('alpha', 1.0)
        hidden = dim // compress
        block: tp.List[nn.Module] = []
        for i, (kernel_size, dilation) in enumerate(zip(kernel_sizes, dilations)):
            in_chs = dim if i == 0 else hidden
            out_chs = dim if i == len(kernel_sizes) - 1 else hidden
            block += [_ELU(self.alpha),
                      StreamableConv1d(in_chs, out_chs, kernel_size=kernel_size, dilation=dilation, norm=norm,
                                       norm_kwargs=norm_params, causal=causal, pad_mode=pad_mode, device=device)]
        self.block = nn.ModuleList(block)
        self.shortcut: nn.Module
        if true_skip:
            self.shortcut = nn.Identity()
        else:
            self.shortcut = StreamableConv1d(dim, dim, kernel_size=1, norm=norm, norm_kwargs=norm_params,
                                             causal=causal, pad_mode=pad_mode, device=device)

    def run(self, x: torch.Tensor) -> torch.Tensor:
        skip = x if isinstance(self.shortcut, nn.Identity) else self.shortcut.run(x)
        convs = [m for m in self.block if isinstance(m, StreamableConv1d)]
        y = x
        for i, conv in enumerate(convs):
            y = conv.run(y, elu_alpha=self.alpha, residual=skip if i == len(convs) - 1 else None)
        return y


def _run_sequence(model: nn.ModuleList, x: torch.Tensor) -> torch.Tensor:
    pending_alpha: tp.Optional[float] = None
    for m in model:
        if isinstance(m, _ELU):
            pending_alpha = m.alpha
            continue
        if isinstance(m, (StreamableConv1d, StreamableConvTranspose1d)):
            x = m.run(x, elu_alpha=pending_alpha)
            pending_alpha = None
            continue
        assert pending_alpha is None, "an activation must be followed by a convolution"
        x = m.run(x)
    assert pending_alpha is None
    return x


class SEANetEncoder(nn.Module):
    """reference seanet.py:63-153 (same constructor arguments)."""

    def __init__(self, channels: int = 1, dimension: int = 128, n_filters: int = 32, n_residual_layers: int = 3,
                 ratios: tp.List[int] = [8, 5, 4, 2], activation: str = 'ELU',
                 activation_params: dict = {'alpha': 1.0}, norm: str = 'none',
                 norm_params: tp.Dict[str, tp.Any] = {}, kernel_size: int = 7, last_kernel_size: int = 7,
                 residual_kernel_size: int = 3, dilation_base: int = 2, causal: bool = False,
                 pad_mode: str = 'reflect', true_skip: bool = True, compress: int = 2, lstm: int = 0,
                 disable_norm_outer_blocks: int = 0, device=None):
        super().__init__()
        assert activation == 'ELU'
        self.channels, self.dimension, self.n_filters = channels, dimension, n_filters
        self.ratios = list(reversed(ratios))
        self.n_residual_layers = n_residual_layers
        self.hop_length = int(math.prod(self.ratios))
        self.n_blocks = len(self.ratios) + 2
        self.disable_norm_outer_blocks = disable_norm_outer_blocks
        assert 0 <= disable_norm_outer_blocks <= self.n_blocks
        alpha = activation_params.get('alpha', 1.0)
        kw = dict(norm_kwargs=norm_params, causal=causal, pad_mode=pad_mode, device=device)
        mult = 1
        model: tp.List[nn.Module] = [
            StreamableConv1d(channels, mult * n_filters, kernel_size,
                             norm='none' if disable_norm_outer_blocks >= 1 else norm, **kw)]
        for i, ratio in enumerate(self.ratios):
            block_norm = 'none' if disable_norm_outer_blocks >= i + 2 else norm
            for j in range(n_residual_layers):
                model += [SEANetResnetBlock(mult * n_filters, kernel_sizes=[residual_kernel_size, 1],
                                            dilations=[dilation_base ** j, 1], norm=block_norm,
                                            norm_params=norm_params, activation=activation,
                                            activation_params=activation_params, causal=causal, pad_mode=pad_mode,
                                            compress=compress, true_skip=true_skip, device=device)]
            model += [_ELU(alpha),
                      StreamableConv1d(mult * n_filters, mult * n_filters * 2, kernel_size=ratio * 2, stride=ratio,
                                       norm=block_norm, **kw)]
            mult *= 2
        if lstm:
            model += [StreamableLSTM(mult * n_filters, num_layers=lstm, device=device)]
        model += [_ELU(alpha),
                  StreamableConv1d(mult * n_filters, dimension, last_kernel_size,
                                   norm='none' if disable_norm_outer_blocks == self.n_blocks else norm, **kw)]
        self.model = nn.ModuleList(model)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _run_sequence(self.model, x.float().contiguous())


class SEANetDecoder(nn.Module):
    """reference seanet.py:156-258 (same constructor arguments)."""

    def __init__(self, channels: int = 1, dimension: int = 128, n_filters: int = 32, n_residual_layers: int = 3,
                 ratios: tp.List[int] = [8, 5, 4, 2], activation: str = 'ELU',
                 activation_params: dict = {'alpha': 1.0}, final_activation: tp.Optional[str] = None,
                 final_activation_params: tp.Optional[dict] = None, norm: str = 'none',
                 norm_params: tp.Dict[str, tp.Any] = {}, kernel_size: int = 7, last_kernel_size: int = 7,
                 residual_kernel_size: int = 3, dilation_base: int = 2, causal: bool = False,
                 pad_mode: str = 'reflect', true_skip: bool = True, compress: int = 2, lstm: int = 0,
                 disable_norm_outer_blocks: int = 0, trim_right_ratio: float = 1.0, device=None):
        super().__init__()
        assert activation == 'ELU'
        if final_activation is not None:
            raise NotImplementedError("final_activation is not used by the MusicGen codecs")
        self.dimension, self.channels, self.n_filters = dimension, channels, n_filters
        self.ratios = list(ratios)
        self.n_residual_layers = n_residual_layers
        self.hop_length = int(math.prod(self.ratios))
        self.n_blocks = len(self.ratios) + 2
        self.disable_norm_outer_blocks = disable_norm_outer_blocks
        assert 0 <= disable_norm_outer_blocks <= self.n_blocks
        alpha = activation_params.get('alpha', 1.0)
        kw = dict(norm_kwargs=norm_params, causal=causal, pad_mode=pad_mode, device=device)
        mult = int(2 ** len(self.ratios))
        model: tp.List[nn.Module] = [
            StreamableConv1d(dimension, mult * n_filters, kernel_size,
                             norm='none' if disable_norm_outer_blocks == self.n_blocks else norm, **kw)]
        if lstm:
            model += [StreamableLSTM(mult * n_filters, num_layers=lstm, device=device)]
        for i, ratio in enumerate(self.ratios):
            block_norm = 'none' if disable_norm_outer_blocks >= self.n_blocks - (i + 1) else norm
            model += [_ELU(alpha),
                      StreamableConvTranspose1d(mult * n_filters, mult * n_filters // 2, kernel_size=ratio * 2,
                                                stride=ratio, norm=block_norm, norm_kwargs=norm_params,
                                                causal=causal, trim_right_ratio=trim_right_ratio, device=device)]
            for j in range(n_residual_layers):
                model += [SEANetResnetBlock(mult * n_filters // 2, kernel_sizes=[residual_kernel_size, 1],
                                            dilations=[dilation_base ** j, 1], activation=activation,
                                            activation_params=activation_params, norm=block_norm,
                                            norm_params=norm_params, causal=causal, pad_mode=pad_mode,
                                            compress=compress, true_skip=true_skip, device=device)]
            mult //= 2
        model += [_ELU(alpha),
                  StreamableConv1d(n_filters, channels, last_kernel_size,
                                   norm='none' if disable_norm_outer_blocks >= 1 else norm, **kw)]
        self.model = nn.ModuleList(model)

    def forward(self, z: torch.Tensor) -> torch.Tensor:
        return _run_sequence(self.model, z.float().contiguous())


def invalidate_prepared(module: nn.Module):
    """Drop folded-weight caches (after load_state_dict / .to())."""
    for m in module.modules():
        if hasattr(m, '_prep'):
            m._prep = None
