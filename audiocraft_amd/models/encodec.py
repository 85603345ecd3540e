"""EnCodec compression model on MI355X.

API mirror of `audiocraft.models.encodec.CompressionModel` / `EncodecModel`
(reference audiocraft/models/encodec.py:28-259): encode / decode / decode_latent / forward and the
channels / frame_rate / sample_rate / cardinality / num_codebooks / total_codebooks properties.
The HF / DAC / stereo-interleave wrappers of the reference are outside this path.
"""
import math
import os
import re
import typing as tp
from abc import ABC, abstractmethod

import torch
from torch import nn

from .. import _C
from ..modules.seanet import invalidate_prepared
from ..quantization.vq import BaseQuantizer, QuantizedResult  # noqa: F401  (QuantizedResult: also exported from here)


def _abstract_property(name: str, doc: str):
    def getter(self):
        raise NotImplementedError(f"{type(self).__name__} must define `{name}`")
    getter.__isabstractmethod__ = True
    return property(getter, doc=doc)


class CompressionModel(ABC, nn.Module):
    """Audio tokenizer interface (reference `CompressionModel`, encodec.py:28-85): `encode(x[B, C, T]) ->
    (codes[B, K, T'], scale | None)`, `decode(codes, scale) -> wav[B, C, T]`, `decode_latent(codes)`,
    `forward(x) -> QuantizedResult`, `set_num_codebooks(n)` and the read-only properties below."""

    channels = _abstract_property('channels', "audio channels")
    frame_rate = _abstract_property('frame_rate', "token frames per second")
    sample_rate = _abstract_property('sample_rate', "audio sample rate")
    cardinality = _abstract_property('cardinality', "entries per codebook")
    num_codebooks = _abstract_property('num_codebooks', "codebooks in use")
    total_codebooks = _abstract_property('total_codebooks', "codebooks available")

    @abstractmethod
    def forward(self, x: torch.Tensor) -> QuantizedResult: ...

    @abstractmethod
    def encode(self, x: torch.Tensor) -> tp.Tuple[torch.Tensor, tp.Optional[torch.Tensor]]: ...

    @abstractmethod
    def decode(self, codes: torch.Tensor, scale: tp.Optional[torch.Tensor] = None): ...

    @abstractmethod
    def decode_latent(self, codes: torch.Tensor): ...

    @abstractmethod
    def set_num_codebooks(self, n: int): ...

    @staticmethod
    def get_pretrained(name: str, device='cuda') -> 'CompressionModel':
        """reference encodec.py:88-122.  `debug_compression_model`; a path (file / directory holding a
        `compression_state_dict.bin` written by `audiocraft.utils.export`) or a released name resolved on disk under
        $AUDIOCRAFT_CACHE_DIR; else -- like the reference's last branch -- a HuggingFace EnCodec (`facebook/encodec_24khz`, or a
        directory in HuggingFace's format) through `transformers.EncodecModel.from_pretrained`, whose weights are re-keyed into
        this package's EncodecModel (`HFEncodecCompressionModel`).  There is no network here: HuggingFace resolves names in
        its local cache only.  The DAC wrappers (third-party codecs outside the path) raise."""
        from . import builders, loaders
        if name in ('dac_44khz', 'dac_24khz'):
            raise NotImplementedError("DAC codecs are third-party models outside the MusicGen path")
        if name == 'debug_compression_model':
            return builders.get_debug_compression_model(device).eval()
        hf_dir = os.path.isdir(name) and os.path.isfile(os.path.join(name, 'config.json')) and \
            not os.path.isfile(os.path.join(name, 'compression_state_dict.bin'))
        if not hf_dir:
            try:
                return loaders.load_compression_model(name, device=device).eval()
            except FileNotFoundError as exc:
                not_on_disk = exc
        else:
            not_on_disk = None
        tried = f"compression model '{name}': not an audiocraft export on disk ({not_on_disk})"
        try:   # the optional dependency itself: absent, or broken (transformers' lazy import raises RuntimeError then)
            from transformers import EncodecModel as HFEncodecModel
        except (ImportError, RuntimeError) as exc:
            raise FileNotFoundError(f"{tried} and transformers' EncodecModel cannot be imported "
                                    f"({type(exc).__name__}: {exc})") from exc
        try:   # not in the HuggingFace cache (OSError), or a repository that is not an EnCodec (ValueError)
            hf_model = HFEncodecModel.from_pretrained(name)
        except (OSError, ValueError) as exc:
            raise FileNotFoundError(f"{tried} and not loadable as a HuggingFace EnCodec "
                                    f"({type(exc).__name__}: {exc})") from exc
        return HFEncodecCompressionModel(hf_model, device).eval()


class EncodecModel(CompressionModel):
    """SEANet encoder -> residual vector quantizer -> SEANet decoder on the waveform (reference
    `EncodecModel`, encodec.py:125-259).  `renormalize=True` (divide by the mono RMS before encoding,
    multiply back after decoding; encodec.py:186-204) is kept for completeness -- no MusicGen codec uses
    it -- and runs as three elementwise torch ops."""
    frame_rate: float = 0
    sample_rate: int = 0
    channels: int = 0

    def __init__(self, encoder: nn.Module, decoder: nn.Module, quantizer: BaseQuantizer, frame_rate: int,
                 sample_rate: int, channels: int, causal: bool = False, renormalize: bool = False):
        super().__init__()
        assert not (causal and renormalize), 'Causal model does not support renormalize'
        self.encoder, self.decoder, self.quantizer = encoder, decoder, quantizer
        self.frame_rate, self.sample_rate, self.channels = frame_rate, sample_rate, channels
        self.causal, self.renormalize = causal, renormalize
        self.eval()

    # folded weights are cached inside the conv / quantizer modules: drop them whenever parameters change
    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        result = super().load_state_dict(state_dict, strict=strict, **kw)
        invalidate_prepared(self)
        self._graphs = {}
        return result

    def _apply(self, fn, *args, **kw):
        result = super()._apply(fn, *args, **kw)
        invalidate_prepared(self)
        self._graphs = {}
        return result

    # ---- short inputs: the SEANet passes as hipGraph replays.  Below ~1 M samples per call the ~70 convolution launches of a
    # pass (two kernels, two allocations, one ctypes call each) cost more host time than GPU time (EnCodec-24k, 1 x 10 s:
    # 1.9 ms of a 5 ms decode).  The second call with a shape captures the pass (torch.cuda.CUDAGraph is only the capture /
    # replay plumbing; every node is an acmi kernel or a memset), later calls copy the input in and replay.
    GRAPH_MAX_SAMPLES = int(os.environ.get('ACMI_CODEC_GRAPH_MAX', str(1 << 20)))   # 0: never
    GRAPH_SLOTS = 6

    def _seanet(self, which: str, net: nn.Module, x: torch.Tensor) -> torch.Tensor:
        x = x.float().contiguous()
        samples = x.shape[0] * x.shape[-1] * (self.decoder.hop_length if which == 'dec' else 1)
        if not x.is_cuda or samples == 0 or samples > self.GRAPH_MAX_SAMPLES or torch.cuda.is_current_stream_capturing():
            return net(x)   # (inside somebody else's capture: plain launches, which that capture records)
        graphs = self.__dict__.setdefault('_graphs', {})
        key = (which, tuple(x.shape))
        entry = graphs.get(key)
        if entry is None:          # first sight of the shape: eager (tiles the weights, fills the host-side caches)
            while len(graphs) >= self.GRAPH_SLOTS:
                graphs.pop(next(iter(graphs)))
            graphs[key] = False
            return net(x)
        if entry is False:
            entry = graphs[key] = self._capture(net, x)
        graph, static_in, static_out, checks = entry
        static_in.copy_(x)
        graph.replay()
        out = static_out.clone()
        if checks and bool(torch.stack([w.view(torch.int32)[0] for w, _ in checks]).any()):
            if _C._lstm_xcd_enabled:
                # the XCD-local recurrence lost residency / placement (shared device): drop the captured passes, keep the all-CU
                # form for the rest of the process and run this call again, eagerly (its own check raises if that fails too)
                _C.disable_lstm_xcd("its bounded waits gave up inside a captured codec pass")
                graphs.clear()
                return net(x)
            raise _C.AcmiError(f"{checks[0][1]}: the persistent LSTM kernel gave up waiting for a workgroup "
                               "(set ACMI_LSTM_WAVE=0 / ACMI_LSTM_PERSISTENT=0)")
        return out

    @staticmethod
    def _capture(net: nn.Module, x: torch.Tensor):
        static_in = x.clone()
        checks: list = []
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            _C.defer_lstm_checks(checks)   # reading the LSTM's give-up word synchronises: done after every replay instead
            graph.capture_begin()
            try:
                static_out = net(static_in)
            finally:
                graph.capture_end()
                _C.defer_lstm_checks(None)
        torch.cuda.current_stream().wait_stream(side)
        return graph, static_in, static_out, checks

    total_codebooks = property(lambda self: self.quantizer.total_codebooks)
    num_codebooks = property(lambda self: self.quantizer.num_codebooks)
    cardinality = property(lambda self: self.quantizer.bins)

    def set_num_codebooks(self, n: int):
        self.quantizer.set_num_codebooks(n)

    def preprocess(self, x: torch.Tensor) -> tp.Tuple[torch.Tensor, tp.Optional[torch.Tensor]]:
        if not self.renormalize:
            return x, None
        rms = x.mean(dim=1, keepdim=True).pow(2).mean(dim=2, keepdim=True).sqrt()
        scale = 1e-8 + rms
        return x / scale, scale.view(-1, 1)

    def postprocess(self, x: torch.Tensor, scale: tp.Optional[torch.Tensor] = None) -> torch.Tensor:
        if scale is None:
            return x
        assert self.renormalize
        return x * scale.view(-1, 1, 1)

    @_C.exclusive
    @torch.no_grad()
    def encode(self, x: torch.Tensor) -> tp.Tuple[torch.Tensor, tp.Optional[torch.Tensor]]:
        assert x.dim() == 3
        x, scale = self.preprocess(x)
        return self.quantizer.encode(self._seanet('enc', self.encoder, x)), scale

    @_C.exclusive
    @torch.no_grad()
    def decode_latent(self, codes: torch.Tensor):
        return self.quantizer.decode(codes)

    @_C.exclusive
    @torch.no_grad()
    def decode(self, codes: torch.Tensor, scale: tp.Optional[torch.Tensor] = None):
        # the result keeps the encoder's extra right padding; callers trim to the length they expect
        return self.postprocess(self._seanet('dec', self.decoder, self.decode_latent(codes)), scale)

    @_C.exclusive
    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> QuantizedResult:
        assert x.dim() == 3
        codes, scale = self.encode(x)
        out = self.decode(codes, scale)
        assert out.shape[-1] >= x.shape[-1], (out.shape[-1], x.shape[-1])
        kbps = codes.shape[1] * math.log2(self.cardinality) * self.frame_rate / 1000
        return QuantizedResult(out[..., :x.shape[-1]], codes, torch.tensor(kbps).to(out))


def hf_encodec_cfg(config) -> dict:
    """`transformers.EncodecConfig` -> builders.get_compression_model cfg.  HuggingFace's EnCodec (modeling_encodec.py) is the
    same SEANet / RVQ as the reference's own (`modules/seanet.py`, `quantization/vq.py`) under other names: the layer lists
    of encoder and decoder line up index by index (activations included), `use_conv_shortcut` is `not true_skip`,
    `dilation_growth_rate` is `dilation_base`, `normalize` is `renormalize`."""
    def get(name, default=None):
        return getattr(config, name, default) if not isinstance(config, dict) else config.get(name, default)
    if get('norm_type', 'weight_norm') != 'weight_norm':
        raise NotImplementedError(f"HF EnCodec norm_type {get('norm_type')!r} (only 'weight_norm': facebook/encodec_24khz)")
    if get('chunk_length_s') is not None:
        raise NotImplementedError("HF EnCodec with chunked encoding (chunk_length_s: the 48 kHz model) is not on the MusicGen path")
    if get('codebook_dim', get('hidden_size')) != get('hidden_size'):
        raise NotImplementedError("HF EnCodec with codebook_dim != hidden_size")
    ratios = list(get('upsampling_ratios'))
    hop = 1
    for r in ratios:
        hop *= r
    sr = get('sampling_rate')
    frame_rate = sr / hop
    bins = get('codebook_size')
    # EncodecConfig.num_quantizers (a property there, a plain entry in a dict): the codebooks the largest target bandwidth
    # needs, 1000 * bandwidth // (ceil(frame rate) * bits per codebook)
    n_q = get('num_quantizers') or int(1000 * get('target_bandwidths')[-1] // (math.ceil(frame_rate) * int(math.log2(bins))))
    seanet = dict(channels=get('audio_channels'), dimension=get('hidden_size'), n_filters=get('num_filters'),
                  n_residual_layers=get('num_residual_layers'), ratios=ratios, activation='ELU', activation_params={'alpha': 1.0},
                  norm='weight_norm', norm_params={}, kernel_size=get('kernel_size'), last_kernel_size=get('last_kernel_size'),
                  residual_kernel_size=get('residual_kernel_size'), dilation_base=get('dilation_growth_rate'),
                  causal=bool(get('use_causal_conv')), pad_mode=get('pad_mode'), true_skip=not get('use_conv_shortcut'),
                  compress=get('compress'), lstm=get('num_lstm_layers'), disable_norm_outer_blocks=0)
    return dict(seanet=seanet, rvq=dict(n_q=n_q, bins=bins), sample_rate=sr, frame_rate=frame_rate,
                channels=get('audio_channels'), causal=bool(get('use_causal_conv')), renormalize=bool(get('normalize')),
                trim_right_ratio=get('trim_right_ratio', 1.0))


def convert_hf_encodec_state_dict(hf_state: tp.Dict[str, torch.Tensor], native_keys: tp.Iterable[str]) -> tp.Dict[str, torch.Tensor]:
    """State dict of `transformers.EncodecModel` -> the reference's EnCodec key names (what `EncodecModel.load_state_dict` of
    this package takes): `encoder.layers.{i}.[block.{j}. | shortcut.]conv.{parametrizations.weight.original0 | original1 |
    weight_g | weight_v | bias}` -> `encoder.model.{i}...conv.conv.{weight_g | weight_v | bias}` (`convtr.convtr.` where the
    native model holds a transposed convolution at that index), LSTMs unchanged, `quantizer.layers.{q}.codebook.*` ->
    `quantizer.vq.layers.{q}._codebook.*`.  No tensor is touched."""
    native = set(native_keys)
    leaf = {'parametrizations.weight.original0': 'weight_g', 'parametrizations.weight.original1': 'weight_v',
            'weight_g': 'weight_g', 'weight_v': 'weight_v', 'weight': 'weight', 'bias': 'bias'}
    out: tp.Dict[str, torch.Tensor] = {}
    for k, v in hf_state.items():
        m = re.match(r'^(encoder|decoder)\.layers\.(\d+)\.(.*)$', k)
        if m:
            side, idx, rest = m.groups()
            base = f'{side}.model.{idx}.'
            if rest.startswith('lstm.'):
                new = base + rest
            else:
                cm = re.match(r'^((?:block\.\d+\.|shortcut\.)?)conv\.(.*)$', rest)
                if cm is None or cm.group(2) not in leaf:
                    raise KeyError(f"unexpected HF EnCodec key {k!r}")
                sub, name = cm.group(1), leaf[cm.group(2)]
                new = base + sub + 'conv.conv.' + name
                if new not in native:
                    new = base + sub + 'convtr.convtr.' + name
        else:
            qm = re.match(r'^quantizer\.layers\.(\d+)\.codebook\.(.*)$', k)
            if qm is None:
                raise KeyError(f"unexpected HF EnCodec key {k!r}")
            new = f'quantizer.vq.layers.{qm.group(1)}._codebook.{qm.group(2)}'
        if new not in native:
            raise KeyError(f"HF EnCodec key {k!r} has no counterpart ({new!r}) in the native model")
        out[new] = v
    return out


class HFEncodecCompressionModel(CompressionModel):
    """The reference's wrapper around HuggingFace EnCodec (audiocraft/models/encodec.py:323-394: `facebook/encodec_24khz`, the
    codec of `MultiBandDiffusion.get_mbd_24khz`) -- same constructor argument (a `transformers.EncodecModel`), methods and
    properties; `set_num_codebooks` only takes the codebook counts of `config.target_bandwidths`, `forward` raises.
    The wrapped HF module is NOT what runs: its weights are re-keyed (`convert_hf_encodec_state_dict`) into this package's
    `EncodecModel` of the same geometry (`hf_encodec_cfg`), i.e. the SEANet / LSTM / RVQ kernels of libacmi."""

    def __init__(self, model, device=None):
        super().__init__()
        from . import builders
        config = model.config
        if device is None:
            p = next(iter(model.parameters()), None)
            device = p.device if p is not None else 'cpu'
        cfg = hf_encodec_cfg(config)
        self.config = config
        self.model = builders.get_compression_model(cfg, device)
        self.model.load_state_dict(convert_hf_encodec_state_dict(model.state_dict(), self.model.state_dict().keys()))
        self._sample_rate, self._channels, self._frame_rate = cfg['sample_rate'], cfg['channels'], cfg['frame_rate']
        self._cardinality = cfg['rvq']['bins']
        bws = list(getattr(config, 'target_bandwidths'))
        num_codebooks = [bw * 1000 / (self.frame_rate * math.log2(self.cardinality)) for bw in bws]
        deltas = [nc - int(nc) for nc in num_codebooks]
        assert all(d <= 1e-3 for d in deltas), deltas   # "we indeed have integers"
        self.possible_num_codebooks = [int(nc) for nc in num_codebooks]
        assert max(self.possible_num_codebooks) <= self.model.total_codebooks
        self.set_num_codebooks(max(self.possible_num_codebooks))
        self.eval()

    def forward(self, x: torch.Tensor) -> QuantizedResult:
        raise NotImplementedError("Forward and training with HF EncodecModel not supported.")

    @_C.exclusive
    @torch.no_grad()
    def encode(self, x: torch.Tensor) -> tp.Tuple[torch.Tensor, tp.Optional[torch.Tensor]]:
        self.model.set_num_codebooks(self.num_codebooks)
        return self.model.encode(x)     # (codes [B, K, T], scale [B, 1] | None): what `res[0][0], res[1][0]` are in the reference

    @_C.exclusive
    @torch.no_grad()
    def decode(self, codes: torch.Tensor, scale: tp.Optional[torch.Tensor] = None):
        # (with a scale the reference hands `scale` to HF as its per-frame list, i.e. it applies scale[0] to every item of the
        # batch; here every item gets its own -- identical for `normalize: false` codecs such as facebook/encodec_24khz)
        return self.model.decode(codes, scale)

    @_C.exclusive
    @torch.no_grad()
    def decode_latent(self, codes: torch.Tensor):
        return self.model.decode_latent(codes)

    channels = property(lambda self: self._channels)
    frame_rate = property(lambda self: self._frame_rate)
    sample_rate = property(lambda self: self._sample_rate)
    cardinality = property(lambda self: self._cardinality)
    num_codebooks = property(lambda self: self._num_codebooks)
    total_codebooks = property(lambda self: max(self.possible_num_codebooks))

    def set_num_codebooks(self, n: int):
        if n not in self.possible_num_codebooks:
            raise ValueError(f"Allowed values for num codebooks: {self.possible_num_codebooks}")
        self._num_codebooks = n


class InterleaveStereoCompressionModel(CompressionModel):
    """Stereo on top of a mono codec (reference audiocraft/models/encodec.py:397-506): left / right are
    encoded independently and their codebooks interleaved ([B, K, T] x 2 -> [B, 2K, T], left first per level);
    `per_timestep=True` interleaves along time instead.  Pure host logic over `EncodecModel`."""

    def __init__(self, model: CompressionModel, per_timestep: bool = False):
        super().__init__()
        self.model = model
        self.per_timestep = per_timestep
        assert self.model.channels == 1, "Wrapped model is expected to be for monophonic audio"

    @property
    def total_codebooks(self):
        return self.model.total_codebooks

    @property
    def num_codebooks(self):
        """With K the virtual number of codebooks: the wrapped model runs K // 2 per channel."""
        return self.model.num_codebooks if self.per_timestep else self.model.num_codebooks * 2

    def set_num_codebooks(self, n: int):
        """reference encodec.py:428-433: `n` is the WRAPPED model's number of codebooks, i.e. before the interleaving (what
        `compression_model_n_q` of an experiment config means, builders.py:345-348); `num_codebooks` then reads 2 n."""
        self.model.set_num_codebooks(n)

    @property
    def num_virtual_steps(self) -> float:
        return 2 if self.per_timestep else 1

    @property
    def frame_rate(self) -> float:
        return self.model.frame_rate * self.num_virtual_steps

    @property
    def sample_rate(self) -> int:
        return self.model.sample_rate

    @property
    def channels(self) -> int:
        return 2

    @property
    def cardinality(self):
        return self.model.cardinality

    def forward(self, x: torch.Tensor) -> QuantizedResult:
        raise NotImplementedError("Not supported, use encode and decode.")

    def encode(self, x: torch.Tensor) -> tp.Tuple[torch.Tensor, tp.Optional[torch.Tensor]]:
        B, C, T = x.shape
        assert C == self.channels, f"Expecting stereo audio but audio num channels is {C}"
        codes, scale = self.model.encode(x.reshape(B * C, 1, T))          # [(B C), K, T']
        K, Tp = codes.shape[1], codes.shape[2]
        codes = codes.view(B, C, K, Tp)
        if self.per_timestep:
            codes = codes.permute(0, 2, 3, 1).reshape(B, K, Tp * C)       # b k (t c)
        else:
            codes = codes.permute(0, 2, 1, 3).reshape(B, K * C, Tp)       # b (k c) t
        return codes.contiguous(), None if scale is None else scale.view(B, C)

    def get_left_right_codes(self, codes: torch.Tensor) -> tp.Tuple[torch.Tensor, torch.Tensor]:
        B = codes.shape[0]
        if self.per_timestep:
            c = codes.view(B, codes.shape[1], -1, 2).permute(0, 3, 1, 2)  # b k (t c) -> b c k t
        else:
            c = codes.view(B, -1, 2, codes.shape[2]).permute(0, 2, 1, 3)  # b (k c) t -> b c k t
        return c[:, 0].contiguous(), c[:, 1].contiguous()

    def decode(self, codes: torch.Tensor, scale: tp.Optional[torch.Tensor] = None):
        B, K, T = codes.shape
        assert T % self.num_virtual_steps == 0 and K == self.num_codebooks
        left, right = self.get_left_right_codes(codes)
        both = torch.cat([left, right], dim=0)
        sc = None if scale is None else torch.cat([scale[:, 0], scale[:, 1]], dim=0).reshape(-1, 1)
        out = self.model.decode(both, sc)                                 # [2B, 1, T']
        return torch.cat([out[:B], out[B:]], dim=1)

    def decode_latent(self, codes: torch.Tensor):
        raise NotImplementedError("Not supported by interleaved stereo wrapped models.")
