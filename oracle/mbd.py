"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the MultiBandDiffusion decoder option:

  unet_forward          audiocraft/models/unet.py:23-214 (DiffusionUnet: encoder / decoder layers of strided convolutions,
                        GroupNorm + ReLU residual blocks, per-step embeddings, codec conditioning in the bottleneck, optional
                        BiLSTM bottleneck; `transformer=True` is not restated)
  generate / generate_subsampled, betas_from_alpha_bar
                        audiocraft/modules/diffusion_schedule.py:19-21, 113-272 (NoiseSchedule: power schedule, DDPM reverse
                        process, the sub-sampled Markov chain the released models are run with)
  MultiBandProcessor    :35-110 (per-band mean / std rescaling of the sample space)
  split_bands           julius.SplitBands / julius.LowPassFilters (third party, pinned `julius` of the reference's
                        requirements.txt; absent here, restated from its published algorithm: windowed-sinc low-pass filters at
                        mel-spaced cut-offs, replicate padding, bands = differences of successive low-passes) -- PARITY UNPINNED
                        against the julius binary, checked against closed forms only (bands sum to the input)
  re_eq                 audiocraft/models/multibanddiffusion.py:150-164; MultiBandDiffusion.generate (:120-139) is the sum of
                        generate_subsampled over the bands (composed in the tests)

The reference's own modules (unet, diffusion_schedule, and multibanddiffusion.py executed from its source text, with julius
stubbed by this module's split_bands) generate the golden fixtures of tests/golden/mbd_*.npz (make_mbd_golden.py);
oracle/validate_against_reference.py `mbd` re-checks at a larger size.
"""
import math
import typing as tp
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F


@dataclass
class UnetConfig:
    chin: int = 1
    hidden: int = 24
    depth: int = 3
    growth: float = 2.
    max_channels: int = 10_000
    num_steps: int = 1000
    emb_all_layers: bool = False
    bilstm: bool = False
    codec_dim: tp.Optional[int] = None
    kernel: int = 4
    stride: int = 2
    norm_groups: int = 4
    res_blocks: int = 1


@dataclass
class ScheduleConfig:
    beta_t0: float = 1e-4
    beta_t1: float = 0.02
    num_steps: int = 1000
    variance: str = 'beta'
    clip: float = 5.
    rescale: float = 1.
    beta_exp: float = 1.
    noise_scale: float = 1.0
    betas: torch.Tensor = field(default=None, repr=False)

    def __post_init__(self):
        # diffusion_schedule.py:147-149 (power schedule)
        self.betas = torch.linspace(self.beta_t0 ** (1 / self.beta_exp), self.beta_t1 ** (1 / self.beta_exp), self.num_steps,
                                    dtype=torch.float) ** self.beta_exp


# ------------------------------------------------------------------------------------------ U-Net

def _res_block(sd, p, x, groups, dilation):
    """unet.py:32-53: x + conv2(relu(norm2(conv1(relu(norm1(x))))))  (kernel 3, padding = dilation)"""
    h = F.conv1d(F.relu(F.group_norm(x, groups, sd[p + '.norm1.weight'], sd[p + '.norm1.bias'])),
                 sd[p + '.conv1.weight'], sd[p + '.conv1.bias'], padding=dilation, dilation=dilation)
    h = F.conv1d(F.relu(F.group_norm(h, groups, sd[p + '.norm2.weight'], sd[p + '.norm2.bias'])),
                 sd[p + '.conv2.weight'], sd[p + '.conv2.bias'], padding=dilation, dilation=dilation)
    return x + h


def _lstm_direction(x, w_ih, w_hh, b_ih, b_hh):
    """one direction of one nn.LSTM layer, x [T, B, I] -> [T, B, H] (gate order i, f, g, o)"""
    T, B, _ = x.shape
    H = w_hh.shape[1]
    h = torch.zeros(B, H)
    c = torch.zeros(B, H)
    out = []
    for t in range(T):
        g = x[t] @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
        i, f, gg, o = g.chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        out.append(h)
    return torch.stack(out)


def _blstm(sd, x, layers=2):
    """unet.py:106-120: bidirectional nn.LSTM (2 layers, hidden = dim) + Linear(2 dim -> dim); x [B, C, T]"""
    y = x.permute(2, 0, 1)
    for layer in range(layers):
        fw = _lstm_direction(y, *(sd[f'bilstm.lstm.{n}_l{layer}'] for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')))
        bw = _lstm_direction(y.flip(0), *(sd[f'bilstm.lstm.{n}_l{layer}_reverse']
                                          for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh'))).flip(0)
        y = torch.cat([fw, bw], dim=-1)
    y = F.linear(y, sd['bilstm.linear.weight'], sd['bilstm.linear.bias'])
    return y.permute(1, 2, 0)


def unet_forward(sd: dict, cfg: UnetConfig, x: torch.Tensor, step: tp.Union[int, torch.Tensor],
                 condition: tp.Optional[torch.Tensor] = None) -> torch.Tensor:
    """DiffusionUnet.forward (unet.py:165-214).  x [B, chin, T] -> estimate [B, chin, T]."""
    B = x.shape[0]
    steps = step if isinstance(step, torch.Tensor) else torch.full((B,), int(step), dtype=torch.long)
    g, pad_k = cfg.norm_groups, (cfg.kernel - cfg.stride) // 2
    z, skips = x, []
    for d in range(cfg.depth):
        p = f'encoders.{d}'
        T = z.shape[-1]
        z = F.pad(z, (0, (cfg.stride - (T % cfg.stride)) % cfg.stride))          # EncoderLayer.forward :96-100
        z = F.conv1d(z, sd[p + '.conv.weight'], None, stride=cfg.stride, padding=pad_k)
        z = F.relu(F.group_norm(z, g, sd[p + '.norm.weight'], sd[p + '.norm.bias']))
        for r in range(cfg.res_blocks):
            z = _res_block(sd, f'{p}.res_blocks.{r}', z, g, 2 ** r)
        if d == 0:
            z = z + sd['embedding.weight'][steps].view(B, -1, 1)
        elif cfg.emb_all_layers:
            z = z + sd[f'embeddings.{d - 1}.weight'][steps].view(B, -1, 1)
        skips.append(z)
    if cfg.codec_dim is not None:   # :186-195 (no cross-attention: interpolate to the bottleneck length and add)
        assert condition is not None
        ce = F.conv1d(condition, sd['conv_codec.weight'], sd['conv_codec.bias'])
        assert ce.shape[-1] <= 2 * z.shape[-1]
        # `z += condition_emb` (:193) is IN PLACE on the tensor that was just appended to `skips`: the condition therefore
        # also enters through the deepest skip connection -- the only way it enters at all when the bottleneck is zeroed
        z = z + F.interpolate(ce, z.shape[-1])
        skips[-1] = z
    z = _blstm(sd, z) if cfg.bilstm else torch.zeros_like(z)                      # :202-207
    for d in reversed(range(cfg.depth)):
        p = f'decoders.{cfg.depth - 1 - d}'
        s = skips.pop(-1)
        z = z[:, :, :s.shape[2]] + s
        for r in range(cfg.res_blocks):
            z = _res_block(sd, f'{p}.res_blocks.{r}', z, g, 2 ** r)
        z = F.relu(F.group_norm(z, g, sd[p + '.norm.weight'], sd[p + '.norm.bias']))
        z = F.conv_transpose1d(z, sd[p + '.convtr.weight'], None, stride=cfg.stride, padding=pad_k)
    return z[:, :, :x.shape[2]]


# ------------------------------------------------------------------------------------------ band splitting (julius, restated)

def mel_frequencies(n_mels: int, fmin: float, fmax: float) -> torch.Tensor:
    """julius.bands.mel_frequencies: n_mels points evenly spaced on the (HTK) mel scale."""
    to_mel = lambda f: 2595.0 * math.log10(1.0 + f / 700.0)   # noqa: E731
    mels = torch.linspace(to_mel(fmin), to_mel(fmax), n_mels)
    return 700.0 * (10.0 ** (mels / 2595.0) - 1.0)


def lowpass_filters(cutoffs: torch.Tensor, zeros: float = 8) -> tp.Tuple[torch.Tensor, int]:
    """julius.lowpass.LowPassFilters: one windowed-sinc filter per normalised cut-off (f / sr), all of half size
    int(zeros / min(cutoffs) / 2), Hann window (symmetric), each normalised to sum 1.  -> ([n, 2 half + 1], half)"""
    half = int(zeros / float(cutoffs.min()) / 2)
    window = torch.hann_window(2 * half + 1, periodic=False)
    t = torch.arange(-half, half + 1, dtype=torch.float32)
    filters = []
    for c in cutoffs.tolist():
        arg = 2 * c * math.pi * t
        sinc = torch.where(arg == 0, torch.ones_like(arg), torch.sin(arg) / arg)
        f = 2 * c * window * sinc
        filters.append(f / f.sum())
    return torch.stack(filters), half


def split_bands(x: torch.Tensor, sample_rate: float, n_bands: int) -> torch.Tensor:
    """julius.SplitBands(sample_rate, n_bands)(x): x [..., T] -> [n_bands, ..., T]; the bands sum to x exactly."""
    if n_bands == 1:
        return x[None]
    cutoffs = mel_frequencies(n_bands + 1, 0, sample_rate / 2)[1:-1] / sample_rate
    filt, half = lowpass_filters(cutoffs)
    shape = x.shape
    flat = F.pad(x.reshape(-1, 1, shape[-1]), (half, half), mode='replicate')
    lows = F.conv1d(flat, filt[:, None]).permute(1, 0, 2).reshape(len(cutoffs), *shape)
    bands, low = [lows[0]], lows[0]
    for low_and_band in lows[1:]:
        bands.append(low_and_band - low)
        low = low_and_band
    bands.append(x - low)
    return torch.stack(bands)


# ------------------------------------------------------------------------------------------ sample processor

@dataclass
class ProcessorState:
    """Buffers of MultiBandProcessor (diffusion_schedule.py:63-70); n_bands = 0 means the identity SampleProcessor."""
    n_bands: int = 0
    sample_rate: float = 24000.
    power_std: tp.Union[float, torch.Tensor] = 1.
    counts: torch.Tensor = None
    sum_x: torch.Tensor = None
    sum_x2: torch.Tensor = None
    sum_target_x2: torch.Tensor = None

    @property
    def mean(self):
        return self.sum_x / self.counts

    @property
    def std(self):
        return (self.sum_x2 / self.counts - self.mean ** 2).clamp(min=0).sqrt()

    @property
    def target_std(self):
        return self.sum_target_x2 / self.counts


def return_sample(ps: ProcessorState, z: torch.Tensor) -> torch.Tensor:
    """MultiBandProcessor.return_sample (:104-109)"""
    if ps.n_bands == 0:
        return z
    bands = split_bands(z, ps.sample_rate, ps.n_bands)
    rescale = (ps.std / ps.target_std) ** ps.power_std
    return (bands * rescale.view(-1, 1, 1, 1) + ps.mean.view(-1, 1, 1, 1)).sum(dim=0)


def project_sample(ps: ProcessorState, x: torch.Tensor) -> torch.Tensor:
    """MultiBandProcessor.project_sample (:91-102) once the statistics are frozen (counts >= num_samples)"""
    if ps.n_bands == 0:
        return x
    bands = split_bands(x, ps.sample_rate, ps.n_bands)
    rescale = (ps.target_std / ps.std.clamp(min=1e-12)) ** ps.power_std
    return ((bands - ps.mean.view(-1, 1, 1, 1)) * rescale.view(-1, 1, 1, 1)).sum(dim=0)


# ------------------------------------------------------------------------------------------ reverse process

def betas_from_alpha_bar(alpha_bar: torch.Tensor) -> torch.Tensor:
    alphas = torch.cat([torch.Tensor([alpha_bar[0]]), alpha_bar[1:] / alpha_bar[:-1]])
    return 1 - alphas


def alpha_bar_at(sc: ScheduleConfig, step: int) -> torch.Tensor:
    return (1 - sc.betas[:step + 1]).prod()


def generate_subsampled(model: tp.Callable, sc: ScheduleConfig, initial: torch.Tensor, step_list: tp.Optional[tp.List[int]] = None,
                        condition: tp.Optional[torch.Tensor] = None, noises: tp.Optional[tp.List[torch.Tensor]] = None,
                        ps: tp.Optional[ProcessorState] = None) -> torch.Tensor:
    """NoiseSchedule.generate_subsampled (:239-272).  model(x, step, condition) -> estimate; `noises[i]` replaces the
    reference's `torch.randn_like` of iteration i (generator streams cannot be shared with a device)."""
    if step_list is None:
        step_list = list(range(1000))[::-50] + [0]
    alpha_bar = alpha_bar_at(sc, sc.num_steps - 1)
    alpha_bars_sub = (1 - sc.betas).cumprod(dim=0)[list(reversed(step_list))]
    betas_sub = betas_from_alpha_bar(alpha_bars_sub)
    current = initial * sc.noise_scale
    previous = current
    for idx, step in enumerate(step_list[:-1]):
        estimate = model(current, step, condition) * sc.noise_scale
        alpha = 1 - betas_sub[-1 - idx]
        previous = (current - (1 - alpha) / (1 - alpha_bar).sqrt() * estimate) / alpha.sqrt()
        previous_alpha_bar = alpha_bar_at(sc, step_list[idx + 1])
        if step == step_list[-2]:
            sigma2 = 0
            previous_alpha_bar = torch.tensor(1.0)
        else:
            sigma2 = (1 - previous_alpha_bar) / (1 - alpha_bar) * (1 - alpha)
        if sigma2 > 0:
            noise = noises[idx] if noises is not None else torch.randn_like(previous)
            previous = previous + sigma2 ** 0.5 * noise * sc.noise_scale
        if sc.clip:
            previous = previous.clamp(-sc.clip, sc.clip)
        current = previous
        alpha_bar = previous_alpha_bar
        if step == 0:
            previous = previous * sc.rescale
    return return_sample(ps, previous) if ps is not None else previous


def re_eq(wav: torch.Tensor, ref: torch.Tensor, sample_rate: float, n_bands: int = 32, strictness: float = 1) -> torch.Tensor:
    """MultiBandDiffusion.re_eq (multibanddiffusion.py:150-164): per-band std matching to the codec's own output"""
    bands, bands_ref = split_bands(wav, sample_rate, n_bands), split_bands(ref, sample_rate, n_bands)
    out = torch.zeros_like(ref)
    for i in range(n_bands):
        out = out + bands[i] * (bands_ref[i].std() / bands[i].std()) ** strictness
    return out
