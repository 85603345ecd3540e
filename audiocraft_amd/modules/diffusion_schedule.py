"""Noise schedule, reverse diffusion process and sample processors of MultiBandDiffusion on MI355X -- host side.

Mirrors `audiocraft/modules/diffusion_schedule.py:20-272` (`betas_from_alpha_bar`, `SampleProcessor`, `MultiBandProcessor`,
`NoiseSchedule`, `TrainingItem`).  The scalar schedule arithmetic (betas, cumulative products, per-step coefficients: a few
floats per step) is host code; everything that touches the samples runs in libacmi:

 * one reverse step  previous = clamp((current - c * estimate) / sqrt(alpha) + sigma * noise)      acmi_ddpm_step (one pass)
 * julius.SplitBands (n_bands - 1 windowed-sinc low-passes at mel-spaced cut-offs, bands = differences)  acmi_fir_bank
 * per-band rescaling + re-summation (the bands themselves are never materialised)                  acmi_band_mix
 * per-band statistics                                                                              acmi_band_stats

The band filters follow julius 0.2.7 (`julius/bands.py`, `julius/lowpass.py`; a third-party dependency that is not part of the
reference tree): cut-offs `mel_frequencies(n_bands + 1, 0, sr / 2)[1:-1] / sr`, Hann-windowed sinc of half size
int(8 / min(cutoff) / 2), each filter normalised to sum 1, replicate padding, and the low-pass outputs differenced.
"""
import math
import random
import typing as tp
from collections import namedtuple

import torch

from .. import _C

TrainingItem = namedtuple("TrainingItem", "noisy noise step")


def betas_from_alpha_bar(alpha_bar):
    alphas = torch.cat([torch.Tensor([alpha_bar[0]]), alpha_bar[1:] / alpha_bar[:-1]])
    return 1 - alphas


def _hz_to_mel(f: float) -> float:
    return 2595.0 * math.log10(1.0 + f / 700.0)


def band_filters(sample_rate: float, n_bands: int) -> tp.Tuple[torch.Tensor, int]:
    """-> ([n_bands - 1, 2 half + 1] f32 low-pass filters of julius.SplitBands(sample_rate, n_bands), half)"""
    mels = torch.linspace(_hz_to_mel(0.), _hz_to_mel(sample_rate / 2), n_bands + 1)
    cutoffs = (700.0 * (10.0 ** (mels / 2595.0) - 1.0))[1:-1] / sample_rate
    half = int(8 / float(cutoffs.min()) / 2)
    window = torch.hann_window(2 * half + 1, periodic=False)
    t = torch.arange(-half, half + 1, dtype=torch.float32)
    rows = []
    for c in cutoffs.tolist():
        arg = 2 * c * math.pi * t
        f = 2 * c * window * torch.where(arg == 0, torch.ones_like(arg), torch.sin(arg) / arg)
        rows.append(f / f.sum())
    return torch.stack(rows).contiguous(), half


class SplitBands:
    """julius.SplitBands on the accelerator: `lows(x)` = the n_bands - 1 low-pass outputs [n_bands - 1, *x.shape];
    band 0 = low 0, band i = low i - low i-1, band n-1 = x - low n-2 (so that the bands sum to x)."""

    def __init__(self, sample_rate: float, n_bands: int):
        self.sample_rate, self.n_bands = sample_rate, n_bands
        self._host = band_filters(sample_rate, n_bands) if n_bands > 1 else (None, 0)
        self._dev: tp.Dict[torch.device, torch.Tensor] = {}

    def lows(self, x: torch.Tensor) -> tp.Optional[torch.Tensor]:
        if self.n_bands == 1:
            return None
        if not x.is_cuda:
            raise RuntimeError("SplitBands runs on the MI355X only (acmi_fir_bank); no CPU fallback")
        if x.device not in self._dev:
            self._dev[x.device] = self._host[0].to(x.device)
        flat = x.reshape(-1, x.shape[-1])
        return _C.fir_bank(flat, self._dev[x.device]).reshape(self.n_bands - 1, *x.shape)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        """The materialised bands [n_bands, *x.shape] (API parity with julius; the processors below never need them)."""
        x = x.float().contiguous()
        lows = self.lows(x)
        if lows is None:
            return x[None]
        out = []
        for i in range(self.n_bands):
            g = torch.zeros(self.n_bands, device=x.device)
            g[i] = 1.
            out.append(_C.band_mix(x, lows, g))
        return torch.stack(out)

    def stats(self, x: torch.Tensor, lows: tp.Optional[torch.Tensor]) -> torch.Tensor:
        """-> host f64 [n_bands, 2]: per band (sum, sum of squares) over every element"""
        if lows is None:
            raise NotImplementedError("band statistics of a single band (n_bands = 1)")
        return _C.band_stats(x, lows)


class SampleProcessor(torch.nn.Module):
    def project_sample(self, x: torch.Tensor):
        """Project the original sample to the 'space' where the diffusion will happen."""
        return x

    def return_sample(self, z: torch.Tensor):
        """Project back from diffusion space to the actual sample space."""
        return z


class MultiBandProcessor(SampleProcessor):
    """diffusion_schedule.py:35-109: per mel band, rescale to the power of Gaussian noise in that band (running statistics
    over the first `num_samples` samples; frozen afterwards and in every released checkpoint)."""

    def __init__(self, n_bands: int = 8, sample_rate: float = 24_000, num_samples: int = 10_000,
                 power_std: tp.Union[float, tp.List[float], torch.Tensor] = 1.):
        super().__init__()
        self.n_bands = n_bands
        self.split_bands = SplitBands(sample_rate, n_bands=n_bands)
        self.num_samples = num_samples
        if isinstance(power_std, list):
            assert len(power_std) == n_bands
            power_std = torch.tensor(power_std)
        self.power_std = power_std
        self.register_buffer('counts', torch.zeros(1))
        self.register_buffer('sum_x', torch.zeros(n_bands))
        self.register_buffer('sum_x2', torch.zeros(n_bands))
        self.register_buffer('sum_target_x2', torch.zeros(n_bands))

    # The reference's processor holds julius.SplitBands, an nn.Module whose low-pass bank is a persistent buffer: released
    # `processor_state` dicts carry `split_bands.lowpass.filters` [n_bands - 1, 1, 2 half + 1] next to the four statistics
    # (julius/lowpass.py registers it; the reference strict-loads these files, diffusion_schedule.py:35-60).  Same key
    # here, both ways: written by state_dict(), and on load the checkpoint's bank REPLACES the computed one, as
    # load_state_dict does for a buffer in the reference.
    _FILTERS_KEY = 'split_bands.lowpass.filters'

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        if self.n_bands > 1:
            destination[prefix + self._FILTERS_KEY] = self.split_bands._host[0][:, None, :].clone()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        key = prefix + self._FILTERS_KEY
        if key in state_dict:
            f = state_dict.pop(key)   # (the caller's dict is a shallow copy made by load_state_dict)
            want = self.split_bands._host[0]
            if self.n_bands == 1 or f.dim() != 3 or f.shape[0] != self.n_bands - 1 or f.shape[1] != 1 or f.shape[2] % 2 != 1:
                error_msgs.append(f'{key}: shape {tuple(f.shape)} is not a bank of {self.n_bands - 1} odd-length low-pass filters')
            else:
                bank = f[:, 0, :].detach().to('cpu', torch.float32).contiguous()
                self.split_bands._host = (bank, (bank.shape[1] - 1) // 2)
                self.split_bands._dev = {}
                self.filters_max_abs_diff = float((bank - want).abs().max()) if bank.shape == want.shape else float('inf')
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    @property
    def mean(self):
        return self.sum_x / self.counts

    @property
    def std(self):
        return (self.sum_x2 / self.counts - self.mean ** 2).clamp(min=0).sqrt()

    @property
    def target_std(self):
        return self.sum_target_x2 / self.counts

    def _power(self, device):
        p = self.power_std
        return p.to(device) if isinstance(p, torch.Tensor) else p

    def project_sample(self, x: torch.Tensor):
        assert x.dim() == 3
        x = x.float().contiguous()
        lows = self.split_bands.lows(x)
        if self.counts.item() < self.num_samples:
            ref = torch.randn_like(x)
            per = x.shape[1] * x.shape[2]
            st = self.split_bands.stats(x, lows).to(x.device, torch.float32)
            st_ref = self.split_bands.stats(ref, self.split_bands.lows(ref)).to(x.device, torch.float32)
            self.counts += len(x)
            self.sum_x += st[:, 0] / per
            self.sum_x2 += st[:, 1] / per
            self.sum_target_x2 += st_ref[:, 1] / per
        rescale = (self.target_std / self.std.clamp(min=1e-12)) ** self._power(x.device)
        # sum_b (band_b - mean_b) * rescale_b
        return _C.band_mix(x, lows, rescale.float().contiguous().to(x.device), offset=-float((self.mean * rescale).sum()))

    def return_sample(self, x: torch.Tensor):
        assert x.dim() == 3
        x = x.float().contiguous()
        lows = self.split_bands.lows(x)
        rescale = (self.std / self.target_std) ** self._power(x.device)
        return _C.band_mix(x, lows, rescale.float().contiguous().to(x.device), offset=float(self.mean.sum()))


class NoiseSchedule:
    """diffusion_schedule.py:112-272.  `betas` lives on the host (the per-step coefficients are scalars handed to
    acmi_ddpm_step by value); `noise_source(like) -> tensor` replaces `torch.randn_like` when given (tests replay the
    reference's draws through it)."""

    def __init__(self, beta_t0: float = 1e-4, beta_t1: float = 0.02, num_steps: int = 1000, variance: str = 'beta',
                 clip: float = 5., rescale: float = 1., device='cuda', beta_exp: float = 1, repartition: str = "power",
                 alpha_sigmoid: dict = {}, n_bands: tp.Optional[int] = None,
                 sample_processor: SampleProcessor = SampleProcessor(), noise_scale: float = 1.0, **kwargs):
        beta_t0, beta_t1, beta_exp = float(beta_t0), float(beta_t1), float(beta_exp)   # YAML 1.1 reads '1e-05' as a string
        self.beta_t0, self.beta_t1 = beta_t0, beta_t1
        self.variance = variance
        self.num_steps = num_steps
        self.clip = clip
        self.sample_processor = sample_processor
        self.rescale = rescale
        self.n_bands = n_bands
        self.noise_scale = noise_scale
        self.device = device
        assert n_bands is None
        if repartition == "power":
            self.betas = torch.linspace(beta_t0 ** (1 / beta_exp), beta_t1 ** (1 / beta_exp), num_steps, dtype=torch.float) ** beta_exp
        else:
            raise RuntimeError('Not implemented')
        self.rng = random.Random(1234)
        self.noise_source: tp.Optional[tp.Callable[[torch.Tensor], torch.Tensor]] = None

    def get_beta(self, step: tp.Union[int, torch.Tensor]):
        return self.betas[step]

    def get_initial_noise(self, x: torch.Tensor):
        return torch.randn_like(x)

    def get_alpha_bar(self, step: tp.Optional[tp.Union[int, torch.Tensor]] = None) -> torch.Tensor:
        if step is None:
            return (1 - self.betas).cumprod(dim=-1)
        if type(step) is int:
            return (1 - self.betas[:step + 1]).prod()
        return (1 - self.betas).cumprod(dim=0)[step.cpu()].view(-1, 1, 1)

    def _randn_like(self, x: torch.Tensor) -> torch.Tensor:
        return self.noise_source(x) if self.noise_source is not None else torch.randn_like(x)

    def get_training_item(self, x: torch.Tensor, tensor_step: bool = False) -> TrainingItem:
        """Noisy item for training (:168-190).  One step for the whole batch only (`tensor_step=False`)."""
        if tensor_step:
            raise NotImplementedError("get_training_item(tensor_step=True): per-item steps are a training-only feature")
        step = self.rng.randrange(self.num_steps)
        alpha_bar = float(self.get_alpha_bar(step))
        x = self.sample_processor.project_sample(x)
        noise = self._randn_like(x)
        # (sqrt(ab) / rescale) x + sqrt(1 - ab) noise_scale noise, as one acmi_ddpm_step with a zero estimate weight
        noisy = _C.ddpm_step(x, x, noise, torch.empty_like(x), 0.0, self.rescale / math.sqrt(alpha_bar),
                             math.sqrt(1 - alpha_bar) * self.noise_scale, 0.0, 1.0, 1.0)
        return TrainingItem(noisy, noise, step)

    def _step(self, model, current: torch.Tensor, step: int, condition, c_est: float, sqrt_alpha: float, sigma2: float,
              est_scale: float) -> torch.Tensor:
        with torch.no_grad():
            estimate = model(current, step, condition=condition).sample
        noise = self._randn_like(current) if sigma2 > 0 else None
        out_scale = self.rescale if step == 0 else 1.0
        return _C.ddpm_step(current, estimate, noise, torch.empty_like(current), c_est, sqrt_alpha,
                            math.sqrt(sigma2) * self.noise_scale if sigma2 > 0 else 0.0, float(self.clip or 0.0), est_scale, out_scale)

    def generate(self, model: torch.nn.Module, initial: tp.Optional[torch.Tensor] = None,
                 condition: tp.Optional[torch.Tensor] = None, return_list: bool = False):
        """Full DDPM reverse process (:192-237)."""
        alpha_bar = self.get_alpha_bar(step=self.num_steps - 1)
        current = initial.float().contiguous()
        iterates = [initial]
        for step in range(self.num_steps)[::-1]:
            alpha = 1 - self.betas[step]
            previous_alpha_bar = self.get_alpha_bar(step=step - 1)
            if step == 0:
                sigma2 = 0.
            elif self.variance == 'beta':
                sigma2 = float(1 - alpha)
            elif self.variance == 'beta_tilde':
                sigma2 = float((1 - previous_alpha_bar) / (1 - alpha_bar) * (1 - alpha))
            elif self.variance == 'none':
                sigma2 = 0.
            else:
                raise ValueError(f'Invalid variance type {self.variance}')
            current = self._step(model, current, step, condition, float((1 - alpha) / (1 - alpha_bar).sqrt()), float(alpha.sqrt()),
                                 sigma2, 1.0)
            alpha_bar = previous_alpha_bar
            if return_list:
                iterates.append(current.cpu())
        return iterates if return_list else self.sample_processor.return_sample(current)

    def generate_subsampled(self, model: torch.nn.Module, initial: torch.Tensor, step_list: tp.Optional[list] = None,
                            condition: tp.Optional[torch.Tensor] = None, return_list: bool = False):
        """Reverse process through the Markov chain states of `step_list` only (:239-272)."""
        if step_list is None:
            step_list = list(range(1000))[::-50] + [0]
        alpha_bar = self.get_alpha_bar(step=self.num_steps - 1)
        alpha_bars_subsampled = (1 - self.betas).cumprod(dim=0)[list(reversed(step_list))]
        betas_subsampled = betas_from_alpha_bar(alpha_bars_subsampled)
        initial = initial.float().contiguous()
        # current = initial * noise_scale
        current = initial if self.noise_scale == 1.0 else _C.ddpm_step(initial, initial, None, torch.empty_like(initial), 0.0,
                                                                       1.0 / self.noise_scale, 0.0, 0.0, 1.0, 1.0)
        iterates = [current]
        for idx, step in enumerate(step_list[:-1]):
            alpha = 1 - betas_subsampled[-1 - idx]
            previous_alpha_bar = self.get_alpha_bar(step_list[idx + 1])
            if step == step_list[-2]:
                sigma2 = 0.
                previous_alpha_bar = torch.tensor(1.0)
            else:
                sigma2 = float((1 - previous_alpha_bar) / (1 - alpha_bar) * (1 - alpha))
            current = self._step(model, current, step, condition, float((1 - alpha) / (1 - alpha_bar).sqrt()), float(alpha.sqrt()),
                                 sigma2, self.noise_scale)
            alpha_bar = previous_alpha_bar
            if return_list:
                iterates.append(current.cpu())
        return iterates if return_list else self.sample_processor.return_sample(current)
