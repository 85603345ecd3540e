"""MultiBandDiffusion: EnCodec tokens -> waveform through one diffusion process per frequency band.

API mirror of `audiocraft/models/multibanddiffusion.py:25-196` (`DiffusionProcess`, `MultiBandDiffusion` with `get_mbd_musicgen`,
`get_mbd_24khz`, `get_condition`, `get_emb`, `generate`, `re_eq`, `regenerate`, `tokens_to_wav`).  All sample arithmetic runs in
libacmi (models/unet.py, modules/diffusion_schedule.py); checkpoints are looked up on disk only (models/loaders.py).
"""
import typing as tp

import torch

from .. import _C
from ..data_audio_utils import resample_frac
from ..modules.diffusion_schedule import NoiseSchedule, SplitBands
from .encodec import CompressionModel
from .unet import DiffusionUnet


class DiffusionProcess:
    """Sampling for one diffusion model (multibanddiffusion.py:25-46)."""

    def __init__(self, model: DiffusionUnet, noise_schedule: NoiseSchedule) -> None:
        self.model = model
        self.schedule = noise_schedule

    def generate(self, condition: torch.Tensor, initial_noise: torch.Tensor, step_list: tp.Optional[tp.List[int]] = None):
        return self.schedule.generate_subsampled(model=self.model, initial=initial_noise, step_list=step_list, condition=condition)


class MultiBandDiffusion:
    """Sample from several diffusion models, one per band, and sum (multibanddiffusion.py:49-196)."""

    def __init__(self, DPs: tp.List[DiffusionProcess], codec_model: CompressionModel) -> None:
        self.DPs = DPs
        self.codec_model = codec_model
        self.device = next(self.codec_model.parameters()).device
        self._split: tp.Dict[tp.Tuple[int, int], SplitBands] = {}
        # `noise_source(like) -> tensor` replaces torch.randn_like for the initial noise when set (tests replay fixed draws)
        self.noise_source: tp.Optional[tp.Callable[[torch.Tensor], torch.Tensor]] = None

    @property
    def sample_rate(self) -> int:
        return self.codec_model.sample_rate

    @staticmethod
    def _assemble(codec_model, path: str, filename: str, device) -> 'MultiBandDiffusion':
        from .loaders import load_diffusion_models
        models, processors, cfgs = load_diffusion_models(path, filename=filename, device=device)
        DPs = []
        for model, processor, cfg in zip(models, processors, cfgs):
            schedule = NoiseSchedule(**cfg['schedule'], sample_processor=processor, device=device)
            DPs.append(DiffusionProcess(model=model, noise_schedule=schedule))
        return MultiBandDiffusion(DPs=DPs, codec_model=codec_model)

    @staticmethod
    def get_mbd_musicgen(device=None):
        """The diffusion decoders trained for MusicGen's 32 kHz codec (`facebook/multiband-diffusion`, mbd_musicgen_32khz.th)."""
        from .loaders import load_compression_model
        device = device or 'cuda'
        codec_model = load_compression_model('facebook/musicgen-small', device=device)
        return MultiBandDiffusion._assemble(codec_model, 'facebook/multiband-diffusion', 'mbd_musicgen_32khz.th', device)

    @staticmethod
    def get_mbd_24khz(bw: float = 3.0, device: tp.Optional[tp.Union[torch.device, str]] = None, n_q: tp.Optional[int] = None):
        """The diffusion decoders for EnCodec 24 kHz at 1.5 / 3 / 6 kbps (mbd_comp_{n_q}.pt).  The reference resolves the codec
        through `CompressionSolver.model_from_checkpoint('//pretrained/facebook/encodec_24khz')`, i.e. the HuggingFace
        EncodecModel wrapper: `CompressionModel.get_pretrained` does the same here (an audiocraft export on disk if there is
        one, else `transformers.EncodecModel.from_pretrained` from HuggingFace's local cache, its weights re-keyed into the
        SEANet / RVQ kernels of this package: `HFEncodecCompressionModel`)."""
        device = device or 'cuda'
        assert bw in [1.5, 3.0, 6.0], f"bandwidth {bw} not available"
        if n_q is not None:
            assert n_q in [2, 4, 8]
            assert {1.5: 2, 3.0: 4, 6.0: 8}[bw] == n_q, \
                f"bandwidth and number of codebooks missmatch to use n_q = {n_q} bw should be {n_q * (1.5 / 2)}"
        n_q = {1.5: 2, 3.0: 4, 6.0: 8}[bw]
        codec_model = CompressionModel.get_pretrained('facebook/encodec_24khz', device=device)
        codec_model.set_num_codebooks(n_q)
        return MultiBandDiffusion._assemble(codec_model, 'facebook/multiband-diffusion', f'mbd_comp_{n_q}.pt', device)

    @torch.no_grad()
    def get_condition(self, wav: torch.Tensor, sample_rate: int) -> torch.Tensor:
        """The conditioning (latent of the compression model) of a waveform."""
        if sample_rate != self.sample_rate:
            wav = resample_frac(wav, sample_rate, self.sample_rate)
        codes, scale = self.codec_model.encode(wav.to(self.device))
        assert scale is None, "Scaled compression models not supported."
        return self.get_emb(codes)

    @torch.no_grad()
    def get_emb(self, codes: torch.Tensor):
        return self.codec_model.decode_latent(codes)

    @_C.exclusive
    @torch.no_grad()
    def generate(self, emb: torch.Tensor, size: tp.Optional[torch.Size] = None, step_list: tp.Optional[tp.List[int]] = None):
        """Waveform from the latent embeddings: the sum over the bands' reverse processes."""
        if size is None:
            upsampling = int(self.codec_model.sample_rate / self.codec_model.frame_rate)
            size = torch.Size([emb.size(0), self.codec_model.channels, emb.size(-1) * upsampling])
        assert size[0] == emb.size(0)
        out = None
        like = torch.empty(size, device=self.device, dtype=torch.float32)
        for DP in self.DPs:
            noise = self.noise_source(like) if self.noise_source is not None else torch.randn_like(like)
            band = DP.generate(condition=emb, step_list=step_list, initial_noise=noise)
            out = band if out is None else _C.add_cropped(out, band, out=out)
        return out if out is not None else torch.zeros(size, device=self.device)

    @torch.no_grad()
    def re_eq(self, wav: torch.Tensor, ref: torch.Tensor, n_bands: int = 32, strictness: float = 1):
        """Match the per-band standard deviation of `wav` to `ref` (:150-164): two filter-bank passes, two statistics
        passes, one re-mix -- the 2 x n_bands band signals are never written."""
        key = (int(self.codec_model.sample_rate), n_bands)
        if key not in self._split:
            self._split[key] = SplitBands(sample_rate=self.codec_model.sample_rate, n_bands=n_bands)
        split = self._split[key]
        wav, ref = wav.float().contiguous(), ref.float().contiguous()
        lows = split.lows(wav)
        st, st_ref = split.stats(wav, lows), split.stats(ref, split.lows(ref))   # host f64 [n_bands, 2]

        def std(s, n):   # torch.std: unbiased
            return ((s[:, 1] - s[:, 0] ** 2 / n) / (n - 1)).clamp(min=0).sqrt()

        gains = (std(st_ref, ref.numel()) / std(st, wav.numel())) ** strictness
        return _C.band_mix(wav, lows, gains.float().to(wav.device))

    def regenerate(self, wav: torch.Tensor, sample_rate: int):
        """Compress and regenerate a waveform through the diffusion decoders."""
        if sample_rate != self.codec_model.sample_rate:
            wav = resample_frac(wav, sample_rate, self.codec_model.sample_rate)
        emb = self.get_condition(wav, sample_rate=self.codec_model.sample_rate)
        out = self.generate(emb, size=wav.size())[..., :wav.size(-1)]
        if sample_rate != self.codec_model.sample_rate:
            out = resample_frac(out, self.codec_model.sample_rate, sample_rate)
        return out

    def tokens_to_wav(self, tokens: torch.Tensor, n_bands: int = 32):
        """Waveform from discrete codes: diffusion decode, then EQ matching against the codec's own decode."""
        wav_encodec = self.codec_model.decode(tokens)
        condition = self.get_emb(tokens)
        wav_diffusion = self.generate(emb=condition, size=wav_encodec.size())
        return self.re_eq(wav=wav_diffusion, ref=wav_encodec, n_bands=n_bands)
