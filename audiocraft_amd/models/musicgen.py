"""MusicGen on MI355X: the user-facing text / melody / continuation generation API.

API mirror of `audiocraft.models.musicgen.MusicGen` (reference audiocraft/models/musicgen.py:40-338):
`get_pretrained`, `set_generation_params`, `generate`, `generate_unconditional`, `generate_continuation`,
`generate_with_chroma`, `generate_audio`, the progress callback and generation beyond `max_duration`.
Behavioural contract kept from the reference: every sample carries a `self_wav` condition (a null
1-sample wav when no melody is given, musicgen.py:211-218); durations above `max_duration` are served
by overlapping windows that advance by `extend_stride` seconds, each window prompted with the tail of
the previous one, a melody being tiled periodically so every window sees a full-length condition
(musicgen.py:290-337).
"""
import typing as tp
import warnings

import torch

from ..modules.conditioners import ConditioningAttributes, WavCondition
from . import builders
from .encodec import CompressionModel, InterleaveStereoCompressionModel
from .genmodel import BaseGenModel, convert_audio
from .lm import LMModel

MelodyList = tp.List[tp.Optional[torch.Tensor]]
MelodyType = tp.Union[torch.Tensor, MelodyList]

# released architectures: name -> (LM scale, melody conditioning?); a "-stereo-" name adds the stereo variant
_ARCH = {
    'facebook/musicgen-small': ('small', False), 'facebook/musicgen-medium': ('medium', False),
    'facebook/musicgen-large': ('large', False), 'facebook/musicgen-melody': ('medium', True),
    'facebook/musicgen-melody-large': ('large', True),
}
_ARCH.update({k.replace('musicgen-', 'musicgen-stereo-'): v for k, v in list(_ARCH.items())})
_SHORT_NAMES = {"small": "facebook/musicgen-small", "medium": "facebook/musicgen-medium",
                "large": "facebook/musicgen-large", "melody": "facebook/musicgen-melody"}


class MusicGen(BaseGenModel):
    def __init__(self, name: str, compression_model: CompressionModel, lm: LMModel,
                 max_duration: tp.Optional[float] = None):
        super().__init__(name, compression_model, lm, max_duration)
        self.set_generation_params(duration=15)  # reference default

    # ------------------------------------------------------------------------------------- construction
    @staticmethod
    def get_pretrained(name: str = 'facebook/musicgen-melody', device=None, weight_dtype=None):
        """'debug' builds the reference's debug geometry (musicgen.py:76-80).  Anything else is a released
        model name or a directory holding `state_dict.bin` + `compression_state_dict.bin` in the reference
        export format, resolved on disk only (see `loaders.py`; there is no network here)."""
        device = 'cuda' if device is None else device
        if name in _SHORT_NAMES:     # reference musicgen.py:83-87
            warnings.warn("MusicGen pretrained model relying on deprecated checkpoint mapping. "
                          f"Please use full pre-trained id instead: facebook/musicgen-{name}")
            name = _SHORT_NAMES[name]
        if name == 'debug':
            return MusicGen(name, builders.get_debug_compression_model(device), builders.get_debug_lm_model(device),
                            max_duration=30)
        from . import loaders
        lm = loaders.load_lm_model(name, device=device, weight_dtype=weight_dtype)
        # max_duration and the stereo codec wrapper come from the checkpoint's experiment config (BaseGenModel)
        return MusicGen(name, loaders.load_compression_model(name, device=device), lm)

    @staticmethod
    def get_random_init(name: str = 'facebook/musicgen-medium', device='cuda', weight_dtype=torch.bfloat16,
                        text_len: int = 16, seed: int = 0):
        """Architecture of a released model with seeded random weights and synthetic conditioners: what
        bench.py runs, since neither checkpoints nor T5 weights exist offline."""
        scale, melody = _ARCH[name]
        stereo = '-stereo-' in name
        torch.manual_seed(seed)
        lm = builders.get_lm_model(builders.musicgen_lm_cfg(scale, melody, synthetic=True, text_len=text_len,
                                                            stereo=stereo), device, weight_dtype)
        codec = builders.get_compression_model(builders.ENCODEC_32KHZ, device)
        if stereo:  # left / right through the mono codec, codebooks interleaved (reference encodec.py:397-506)
            codec = InterleaveStereoCompressionModel(codec)
        return MusicGen(name, codec, lm, max_duration=30)

    # ------------------------------------------------------------------------------------- parameters
    def set_style_conditioner_params(self, eval_q: int = 3, excerpt_length: float = 3.0, ds_factor: tp.Optional[int] = None,
                                     encodec_n_q: tp.Optional[int] = None) -> None:
        """reference musicgen.py:134-153 (MusicGen-Style): forwarded to the `self_wav` conditioner's `set_params`.  The style
        conditioner is a model of its own outside this path; the call is accepted for any plugged-in
        `self_wav` conditioner that offers `set_params` and fails like the reference's assert otherwise."""
        cond = self.lm.condition_provider.conditioners['self_wav'] if 'self_wav' in self.lm.condition_provider.conditioners else None
        assert cond is not None and hasattr(cond, 'set_params'), "Only use this function if you model is MusicGen-Style"
        cond.set_params(eval_q=eval_q, excerpt_length=excerpt_length, ds_factor=ds_factor, encodec_n_q=encodec_n_q)

    def set_generation_params(self, use_sampling: bool = True, top_k: int = 250, top_p: float = 0.0,
                              temperature: float = 1.0, duration: float = 30.0, cfg_coef: float = 3.0,
                              cfg_coef_beta: tp.Optional[float] = None, two_step_cfg: bool = False,
                              extend_stride: float = 18):
        """Same knobs and defaults as the reference (musicgen.py:96-132)."""
        assert extend_stride < self.max_duration, "Cannot stride by more than max generation duration."
        self.extend_stride, self.duration = extend_stride, duration
        self.generation_params = dict(use_sampling=use_sampling, temp=temperature, top_k=top_k, top_p=top_p,
                                      cfg_coef=cfg_coef, two_step_cfg=two_step_cfg, cfg_coef_beta=cfg_coef_beta)

    # ------------------------------------------------------------------------------------- melody entry point
    def generate_with_chroma(self, descriptions: tp.List[str], melody_wavs: MelodyType, melody_sample_rate: int,
                             progress: bool = False, return_tokens: bool = False):
        """Text + melody conditioning (reference musicgen.py:155-191).  `melody_wavs`: [B, C, T], [C, T] or a
        list of [C, T] / None."""
        if torch.is_tensor(melody_wavs):
            if melody_wavs.dim() == 2:
                melody_wavs = melody_wavs[None]
            if melody_wavs.dim() != 3:
                raise ValueError("Melody wavs should have a shape [B, C, T].")
            melody_wavs = list(melody_wavs)
        elif any(m is not None and m.dim() != 2 for m in melody_wavs):
            raise AssertionError("One melody in the list has the wrong number of dims.")
        melody_wavs = [None if m is None else convert_audio(m, melody_sample_rate, self.sample_rate, self.audio_channels)
                       for m in melody_wavs]
        return self._run(descriptions, None, progress, return_tokens, expect_prompt=False, melody_wavs=melody_wavs)

    # ------------------------------------------------------------------------------------- inputs
    def _wav_condition(self, melody: tp.Optional[torch.Tensor]) -> WavCondition:
        if melody is None:  # the null condition every non-melody sample carries
            return WavCondition(torch.zeros((1, 1, 1), device=self.device), torch.tensor([0], device=self.device),
                                sample_rate=[self.sample_rate], path=[None])
        return WavCondition(melody[None].to(device=self.device), torch.tensor([melody.shape[-1]], device=self.device),
                            sample_rate=[self.sample_rate], path=[None])

    def _prepare_tokens_and_attributes(self, descriptions: tp.Sequence[tp.Optional[str]],
                                       prompt: tp.Optional[torch.Tensor], melody_wavs: tp.Optional[MelodyList] = None):
        """-> (attributes, prompt tokens); reference musicgen.py:193-249.  Every sample carries a `self_wav`
        condition: its melody, or the null 1-sample wav."""
        if melody_wavs is not None:
            if 'self_wav' not in self.lm.condition_provider.conditioners:
                raise RuntimeError("This model doesn't support melody conditioning. Use the `melody` model.")
            assert len(melody_wavs) == len(descriptions), \
                f"number of melody wavs must match number of descriptions! " \
                f"got melody len={len(melody_wavs)}, and descriptions len={len(descriptions)}"
        attributes = []
        for i, text in enumerate(descriptions):
            attr = ConditioningAttributes(text={'description': text})
            attr.wav['self_wav'] = self._wav_condition(None if melody_wavs is None else melody_wavs[i])
            attributes.append(attr)
        return attributes, self._encode_prompt(descriptions, prompt)

    # ------------------------------------------------------------------------------------- windows
    def _generate_tokens(self, attributes, prompt_tokens, progress: bool = False) -> torch.Tensor:
        self._melodies = [a.wav['self_wav'] for a in attributes]   # the un-tiled melodies of this call
        try:
            return super()._generate_tokens(attributes, prompt_tokens, progress)
        finally:
            self._melodies = None

    def _window_attributes(self, attributes: tp.List[ConditioningAttributes], t_start: float) -> None:
        """Periodic extension of each melody so that the window starting at `t_start` sees `max_duration`
        seconds of it (reference musicgen.py:309-324)."""
        want = int(self.max_duration * self.sample_rate)
        for attr, mel in zip(attributes, self._melodies):
            n = int(mel.length.item())
            if n == 0:
                continue
            idx = (int(t_start * self.sample_rate) + torch.arange(want, device=self.device)) % n
            attr.wav['self_wav'] = WavCondition(mel[0][..., idx], torch.full_like(mel[1], want),
                                                [self.sample_rate] * mel[0].size(0), [None], [0.])
