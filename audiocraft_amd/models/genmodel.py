"""Generation front-end shared by the generative models.

API mirror of `audiocraft.models.genmodel.BaseGenModel` (reference audiocraft/models/genmodel.py:28-267):
same public methods, properties and argument meaning.  The implementation is organised around one
private driver (`_run`) that every `generate*` entry point funnels into, and there is no autocast
context: precision is a property of the packed weights (`LMModel.weight_dtype`), not of a tracing mode.
"""
import typing as tp
from abc import ABC, abstractmethod

import torch

from .encodec import CompressionModel
from .lm import LMModel

ProgressFn = tp.Callable[[int, int], None]


def convert_audio(wav: torch.Tensor, from_rate: float, to_rate: float, to_channels: int) -> torch.Tensor:
    """Channel handling of `audiocraft.data.audio_utils.convert_audio` (reference
    audiocraft/data/audio_utils.py:18-59): down-mix to mono, replicate mono, or keep the first channels.
    Resampling is `julius.resample_frac` in the reference -- a third-party routine that is not part of
    this package -- so the two rates must already agree."""
    have = wav.shape[-2]
    if have != to_channels:
        if to_channels == 1:
            wav = wav.mean(dim=-2, keepdim=True)
        elif have == 1:
            wav = wav.expand(*wav.shape[:-2], to_channels, wav.shape[-1])
        elif have > to_channels:
            wav = wav[..., :to_channels, :]
        else:
            raise ValueError('The audio file has less channels than requested but is not mono.')
    if int(from_rate) != int(to_rate):
        raise NotImplementedError(f"resample {from_rate} -> {to_rate} Hz before calling: sample-rate conversion "
                                  "(julius in the reference) is outside this package")
    return wav


class BaseGenModel(ABC):
    """Tokens-from-LM + audio-from-codec generator.

    Args (as in the reference): name, compression_model, lm, max_duration (seconds the LM was trained on;
    longer requests are served by windowed generation in the subclass)."""

    def __init__(self, name: str, compression_model: CompressionModel, lm: LMModel,
                 max_duration: tp.Optional[float] = None):
        if max_duration is None:
            raise ValueError("You must provide max_duration when building directly your GenModel")
        self.name = name
        self.cfg = None
        self.compression_model = compression_model.eval()
        self.lm = lm.eval()
        self.max_duration: float = float(max_duration)
        self.duration: float = self.max_duration
        self.extend_stride: tp.Optional[float] = None
        self.generation_params: dict = {}
        self._progress_callback: tp.Optional[ProgressFn] = None
        self.device = next(iter(lm.parameters())).device

    # -- codec properties ---------------------------------------------------------------------------
    frame_rate = property(lambda self: self.compression_model.frame_rate, doc="Token frames per second.")
    sample_rate = property(lambda self: self.compression_model.sample_rate, doc="Audio sample rate.")
    audio_channels = property(lambda self: self.compression_model.channels, doc="Audio channels.")

    def set_custom_progress_callback(self, progress_callback: tp.Optional[ProgressFn] = None):
        """Override the default progress printer (called with (generated_tokens, tokens_to_generate))."""
        self._progress_callback = progress_callback

    # -- to be provided by the concrete model -------------------------------------------------------
    @abstractmethod
    def set_generation_params(self, *args, **kwargs):
        raise NotImplementedError("No base implementation for setting generation params.")

    @staticmethod
    @abstractmethod
    def get_pretrained(name: str, device=None):
        raise NotImplementedError("No base implementation for getting pretrained model")

    @abstractmethod
    def _prepare_tokens_and_attributes(self, descriptions, prompt):
        """-> (list of ConditioningAttributes, prompt tokens [B, K, T0] or None)"""

    @abstractmethod
    def _generate_tokens(self, attributes, prompt_tokens, progress: bool = False) -> torch.Tensor:
        """-> tokens [B, K, T]"""

    # -- single driver ------------------------------------------------------------------------------
    def _run(self, descriptions, prompt_wav, progress: bool, return_tokens: bool, expect_prompt: bool, **prep_kw):
        with torch.no_grad():
            attributes, prompt_tokens = self._prepare_tokens_and_attributes(descriptions, prompt_wav, **prep_kw)
            assert (prompt_tokens is not None) == expect_prompt
            tokens = self._generate_tokens(attributes, prompt_tokens, progress)
            audio = self.generate_audio(tokens)
        return (audio, tokens) if return_tokens else audio

    # -- public entry points (reference genmodel.py:135-191, 262-267) --------------------------------
    def generate_unconditional(self, num_samples: int, progress: bool = False, return_tokens: bool = False):
        """`num_samples` generations with no text conditioning (descriptions are all None)."""
        return self._run([None] * num_samples, None, progress, return_tokens, expect_prompt=False)

    def generate(self, descriptions: tp.List[str], progress: bool = False, return_tokens: bool = False):
        """One generation per text description."""
        return self._run(descriptions, None, progress, return_tokens, expect_prompt=False)

    def generate_continuation(self, prompt: torch.Tensor, prompt_sample_rate: int,
                              descriptions: tp.Optional[tp.List[tp.Optional[str]]] = None,
                              progress: bool = False, return_tokens: bool = False):
        """Continue the audio `prompt` ([B, C, T] or [C, T]), optionally guided by descriptions."""
        if prompt.dim() == 2:
            prompt = prompt[None]
        if prompt.dim() != 3:
            raise ValueError("prompt should have 3 dimensions: [B, C, T] (C = 1).")
        prompt = convert_audio(prompt, prompt_sample_rate, self.sample_rate, self.audio_channels)
        if descriptions is None:
            descriptions = [None] * len(prompt)
        return self._run(descriptions, prompt, progress, return_tokens, expect_prompt=True)

    def generate_audio(self, gen_tokens: torch.Tensor) -> torch.Tensor:
        """Decode tokens [B, K, T] to a waveform [B, C, T * hop] with the compression model."""
        assert gen_tokens.dim() == 3
        with torch.no_grad():
            return self.compression_model.decode(gen_tokens, None)
