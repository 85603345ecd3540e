"""Base API of the generative models (`audiocraft.models.genmodel.BaseGenModel`, reference
audiocraft/models/genmodel.py:28-267): generation parameters, prompt/attribute preparation,
token generation and decoding to audio.  No autocast context: precision is a property of the packed
weights (`LMModel.weight_dtype`), not of a tracing mode."""
import typing as tp
from abc import ABC, abstractmethod

import torch

from ..modules.conditioners import ConditioningAttributes
from .encodec import CompressionModel
from .lm import LMModel


def convert_audio(wav: torch.Tensor, from_rate: float, to_rate: float, to_channels: int) -> torch.Tensor:
    """Channel conversion of `audiocraft.data.audio_utils.convert_audio` (reference
    audiocraft/data/audio_utils.py:18-59).  Resampling is `julius.resample_frac` in the reference, a
    third-party routine outside this package: rates must already match."""
    *shape, src_channels, length = wav.shape
    if src_channels != to_channels:
        if to_channels == 1:
            wav = wav.mean(dim=-2, keepdim=True)
        elif src_channels == 1:
            wav = wav.expand(*shape, to_channels, length)
        elif src_channels >= to_channels:
            wav = wav[..., :to_channels, :]
        else:
            raise ValueError('The audio file has less channels than requested but is not mono.')
    if int(from_rate) != int(to_rate):
        raise NotImplementedError("sample-rate conversion (julius.resample_frac in the reference) is outside "
                                  f"this package: resample {from_rate} -> {to_rate} Hz before calling")
    return wav


class BaseGenModel(ABC):
    def __init__(self, name: str, compression_model: CompressionModel, lm: LMModel,
                 max_duration: tp.Optional[float] = None):
        self.name = name
        self.compression_model = compression_model
        self.lm = lm
        self.cfg = None
        self.compression_model.eval()
        self.lm.eval()
        if max_duration is None:
            raise ValueError("You must provide max_duration when building directly your GenModel")
        assert max_duration is not None
        self.max_duration: float = max_duration
        self.duration = self.max_duration
        self.extend_stride: tp.Optional[float] = None
        self.device = next(iter(lm.parameters())).device
        self.generation_params: dict = {}
        self._progress_callback: tp.Optional[tp.Callable[[int, int], None]] = None

    @property
    def frame_rate(self) -> float:
        return self.compression_model.frame_rate

    @property
    def sample_rate(self) -> int:
        return self.compression_model.sample_rate

    @property
    def audio_channels(self) -> int:
        return self.compression_model.channels

    def set_custom_progress_callback(self, progress_callback: tp.Optional[tp.Callable[[int, int], None]] = None):
        self._progress_callback = progress_callback

    @abstractmethod
    def set_generation_params(self, *args, **kwargs):
        raise NotImplementedError("No base implementation for setting generation params.")

    @staticmethod
    @abstractmethod
    def get_pretrained(name: str, device=None):
        raise NotImplementedError("No base implementation for getting pretrained model")

    @abstractmethod
    def _prepare_tokens_and_attributes(self, descriptions, prompt):
        ...

    @abstractmethod
    def _generate_tokens(self, attributes, prompt_tokens, progress: bool = False) -> torch.Tensor:
        ...

    @torch.no_grad()
    def generate_unconditional(self, num_samples: int, progress: bool = False, return_tokens: bool = False):
        descriptions: tp.List[tp.Optional[str]] = [None] * num_samples
        attributes, prompt_tokens = self._prepare_tokens_and_attributes(descriptions, None)
        tokens = self._generate_tokens(attributes, prompt_tokens, progress)
        if return_tokens:
            return self.generate_audio(tokens), tokens
        return self.generate_audio(tokens)

    @torch.no_grad()
    def generate(self, descriptions: tp.List[str], progress: bool = False, return_tokens: bool = False):
        attributes, prompt_tokens = self._prepare_tokens_and_attributes(descriptions, None)
        assert prompt_tokens is None
        tokens = self._generate_tokens(attributes, prompt_tokens, progress)
        if return_tokens:
            return self.generate_audio(tokens), tokens
        return self.generate_audio(tokens)

    @torch.no_grad()
    def generate_continuation(self, prompt: torch.Tensor, prompt_sample_rate: int,
                              descriptions: tp.Optional[tp.List[tp.Optional[str]]] = None,
                              progress: bool = False, return_tokens: bool = False):
        if prompt.dim() == 2:
            prompt = prompt[None]
        if prompt.dim() != 3:
            raise ValueError("prompt should have 3 dimensions: [B, C, T] (C = 1).")
        prompt = convert_audio(prompt, prompt_sample_rate, self.sample_rate, self.audio_channels)
        if descriptions is None:
            descriptions = [None] * len(prompt)
        attributes, prompt_tokens = self._prepare_tokens_and_attributes(descriptions, prompt)
        assert prompt_tokens is not None
        tokens = self._generate_tokens(attributes, prompt_tokens, progress)
        if return_tokens:
            return self.generate_audio(tokens), tokens
        return self.generate_audio(tokens)

    @torch.no_grad()
    def generate_audio(self, gen_tokens: torch.Tensor) -> torch.Tensor:
        """Generate Audio from tokens (reference genmodel.py:262-267)."""
        assert gen_tokens.dim() == 3
        return self.compression_model.decode(gen_tokens, None)
