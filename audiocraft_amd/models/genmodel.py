"""Generation front-end shared by the generative models.

API mirror of `audiocraft.models.genmodel.BaseGenModel` (reference audiocraft/models/genmodel.py:28-267):
same public methods, properties and argument meaning.  The implementation is organised around one
private driver (`_run`) that every `generate*` entry point funnels into and one windowed token generator with
a per-window hook, and there is no autocast
context: precision is a property of the packed weights (`LMModel.weight_dtype`), not of a tracing mode.
"""
import typing as tp
from abc import ABC, abstractmethod

import torch

from .. import _C
from ..modules.conditioners import ConditioningAttributes
from .encodec import CompressionModel, InterleaveStereoCompressionModel
from .lm import LMModel

ProgressFn = tp.Callable[[int, int], None]


from ..data_audio_utils import convert_audio  # noqa: E402,F401  (re-exported: MusicGen imports it from here)


def get_wrapped_compression_model(compression_model: CompressionModel, cfg: dict) -> CompressionModel:
    """Stereo wrapper / codebook count requested by an experiment config (reference builders.py:338-351)."""
    stereo = cfg.get('interleave_stereo_codebooks') or {}
    if stereo.get('use'):
        compression_model = InterleaveStereoCompressionModel(compression_model,
                                                             per_timestep=bool(stereo.get('per_timestep', False)))
    if cfg.get('compression_model_n_q') is not None:
        compression_model.set_num_codebooks(cfg['compression_model_n_q'])
    return compression_model


class BaseGenModel(ABC):
    """Tokens-from-LM + audio-from-codec generator.

    Args (as in the reference): name, compression_model, lm, max_duration (seconds the LM was trained on;
    longer requests are served by windowed generation in the subclass)."""

    def __init__(self, name: str, compression_model: CompressionModel, lm: LMModel,
                 max_duration: tp.Optional[float] = None):
        self.name = name
        # the LM of a released checkpoint carries the experiment config (`loaders.load_lm_model`): it says whether
        # the codec is the stereo wrapper and what duration the model was trained on (reference genmodel.py:49-62)
        self.cfg: tp.Optional[dict] = getattr(lm, 'cfg', None)
        if self.cfg is not None:
            compression_model = get_wrapped_compression_model(compression_model, self.cfg)
            if max_duration is None:
                max_duration = (self.cfg.get('dataset') or {}).get('segment_duration')
        if max_duration is None:
            raise ValueError("You must provide max_duration when building directly your GenModel")
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

    # -- inputs and tokens (reference genmodel.py:109-133, 193-260) ------------------------------------
    def _prepare_tokens_and_attributes(self, descriptions: tp.Sequence[tp.Optional[str]],
                                       prompt: tp.Optional[torch.Tensor]):
        """-> (one ConditioningAttributes per description, prompt tokens [B, K, T0] or None)."""
        attributes = [ConditioningAttributes(text={'description': text}) for text in descriptions]
        return attributes, self._encode_prompt(descriptions, prompt)

    def _encode_prompt(self, descriptions, prompt: tp.Optional[torch.Tensor]) -> tp.Optional[torch.Tensor]:
        if prompt is None:
            return None
        assert descriptions is None or len(descriptions) == len(prompt), "Prompt and nb. descriptions doesn't match"
        prompt_tokens, scale = self.compression_model.encode(prompt.to(self.device))
        assert scale is None
        return prompt_tokens

    def _lm_generate(self, prompt_tokens, attributes, n_frames: int, callback):
        return self.lm.generate(prompt_tokens, attributes, callback=callback, max_gen_len=n_frames,
                                **self.generation_params)

    def _window_attributes(self, attributes: tp.List[ConditioningAttributes], t_start: float) -> None:
        """Hook of the windowed generation: adapt the conditions to the window starting at `t_start` seconds
        (MusicGen tiles the melody); the default conditions do not depend on time."""

    def _generate_tokens(self, attributes: tp.List[ConditioningAttributes],
                         prompt_tokens: tp.Optional[torch.Tensor], progress: bool = False) -> torch.Tensor:
        """-> tokens [B, K, T].  Durations above `max_duration` are served by overlapping windows that advance by
        `extend_stride` seconds, each prompted with the tail of the previous one."""
        fps = self.frame_rate
        total_frames = int(self.duration * fps)
        if prompt_tokens is not None:
            assert prompt_tokens.shape[-1] <= int(min(self.duration, self.max_duration) * fps), \
                "Prompt is longer than audio to generate"
        frames_done = 0  # frames produced by earlier windows (offsets the progress report)

        def report(generated: int, to_generate: int):
            generated += frames_done
            if self._progress_callback is not None:
                self._progress_callback(generated, to_generate)
            else:
                print(f'{generated: 6d} / {to_generate: 6d}', end='\r')

        callback = report if progress else None
        if self.duration <= self.max_duration:
            return self._lm_generate(prompt_tokens, attributes, total_frames, callback)

        assert self.extend_stride is not None, "Stride should be defined to generate beyond max_duration"
        assert self.extend_stride < self.max_duration, "Cannot stride by more than max generation duration."
        stride_frames = int(fps * self.extend_stride)
        pieces = [] if prompt_tokens is None else [prompt_tokens]
        prompt_len = 0 if prompt_tokens is None else prompt_tokens.shape[-1]
        while frames_done + prompt_len < total_frames:
            t_start = frames_done / fps
            window_frames = int(min(self.duration - t_start, self.max_duration) * fps)
            self._window_attributes(attributes, t_start)
            window = self._lm_generate(prompt_tokens, attributes, window_frames, callback)
            pieces.append(window if prompt_tokens is None else window[..., prompt_tokens.shape[-1]:])
            prompt_tokens = window[..., stride_frames:]
            prompt_len = prompt_tokens.shape[-1]
            frames_done += stride_frames
        return torch.cat(pieces, dim=-1)

    # -- single driver ------------------------------------------------------------------------------
    @_C.exclusive   # one request at a time per process: conditioning, token generation and decode are not interleaved with another thread's
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
