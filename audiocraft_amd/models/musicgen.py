"""MusicGen on MI355X: the user-facing generation API.

Mirror of `audiocraft.models.musicgen.MusicGen` (reference audiocraft/models/musicgen.py:40-338):
`get_pretrained`, `set_generation_params`, `generate`, `generate_unconditional`,
`generate_continuation`, `generate_with_chroma`, `generate_audio`, windowed generation beyond
`max_duration`, the progress callback.
"""
import os
import typing as tp

import torch

from ..modules.conditioners import ConditioningAttributes, WavCondition
from . import builders
from .encodec import CompressionModel
from .genmodel import BaseGenModel, convert_audio
from .lm import LMModel

MelodyList = tp.List[tp.Optional[torch.Tensor]]
MelodyType = tp.Union[torch.Tensor, MelodyList]

_ARCH = {  # name -> (LM scale, melody?)
    'facebook/musicgen-small': ('small', False), 'facebook/musicgen-medium': ('medium', False),
    'facebook/musicgen-large': ('large', False), 'facebook/musicgen-melody': ('medium', True),
    'facebook/musicgen-melody-large': ('large', True),
}
_HF_MODEL_CHECKPOINTS_MAP = {"small": "facebook/musicgen-small", "medium": "facebook/musicgen-medium",
                             "large": "facebook/musicgen-large", "melody": "facebook/musicgen-melody"}


class MusicGen(BaseGenModel):
    """MusicGen main model with convenient generation API (reference musicgen.py:40-54)."""

    def __init__(self, name: str, compression_model: CompressionModel, lm: LMModel,
                 max_duration: tp.Optional[float] = None):
        super().__init__(name, compression_model, lm, max_duration)
        self.set_generation_params(duration=15)  # default duration

    @staticmethod
    def get_pretrained(name: str = 'facebook/musicgen-melody', device=None, weight_dtype=None):
        """'debug' -> the reference's debug geometry (musicgen.py:76-80).  A released model name or a
        directory -> `state_dict.bin` + `compression_state_dict.bin` in the reference export format
        (`loaders.py`), looked up on disk only (no network in this environment)."""
        if device is None:
            device = 'cuda'
        if name in _HF_MODEL_CHECKPOINTS_MAP:
            name = _HF_MODEL_CHECKPOINTS_MAP[name]
        if name == 'debug':
            compression_model = builders.get_debug_compression_model(device)
            lm = builders.get_debug_lm_model(device)
            return MusicGen(name, compression_model, lm, max_duration=30)
        from . import loaders
        lm = loaders.load_lm_model(name, device=device, weight_dtype=weight_dtype)
        compression_model = loaders.load_compression_model(name, device=device)
        return MusicGen(name, compression_model, lm, max_duration=30)

    @staticmethod
    def get_random_init(name: str = 'facebook/musicgen-medium', device='cuda', weight_dtype=torch.bfloat16,
                        text_len: int = 16, seed: int = 0):
        """Architecture of a released model with seeded random weights and synthetic conditioners
        (no checkpoints / no T5 weights exist offline; used by bench.py)."""
        scale, melody = _ARCH[name]
        torch.manual_seed(seed)
        lm = builders.get_lm_model(builders.musicgen_lm_cfg(scale, melody, synthetic=True, text_len=text_len),
                                   device, weight_dtype)
        compression_model = builders.get_compression_model(builders.ENCODEC_32KHZ, device)
        return MusicGen(name, compression_model, lm, max_duration=30)

    def set_generation_params(self, use_sampling: bool = True, top_k: int = 250, top_p: float = 0.0,
                              temperature: float = 1.0, duration: float = 30.0, cfg_coef: float = 3.0,
                              cfg_coef_beta: tp.Optional[float] = None, two_step_cfg: bool = False,
                              extend_stride: float = 18):
        """reference musicgen.py:96-132"""
        assert extend_stride < self.max_duration, "Cannot stride by more than max generation duration."
        self.extend_stride = extend_stride
        self.duration = duration
        self.generation_params = {
            'use_sampling': use_sampling, 'temp': temperature, 'top_k': top_k, 'top_p': top_p,
            'cfg_coef': cfg_coef, 'two_step_cfg': two_step_cfg, 'cfg_coef_beta': cfg_coef_beta,
        }

    @torch.no_grad()
    def generate_with_chroma(self, descriptions: tp.List[str], melody_wavs: MelodyType, melody_sample_rate: int,
                             progress: bool = False, return_tokens: bool = False):
        """reference musicgen.py:155-191"""
        if isinstance(melody_wavs, torch.Tensor):
            if melody_wavs.dim() == 2:
                melody_wavs = melody_wavs[None]
            if melody_wavs.dim() != 3:
                raise ValueError("Melody wavs should have a shape [B, C, T].")
            melody_wavs = list(melody_wavs)
        else:
            for melody in melody_wavs:
                if melody is not None:
                    assert melody.dim() == 2, "One melody in the list has the wrong number of dims."
        melody_wavs = [convert_audio(wav, melody_sample_rate, self.sample_rate, self.audio_channels)
                       if wav is not None else None for wav in melody_wavs]
        attributes, prompt_tokens = self._prepare_tokens_and_attributes(descriptions=descriptions, prompt=None,
                                                                        melody_wavs=melody_wavs)
        assert prompt_tokens is None
        tokens = self._generate_tokens(attributes, prompt_tokens, progress)
        if return_tokens:
            return self.generate_audio(tokens), tokens
        return self.generate_audio(tokens)

    def _null_wav(self) -> WavCondition:
        return WavCondition(torch.zeros((1, 1, 1), device=self.device), torch.tensor([0], device=self.device),
                            sample_rate=[self.sample_rate], path=[None])

    @torch.no_grad()
    def _prepare_tokens_and_attributes(self, descriptions: tp.Sequence[tp.Optional[str]],
                                       prompt: tp.Optional[torch.Tensor],
                                       melody_wavs: tp.Optional[MelodyList] = None):
        """reference musicgen.py:193-249"""
        attributes = [ConditioningAttributes(text={'description': description}) for description in descriptions]
        if melody_wavs is None:
            for attr in attributes:
                attr.wav['self_wav'] = self._null_wav()
        else:
            if 'self_wav' not in self.lm.condition_provider.conditioners:
                raise RuntimeError("This model doesn't support melody conditioning. Use the `melody` model.")
            assert len(melody_wavs) == len(descriptions), \
                f"number of melody wavs must match number of descriptions! " \
                f"got melody len={len(melody_wavs)}, and descriptions len={len(descriptions)}"
            for attr, melody in zip(attributes, melody_wavs):
                if melody is None:
                    attr.wav['self_wav'] = self._null_wav()
                else:
                    attr.wav['self_wav'] = WavCondition(
                        melody[None].to(device=self.device), torch.tensor([melody.shape[-1]], device=self.device),
                        sample_rate=[self.sample_rate], path=[None])
        if prompt is not None:
            if descriptions is not None:
                assert len(descriptions) == len(prompt), "Prompt and nb. descriptions doesn't match"
            prompt = prompt.to(self.device)
            prompt_tokens, scale = self.compression_model.encode(prompt)
            assert scale is None
        else:
            prompt_tokens = None
        return attributes, prompt_tokens

    def _generate_tokens(self, attributes: tp.List[ConditioningAttributes],
                         prompt_tokens: tp.Optional[torch.Tensor], progress: bool = False) -> torch.Tensor:
        """reference musicgen.py:251-338, including the sliding-window extension beyond max_duration."""
        total_gen_len = int(self.duration * self.frame_rate)
        max_prompt_len = int(min(self.duration, self.max_duration) * self.frame_rate)
        current_gen_offset: int = 0

        def _progress_callback(generated_tokens: int, tokens_to_generate: int):
            generated_tokens += current_gen_offset
            if self._progress_callback is not None:
                self._progress_callback(generated_tokens, tokens_to_generate)
            else:
                print(f'{generated_tokens: 6d} / {tokens_to_generate: 6d}', end='\r')

        if prompt_tokens is not None:
            assert max_prompt_len >= prompt_tokens.shape[-1], "Prompt is longer than audio to generate"
        callback = _progress_callback if progress else None

        if self.duration <= self.max_duration:
            return self.lm.generate(prompt_tokens, attributes, callback=callback, max_gen_len=total_gen_len,
                                    **self.generation_params)

        ref_wavs = [attr.wav['self_wav'] for attr in attributes]
        all_tokens = []
        if prompt_tokens is None:
            prompt_length = 0
        else:
            all_tokens.append(prompt_tokens)
            prompt_length = prompt_tokens.shape[-1]
        assert self.extend_stride is not None, "Stride should be defined to generate beyond max_duration"
        assert self.extend_stride < self.max_duration, "Cannot stride by more than max generation duration."
        stride_tokens = int(self.frame_rate * self.extend_stride)
        while current_gen_offset + prompt_length < total_gen_len:
            time_offset = current_gen_offset / self.frame_rate
            chunk_duration = min(self.duration - time_offset, self.max_duration)
            max_gen_len = int(chunk_duration * self.frame_rate)
            for attr, ref_wav in zip(attributes, ref_wavs):
                wav_length = int(ref_wav.length.item())
                if wav_length == 0:
                    continue
                # tile the melody periodically so that every window sees a full-length condition
                initial_position = int(time_offset * self.sample_rate)
                wav_target_length = int(self.max_duration * self.sample_rate)
                positions = torch.arange(initial_position, initial_position + wav_target_length, device=self.device)
                attr.wav['self_wav'] = WavCondition(
                    ref_wav[0][..., positions % wav_length], torch.full_like(ref_wav[1], wav_target_length),
                    [self.sample_rate] * ref_wav[0].size(0), [None], [0.])
            gen_tokens = self.lm.generate(prompt_tokens, attributes, callback=callback, max_gen_len=max_gen_len,
                                          **self.generation_params)
            if prompt_tokens is None:
                all_tokens.append(gen_tokens)
            else:
                all_tokens.append(gen_tokens[:, :, prompt_tokens.shape[-1]:])
            prompt_tokens = gen_tokens[:, :, stride_tokens:]
            prompt_length = prompt_tokens.shape[-1]
            current_gen_offset += stride_tokens
        return torch.cat(all_tokens, dim=-1)
