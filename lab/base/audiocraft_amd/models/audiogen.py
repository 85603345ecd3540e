"""AudioGen on MI355X: text-to-sound generation API.

API mirror of `audiocraft.models.audiogen.AudioGen` (reference audiocraft/models/audiogen.py:23-93): the same
`LMModel` / EnCodec machinery as MusicGen at other shapes (16 kHz EnCodec with hop 320 -> 50 frames/s, RVQ
4 x 2048; text conditioning only), default duration 5 s, windows of at most `max_duration` = 10 s advanced
by `extend_stride` = 2 s.  Everything else -- generate / generate_continuation / generate_unconditional, the
windowed token generation -- is `BaseGenModel`.
"""
import typing as tp

from . import builders
from .encodec import CompressionModel
from .genmodel import BaseGenModel
from .lm import LMModel


class AudioGen(BaseGenModel):
    def __init__(self, name: str, compression_model: CompressionModel, lm: LMModel,
                 max_duration: tp.Optional[float] = None):
        super().__init__(name, compression_model, lm, max_duration)
        self.set_generation_params(duration=5)  # reference default

    @staticmethod
    def get_pretrained(name: str = 'facebook/audiogen-medium', device=None, weight_dtype=None):
        """'debug' builds the reference's debug geometry at 16 kHz (audiogen.py:50-54).  Anything else is a
        directory (or cache entry) holding `state_dict.bin` + `compression_state_dict.bin` in the reference
        export format, resolved on disk only (see `loaders.py`; there is no network here)."""
        device = 'cuda' if device is None else device
        if name == 'debug':
            return AudioGen(name, builders.get_debug_compression_model(device, sample_rate=16000),
                            builders.get_debug_lm_model(device), max_duration=10)
        from . import loaders
        lm = loaders.load_lm_model(name, device=device, weight_dtype=weight_dtype)
        assert 'self_wav' not in lm.condition_provider.conditioners, \
            "AudioGen do not support waveform conditioning for now"
        return AudioGen(name, loaders.load_compression_model(name, device=device), lm)

    @staticmethod
    def get_random_init(name: str = 'facebook/audiogen-medium', device='cuda', weight_dtype=None, text_len: int = 16,
                        seed: int = 0):
        """Architecture of the released model (1.5 B LM, 16 kHz EnCodec) with seeded random weights and a synthetic
        text conditioner: neither checkpoints nor T5 weights exist offline."""
        import torch
        assert name == 'facebook/audiogen-medium', name
        torch.manual_seed(seed)
        lm = builders.get_lm_model(builders.audiogen_lm_cfg('medium', synthetic=True, text_len=text_len), device,
                                   torch.bfloat16 if weight_dtype is None else weight_dtype)
        return AudioGen(name, builders.get_compression_model(builders.ENCODEC_16KHZ, device), lm, max_duration=10)

    def set_generation_params(self, use_sampling: bool = True, top_k: int = 250, top_p: float = 0.0,
                              temperature: float = 1.0, duration: float = 10.0, cfg_coef: float = 3.0,
                              two_step_cfg: bool = False, extend_stride: float = 2):
        """Same knobs and defaults as the reference (audiogen.py:63-93)."""
        assert extend_stride < self.max_duration, "Cannot stride by more than max generation duration."
        self.extend_stride, self.duration = extend_stride, duration
        self.generation_params = dict(use_sampling=use_sampling, temp=temperature, top_k=top_k, top_p=top_p,
                                      cfg_coef=cfg_coef, two_step_cfg=two_step_cfg)
