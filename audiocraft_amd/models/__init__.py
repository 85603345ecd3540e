"""Models of the generation path, under the reference's names (`audiocraft.models`): resolved lazily so that
importing a submodule does not pull in the whole package."""
import importlib

_EXPORTS = {
    'MusicGen': 'musicgen', 'AudioGen': 'audiogen', 'BaseGenModel': 'genmodel', 'LMModel': 'lm',
    'MultiBandDiffusion': 'multibanddiffusion', 'DiffusionUnet': 'unet',
    'CompressionModel': 'encodec', 'EncodecModel': 'encodec', 'InterleaveStereoCompressionModel': 'encodec',
}
__all__ = sorted(_EXPORTS)


def __getattr__(name):
    if name in _EXPORTS:
        return getattr(importlib.import_module(f'{__name__}.{_EXPORTS[name]}'), name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
