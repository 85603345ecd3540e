"""TEST INFRASTRUCTURE ONLY -- import-stub harness that lets the *unmodified* reference
sources under /root/reference be imported on CPU in the build container.

Only the fixture generators under `tests/golden/`, `oracle/validate_against_reference.py` and
`oracle/ref_baseline.py` (bench.py's cpu_baseline leg, taken only where `available()`) import this
module.  It is never imported by the product package (`audiocraft_amd/`), by
`__graft_entry__.smoke()` or by the `-m gpu` tests: /root/reference does not exist on the GPU box.  Recipe documented in SURVEY.md section 8(c) / Appendix A.

Importing this module registers stub modules for the third-party packages the reference imports
but that are not installed here (xformers, flashy, omegaconf, julius, ...).  No arithmetic on the
hot path is stubbed: only `xformers.ops.unbind` (= torch.unbind) is *executed* on the torch backend.
"""
import importlib.machinery as _im
import os
import sys
import types
import warnings

warnings.filterwarnings('ignore')

REF_ROOT = os.environ.get('AUDIOCRAFT_REFERENCE', '/root/reference')
REF = os.path.join(REF_ROOT, 'audiocraft')


def available() -> bool:
    return os.path.isdir(REF)


def install():
    if 'audiocraft' in sys.modules:
        return
    if not available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    import torch
    try:  # must precede the librosa stub, transformers' lazy imports get confused otherwise
        from transformers import T5EncoderModel, T5Tokenizer  # noqa: F401
    except Exception:
        pass

    def pkg(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        m.__package__ = name
        sys.modules[name] = m
        return m

    ac = pkg('audiocraft', REF)
    ac.__version__ = '1.4.0a2'
    for s in ['data', 'models']:
        setattr(ac, s, pkg('audiocraft.' + s, REF + '/' + s))

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__spec__ = _im.ModuleSpec(name, None)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class Lazy(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith('__'):
                raise AttributeError(k)
            return type(k, (), {})

    def lazy(name):
        m = Lazy(name)
        m.__path__ = []
        m.__spec__ = _im.ModuleSpec(name, None, is_package=True)
        sys.modules[name] = m
        return m

    xops = stub('xformers.ops', unbind=torch.unbind, memory_efficient_attention=None,
                LowerTriangularMask=object)
    stub('xformers', ops=xops)
    fd = stub('flashy.distrib', broadcast_tensors=lambda *a, **k: None, rank=lambda: 0,
              world_size=lambda: 1, is_distributed=lambda: False)
    stub('flashy', distrib=fd)
    for n in ['omegaconf', 'num2words', 'spacy', 'librosa', 'librosa.filters', 'torchaudio',
              'torchaudio.transforms', 'av', 'julius', 'soundfile', 'demucs', 'dora', 'dora.log',
              'hydra', 'torchdiffeq', 'audioseal', 'torchmetrics', 'encodec', 'pesq', 'pystoi',
              'torchvision', 'gradio', 'einops_exts']:
        if n not in sys.modules:
            lazy(n)
    sys.modules['omegaconf'].DictConfig = type('DictConfig', (), {})

    class _OmegaConf:   # utils/export.py only needs the YAML dump of an (already plain) config mapping
        @staticmethod
        def to_yaml(cfg):
            import yaml
            return yaml.safe_dump(cfg, sort_keys=False)
    sys.modules['omegaconf'].OmegaConf = _OmegaConf

    class _Tok:
        def __init__(s, t):
            s.text = t
            s.lemma_ = t
            s.is_stop = False
    sys.modules['spacy'].load = lambda lang: (lambda text: [_Tok(w) for w in text.split()])
    sys.modules['num2words'].num2words = lambda n: str(n)

    def _resample_frac(x, old_sr, new_sr, *a, **k):
        assert old_sr == new_sr, "julius stub: only identity resample supported"
        return x
    sys.modules['julius'].resample_frac = _resample_frac


install()
