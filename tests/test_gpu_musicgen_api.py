"""-m gpu: the user-facing MusicGen API on the debug geometry, mirroring the reference's
tests/models/test_musicgen.py (properties, unconditional, continuation incl. the mismatched-descriptions
assertion, text, long generation with windowing) plus melody conditioning and both LayerNorm execution modes."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def mg():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from audiocraft_amd.models.musicgen import MusicGen
    return MusicGen.get_pretrained('debug', 'cuda')


def test_base(mg):
    assert mg.frame_rate == 25 and mg.sample_rate == 32000 and mg.audio_channels == 1


def test_generate_unconditional(mg):
    mg.set_generation_params(duration=2.0, extend_stride=2.)
    wav = mg.generate_unconditional(3)
    assert list(wav.shape) == [3, 1, 64000]


def test_generate_continuation(mg):
    mg.set_generation_params(duration=2.0, extend_stride=2.)
    prompt = torch.randn(3, 1, 32000)
    wav = mg.generate_continuation(prompt, 32000)
    assert list(wav.shape) == [3, 1, 64000]
    prompt = torch.randn(2, 1, 32000)
    wav, toks = mg.generate_continuation(prompt, 32000, ['youpi', 'lapin dort'], return_tokens=True)
    assert list(wav.shape) == [2, 1, 64000] and list(toks.shape) == [2, 4, 50]
    # the continuation starts with the prompt's own codes
    codes, _ = mg.compression_model.encode(prompt.cuda())
    assert torch.equal(toks[..., :codes.shape[-1]], codes)
    prompt = torch.randn(2, 1, 32000)
    with pytest.raises(AssertionError):
        mg.generate_continuation(prompt, 32000, ['youpi', 'lapin dort', 'one too many'])


def test_generate(mg):
    mg.set_generation_params(duration=2.0, extend_stride=2.)
    wav = mg.generate(['youpi', 'lapin dort'])
    assert list(wav.shape) == [2, 1, 64000]
    assert torch.isfinite(wav).all()


def test_generate_long(mg):
    """reference test_musicgen.py:52-58: duration > max_duration -> sliding windows of `extend_stride`."""
    mg.max_duration = 3.
    mg.set_generation_params(duration=4., extend_stride=2.)
    try:
        calls = []
        mg.set_custom_progress_callback(lambda a, b: calls.append((a, b)))
        wav, toks = mg.generate(['youpi', 'lapin dort'], progress=True, return_tokens=True)
        assert list(wav.shape) == [2, 1, 32000 * 4] and list(toks.shape) == [2, 4, 100]
        assert calls and calls[-1][0] > 75
    finally:
        mg.max_duration = 30.
        mg.set_custom_progress_callback(None)


def test_greedy_is_deterministic_and_seeded_sampling_differs(mg):
    mg.set_generation_params(duration=1.0, use_sampling=False)
    a = mg.generate(['x'], return_tokens=True)[1]
    b = mg.generate(['x'], return_tokens=True)[1]
    assert torch.equal(a, b)
    mg.set_generation_params(duration=1.0, use_sampling=True, top_k=20)
    torch.manual_seed(0)
    c = mg.generate(['x'], return_tokens=True)[1]
    torch.manual_seed(1)
    d = mg.generate(['x'], return_tokens=True)[1]
    assert not torch.equal(c, d)


def test_melody_model_with_chroma():
    """musicgen-melody wiring (chroma2music.yaml): prepend [self_wav ; description], no cross-attention."""
    from audiocraft_amd.models import builders
    from audiocraft_amd.models.musicgen import MusicGen
    cfg = dict(dim=32, num_heads=4, num_layers=2, n_q=4, card=400, cfg_coef=3.0,
               conditioners={'description': {'kind': 't5', 'embedder': 'synthetic', 'dim': 16, 'length': 3},
                             'self_wav': {'kind': 'chroma', 'embedder': 'synthetic', 'n_frames': 5, 'duration': 30.}},
               fuser={'prepend': ['self_wav', 'description']})
    lm = builders.get_lm_model(cfg, 'cuda', torch.float32)
    mg = MusicGen('melody-debug', builders.get_debug_compression_model('cuda'), lm, max_duration=30)
    mg.set_generation_params(duration=1.0, use_sampling=False)
    melody = torch.randn(2, 1, 32000)
    wav = mg.generate_with_chroma(['a', 'b'], melody, 32000)
    assert list(wav.shape) == [2, 1, 32000]
    wav2 = mg.generate_with_chroma(['a', 'b'], [None, melody[0]], 32000)
    assert list(wav2.shape) == [2, 1, 32000]
    with pytest.raises(RuntimeError):
        MusicGen.get_pretrained('debug', 'cuda').generate_with_chroma(['a'], melody[:1], 32000)


def test_stereo_musicgen_debug_geometry():
    """Stereo MusicGen (reference encodec.py:397-506 + 8-codebook delay pattern [0,0,1,1,2,2,3,3]): left / right
    through the mono codec, codebooks interleaved per RVQ level; generation and continuation give 2-channel audio."""
    from audiocraft_amd.models import MusicGen, builders
    from audiocraft_amd.models.encodec import InterleaveStereoCompressionModel
    torch.manual_seed(0)
    lm = builders.get_lm_model(dict(dim=32, num_heads=4, num_layers=2, n_q=8, card=400,
                                    codebooks_pattern={'modeling': 'delay', 'delay': {'delays': [0, 0, 1, 1, 2, 2, 3, 3]}},
                                    conditioners={'description': {'kind': 't5', 'embedder': 'synthetic', 'dim': 16, 'length': 4}},
                                    fuser={'cross': ['description']}), 'cuda', torch.float32)
    codec = InterleaveStereoCompressionModel(builders.get_debug_compression_model('cuda'))
    mg = MusicGen('stereo-debug', codec, lm, max_duration=30)
    mg.set_generation_params(duration=2.0, extend_stride=2.)
    assert mg.audio_channels == 2 and codec.num_codebooks == 8
    wav, toks = mg.generate(['youpi', 'lapin dort'], return_tokens=True)
    assert list(wav.shape) == [2, 2, 64000] and list(toks.shape) == [2, 8, 50] and bool(torch.isfinite(wav).all())
    assert int(toks.min()) >= 0 and int(toks.max()) < 400
    # the wrapper's decode of the interleaved codes == the mono codec on the de-interleaved halves
    left, right = codec.model.decode(toks[:, 0::2]), codec.model.decode(toks[:, 1::2])
    assert torch.equal(wav, torch.cat([left, right], dim=1))
    wav = mg.generate_continuation(torch.randn(2, 2, 32000), 32000, ['a', 'b'])
    assert list(wav.shape) == [2, 2, 64000]
    wav = mg.generate_continuation(torch.randn(2, 1, 32000), 32000)   # mono prompt is replicated to both channels
    assert list(wav.shape) == [2, 2, 64000]


# ---- AudioGen (reference tests/models/test_audiogen.py): same machinery, 16 kHz codec, 10 s windows, stride 2 s
@pytest.fixture(scope='module')
def ag():
    from audiocraft_amd.models.audiogen import AudioGen
    model = AudioGen.get_pretrained('debug', 'cuda')
    model.set_generation_params(duration=2.0, extend_stride=2.)
    return model


def test_audiogen_base(ag):
    assert ag.frame_rate == 25 and ag.sample_rate == 16000 and ag.audio_channels == 1
    assert ag.max_duration == 10 and ag.generation_params['top_k'] == 250


def test_audiogen_generate_and_continuation(ag):
    wav = ag.generate(['youpi', 'lapin dort'])
    assert list(wav.shape) == [2, 1, 32000] and bool(torch.isfinite(wav).all())
    wav = ag.generate_continuation(torch.randn(3, 1, 16000), 16000)
    assert list(wav.shape) == [3, 1, 32000]
    wav, toks = ag.generate_continuation(torch.randn(2, 1, 16000), 16000, ['youpi', 'lapin dort'], return_tokens=True)
    assert list(wav.shape) == [2, 1, 32000] and list(toks.shape) == [2, 4, 50]
    with pytest.raises(AssertionError):
        ag.generate_continuation(torch.randn(2, 1, 16000), 16000, ['youpi', 'lapin dort', 'one too many'])


def test_audiogen_generate_long(ag):
    ag.max_duration = 3.
    ag.set_generation_params(duration=4., extend_stride=2.)
    try:
        wav = ag.generate(['youpi', 'lapin dort'])
        assert list(wav.shape) == [2, 1, 16000 * 4]
    finally:
        ag.max_duration = 10.
        ag.set_generation_params(duration=2.0, extend_stride=2.)


def test_ln_modes_agree(monkeypatch):
    """LayerNorm folded into the consuming GEMM's epilogue (producer statistics + raw fragment-order row)
    -- with the cross-attention query riding in the out-projection launch (default) or projected by its own
    launch (ACMI_CROSS_FUSED=0) -- == separate standardisation kernel (tokens identical,
    logits within f32 round-off): run in a subprocess per mode because the mode is latched at first use."""
    import subprocess
    import sys
    code = ("import torch, sys; sys.path.insert(0, '.');"
            "from audiocraft_amd.models import builders;"
            "torch.manual_seed(0);"
            "lm = builders.get_lm_model(dict(dim=512, num_heads=8, num_layers=2, n_q=4, card=256,"
            " conditioners={'description': {'kind': 't5', 'embedder': 'synthetic', 'dim': 32, 'length': 4}},"
            " fuser={'cross': ['description']}), 'cuda', torch.float32);"
            "g = torch.Generator().manual_seed(1); c = torch.randn(4, 4, 512, generator=g); c[2:] = 0;"
            "ct = {'description': (c.cuda(), torch.ones(4, 4, dtype=torch.int64).cuda())};"
            "t, l = lm.generate(None, [], num_samples=2, max_gen_len=8, use_sampling=False, condition_tensors=ct, return_logits=True);"
            "torch.save((t.cpu(), l.cpu()), sys.argv[1])")
    outs = []
    import os
    import tempfile
    for mode, fused in (('tile', '1'), ('fold', '1'), ('fold', '0')):
        f = tempfile.mktemp(suffix='.pt')
        env = dict(os.environ, ACMI_LN_MODE=mode, ACMI_CROSS_FUSED=fused)
        subprocess.run([sys.executable, '-c', code, f], check=True, env=env, cwd=os.path.dirname(os.path.dirname(__file__)))
        outs.append(torch.load(f))
        os.remove(f)
    for other in outs[1:]:
        assert torch.equal(outs[0][0], other[0])
        assert (outs[0][1] - other[1]).abs().max() < 1e-3 * outs[0][1].abs().max()
