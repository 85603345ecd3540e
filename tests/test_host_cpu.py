"""CPU suite (-m "not gpu"): host logic of the product package + the C ABI surface (no compute calls)."""
import ctypes
import os
import re

import pytest
import torch

from oracle import patterns as opat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------ C ABI

def _declared_functions():
    hdr = open(os.path.join(ROOT, 'include', 'acmi.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    return sorted(set(re.findall(r'\b(acmi_[a-z0-9_]+)\s*\(', hdr)))


def test_library_exports_every_declared_symbol():
    from audiocraft_amd import _C
    names = _declared_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(_C.lib, n), f"include/acmi.h declares {n} but libacmi.so does not export it"
    assert set(_C.EXPORTS) == set(names)
    assert _C.version() >= 110


def _header_struct_fields(cname):
    hdr = re.sub(r'/\*.*?\*/', '', open(os.path.join(ROOT, 'include', 'acmi.h')).read(), flags=re.S)
    m = re.search(r'typedef struct\s*\{([^{}]*)\}\s*' + cname + r'\s*;', hdr)
    assert m, f"include/acmi.h has no struct {cname}"
    names = []
    for decl in m.group(1).split(';'):
        for part in ' '.join(decl.split()).split(','):
            if part.strip():
                names.append(re.findall(r'([A-Za-z_][A-Za-z0-9_]*)\s*(?:\[\d+\])?\s*$', part.strip())[0])
    return names


def test_ctypes_mirrors_have_the_layout_of_the_header(tmp_path):
    """Every descriptor crosses the C ABI as a ctypes mirror written by hand (_C.py); a field added to one side only would
    shift everything behind it and corrupt calls silently.  The header is compiled as C (gcc) and every field's offset and
    each struct's size are compared with the mirror's."""
    import subprocess
    from audiocraft_amd import _C
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "acmi.h"', 'int main(void) {']
    for cname, cls in _C.STRUCT_MIRRORS.items():
        fields = _header_struct_fields(cname)
        assert fields == [f[0] for f in cls._fields_], f"{cname}: field names / order differ from _C.{cls.__name__}"
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        lines += [f'  printf("{cname}.{f} %zu\\n", offsetof({cname}, {f}));' for f in fields]
    lines += ['  return 0;', '}']
    src, exe = tmp_path / 'layout.c', tmp_path / 'layout'
    src.write_text('\n'.join(lines))
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)], check=True)
    got = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in _C.STRUCT_MIRRORS.items():
        assert int(got[cname]) == ctypes.sizeof(cls), f"sizeof({cname}) = {got[cname]}, the mirror has {ctypes.sizeof(cls)}"
        for f, _ in cls._fields_:
            assert int(got[f'{cname}.{f}']) == getattr(cls, f).offset, f"{cname}.{f}"


def test_argument_validation_without_gpu():
    """Error paths never touch the device: negative return code + thread-local message."""
    from audiocraft_amd import _C
    lib = _C.lib
    rc = lib.acmi_rvq_decode(None, None, None, 1, 16, 4, 64, 32, None)
    assert rc == -1 and b'K=64' in lib.acmi_last_error()
    rc = lib.acmi_rvq_encode(None, None, None, None, 1, 100, 4, 4, 32, None)
    assert rc == -1 and b'dimension 100' in lib.acmi_last_error()
    d = _C.ConvDesc()
    d.B, d.Cin, d.Tin, d.Cout, d.Tout, d.ksize, d.stride, d.dilation = 1, 4, 10, 6, 10, 3, 1, 1
    d.shuffle = 4  # 6 % 4 != 0
    rc = lib.acmi_conv1d(ctypes.byref(d), None, None, None, None, None, None, None)
    assert rc == -1 and b'shuffle' in lib.acmi_last_error()
    assert lib.acmi_conv1d_weight_floats(ctypes.byref(d)) == 0 and lib.acmi_conv1d_work_floats(ctypes.byref(d)) == 0   # same refusal
    d.shuffle = 1
    # sizes of the tiled weights / the packed-input scratch are pure host arithmetic: 42 channels x 3 taps per 128-wide K chunk
    d.B, d.Cin, d.Tin, d.Cout, d.Tout, d.ksize = 2, 100, 1000, 70, 1000, 3
    assert lib.acmi_conv1d_weight_floats(ctypes.byref(d)) == 2 * 3 * 2 * 64 * 68      # 2 row tiles x 3 chunks x 2 parities x 64 x KCP
    assert lib.acmi_conv1d_work_floats(ctypes.byref(d)) == 2 * 126 * (1024 + 68)       # B x Cin padded to whole chunks x (tiles + halo)
    # few output tiles, long K: K split over 8 workgroups per tile, partial sums behind the packed input
    d.B, d.Cin, d.Tin, d.Cout, d.Tout, d.ksize = 1, 3072, 125, 3072, 125, 3
    assert lib.acmi_conv1d_work_floats(ctypes.byref(d)) == 74 * 42 * (4 * 64 + 68) + 8 * 3072 * 2 * 64
    d.B, d.Tin, d.Tout = 2, 1000, 1000
    d.Cout, d.Cin, d.ksize = 1, 64, 7      # the few-output kernel: raw weights, no scratch
    assert lib.acmi_conv1d_weight_floats(ctypes.byref(d)) == 64 * 7 and lib.acmi_conv1d_work_floats(ctypes.byref(d)) == 0
    assert lib.acmi_lm_step(None, None, 0, None) == -1
    assert lib.acmi_layer_norm_rows(None, None, None, None, 4, 64, ctypes.c_float(1e-5), None) == -1
    assert b'acmi_layer_norm_rows' in lib.acmi_last_error()
    assert lib.acmi_ln_tile(None, None, 0, 4, 4096, ctypes.c_float(1e-5), None) == -1
    assert _C.lstm_work_floats(3, 8) == 5 * 3 * 8 + 4   # c + three hidden-state buffers + give-up counter
    # descriptor entry points: null descriptors, contradictory operands, bad placement
    assert lib.acmi_linear_ex(None, None) == -1 and lib.acmi_linear_pair(None, None, None) == -1
    assert lib.acmi_attn_decode_ex(None, None) == -1
    ld = _C.LinearDesc()
    ld.a_mode, ld.wdtype, ld.M, ld.N, ld.K = _C.A_ROWMAJOR_F32, _C.BF16, 4, 32, 64
    ld.colsum = 1  # folded LayerNorm asked for on a row-major activation
    assert lib.acmi_linear_ex(ctypes.byref(ld), None) == -1 and b'colsum' in lib.acmi_last_error()
    ld = _C.LinearDesc()
    ld.a_mode, ld.wdtype, ld.M, ld.N, ld.K, ld.a_lo, ld.lo_K = _C.A_TILED, _C.BF16, 4, 32, 64, 1, 40  # 40 % 32 != 0
    assert lib.acmi_linear_ex(ctypes.byref(ld), None) == -1 and b'lo_K' in lib.acmi_last_error()
    ad = _C.AttnDesc()
    ad.out_mode, ad.out_dtype, ad.Beff, ad.H, ad.hd, ad.Tcap, ad.len = _C.OUT_TILED, _C.BF16, 4, 2, 64, 8, 8
    ad.out_col0, ad.out_rbs = 48, 8   # not a multiple of the K tile
    assert lib.acmi_attn_decode_ex(ctypes.byref(ad), None) == -1 and b'placement' in lib.acmi_last_error()
    ad.out_col0, ad.out_rbs, ad.cache_rows = 0, 0, 3   # 4 query rows over 3 cache rows
    assert lib.acmi_attn_decode_ex(ctypes.byref(ad), None) == -1 and b'cache rows' in lib.acmi_last_error()
    ad.cache_rows, ad.q_colsum = 0, 1   # LayerNorm hook without statistics
    assert lib.acmi_attn_decode_ex(ctypes.byref(ad), None) == -1 and b'statistics' in lib.acmi_last_error()


def test_tiling_roundtrip():
    from audiocraft_amd import _C
    w = torch.randn(40, 70)
    for dt in (torch.float32, torch.bfloat16):
        t = _C.tile_matrix(w, dt)
        assert t.shape[0] == 3 and t.shape[3] == 16
        assert torch.equal(_C.untile_matrix(t, 40, 70), w.to(dt))
        # documented layout: T[rt][kc][kg*16 + r][j] = w[rt*16 + r][kc*KT + kg*e + j]
        e = t.shape[4]
        assert t[1, 1, 3, 5, 1] == w.to(dt)[16 + 5, 1 * 4 * e + 3 * e + 1]


def test_half_tile_order_matches_the_header():
    """acmi_linear_desc.w_half (include/acmi.h): unit u of half-tile j is 64 lanes x 16 B; lane = kg*16 + s*8 + f holds
    w[j*8 + f][u*2KT + s*KT + kg*e .. + e - 1]."""
    from audiocraft_amd import _C
    for dt, e in ((torch.float32, 4), (torch.bfloat16, 8)):
        kt = 4 * e
        w = torch.randn(24, 6 * kt)
        t = _C.tile_matrix_half(w, dt).reshape(3, 3, 64, e)     # [half-tile j][unit u][lane][element]
        wd = w.to(dt)
        for j, u, kg, s_, f in ((0, 0, 0, 0, 0), (2, 1, 3, 1, 7), (1, 2, 2, 0, 5), (2, 2, 1, 1, 0)):
            lane = kg * 16 + s_ * 8 + f
            k0 = u * 2 * kt + s_ * kt + kg * e
            assert torch.equal(t[j, u, lane], wd[j * 8 + f, k0:k0 + e])
        assert _C.TiledWeight(w, dt, half=True).half and not _C.TiledWeight(w, dt).half
    with pytest.raises(AssertionError):
        _C.tile_matrix_half(torch.randn(20, 64), torch.bfloat16)     # N % 8


# ------------------------------------------------------------------------------------------ patterns

@pytest.mark.parametrize('T,delays', [(7, None), (1, None), (12, [0, 1, 2, 3]), (9, [0, 0, 1, 3]), (5, [0, 0, 0, 0])])
def test_pattern_matches_naive_loops(T, delays):
    """reference tests/modules/test_codebooks_patterns.py:107-246: gathers == naive python loops."""
    from audiocraft_amd.modules.codebooks_patterns import DelayedPatternProvider
    K = 4
    p = DelayedPatternProvider(K, delays).get_pattern(T)
    z = torch.randint(0, 100, (3, K, T))
    v, idx, m = p.build_pattern_sequence(z, 777)
    v2, m2 = opat.build_pattern_sequence(z, 777, delays)
    assert torch.equal(v, v2) and torch.equal(m, m2)
    assert v.shape[-1] == T + max(delays or range(K)) + 1 == p.num_sequence_steps + 1
    back, _, bm = p.revert_pattern_sequence(v, -1)
    b2, bm2 = opat.revert_pattern_sequence(v2, -1, T, delays)
    assert torch.equal(back, b2) and torch.equal(bm, bm2) and torch.equal(back, z)
    # truncated sequences revert with the special token where steps are missing
    back_t, _, bm_t = p.revert_pattern_sequence(v[..., :T], -1)
    b3, bm3 = opat.revert_pattern_sequence(v2[..., :T], -1, T, delays)
    assert torch.equal(back_t, b3) and torch.equal(bm_t, bm3)
    for t in range(T + 1):   # t == T: the coordinates the layout's max_delay tail still holds (as the reference's layout does)
        assert p.get_first_step_with_timesteps(t) == opat.first_step_with_timestep(K, T, t, delays)
    # layout view agrees with the closed form
    lay = opat.delayed_layout(K, T, delays)
    assert [[tuple(c) for c in s] for s in p.layout] == lay
    # logits revert: [B, card, K, S] -> [B, card, K, T]; position t of codebook q comes from step t + delay
    S = v.shape[-1]
    logits = torch.randn(2, 5, K, S)
    lv, _, lm = p.revert_pattern_logits(logits, float('nan'))
    dl = delays or list(range(K))
    for q in range(K):
        for t in range(T):
            if t + dl[q] < S - 1 + 1 and t + dl[q] < S:
                assert lm[q, t] == (t + dl[q] < S)
                if lm[q, t]:
                    assert torch.equal(lv[:, :, q, t], logits[:, :, q, t + dl[q]])


def test_pattern_providers_match_reference_golden():
    """Every provider of the reference's builder (`delay` incl. flatten_first / empty_initial, `parallel`, `unroll`,
    `coarse_first`, `musiclm`): layouts, build / revert values + indexes + masks (all steps and valid steps only, full and
    truncated sequences), the logits maps and first steps, against what the unmodified reference recorded
    (tests/golden/make_pattern_golden.py; the reference's own checks: tests/modules/test_codebooks_patterns.py)."""
    import json
    from audiocraft_amd.models import builders
    recs = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'patterns.json')))
    assert len(recs) >= 70 and {r['provider'] for r in recs} == {'delay', 'parallel', 'unroll', 'coarse_first', 'musiclm'}
    for r in recs:
        what = (r['provider'], r['kwargs'], r['n_q'], r['timesteps'])
        prov = builders.get_codebooks_pattern_provider(r['n_q'], {'modeling': r['provider'], r['provider']: r['kwargs']})
        p = prov.get_pattern(r['timesteps'])
        assert [[[c.t, c.q] for c in s] for s in p.layout] == r['layout'], what
        assert (p.num_sequence_steps, p.max_delay, len(p.valid_layout)) == (r['num_sequence_steps'], r['max_delay'], r['valid_steps']), what
        assert p.starts_with_special_token()
        firsts = [[p.get_first_step_with_timesteps(t, q) for q in [None] + list(range(r['n_q']))] for t in range(r['timesteps'] + 1)]
        assert firsts == r['first_steps'], what
        z = torch.tensor(r['z'], dtype=torch.long).reshape(2, r['n_q'], r['timesteps'])
        for m in r['maps']:
            v, i, k = p.build_pattern_sequence(z, 99, m['keep'])
            assert v.tolist() == m['values'] and i.tolist() == m['indexes'] and k.int().tolist() == m['mask'], what
            for rv in m['revert']:
                s = torch.tensor(rv['s'], dtype=torch.long)
                v, i, k = p.revert_pattern_sequence(s, -1, m['keep'])
                assert v.tolist() == rv['values'] and i.tolist() == rv['indexes'] and k.int().tolist() == rv['mask'], (what, rv['S'])
                lv, li, lk = p.revert_pattern_logits(torch.arange(float(2 * 3 * r['n_q'] * rv['S'])).reshape(2, 3, r['n_q'], rv['S']),
                                                     float('nan'), m['keep'])
                assert li.tolist() == rv['logits_indexes'] and lk.int().tolist() == rv['logits_mask'], (what, rv['S'])
                assert torch.isnan(lv[:, :, ~lk]).all() and not torch.isnan(lv[:, :, lk]).any()
    with pytest.raises(KeyError):
        builders.get_codebooks_pattern_provider(4, {'modeling': 'spiral'})
    with pytest.raises(AssertionError):    # two codebooks of one inner step must share their delay (codebooks_patterns.py:441-445)
        builders.get_codebooks_pattern_provider(3, {'modeling': 'unroll', 'unroll': {'flattening': [0, 1, 1], 'delays': [0, 1, 2]}})


REFERENCE_ROOT = '/root/reference'
_ALIAS_PATTERNS = """
import audiocraft_amd.modules.codebooks_patterns as mine
pkg = types.ModuleType('audiocraft'); pkg.__path__ = []
mods = types.ModuleType('audiocraft.modules'); mods.__path__ = []
sys.modules.update({'audiocraft': pkg, 'audiocraft.modules': mods, 'audiocraft.modules.codebooks_patterns': mine})
mods.codebooks_patterns = mine
"""
_ALIAS_AUDIO_UTILS = """
import audiocraft_amd.data_audio as da, audiocraft_amd.data_audio_utils as du
m = types.ModuleType('audiocraft.data.audio_utils')
for src in (da, du):
    for k in dir(src):
        if not k.startswith('__'):
            setattr(m, k, getattr(src, k))
pkg = types.ModuleType('audiocraft'); pkg.__path__ = []
data = types.ModuleType('audiocraft.data'); data.__path__ = []
audio = types.ModuleType('audiocraft.data.audio'); audio.audio_write = da.audio_write
sys.modules.update({'audiocraft': pkg, 'audiocraft.data': data, 'audiocraft.data.audio_utils': m, 'audiocraft.data.audio': audio,
                    'julius': types.ModuleType('julius')})
"""


@pytest.mark.skipif(not os.path.isdir(REFERENCE_ROOT), reason="the reference tree only exists in the build container")
@pytest.mark.parametrize('test_file,alias,select,n_min', [
    ('tests/modules/test_codebooks_patterns.py', _ALIAS_PATTERNS, '', 120),
    # (the two deselected cases compare with julius' resampler, absent here; resampling itself runs on the device)
    ('tests/data/test_audio_utils.py', _ALIAS_AUDIO_UTILS, 'not upsample and not resample', 11),
])
def test_reference_own_tests_pass_on_this_implementation(test_file, alias, select, n_min):
    """The reference's OWN test files for host-side pieces of the path -- the codebook patterns (parallel, delayed, unrolled
    providers, layouts, build / revert maps against its naive loops) and the audio utilities (channel conversion, PCM
    conversion, the normalisation strategies) -- run unmodified with the `audiocraft.*` module they import aliased to this
    package's module; in a subprocess, so that the alias does not leak into this session."""
    import subprocess
    import sys
    code = ("import sys, types, pytest\n" + f"sys.path[:0] = [{ROOT!r}, {REFERENCE_ROOT!r}]\n" + alias +
            f"sys.exit(int(pytest.main(['-q', '-p', 'no:cacheprovider', {os.path.join(REFERENCE_ROOT, test_file)!r}, '-k', {select!r}])))\n")
    out = subprocess.run([sys.executable, '-c', code], cwd='/tmp', capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    m = re.search(r'(\d+) passed', out.stdout)
    assert m and int(m.group(1)) >= n_min and 'failed' not in out.stdout, out.stdout[-500:]


def test_delay_pattern_closed_form_equals_its_layout_form():
    """The closed-form `Pattern` (what MusicGen runs) and the coordinate-array `LayoutPattern` are two implementations of
    the same delay rule: 300 random (codebooks, timesteps, delays) -- delays that do not start at 0 included, where the
    reference's max_delay is max - min (codebooks_patterns.py:84-90) -- give identical layouts, maps and first steps."""
    import random
    from audiocraft_amd.modules import codebooks_patterns as cp
    rng = random.Random(0)
    g = torch.Generator().manual_seed(0)
    for _ in range(300):
        K, T = rng.randint(1, 6), rng.randint(1, 12)
        delays = sorted(rng.randint(0, 4) for _ in range(K))
        a = cp.Pattern(K, T, delays)
        st, t, q, used = cp._delayed_coords(K, delays, T, 0, 1)
        b = cp.LayoutPattern(K, T, 1 + used, st, t, q)
        what = (K, T, delays)
        assert [[tuple(c) for c in s] for s in a.layout] == [[tuple(c) for c in s] for s in b.layout], what
        assert (a.max_delay, a.num_sequence_steps, len(a.valid_layout)) == (b.max_delay, b.num_sequence_steps, len(b.valid_layout)), what
        z = torch.randint(0, 9, (2, K, T), generator=g)
        for keep in (False, True):
            x, y = a.build_pattern_sequence(z, 99, keep), b.build_pattern_sequence(z, 99, keep)
            assert all(torch.equal(i, j) for i, j in zip(x, y)), (what, keep)
            for S in {x[0].shape[-1], max(x[0].shape[-1] - 1, 1), 1}:
                s = torch.randint(0, 9, (2, K, S), generator=g)
                assert all(torch.equal(i, j) for i, j in zip(a.revert_pattern_sequence(s, -1, keep),
                                                              b.revert_pattern_sequence(s, -1, keep))), (what, keep, S)
                lg = torch.randn(1, 2, K, S, generator=g)
                xa, xb = a.revert_pattern_logits(lg, float('nan'), keep), b.revert_pattern_logits(lg, float('nan'), keep)
                assert torch.equal(xa[1], xb[1]) and torch.equal(xa[2], xb[2]) and torch.equal(torch.nan_to_num(xa[0]), torch.nan_to_num(xb[0]))
        for tt in range(T + 1):
            for qq in [None] + list(range(K)):
                assert a.get_first_step_with_timesteps(tt, qq) == b.get_first_step_with_timesteps(tt, qq), (what, tt, qq)


def test_pattern_valid_steps_only():
    from audiocraft_amd.modules.codebooks_patterns import DelayedPatternProvider
    p = DelayedPatternProvider(4).get_pattern(6)
    z = torch.arange(2 * 4 * 6).view(2, 4, 6)
    v, _, m = p.build_pattern_sequence(z, 99, keep_only_valid_steps=True)
    assert v.shape[-1] == 7 and len(p.valid_layout) == 7
    assert torch.equal(v, p.build_pattern_sequence(z, 99)[0][..., :7])


# ------------------------------------------------------------------------------------------ conditioning

def test_cfg_dropout_and_fuser_order():
    from audiocraft_amd.modules.conditioners import (ClassifierFreeGuidanceDropout, ConditionFuser,
                                                     ConditioningAttributes, WavCondition)
    c = ConditioningAttributes(text={'description': 'hello'})
    c.wav['self_wav'] = WavCondition(torch.randn(1, 1, 50), torch.tensor([50]), [32000], [None], [0.])
    null = ClassifierFreeGuidanceDropout(p=1.0)([c])
    assert c.text['description'] == 'hello' and c.wav['self_wav'].wav.shape[-1] == 50   # deep-copied
    assert null[0].text['description'] is None
    assert null[0].wav['self_wav'].wav.shape == (1, 1, 1) and int(null[0].wav['self_wav'].length) == 0
    assert ClassifierFreeGuidanceDropout(p=0.0)([c])[0] is c
    flat = c.to_flat_dict()
    assert ConditioningAttributes.from_flat_dict(flat).text == c.text
    # reference fuser loop (conditioners.py:1730-1748): later 'prepend' conditions go in FRONT
    fuser = ConditionFuser({'prepend': ['self_wav', 'description'], 'cross': []})
    d = {'description': (torch.ones(2, 3, 4), torch.ones(2, 3)), 'self_wav': (2 * torch.ones(2, 5, 4), torch.ones(2, 5))}
    prepend, cross = fuser.fuse(d)
    assert cross is None and prepend.shape == (2, 8, 4)
    assert (prepend[:, :5] == 2).all() and (prepend[:, 5:] == 1).all()
    fuser = ConditionFuser({'cross': ['description']})
    prepend, cross = fuser.fuse({'description': d['description']})
    assert prepend is None and cross.shape == (2, 3, 4)
    with pytest.raises(AssertionError):
        fuser.fuse(d)


def test_wav_collation_pads_to_longest():
    from audiocraft_amd.modules.conditioners import (ChromaStemConditioner, ConditioningAttributes,
                                                     ConditioningProvider, WavCondition)
    prov = ConditioningProvider({'self_wav': ChromaStemConditioner(8, 32000, 12, 14, 30.)})
    a = ConditioningAttributes()
    a.wav['self_wav'] = WavCondition(torch.ones(1, 2, 40), torch.tensor([40]), [32000], [None], [0.])
    b = ConditioningAttributes()
    b.wav['self_wav'] = WavCondition(torch.zeros(1, 1, 1), torch.tensor([0]), [32000], [None], [None])
    tok = prov.tokenize([a, b])
    w = tok['self_wav']
    assert w.wav.shape == (2, 1, 40) and w.length.tolist() == [40, 0]
    assert (w.wav[0] == 1).all() and (w.wav[1] == 0).all()
    assert prov.conditioners['self_wav'].chroma_len == 235    # 30 s * 32 kHz / 4096 + 1


def test_models_refuse_to_run_without_gpu():
    from audiocraft_amd.models import builders
    lm = builders.get_debug_lm_model('cpu')
    with pytest.raises(RuntimeError, match="MI355X"):
        ct = {'description': (torch.zeros(2, 3, 16), torch.ones(2, 3, dtype=torch.int64))}
        lm.generate(None, [], num_samples=1, max_gen_len=4, condition_tensors=ct)
    assert [k for k in lm.state_dict() if k.startswith('transformer.layers.0.')] == [
        'transformer.layers.0.self_attn.in_proj_weight', 'transformer.layers.0.self_attn.out_proj.weight',
        'transformer.layers.0.linear1.weight', 'transformer.layers.0.linear2.weight',
        'transformer.layers.0.norm1.weight', 'transformer.layers.0.norm1.bias',
        'transformer.layers.0.norm2.weight', 'transformer.layers.0.norm2.bias',
        'transformer.layers.0.cross_attention.in_proj_weight', 'transformer.layers.0.cross_attention.out_proj.weight',
        'transformer.layers.0.norm_cross.weight', 'transformer.layers.0.norm_cross.bias']


def test_state_dict_keys_match_reference_golden():
    """Checkpoint compatibility: our module trees expose exactly the reference's parameter names."""
    from conftest import load_golden
    from audiocraft_amd.models import builders
    cfg, sd, _ = load_golden('lm_text')
    lm = builders.get_lm_model(dict(dim=cfg['dim'], num_heads=cfg['num_heads'], num_layers=cfg['num_layers'],
                                    n_q=cfg['n_q'], card=cfg['card'],
                                    conditioners={'description': {'kind': 't5', 'embedder': 'synthetic', 'dim': cfg['cond_dim']}},
                                    fuser={'cross': ['description']}), 'cpu', torch.float32)
    assert set(lm.state_dict().keys()) == set(sd.keys())
    for k, v in lm.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    # the transformer options no release uses: rotary buffers under transformer.rope.* AND every self_attn.rope.*
    # (the reference shares one module), xPos decay rates, LayerScale parameters -- same names, shapes and VALUES
    for name in ('lm_rope', 'lm_sin_rope'):
        cfg, sd, _ = load_golden(name)
        lm = builders.get_lm_model(dict(dim=cfg['dim'], num_heads=cfg['num_heads'], num_layers=cfg['num_layers'],
                                        n_q=cfg['n_q'], card=cfg['card'],
                                        conditioners={'description': {'kind': 't5', 'embedder': 'synthetic', 'dim': cfg['cond_dim']}},
                                        fuser={'cross': ['description']},
                                        **{k: cfg[k] for k in ('positional_embedding', 'xpos', 'past_context', 'layer_scale',
                                                               'positional_scale') if k in cfg}), 'cpu', torch.float32)
        own = lm.state_dict()
        assert set(own.keys()) == set(sd.keys()), set(own.keys()) ^ set(sd.keys())
        for k in own:
            if 'rope' in k:
                assert torch.allclose(own[k], sd[k], rtol=1e-6, atol=0), k
        lm.load_state_dict(sd)   # strict
    cfg, sd, _ = load_golden('codec_causal')
    sk = dict(channels=cfg['channels'], dimension=cfg['dimension'], n_filters=cfg['n_filters'],
              n_residual_layers=cfg['n_residual_layers'], ratios=cfg['ratios'], norm=cfg['norm'], causal=True,
              pad_mode=cfg['pad_mode'], true_skip=cfg['true_skip'], lstm=cfg['lstm'])
    m = builders.get_compression_model(dict(seanet=sk, rvq=dict(n_q=cfg['n_q'], bins=cfg['bins']), sample_rate=1200,
                                            frame_rate=75, channels=1, causal=True), 'cpu')
    assert set(m.state_dict().keys()) == set(sd.keys())


OPTION_KEYS = ('kv_repeat', 'qk_layer_norm', 'qk_layer_norm_cross', 'bias_attn', 'bias_ff', 'cross_attention_pos_emb',
               'cross_attention_pos_emb_scale', 'norm_first', 'bias_proj')


def options_lm_cfg(cfg):
    """builders.get_lm_model cfg of an options golden (tests/golden/make_options_golden.py)."""
    fuser = {'cross': ['description']} if cfg.get('cross_attention', True) else {'prepend': ['description']}
    if 'curve_frames' in cfg:
        fuser.update({'sum': ['genre'], 'input_interpolate': ['curve']})
    return dict(dim=cfg['dim'], num_heads=cfg['num_heads'], num_layers=cfg['num_layers'], n_q=cfg['n_q'], card=cfg['card'],
                hidden_scale=cfg['hidden_scale'], cfg_coef=cfg['cfg_coef'],
                conditioners={'description': {'kind': 't5', 'embedder': 'synthetic', 'dim': cfg['cond_dim'], 'length': cfg['Lc']}},
                fuser=fuser, codebooks_pattern={'modeling': 'delay', 'delay': {'delays': cfg['delays']}},
                **{k: cfg[k] for k in OPTION_KEYS if k in cfg})


@pytest.mark.parametrize('name', ['lm_kv_repeat', 'lm_qk_ln', 'lm_fuser_sum', 'lm_post_norm'])
def test_option_goldens_load_strictly(name):
    """kv_repeat narrows in_proj_weight, qk_layer_norm adds q_layer_norm / k_layer_norm under self_attn (and under
    cross_attention for qk_layer_norm_cross): same names and shapes as the reference's modules (transformer.py:196-222);
    a post-norm model (norm_first=False) has no out_norm (lm.py:171-173)."""
    from conftest import load_golden
    from audiocraft_amd.models import builders
    cfg, sd, _ = load_golden(name)
    lm = builders.get_lm_model(options_lm_cfg(cfg), 'cpu', torch.float32)
    own = lm.state_dict()
    extra = {k for k in sd if k.startswith('condition_provider.conditioners.') and '.description.' not in k}   # test-only conditioners
    assert set(own.keys()) == set(sd.keys()) - extra, set(own.keys()) ^ (set(sd.keys()) - extra)
    for k, v in own.items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    lm.load_state_dict({k: v for k, v in sd.items() if k not in extra})   # strict


def test_kv_repeat_expansion_is_the_reference_attention():
    """expand_kv_in_proj lays every shared key / value head out once per query head: an ordinary attention on the expanded
    projection IS the kv_repeat attention (checked through the oracle, which is pinned to the reference's golden)."""
    from conftest import load_golden
    from audiocraft_amd.models.lm import expand_kv_in_proj
    from oracle import lm as olm
    from test_oracle_golden import lm_cfg
    cfg, sd, a = load_golden('lm_kv_repeat')
    c = lm_cfg(cfg)
    assert c.kv_repeat == 2
    ex = dict(sd)
    for li in range(c.num_layers):
        p = f'transformer.layers.{li}.self_attn.'
        w, b = expand_kv_in_proj(sd[p + 'in_proj_weight'], sd[p + 'in_proj_bias'], c.dim, c.num_heads, c.kv_repeat)
        assert w.shape == (3 * c.dim, c.dim) and b.shape == (3 * c.dim,)
        ex[p + 'in_proj_weight'], ex[p + 'in_proj_bias'] = w, b
    import dataclasses
    plain = dataclasses.replace(c, kv_repeat=1)
    got = olm.lm_forward(ex, plain, a['tf_sequence'], a['cond_description'])
    assert torch.allclose(got, a['tf_logits'], atol=2e-5, rtol=1e-4)
    w, b = expand_kv_in_proj(sd[p + 'in_proj_weight'], None, c.dim, c.num_heads, 1)
    assert w is sd[p + 'in_proj_weight'] and b is None


def test_fuser_sum_interpolate_and_cross_pos_emb_host_side():
    """ConditionFuser: 'sum' / 'input_interpolate' conditions become the rows the embedding kernel adds (input_add_rows ==
    what the reference's in-place add / F.interpolate give for a call of T steps); cross_attention_pos_emb == the oracle's
    restatement (pinned to the reference by lm_fuser_sum.npz)."""
    import torch.nn.functional as F
    from audiocraft_amd.modules.conditioners import ConditionFuser
    from oracle import lm as olm
    g = torch.Generator().manual_seed(0)
    one, many = torch.randn(3, 1, 8, generator=g), torch.randn(3, 5, 8, generator=g)
    for T in (1, 2, 3, 4, 5, 7, 9, 10, 11, 13, 64, 1503):
        rows = ConditionFuser.input_add_rows([('sum', one), ('input_interpolate', many)], T)
        ref = one.expand(-1, T, -1) + F.interpolate(many.transpose(1, 2), size=T).transpose(1, 2)
        assert torch.equal(rows, ref), T
    for Tc, T in ((3, 9), (7, 3), (11, 1000), (1500, 1499), (6, 36), (29, 87)):   # ATen's f32 index arithmetic, not exact integers
        cond = torch.randn(2, Tc, 4, generator=g)
        assert torch.equal(ConditionFuser.input_add_rows([('input_interpolate', cond)], T),
                           F.interpolate(cond.transpose(1, 2), size=T).transpose(1, 2)), (Tc, T)
    assert torch.equal(ConditionFuser.input_add_rows([('sum', many)], 5), many)   # one frame per step
    with pytest.raises(RuntimeError):
        ConditionFuser.input_add_rows([('sum', many)], 4)
    assert ConditionFuser.input_add_rows([], 4) is None
    fuse = {'cross': ['description'], 'sum': ['genre'], 'input_interpolate': ['curve'], 'prepend': ['wav']}
    fuser = ConditionFuser(fuse, cross_attention_pos_emb=True, cross_attention_pos_emb_scale=0.7)
    src = torch.randn(3, 6, 8, generator=g)
    m = torch.ones(3, 6)
    conds = {'description': (src, m), 'genre': (one, m[:, :1]), 'curve': (many, m[:, :5])}
    prepend, cross = fuser.fuse(conds)
    assert prepend is None and torch.allclose(cross, olm.cross_pos_emb(src, 0.7), atol=1e-6)
    assert [op for op, _ in fuser.input_ops(conds)] == ['sum', 'input_interpolate']
    with pytest.raises(NotImplementedError):   # the reference would add to the prepended rows too: not built
        fuser.fuse({'wav': (src, m), 'genre': (one, m[:, :1])})
    with pytest.raises(AssertionError):
        fuser.input_ops({'unknown': (src, m)})


def test_loader_passes_transformer_options_and_bench_tags():
    from audiocraft_amd.models import loaders
    xp = {'transformer_lm': {'dim': 16, 'num_heads': 4, 'num_layers': 2, 'positional_embedding': 'sin_rope', 'xpos': False,
                             'past_context': 12, 'layer_scale': None, 'positional_scale': 0.5},
          'conditioners': {'args': {'merge_text_conditions_p': 0.25}}, 'fuser': {'cross': []}}
    cfg = loaders.lm_cfg_from_xp(loaders.parse_cfg(xp))
    assert cfg['positional_embedding'] == 'sin_rope' and cfg['past_context'] == 12 and cfg['positional_scale'] == 0.5
    assert 'layer_scale' not in cfg and cfg['xpos'] is False
    xp['transformer_lm'].update(kv_repeat=2, qk_layer_norm_cross=True)
    xp['fuser'] = {'cross': ['description'], 'sum': ['genre'], 'cross_attention_pos_emb': True, 'cross_attention_pos_emb_scale': 0.5}
    cfg = loaders.lm_cfg_from_xp(loaders.parse_cfg(xp))
    assert cfg['kv_repeat'] == 2 and cfg['qk_layer_norm_cross'] is True and 'qk_layer_norm' not in cfg
    assert cfg['fuser'] == {'cross': ['description'], 'sum': ['genre']} and cfg['cross_attention_pos_emb_scale'] == 0.5
    import argparse
    import bench
    ns = argparse.Namespace(model='facebook/musicgen-medium', batch=8, duration=30.0, greedy=False)
    assert bench.config_tag(ns).endswith('configs[2]')
    ns = argparse.Namespace(model='facebook/musicgen-small', batch=1, duration=10.0, greedy=True)
    assert bench.config_tag(ns).endswith('configs[1]')
    assert 'not a BASELINE' in bench.config_tag(argparse.Namespace(model='x/y', batch=3, duration=5.0, greedy=False))


def test_loader_finds_checkpoints_in_the_huggingface_hub_cache(tmp_path, monkeypatch):
    """The reference resolves model names with hf_hub_download (loaders.py:63-70), so a checkpoint it fetched earlier sits in
    the hub's cache layout; the loaders find it there (under $AUDIOCRAFT_CACHE_DIR or HuggingFace's default cache) without
    any network access."""
    pytest.importorskip('huggingface_hub')
    from audiocraft_amd.models import loaders
    repo = tmp_path / 'models--facebook--musicgen-tiny'
    (repo / 'snapshots' / 'abc123').mkdir(parents=True)
    (repo / 'refs').mkdir()
    (repo / 'refs' / 'main').write_text('abc123')
    torch.save({'marker': 7}, repo / 'snapshots' / 'abc123' / 'state_dict.bin')
    monkeypatch.setenv('AUDIOCRAFT_CACHE_DIR', str(tmp_path))
    assert loaders._get_state_dict('facebook/musicgen-tiny', 'state_dict.bin') == {'marker': 7}
    with pytest.raises(FileNotFoundError):
        loaders._get_state_dict('facebook/musicgen-tiny', 'compression_state_dict.bin')
    with pytest.raises(FileNotFoundError):
        loaders._get_state_dict('facebook/musicgen-absent', 'state_dict.bin')


def test_loader_roundtrip(tmp_path):
    """Reference export format (utils/export.py:58-79) -> loaders.load_lm_model (construction on CPU only)."""
    from audiocraft_amd.models import builders, loaders
    xp = {'transformer_lm': {'dim': 16, 'num_heads': 4, 'num_layers': 2, 'hidden_scale': 4, 'n_q': 4, 'card': 400},
          'conditioners': {'description': {'model': 't5', 't5': {'name': 't5-small'}}},
          'fuser': {'cross': ['description'], 'prepend': [], 'sum': [], 'input_interpolate': []},
          'codebooks_pattern': {'modeling': 'delay', 'delay': {'delays': [0, 1, 2, 3]}},
          'classifier_free_guidance': {'inference_coef': '${cfgc}'}, 'cfgc': 2.5}
    cfg = loaders.lm_cfg_from_xp(loaders.parse_cfg(xp))
    assert cfg['cfg_coef'] == 2.5 and cfg['conditioners']['description']['name'] == 't5-small'
    lm = builders.get_lm_model(cfg, 'cpu', torch.float32)
    path = tmp_path / 'state_dict.bin'
    loaders.export_lm(lm, str(path), xp)
    lm2 = loaders.load_lm_model(str(tmp_path), device='cpu', weight_dtype=torch.float32)
    for (k1, v1), (k2, v2) in zip(lm.state_dict().items(), lm2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    with pytest.raises(FileNotFoundError):
        loaders.load_lm_model('facebook/musicgen-small', device='cpu')


def test_stereo_interleave_wrapper():
    """InterleaveStereoCompressionModel (reference encodec.py:397-506): '(b c) k t -> b (k c) t' and the
    per-timestep variant, checked on a stand-in mono codec and against outputs of the reference class."""
    from audiocraft_amd.models.encodec import CompressionModel, InterleaveStereoCompressionModel

    class Mono(CompressionModel):
        channels, frame_rate, sample_rate, cardinality, num_codebooks, total_codebooks = 1, 50, 32000, 2048, 4, 4

        def __init__(self):
            torch.nn.Module.__init__(self)

        def encode(self, x):
            T = x.shape[-1] // 4
            return (x[:, 0, :T * 4:4].abs() * 1000).long()[:, None].repeat(1, 4, 1) + torch.arange(4).view(1, 4, 1), None

        def decode(self, codes, scale=None):
            return codes.float().sum(1, keepdim=True).repeat_interleave(4, dim=-1)

        def set_num_codebooks(self, n): pass
        def forward(self, x): pass
        def decode_latent(self, c): pass

    x = torch.randn(3, 2, 40)
    mono = Mono()
    left, _ = mono.encode(x[:, :1])
    right, _ = mono.encode(x[:, 1:])
    st = InterleaveStereoCompressionModel(mono)
    codes, scale = st.encode(x)
    assert scale is None and codes.shape == (3, 8, 10) and st.num_codebooks == 8 and st.channels == 2
    assert torch.equal(codes[:, 0::2], left) and torch.equal(codes[:, 1::2], right)
    l2, r2 = st.get_left_right_codes(codes)
    assert torch.equal(l2, left) and torch.equal(r2, right)
    wav = st.decode(codes)
    assert wav.shape == (3, 2, 40) and torch.equal(wav[:, :1], mono.decode(left))
    st = InterleaveStereoCompressionModel(mono, per_timestep=True)
    codes, _ = st.encode(x)
    assert codes.shape == (3, 4, 20) and st.frame_rate == 100 and st.num_codebooks == 4
    assert torch.equal(codes[..., 0::2], left) and torch.equal(codes[..., 1::2], right)
    assert torch.equal(st.decode(codes)[:, 1:], mono.decode(right))
    # and against the reference class itself (tests/golden/stereo.npz, recorded by tests/golden/make_host_golden.py
    # from audiocraft.models.encodec.InterleaveStereoCompressionModel over the same stand-in codec)
    import numpy as np
    gold = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'stereo.npz'))
    xg = torch.from_numpy(gold['x'])
    for tag, per_timestep in (('k', False), ('t', True)):
        st = InterleaveStereoCompressionModel(Mono(), per_timestep=per_timestep)
        codes, _ = st.encode(xg)
        assert torch.equal(codes, torch.from_numpy(gold[f'codes_{tag}']))
        assert torch.equal(st.decode(codes), torch.from_numpy(gold[f'wav_{tag}']))
        assert [st.num_codebooks, st.frame_rate, st.channels, st.total_codebooks] == gold[f'meta_{tag}'].tolist()
    # set_num_codebooks counts the WRAPPED model's codebooks, before the interleaving (reference encodec.py:428-433, as
    # `compression_model_n_q` is applied by builders.py:345-348); found by running the reference class side by side
    seen = []
    mono = Mono()
    mono.set_num_codebooks = seen.append
    InterleaveStereoCompressionModel(mono).set_num_codebooks(4)
    InterleaveStereoCompressionModel(mono, per_timestep=True).set_num_codebooks(3)
    assert seen == [4, 3]
    with pytest.raises(NotImplementedError):   # encodec.py:504-506
        InterleaveStereoCompressionModel(mono).decode_latent(torch.zeros(1, 8, 4, dtype=torch.long))


def test_genmodel_reads_experiment_config_of_the_lm():
    """BaseGenModel (reference genmodel.py:49-62, builders.py:338-351): an LM loaded from a released checkpoint
    carries `cfg`; `interleave_stereo_codebooks.use` wraps the codec, `dataset.segment_duration` is max_duration."""
    from audiocraft_amd.models import AudioGen, MusicGen, builders
    from audiocraft_amd.models.encodec import InterleaveStereoCompressionModel
    lm = builders.get_lm_model(dict(dim=16, num_heads=4, num_layers=1, n_q=8, card=400,
                                    codebooks_pattern={'modeling': 'delay', 'delay': {'delays': [0, 0, 1, 1, 2, 2, 3, 3]}},
                                    conditioners={'description': {'kind': 't5', 'embedder': 'synthetic', 'dim': 16}},
                                    fuser={'cross': ['description']}), 'cpu', torch.float32)
    lm.cfg = {'interleave_stereo_codebooks': {'use': True, 'per_timestep': False}, 'compression_model_n_q': None,
              'dataset': {'segment_duration': 30}}
    mg = MusicGen('stereo-stub', builders.get_debug_compression_model('cpu'), lm)
    assert isinstance(mg.compression_model, InterleaveStereoCompressionModel)
    assert mg.audio_channels == 2 and mg.compression_model.num_codebooks == 8 and mg.max_duration == 30
    del lm.cfg
    with pytest.raises(ValueError):
        MusicGen('no-duration', builders.get_debug_compression_model('cpu'), lm)
    ag = AudioGen('plain', builders.get_debug_compression_model('cpu', sample_rate=16000), lm, max_duration=10)
    assert ag.sample_rate == 16000 and ag.duration == 5 and ag.extend_stride == 2
    # MusicGen-Style's parameter setter (musicgen.py:134-153): forwarded to a `self_wav` conditioner that offers set_params,
    # the reference's assertion otherwise
    plain = MusicGen('plain', builders.get_debug_compression_model('cpu'), lm, max_duration=30)
    with pytest.raises(AssertionError, match='MusicGen-Style'):
        plain.set_style_conditioner_params(eval_q=2)
    assert not hasattr(ag, 'set_style_conditioner_params')    # a MusicGen method only, as in the reference
    seen = {}

    class Style(torch.nn.Module):
        def set_params(self, **kw):
            seen.update(kw)
    lm.condition_provider.conditioners['self_wav'] = Style()
    try:
        mg2 = MusicGen('style-stub', builders.get_debug_compression_model('cpu'), lm, max_duration=30)
        mg2.set_style_conditioner_params(eval_q=2, excerpt_length=1.5)
        assert seen == dict(eval_q=2, excerpt_length=1.5, ds_factor=None, encodec_n_q=None)
    finally:
        del lm.condition_provider.conditioners['self_wav']
    assert builders.ENCODEC_16KHZ['seanet']['ratios'] == [8, 5, 4, 2] and builders.audiogen_lm_cfg()['dim'] == 1536


def test_windowed_generation_matches_reference_trace():
    """Durations above max_duration (reference genmodel.py:193-260, musicgen.py:290-337): the windows' (prompt length,
    max_gen_len), the periodically tiled melody each one carries and the stitched tokens are those recorded from the
    unmodified reference with a deterministic stand-in for `lm.generate` (tests/golden/make_host_golden.py)."""
    import json
    from audiocraft_amd.models import MusicGen
    from audiocraft_amd.modules.conditioners import WavCondition
    cases = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'windowing.json')))
    assert len(cases) >= 6
    for c in cases:
        mg = MusicGen.get_pretrained('debug', device='cpu')
        mg.max_duration = c['max_duration']
        mg.set_generation_params(duration=c['duration'], extend_stride=c['extend_stride'])
        assert mg.frame_rate == c['frame_rate'] and mg.sample_rate == c['sample_rate']
        calls = []

        def generate(prompt, attributes, callback=None, max_gen_len=256, **kw):
            B, T0 = len(attributes), 0 if prompt is None else prompt.shape[-1]
            mel = []
            for a in attributes:
                w = a.wav['self_wav']
                mel.append([int(w.length[0]), [round(float(v), 6) for v in w.wav.flatten()[:3]],
                            round(float(w.wav.double().sum()), 4)])
            calls.append({'prompt_len': T0, 'max_gen_len': int(max_gen_len), 'melody': mel})
            out = torch.empty(B, 4, max_gen_len, dtype=torch.long)
            if prompt is not None:
                out[..., :T0] = prompt
            out[..., T0:] = (torch.arange(T0, max_gen_len) * 7 + len(calls) * 13) % 400
            return out

        mg.lm.generate = generate
        attributes, _ = mg._prepare_tokens_and_attributes(['a', 'b'], None)
        if c['melody_len']:
            for i, attr in enumerate(attributes):
                m = torch.arange(c['melody_len'], dtype=torch.float32)[None] * (i + 1) * 1e-3
                attr.wav['self_wav'] = WavCondition(m[None], torch.tensor([m.shape[-1]]), sample_rate=[mg.sample_rate],
                                                    path=[None])
        prompt = None
        if c['prompt_len']:
            prompt = (torch.arange(c['prompt_len'])[None, None] + torch.arange(4)[None, :, None]).repeat(2, 1, 1) % 400
        tokens = mg._generate_tokens(attributes, prompt)
        assert list(tokens.shape) == c['tokens_shape']
        assert tokens[0, 0].tolist() == c['tokens_row']
        assert len(calls) == len(c['calls'])
        for got, ref in zip(calls, c['calls']):
            assert (got['prompt_len'], got['max_gen_len']) == (ref['prompt_len'], ref['max_gen_len'])
            for (gl, gh, gs), (rl, rh, rs) in zip(got['melody'], ref['melody']):
                assert gl == rl and gh == pytest.approx(rh, abs=1e-5) and gs == pytest.approx(rs, rel=1e-6)


def test_streamable_conv_geometry_and_constructor_checks():
    """Host geometry of the conv wrappers (reference tests/modules/test_conv.py:152-203): output length of
    StreamableConv1d = the reference's "last window is full" formula, StreamableConvTranspose1d's constructor
    rejects trim_right_ratio outside [0, 1] or != 1 for non-causal layers; the SEANet hop follows."""
    import math
    import random
    from audiocraft_amd.modules.seanet import SEANetEncoder, StreamableConv1d, StreamableConvTranspose1d

    def ref_len(length, kernel_size, stride, dilation):
        padding_total = (kernel_size - 1) * dilation - (stride - 1)
        n_frames = (length - kernel_size + padding_total) / stride + 1
        ideal_length = (math.ceil(n_frames) - 1) * stride + (kernel_size - padding_total)
        return ideal_length // stride

    rng = random.Random(0)
    for _ in range(50):
        T = rng.randrange(1, 100_000)
        for causal in (False, True):
            for k, s, d in [(4, 1, 1), (4, 2, 1), (3, 1, 3), (10, 5, 1), (3, 2, 3), (7, 1, 1), (16, 8, 1)]:
                conv = StreamableConv1d(2, 1, kernel_size=k, stride=s, dilation=d, causal=causal, device='cpu')
                assert conv.out_length(T) == ref_len(T, k, s, d), (T, k, s, d, causal)
    with pytest.raises(AssertionError):
        StreamableConvTranspose1d(2, 1, kernel_size=4, causal=False, trim_right_ratio=0.5, device='cpu')
    with pytest.raises(AssertionError):
        StreamableConvTranspose1d(2, 1, kernel_size=4, causal=True, trim_right_ratio=-1., device='cpu')
    with pytest.raises(AssertionError):
        StreamableConvTranspose1d(2, 1, kernel_size=4, causal=True, trim_right_ratio=2, device='cpu')
    enc = SEANetEncoder(channels=1, dimension=16, n_filters=4, ratios=[8, 5, 4, 4], device='cpu')
    assert enc.hop_length == 640


# ------------------------------------------------------------------------------------------ reference-written checkpoints

CKPT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ckpt_ref')


def test_loader_reads_reference_written_lm_checkpoints():
    """Files produced by the reference's own audiocraft/utils/export.py (tests/golden/make_ckpt_golden.py): xp.cfg composed
    from the reference's YAML files incl. `conditioners.args`, `attribute_dropout.args`, `chroma_stem.cache_path`; a
    third-party buffer under the chroma conditioner; and the nn.MultiheadAttention key layout."""
    from audiocraft_amd.models import loaders
    from conftest import load_golden
    for sub, golden in (('text', 'lm_text'), ('melody', 'lm_melody')):
        lm = loaders.load_lm_model(os.path.join(CKPT, sub), device='cpu', weight_dtype=torch.float32)
        cfg, sd, _ = load_golden(golden)
        got = lm.state_dict()
        assert set(got.keys()) == set(sd.keys())
        for k in sd:
            assert torch.equal(got[k], sd[k]), k
        assert lm.dim == cfg['dim'] and lm.num_layers == cfg['num_layers'] and lm.cfg_coef == cfg['cfg_coef']
        assert lm.has_cross_attention == cfg['cross_attention']
        assert lm.pattern_provider.delays == cfg['delays']
    assert list(lm.fuser.fuse2cond['prepend']) == ['self_wav', 'description']     # melody: prepend order of chroma2music.yaml
    cw = lm.condition_provider.conditioners['self_wav']
    assert cw.chroma_len == 235 and cw.chroma.argmax and cw.chroma.nfft == 16384
    assert lm.cfg['conditioners']['self_wav']['chroma_stem']['cache_path'].startswith('/checkpoint')   # cfg kept verbatim
    # nn.MultiheadAttention layout: `self_attn.mha.in_proj_weight` -> `self_attn.in_proj_weight`
    lm = loaders.load_lm_model(os.path.join(CKPT, 'mha'), device='cpu', weight_dtype=torch.float32)
    raw = torch.load(os.path.join(CKPT, 'mha', 'state_dict.bin'), map_location='cpu', weights_only=False)['best_state']
    assert any('.mha.' in k for k in raw)
    assert torch.equal(lm.state_dict()['transformer.layers.1.cross_attention.in_proj_weight'],
                       raw['transformer.layers.1.cross_attention.mha.in_proj_weight'])


def test_loader_reads_reference_written_codec_checkpoints(monkeypatch, tmp_path):
    from audiocraft_amd.models import loaders
    from audiocraft_amd.models.encodec import CompressionModel
    from conftest import load_golden
    m = loaders.load_compression_model(os.path.join(CKPT, 'text'), device='cpu')
    cfg, sd, _ = load_golden('codec_noncausal')
    got = m.state_dict()
    assert set(got.keys()) == set(sd.keys())
    for k in sd:
        assert torch.equal(got[k], sd[k]), k
    assert m.sample_rate == cfg['sample_rate'] and m.frame_rate == cfg['frame_rate'] and m.num_codebooks == cfg['n_q']
    # {'pretrained': 'facebook/encodec_32khz'}: resolved through CompressionModel.get_pretrained -> the cache directory
    stub = torch.load(os.path.join(CKPT, 'stub', 'compression_state_dict.bin'), map_location='cpu', weights_only=False)
    assert stub['pretrained'] == 'facebook/encodec_32khz'
    cache = tmp_path / 'facebook--encodec_32khz'
    cache.mkdir()
    import shutil
    shutil.copy(os.path.join(CKPT, 'text', 'compression_state_dict.bin'), cache / 'compression_state_dict.bin')
    monkeypatch.setenv('AUDIOCRAFT_CACHE_DIR', str(tmp_path))
    m2 = loaders.load_compression_model(os.path.join(CKPT, 'stub'), device='cpu')
    assert torch.equal(m2.state_dict()['quantizer.vq.layers.0._codebook.embed'], sd['quantizer.vq.layers.0._codebook.embed'])
    m3 = CompressionModel.get_pretrained(os.path.join(CKPT, 'text'), device='cpu')
    assert not m3.training and m3.cardinality == cfg['bins']
    with pytest.raises(NotImplementedError):
        CompressionModel.get_pretrained('dac_44khz')


# ------------------------------------------------------------------------------------------ output stage (f4)

def test_audio_write_and_normalisation(tmp_path):
    """audio_write / normalize_audio (reference data/audio.py:159-231, data/audio_utils.py:62-152)."""
    import math
    import wave
    import numpy as np
    from audiocraft_amd.data_audio import audio_write, loudness, normalize_audio
    sr = 32000
    t = torch.arange(2 * sr) / sr
    tone = torch.sin(2 * math.pi * 997.0 * t)[None]
    # ITU-R BS.1770: a 997 Hz sine of amplitude a in one channel reads -3.01 + 20 log10(a) LKFS
    assert abs(loudness(0.5 * tone, sr) - (-9.03)) < 0.05
    assert abs(loudness(0.1 * tone, sr) - (-23.01)) < 0.05
    assert abs(loudness(torch.cat([0.5 * tone, 0.5 * tone]), sr) - (-6.02)) < 0.06    # two channels: +3.01 dB
    # (torchaudio's biquads clamp their output to [-1, 1]: a full-scale tone, boosted by the K-weighting shelf, reads lower)
    assert loudness(tone, sr) < -3.01
    wav = 0.05 * tone
    peak = normalize_audio(wav, strategy='peak')
    assert abs(peak.abs().max().item() - 10 ** (-1 / 20)) < 1e-6
    assert normalize_audio(3 * tone, strategy='clip').abs().max().item() <= 10 ** (-1 / 20) + 1e-6
    rms = normalize_audio(wav.clone(), strategy='rms')
    assert abs(rms.pow(2).mean().sqrt().item() - 10 ** (-18 / 20)) < 1e-4
    loud = normalize_audio(wav.clone(), strategy='loudness', sample_rate=sr)
    assert abs(loudness(loud, sr) - (-14.0)) < 0.05
    assert normalize_audio(1e-4 * tone, strategy='loudness', sample_rate=sr).abs().max() < 2e-4    # below the energy floor
    # against the oracle's independent restatement of torchaudio's meter (oracle/loudness.py), on noise: white, coloured,
    # multi-channel, with near-silent stretches (both gates active), loud enough for the biquads' clamp, other sample rates
    from oracle import loudness as ol
    rng = np.random.default_rng(0)
    for C, amp, rate in ((1, 0.1, 32000), (2, 0.3, 32000), (3, 0.05, 16000), (1, 1.5, 24000), (5, 0.2, 44100)):
        x = rng.standard_normal((C, int(1.7 * rate))) * amp
        x = np.cumsum(x, axis=-1) * 0.05 + x if C == 2 else x       # a coloured case
        x[:, rate // 2:rate] *= 1e-4
        assert abs(loudness(torch.tensor(x), rate) - ol.loudness(x, rate)) < 1e-6
    assert loudness(torch.zeros(1, sr), sr) == -float('inf') == ol.loudness(np.zeros((1, sr)), sr)
    p = audio_write(tmp_path / 'sub' / 'clip', torch.cat([wav, -wav]), sr, strategy='peak')
    assert p.name == 'clip.wav' and p.exists()
    with wave.open(str(p)) as f:
        assert (f.getnchannels(), f.getsampwidth(), f.getframerate(), f.getnframes()) == (2, 2, sr, 2 * sr)
        pcm = np.frombuffer(f.readframes(f.getnframes()), dtype='<i2').reshape(-1, 2)
    assert abs(int(np.abs(pcm).max()) - round(10 ** (-1 / 20) * 32768)) <= 1 and (pcm[:, 0] == -pcm[:, 1]).all()
    with pytest.raises(RuntimeError):
        audio_write(tmp_path / 'x', wav, sr, format='aiff')
    with pytest.raises(ValueError):
        audio_write(tmp_path / 'x', torch.zeros(1, 1, 10), sr)


def test_normalize_audio_vs_reference_golden(capsys):
    """normalize_audio / i16_pcm / f32_pcm against arrays recorded from the unmodified reference
    (tests/golden/make_audio_norm_golden.py; reference data/audio_utils.py:104-192): every strategy that needs no third-party
    meter, normalize on and off, two headroom settings, a quiet, a hot and a two-channel clip."""
    import numpy as np
    from audiocraft_amd.data_audio import f32_pcm, i16_pcm, normalize_audio
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'audio_norm.npz'))
    n = 0
    for key in z.files:
        if key.count('|') != 4:
            continue
        name, strategy, normalize, hp, hr = key.split('|')
        wav = torch.tensor(z[f'in_{name}'])
        keep = wav.clone()
        y = normalize_audio(wav, normalize=bool(int(normalize)), strategy=strategy, peak_clip_headroom_db=float(hp),
                            rms_headroom_db=float(hr))
        np.testing.assert_allclose(y.numpy(), z[key], rtol=2e-6, atol=1e-7, err_msg=key)
        assert torch.equal(wav, keep), f"{key}: the input was modified"
        n += 1
    assert n == 36
    np.testing.assert_array_equal(normalize_audio(torch.tensor(z['in_none']), strategy='none').numpy(), z['none'])
    with pytest.raises(AssertionError):
        normalize_audio(torch.tensor(z['in_hot']), strategy='none')
    with pytest.raises(AssertionError):
        normalize_audio(torch.tensor(z['in_quiet']), strategy='loud')
    with pytest.raises(AssertionError):
        normalize_audio(torch.tensor(z['in_quiet']), strategy='loudness')      # needs the sample rate
    # the overshoot report (stderr) appears only when asked for and only when something is clamped
    capsys.readouterr()
    normalize_audio(torch.tensor(z['in_hot']), strategy='rms', rms_headroom_db=3.0, log_clipping=True, stem_name='take7')
    assert 'take7' in capsys.readouterr().err
    normalize_audio(torch.tensor(z['in_hot']), strategy='rms', rms_headroom_db=3.0)
    normalize_audio(torch.tensor(z['in_quiet']), strategy='rms', rms_headroom_db=30.0, log_clipping=True)
    assert capsys.readouterr().err == ''
    pcm = torch.tensor(z['pcm_in'])
    np.testing.assert_array_equal(i16_pcm(pcm).numpy(), z['pcm_i16'])
    np.testing.assert_array_equal(i16_pcm(pcm[:, :4]).numpy(), z['pcm_i16_noplus'])
    np.testing.assert_array_equal(f32_pcm(torch.tensor([[0, 16384, -32768, 32767]], dtype=torch.int16)).numpy(), z['pcm_f32_from_i16'])


def test_bench_cpu_baseline_imports_without_the_reference_tree():
    """bench.py's cpu_baseline leg imports oracle.ref_baseline on every box; where /root/reference does not exist (the GPU
    box) that import must succeed and report `available() == False` (kind "port"), not raise."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, AUDIOCRAFT_REFERENCE='/nonexistent/reference')
    code = ("import sys; sys.path.insert(0, %r); from oracle import ref_baseline as rb; "
            "assert rb.available() is False; import bench; print('ok')" % ROOT)
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith('ok'), out.stderr[-2000:]


# ------------------------------------------------------------------------------------------ MultiBandDiffusion host logic

@pytest.mark.parametrize('name', ['mbd_unet', 'mbd_unet_bilstm'])
def test_diffusion_unet_takes_reference_state_dicts(name):
    """DiffusionUnet's parameter names / shapes are the reference's (strict load of the golden's reference-made state dict);
    there is no CPU forward."""
    from conftest import load_golden
    from audiocraft_amd.models.unet import DiffusionUnet
    cfg, sd, _ = load_golden(name)
    m = DiffusionUnet(**{k: cfg[k] for k in ('chin', 'hidden', 'depth', 'growth', 'max_channels', 'num_steps', 'emb_all_layers', 'bilstm',
                                             'codec_dim', 'kernel', 'stride', 'norm_groups', 'res_blocks')})
    m.load_state_dict(sd, strict=True)
    assert set(m.state_dict()) == set(sd)
    with pytest.raises(RuntimeError, match="MI355X"):
        m(torch.zeros(1, 1, 64), 3, torch.zeros(1, cfg['codec_dim'], 4))


def test_band_filters_and_schedule_host_side():
    """The host-built SplitBands filters equal the oracle's; the schedule's scalar side (betas, alpha-bar, YAML-string floats)."""
    from oracle import mbd as ombd
    from audiocraft_amd.modules.diffusion_schedule import NoiseSchedule, band_filters, betas_from_alpha_bar
    for sr, n in ((32000, 32), (24000, 8), (16000, 4)):
        cut = ombd.mel_frequencies(n + 1, 0, sr / 2)[1:-1] / sr
        filt, half = ombd.lowpass_filters(cut)
        mine, mhalf = band_filters(sr, n)
        assert mhalf == half and torch.equal(mine, filt)
        assert torch.allclose(mine.sum(1), torch.ones(n - 1), atol=1e-6)
    s = NoiseSchedule(beta_t0='1e-05', beta_t1=0.029, beta_exp=7.5, num_steps=100, device='cpu')
    sc = ombd.ScheduleConfig(beta_t0=1e-5, beta_t1=0.029, beta_exp=7.5, num_steps=100)
    assert torch.equal(s.betas, sc.betas)
    assert torch.equal(s.get_alpha_bar(41), ombd.alpha_bar_at(sc, 41))
    ab = s.get_alpha_bar()[[0, 20, 40, 99]]
    assert torch.allclose(1 - betas_from_alpha_bar(ab), torch.cat([ab[:1], ab[1:] / ab[:-1]]))


def test_multiband_processor_state_dict_has_the_release_layout():
    """Released `processor_state` dicts carry julius' low-pass bank as `split_bands.lowpass.filters` (a persistent buffer of
    the reference's julius.SplitBands; the reference strict-loads those files): the key is written, accepted by a strict load,
    and the checkpoint's bank replaces the computed one like a loaded buffer does in the reference."""
    from audiocraft_amd.modules.diffusion_schedule import MultiBandProcessor, band_filters
    p = MultiBandProcessor(n_bands=4, sample_rate=16000, num_samples=10, power_std=[1., 1., .5, .5])
    sd = p.state_dict()
    assert set(sd) == {'counts', 'sum_x', 'sum_x2', 'sum_target_x2', 'split_bands.lowpass.filters'}
    bank, half = band_filters(16000, 4)
    assert sd['split_bands.lowpass.filters'].shape == (3, 1, 2 * half + 1)
    assert torch.equal(sd['split_bands.lowpass.filters'][:, 0], bank)
    sd = {k: v.clone() for k, v in sd.items()}
    sd['counts'] += 7
    sd['split_bands.lowpass.filters'] = sd['split_bands.lowpass.filters'] * 1.5      # "the checkpoint's" bank
    q = MultiBandProcessor(n_bands=4, sample_rate=16000, num_samples=10, power_std=[1., 1., .5, .5])
    q.load_state_dict(sd, strict=True)
    assert float(q.counts) == 7 and torch.equal(q.split_bands._host[0], bank * 1.5) and q.split_bands._host[1] == half
    assert q.filters_max_abs_diff > 0
    q.load_state_dict({k: v for k, v in sd.items() if not k.startswith('split_bands')}, strict=True)   # hand-built dicts still load
    bad = dict(sd)
    bad['split_bands.lowpass.filters'] = torch.zeros(2, 1, 5)
    with pytest.raises(RuntimeError, match='low-pass'):
        MultiBandProcessor(n_bands=4, sample_rate=16000).load_state_dict(bad)


def test_mbd_cfg_resolves_only_what_the_loader_reads():
    """An xp config whose unrelated nodes cannot be resolved (dora.dir = ${oc.env:USER} with USER unset, `???`) must load."""
    from audiocraft_amd.models.loaders import _pick_mbd_cfg

    class Node(dict):   # stands in for an OmegaConf DictConfig: resolving the poisoned sub-tree raises
        pass

    def to_container(node):
        if node.get('_poison'):
            raise RuntimeError('InterpolationResolutionError')
        return {k: v for k, v in node.items()}

    cfg = Node(channels=1, schedule=Node(num_steps=10), diffusion_unet=Node(hidden=8), processor=Node(use=False),
               dora=Node(_poison=True, dir='${oc.env:USER}'), datasource=Node(_poison=True))
    out = _pick_mbd_cfg(cfg, lambda n: isinstance(n, Node), to_container)
    assert out == {'channels': 1, 'schedule': {'num_steps': 10}, 'diffusion_unet': {'hidden': 8}, 'processor': {'use': False}}


def test_philox_restatement_known_answers():
    """oracle/sampler.py against the published known-answer vectors of Philox4x32-10 (Random123 kat_vectors): the host replay
    of the device sampler's random stream is pinned to the generator's definition, not to the device."""
    import numpy as np
    from oracle.sampler import philox4x32_10, race, uniforms
    kats = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
            ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
            ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
             (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for c, k, want in kats:
        got = philox4x32_10(*[np.uint32(v) for v in c], k[0], k[1])
        assert tuple(int(v) for v in got) == want
    u = uniforms(2048, 5, 77, 0x1234567890abcdef)
    assert u.min() > 0 and u.max() < 1 and abs(u.mean() - 0.5) < 0.03
    # the race is torch.multinomial's algorithm: over many independent positions it samples the top-k renormalised distribution
    rng = np.random.default_rng(0)
    p = rng.dirichlet(np.ones(16) * 0.7)
    kth = np.sort(p)[-5]
    counts = np.zeros(16)
    for step in range(4000):
        tok, margin, _ = race(p, 5, 3, step, 99)
        counts[tok] += 1
        assert 0 <= margin <= 1
    exp = np.where(p >= kth, p, 0) / p[p >= kth].sum() * 4000
    assert counts[p < kth].sum() == 0
    chi2 = ((counts[p >= kth] - exp[p >= kth]) ** 2 / exp[p >= kth]).sum()
    assert chi2 < 20, chi2


def test_fuser_first_call_inputs_is_the_reference_loop():
    """ConditionFuser.first_call_inputs against a literal replay of the reference's fuser loop (conditioners.py:1730-1748) on
    a random input: conditions in dict order, `input += cond` / F.interpolate to the CURRENT input length / prepend in front."""
    import torch.nn.functional as F
    from audiocraft_amd.modules.conditioners import ConditionFuser
    g = torch.Generator().manual_seed(4)
    B, d = 3, 8
    for order in (['description', 'genre', 'curve'], ['genre', 'description', 'curve'], ['curve', 'wav', 'genre', 'description'],
                  ['genre', 'curve', 'description']):
        fuser = ConditionFuser({'prepend': ['description', 'wav'], 'sum': ['genre'], 'input_interpolate': ['curve']})
        conds = {'description': torch.randn(B, 5, d, generator=g), 'wav': torch.randn(B, 3, d, generator=g),
                 'genre': torch.randn(B, 1, d, generator=g), 'curve': torch.randn(B, 7, d, generator=g)}
        ct = {k: (conds[k], torch.ones(B, conds[k].shape[1], dtype=torch.int64)) for k in order}
        for T in (1, 4, 9):
            x = torch.randn(B, T, d, generator=g)
            ref = x.clone()
            for k in order:                       # the reference's loop, first streaming call
                op, cond = fuser.cond2fuse[k], conds[k]
                if op == 'sum':
                    ref += cond
                elif op == 'input_interpolate':
                    ref += F.interpolate(cond.transpose(1, 2), size=ref.shape[1]).transpose(1, 2)
                else:
                    ref = torch.cat([cond, ref], dim=1)
            prepend, add = fuser.first_call_inputs(ct, T)
            P = sum(conds[k].shape[1] for k in order if fuser.cond2fuse[k] == 'prepend')
            assert prepend.shape == (B, P, d) and add.shape == (B, T, d)
            got = torch.cat([prepend, x + add], dim=1)
            assert torch.allclose(got, ref, atol=1e-6), (order, T)
            seen, mixed = False, False
            for k in order:
                seen = seen or fuser.cond2fuse[k] == 'prepend'
                mixed = mixed or (seen and fuser.cond2fuse[k] != 'prepend')
            assert fuser.mixed_order(ct) == mixed
            if not mixed:                          # the common order: the plain concatenation + the token additions
                p0, _ = fuser.fuse(ct)
                assert torch.equal(prepend, p0.float())
                assert torch.allclose(add, ConditionFuser.input_add_rows(fuser.input_ops(ct), T))
            else:
                with pytest.raises(NotImplementedError):
                    fuser.fuse(ct)


@pytest.mark.parametrize('wdt,tol', [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
def test_cross_fold_tables_are_the_cross_attention_block(wdt, tol):
    """modules/cross_fold.py: the per-generate tables G / G2 / U (+ CS, BS) reproduce norm_cross -> q projection -> scores ->
    softmax -> values -> out projection of the reference's layer (transformer.py:344-361, 563-566), the query's LayerNorm
    in its folded form on shifted rows, biases and a non-trivial LayerNorm affine included."""
    from audiocraft_amd.modules import cross_fold
    g = torch.Generator().manual_seed(12)
    R, H, Lc, hd = 5, 4, 6, 16
    d = H * hd
    rn = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)  # noqa: E731
    wq, wo, wco = rn(d, d) / d ** 0.5, rn(d, d) / d ** 0.5, rn(d, d) / d ** 0.5
    bq, bo, bco = 0.1 * rn(d), 0.1 * rn(d), 0.1 * rn(d)
    gam, bet = 1 + 0.2 * rn(d), 0.2 * rn(d)
    kc, vc = rn(R, H, Lc, hd), rn(R, H, Lc, hd)
    x0, att = 2.0 * rn(R, d) + 1.5, rn(R, d)
    x1 = x0 + att @ wo.T + bo
    # the block, directly (f64)
    q = torch.nn.functional.layer_norm(x1, (d,), gam, bet, 1e-5) @ wq.T + bq
    s = torch.einsum('bhf,bhjf->bhj', q.view(R, H, hd), kc) * hd ** -0.5
    o = torch.einsum('bhj,bhjf->bhf', torch.softmax(s, dim=-1), vc).reshape(R, d)
    want = x1 + o @ wco.T + bco
    # through the tables
    f = lambda t: t.float()  # noqa: E731
    t = cross_fold.fold_tables(f(kc), f(vc), f(wq * gam[None, :]), f(wq @ bet + bq), f(wo), f(bo), f(wco), wdt)
    assert t['G'].dtype == wdt and t['G'].shape == (R, H * Lc, d) and t['CS'].shape == (R, H * Lc)
    shift = f(x0.mean(dim=1))
    got = cross_fold.folded_cross_block(t, f(x0) - shift[:, None], f(att), f(x1), shift, H, 1e-5, f(bco))
    err = ((got.double() - want).norm() / want.norm()).item()
    assert err < tol, err
