"""ctypes binding of libacmi.so (include/acmi.h).

There is NO fallback: if the HIP library is missing or fails to load, importing this module raises.
Build it with `python -c "import __graft_entry__ as g; g.build()"` or `python -m audiocraft_amd.build`.
"""
import ctypes as C
import os
import threading
import typing as tp

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('ACMI_LIB') or os.path.join(_HERE, 'csrc', 'libacmi.so')   # ACMI_LIB: A/B builds (dev)

F32, BF16 = 0, 1
PAD_ZERO, PAD_REFLECT = 0, 1
STEP_PREFILL, STEP_DECODE = 0, 1
CFG_NONE, CFG_PAIR, CFG_DOUBLE = 0, 1, 2


class AcmiError(RuntimeError):
    pass


# ---- one device-side call sequence at a time per process ------------------------------------------------------------------
# The generation path mutates per-model run state (KV caches, the device-side position counter, sampler state) and captures
# hipGraphs in the runtime's GLOBAL capture mode, in which another host thread's synchronisation or allocation during the
# capture window is an error for BOTH threads.  The reference is "not re-entrant" (lm.py:536-566) and its demo serves one
# request at a time through a queue (demos/musicgen_app.py:384); here the same contract is enforced instead of assumed: every
# public entry that touches the device (LMModel.generate / forward / streaming steps, EncodecModel.encode / decode,
# MultiBandDiffusion) runs under this re-entrant lock, so a second host thread SERIALISES behind the first -- whatever
# model it uses: the capture mode is process-wide -- and results are what each thread computes alone
# (tests/test_gpu_models.py::test_two_host_threads_generate_serialised).  Throughput across requests comes from batching and
# from one process per GPU (bench.py --gpus N), not from threads.
device_lock = threading.RLock()


def exclusive(fn):
    """Decorator: run `fn` holding the process-wide device lock (re-entrant: entries may nest)."""
    import functools

    @functools.wraps(fn)
    def locked(*args, **kwargs):
        with device_lock:
            return fn(*args, **kwargs)
    return locked


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: the MI355X kernel library has not been built. "
        "Run `python -m audiocraft_amd.build` (needs hipcc). There is no CPU fallback.")
lib = C.CDLL(LIB_PATH)

vp, i32, f32, u64, i64p = C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_void_p


class ConvDesc(C.Structure):
    _fields_ = [('B', i32), ('Cin', i32), ('Tin', i32), ('Cout', i32), ('Tout', i32), ('ksize', i32),
                ('stride', i32), ('dilation', i32), ('pad_left', i32), ('pad_mode', i32), ('reflect_len', i32),
                ('elu_in', i32), ('elu_alpha', f32), ('shuffle', i32), ('trim_left', i32)]


class LMLayer(C.Structure):
    _fields_ = [('w_qkv', vp), ('w_out', vp), ('w_cq', vp), ('w_cout', vp), ('w_xcq', vp), ('w_ff1', vp), ('w_ff2', vp),
                ('b_qkv', vp), ('b_cq', vp), ('b_ff1', vp), ('cs_qkv', vp), ('cs_cq', vp), ('cs_ff1', vp),
                ('k_cache', vp), ('v_cache', vp), ('ck_cache', vp), ('cv_cache', vp),
                ('w_qkvx', vp), ('b_qkvx', vp), ('cs_qkvx', vp), ('w_mq', vp), ('w_ff2h', vp),
                ('b_out', vp), ('b_cout', vp), ('b_ff2', vp), ('b_mq', vp), ('cvt_cache', vp),
                ('q_ln_g', vp), ('q_ln_b', vp), ('k_ln_g', vp), ('k_ln_b', vp), ('cq_ln_g', vp), ('cq_ln_b', vp),
                ('n1_g', vp), ('n1_b', vp), ('nc_g', vp), ('nc_b', vp), ('n2_g', vp), ('n2_b', vp),
                ('w_qkvs', vp), ('b_qkvs', vp), ('cs_qkvs', vp), ('w_g2', vp), ('b_gs', vp), ('xs_u', vp), ('xs_cs', vp), ('xs_bs', vp)]


class LMModelDesc(C.Structure):
    _fields_ = [('dim', i32), ('num_heads', i32), ('num_layers', i32), ('ffn_dim', i32), ('n_q', i32),
                ('card', i32), ('wdtype', i32), ('kvdtype', i32), ('cross_attention', i32), ('eps', f32),
                ('positional_scale', f32), ('layers', C.POINTER(LMLayer)), ('emb', C.POINTER(vp)),
                ('pos_table', vp), ('w_head', vp), ('b_head', vp), ('cs_head', vp), ('rope_freq', vp), ('rope_decay', vp),
                ('rope_scale', f32), ('rope_base', f32), ('past_context', i32), ('post_norm', i32)]


class LMState(C.Structure):
    _fields_ = [('Beff', i32), ('B', i32), ('use_cfg', i32), ('Tmax', i32), ('Lc', i32), ('n_prepend', i32),
                ('S', i32), ('n_pos', i32), ('gen_sequence', vp), ('seq_mask', vp), ('prepend', vp), ('pos', vp),
                ('x', vp), ('q', vp), ('stats', vp), ('xn', vp), ('xlo', vp), ('x_rbs', i32), ('xn2', vp), ('xlo2', vp), ('r', vp), ('att', vp), ('hidden', vp), ('logits', vp), ('step_logits', vp),
                ('use_sampling', i32), ('temp', f32), ('top_k', i32), ('top_p', f32), ('cfg_coef', f32),
                ('seed', u64), ('cfg_coef_beta', f32), ('cross_len_rows', vp), ('rope_first', i32), ('rope_shift', i32),
                ('xshift', vp), ('cross_active_rows', i32), ('pf_xn', vp), ('pf_vt', vp), ('pf_tcap', i32), ('cvt_tcap', i32),
                ('row_off', vp), ('input_add', vp), ('n_add', i32), ('xs_rows', i32), ('qkv_hand', vp), ('hand_err', vp)]


def _sig(name, argtypes, restype=i32):
    fn = getattr(lib, name)
    fn.argtypes = argtypes
    fn.restype = restype
    return fn


_version = _sig('acmi_version', [])
_last_error = _sig('acmi_last_error', [], C.c_char_p)
_rvq_norms = _sig('acmi_rvq_codebook_norms', [vp, vp, i32, i32, i32, vp])
_rvq_encode = _sig('acmi_rvq_encode', [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp])
_rvq_decode = _sig('acmi_rvq_decode', [vp, vp, vp, i32, i32, i32, i32, i32, vp])
_conv1d = _sig('acmi_conv1d', [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp])
_conv1d_gn = _sig('acmi_conv1d_gn', [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, f32, i32, vp])
_conv1d_tile = _sig('acmi_conv1d_tile_weights', [C.POINTER(ConvDesc), vp, vp, vp])
_conv1d_wfloats = _sig('acmi_conv1d_weight_floats', [C.POINTER(ConvDesc)], C.c_size_t)
_conv1d_work = _sig('acmi_conv1d_work_floats', [C.POINTER(ConvDesc)], C.c_size_t)
_lstm_layer = _sig('acmi_lstm_layer', [vp, vp, vp, vp, vp, i32, i32, i32, vp])
_lstm_work = _sig('acmi_lstm_work_floats', [i32, i32], C.c_size_t)
_lstm_layer_ex = _sig('acmi_lstm_layer_ex', [vp, vp, vp, vp, vp, C.c_size_t, i32, i32, i32, vp])
_lstm_layer_work = _sig('acmi_lstm_layer_work_floats', [i32, i32, i32], C.c_size_t)
_lm_step = _sig('acmi_lm_step', [C.POINTER(LMModelDesc), C.POINTER(LMState), i32, vp])
_linear = _sig('acmi_linear', [vp, i32, vp, vp, f32, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp])
_attn = _sig('acmi_attn_decode', [vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp])
_pos_table = _sig('acmi_pos_table', [vp, vp, i32, i32, vp])
_ln_tile = _sig('acmi_ln_tile', [vp, vp, i32, i32, i32, f32, vp])
_layer_norm_rows = _sig('acmi_layer_norm_rows', [vp, vp, vp, vp, i32, i32, f32, vp])


class LinearDesc(C.Structure):
    _fields_ = [('a', vp), ('a_mode', i32), ('ln_g', vp), ('ln_b', vp), ('eps', f32), ('a_stats', vp),
                ('a_stats_np', i32), ('a_stats_cnt', i32), ('w', vp), ('wdtype', i32), ('bias', vp), ('residual', vp),
                ('out', vp), ('out_mode', i32), ('act', i32), ('stats_out', vp), ('ksplit', i32), ('M', i32), ('N', i32), ('K', i32),
                ('a_lo', vp), ('colsum', vp), ('xt_hi', vp), ('xt_lo', vp), ('a_rbs', i32), ('a_lo_rbs', i32),
                ('xt_rbs', i32), ('xt_lo_rbs', i32), ('lo_K', i32), ('w_half', i32), ('a_shift', vp), ('xt_shift', vp), ('mean_out', vp)]


class CrossFoldDesc(C.Structure):
    _fields_ = [('s_raw', vp), ('s_ld', i32), ('stats', vp), ('stats_np', i32), ('stats_cnt', i32), ('shift', vp), ('cs', vp),
                ('bs', vp), ('u', vp), ('wdtype', i32), ('x', vp), ('bias', vp), ('xt', vp), ('xt_nkc', i32), ('xt_shift', vp),
                ('stats_out', vp), ('rows', i32), ('R', i32), ('HL', i32), ('Lc', i32), ('d', i32), ('FB', i32), ('eps', f32)]


class AttnDesc(C.Structure):
    _fields_ = [('q', vp), ('k_cache', vp), ('v_cache', vp), ('kvdtype', i32), ('out', vp), ('out_mode', i32),
                ('out_dtype', i32), ('out_rbs', i32), ('out_col0', i32), ('Beff', i32), ('H', i32), ('hd', i32),
                ('Tcap', i32), ('len', i32), ('len_dev', vp), ('len_bias', i32), ('cache_rows', i32), ('q_stats', vp), ('q_stats_np', i32),
                ('q_stats_cnt', i32), ('eps', f32), ('q_colsum', vp), ('q_bias', vp), ('len_rows', vp), ('past_context', i32),
                ('q_shift', vp), ('active_rows', i32), ('pos_minor_rows', i32), ('start_rows', vp)]


class FfnEngineDesc(C.Structure):
    _fields_ = [('w0', vp), ('w1', vp), ('w2', vp), ('b0', vp), ('b1', vp), ('cs1', vp), ('b2', vp), ('a0', vp), ('x', vp),
                ('xt_mid', vp), ('xt_out', vp), ('xt_rbs', i32), ('hidden', vp), ('shift', vp), ('flags', vp), ('flags_next', vp),
                ('err', vp), ('M', i32), ('d', i32), ('ffn', i32), ('eps', f32), ('acq_mode', i32), ('waves', i32), ('dma_chunk', i32), ('dma_epi', i32), ('poll_sleep', i32), ('trace', vp)]


_linear_ex = _sig('acmi_linear_ex', [C.POINTER(LinearDesc), vp])
FFN_ENGINE_FLAG_BYTES = 16384
_ffn_engine = _sig('acmi_ffn_engine', [C.POINTER(FfnEngineDesc), vp])
_cross_fold = _sig('acmi_cross_fold', [C.POINTER(CrossFoldDesc), vp])
qkv_attn_launches = _sig('acmi_qkv_attn_launches', [], C.c_longlong)
_ffn_engine_supported = _sig('acmi_ffn_engine_supported', [i32, i32, i32, i32])
_linear_pair = _sig('acmi_linear_pair', [C.POINTER(LinearDesc), C.POINTER(LinearDesc), vp])
_attn_ex = _sig('acmi_attn_decode_ex', [C.POINTER(AttnDesc), vp])
_ln_tile_reduce = _sig('acmi_ln_tile_reduce', [vp, vp, i32, vp, i32, i32, i32, f32, vp])
_kv_store = _sig('acmi_kv_store', [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp])
_sample = _sig('acmi_sample', [vp, vp, vp, i32, i32, i32, i32, f32, f32, i32, f32, i32, f32, u64, u64, vp])

_chroma = _sig('acmi_chroma', [vp, i32, i32, i32, i32, vp, vp, i32, i32, vp, vp, vp])
_chroma_frames = _sig('acmi_chroma_frames', [i32, i32])

_linear_big = _sig('acmi_linear_big', [vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp])
_attn_prefill = _sig('acmi_attn_prefill', [vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp])

_gn_work = _sig('acmi_group_norm_work_floats', [i32, i32, i32, i32], C.c_size_t)
_group_norm = _sig('acmi_group_norm', [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp])
_channel_add = _sig('acmi_channel_add', [vp, vp, vp, i32, i32, i32, vp])
_add_cropped = _sig('acmi_add_cropped', [vp, i32, vp, vp, i32, i32, vp])
_interp_add = _sig('acmi_interp_add', [vp, vp, i32, i32, i32, vp])
_ddpm_step = _sig('acmi_ddpm_step', [vp, vp, vp, vp, C.c_size_t, f32, f32, f32, f32, f32, f32, vp])
_fir_bank = _sig('acmi_fir_bank', [vp, vp, vp, i32, i32, i32, i32, vp])
_band_stats = _sig('acmi_band_stats', [vp, vp, vp, i32, C.c_size_t, i32, vp])
_band_mix = _sig('acmi_band_mix', [vp, vp, vp, vp, i32, C.c_size_t, f32, vp])

_resample = _sig('acmi_resample_frac', [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp])

EXPORTS = ['acmi_group_norm_work_floats', 'acmi_group_norm', 'acmi_channel_add', 'acmi_add_cropped', 'acmi_interp_add', 'acmi_ddpm_step',
           'acmi_fir_bank', 'acmi_band_stats', 'acmi_band_mix', 'acmi_linear_big', 'acmi_attn_prefill', 'acmi_resample_frac', 'acmi_chroma', 'acmi_chroma_frames', 'acmi_version', 'acmi_last_error', 'acmi_rvq_codebook_norms', 'acmi_rvq_encode', 'acmi_rvq_decode',
           'acmi_conv1d', 'acmi_conv1d_gn', 'acmi_conv1d_tile_weights', 'acmi_conv1d_weight_floats', 'acmi_conv1d_work_floats', 'acmi_lstm_layer', 'acmi_lstm_work_floats', 'acmi_lstm_layer_ex', 'acmi_lstm_layer_work_floats', 'acmi_lstm_stack2', 'acmi_lstm_stack2_work_floats', 'acmi_lstm_stack2_supported', 'acmi_lm_step', 'acmi_linear',
           'acmi_attn_decode', 'acmi_kv_store', 'acmi_sample', 'acmi_pos_table', 'acmi_ln_tile', 'acmi_linear_ex', 'acmi_ln_tile_reduce', 'acmi_linear_pair', 'acmi_attn_decode_ex', 'acmi_layer_norm_rows',
           'acmi_ffn_engine', 'acmi_ffn_engine_supported', 'acmi_cross_fold', 'acmi_qkv_attn_launches']


# ctypes mirror -> C type of include/acmi.h (tests/test_host_cpu.py compiles the header with gcc and compares every
# field's offset: a mirror that drifts from the header would corrupt every call silently)
STRUCT_MIRRORS = {'acmi_conv_desc': ConvDesc, 'acmi_lm_layer': LMLayer, 'acmi_lm_model': LMModelDesc, 'acmi_lm_state': LMState,
                  'acmi_linear_desc': LinearDesc, 'acmi_attn_desc': AttnDesc, 'acmi_ffn_engine_desc': FfnEngineDesc,
                  'acmi_cross_fold_desc': CrossFoldDesc}


def version() -> int:
    return _version()


def check(rc: int, what: str):
    if rc != 0:
        raise AcmiError(f"{what} failed ({rc}): {_last_error().decode()}")


def ptr(t):
    """Device pointer of a contiguous CUDA/HIP tensor (or None)."""
    if t is None:
        return None
    assert t.is_cuda, "libacmi takes device pointers only (no CPU fallback)"
    assert t.is_contiguous(), "libacmi takes dense row-major tensors"
    return t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    raise ValueError(f"unsupported dtype {dt}")


# ---------------------------------------------------------------------------------- thin wrappers

def rvq_codebook_norms(codebooks: torch.Tensor) -> torch.Tensor:
    K, bins, D = codebooks.shape
    norms = torch.empty(K, bins, device=codebooks.device, dtype=torch.float32)
    check(_rvq_norms(ptr(codebooks), ptr(norms), K, bins, D, stream()), 'acmi_rvq_codebook_norms')
    return norms


def rvq_encode(latents: torch.Tensor, codebooks: torch.Tensor, norms: torch.Tensor, n_q: int) -> torch.Tensor:
    B, D, T = latents.shape
    codes = torch.empty(B, n_q, T, device=latents.device, dtype=torch.int64)
    check(_rvq_encode(ptr(latents), ptr(codebooks), ptr(norms), ptr(codes), B, D, T, n_q, codebooks.shape[1],
                      stream()), 'acmi_rvq_encode')
    return codes


def rvq_decode(codes: torch.Tensor, codebooks: torch.Tensor) -> torch.Tensor:
    B, K, T = codes.shape
    D = codebooks.shape[2]
    if K > codebooks.shape[0]:   # the reference indexes self.layers[i] and raises (core_vq.py:398-404)
        raise IndexError(f"rvq_decode: codes carry {K} codebooks, the quantizer holds {codebooks.shape[0]}")
    out = torch.empty(B, D, T, device=codes.device, dtype=torch.float32)
    check(_rvq_decode(ptr(codes), ptr(codebooks), ptr(out), B, D, T, K, codebooks.shape[1], stream()),
          'acmi_rvq_decode')
    return out


def conv1d_tile_weights(desc: ConvDesc, w: torch.Tensor) -> torch.Tensor:
    """w [Cout_rows, Cin, ksize] f32 -> the tiled image acmi_conv1d stages (depends on the descriptor's Cout, Cin, ksize, stride,
    dilation, shuffle only).  Once per model: modules keep the result."""
    n = int(_conv1d_wfloats(C.byref(desc)))
    if n == 0:
        raise AcmiError(f"acmi_conv1d_weight_floats: {_last_error().decode()}")
    wt = torch.empty(n, device=w.device, dtype=torch.float32)
    check(_conv1d_tile(C.byref(desc), ptr(w), ptr(wt), stream()), 'acmi_conv1d_tile_weights')
    return wt


def conv1d_tiled(desc: ConvDesc, x, wt, bias, residual, y):
    n = int(_conv1d_work(C.byref(desc)))
    work = torch.empty(n, device=x.device, dtype=torch.float32) if n else None
    check(_conv1d(C.byref(desc), ptr(x), ptr(wt), ptr(bias), ptr(residual), ptr(y), ptr(work), stream()), 'acmi_conv1d')


def conv1d_tiled_gn(desc: ConvDesc, x, wt, bias, residual, y, gamma, beta, groups: int, eps: float, relu: bool):
    """conv(relu?(GroupNorm(x))) without materialising the normalised tensor (acmi_conv1d_gn)."""
    n = int(_conv1d_work(C.byref(desc)))
    work = torch.empty(max(n, 1), device=x.device, dtype=torch.float32)
    gn_work = torch.empty(max(int(_gn_work(desc.B, desc.Cin, desc.Tin, groups)), 1), device=x.device, dtype=torch.float32)
    check(_conv1d_gn(C.byref(desc), ptr(x), ptr(wt), ptr(bias), ptr(residual), ptr(y), ptr(work), ptr(gn_work), ptr(gamma), ptr(beta),
                     groups, eps, int(relu), stream()), 'acmi_conv1d_gn')


def conv1d(desc: ConvDesc, x, w, bias, residual, y):
    """Convenience for one-off calls with raw weights (tests, the LSTM input projections): tiles `w` every time."""
    conv1d_tiled(desc, x, conv1d_tile_weights(desc, w), bias, residual, y)


def lstm_layer(gates_in, w_hh, skip, y, work, B, H, T):
    """One LSTM layer; `work` of lstm_layer_work_floats(B, H, T) floats lets H = 1024 run one recurrence per XCD."""
    if work.numel() > 5 * B * H + 4:
        check(_lstm_layer_ex(ptr(gates_in), ptr(w_hh), ptr(skip), ptr(y), ptr(work), work.numel(), B, H, T, stream()), 'acmi_lstm_layer_ex')
    else:
        check(_lstm_layer(ptr(gates_in), ptr(w_hh), ptr(skip), ptr(y), ptr(work), B, H, T, stream()), 'acmi_lstm_layer')


_lstm_xcd_enabled = True


def disable_lstm_xcd(why: str):
    """The XCD-local LSTM form (one recurrence per XCD, acmi_lstm_layer_ex) depends on every workgroup being resident and on
    the dispatcher's round-robin workgroup -> XCD placement; on a shared or partitioned device its bounded waits give up and
    set the err word.  From then on this process runs the all-CU form (what ACMI_LSTM_XCD=0 selects): slower, not unavailable."""
    global _lstm_xcd_enabled
    if _lstm_xcd_enabled:
        import warnings
        warnings.warn(f"libacmi: the XCD-local LSTM recurrence is disabled for this process ({why}); using the all-CU form")
    _lstm_xcd_enabled = False


def lstm_layer_work_floats(B, H, T) -> int:
    """Work floats of one LSTM layer; the legacy size (5 B H + 4) selects the all-CU form in lstm_layer."""
    return int(_lstm_layer_work(B, H, T)) if _lstm_xcd_enabled else int(_lstm_work(B, H))


_lstm2 = _sig('acmi_lstm_stack2', [vp] * 8 + [i32] * 3 + [vp])
_lstm2_work = _sig('acmi_lstm_stack2_work_floats', [i32] * 3, C.c_size_t)
_lstm2_ok = _sig('acmi_lstm_stack2_supported', [i32] * 3)


# LSTM give-up words whose host-side check is postponed (a list while a hipGraph capture is running: reading a device word
# synchronises, which a capture forbids; the owner of the capture checks them after each replay), else None
# (per thread: another thread running an LSTM while this one captures must keep its own immediate check)
_lstm_tls = threading.local()


def defer_lstm_checks(sink):
    _lstm_tls.sink = sink


def lstm_failed(err_word: torch.Tensor) -> bool:
    return int(err_word.view(torch.int32)[0]) != 0


def lstm_check(err_word: torch.Tensor, what: str):
    """err_word: the [1+] int32/f32 view whose first word counts the persistent kernels' bounded-spin give-ups."""
    sink = getattr(_lstm_tls, 'sink', None)
    if sink is not None:
        sink.append((err_word, what))
        return
    if lstm_failed(err_word):
        raise AcmiError(f"{what}: the persistent LSTM kernel gave up waiting for a workgroup or found it on another XCD than expected (set "
                        "ACMI_LSTM_XCD=0 for the all-CU form at H = 1024, ACMI_LSTM_WAVE=0 for one launch per layer, ACMI_LSTM_PERSISTENT=0 for one "
                        "per time step)")


def lstm_stack2_supported(B, H, T) -> bool:
    return bool(_lstm2_ok(B, H, T))


def lstm_stack2(gates_in0, w_hh0, w_ih1, w_hh1, bias1, skip, y, B, H, T):
    """Two-layer LSTM stack in one launch (layer 1 one step behind layer 0); raises if a workgroup gave up waiting."""
    work = torch.empty(int(_lstm2_work(B, H, T)), device=y.device, dtype=torch.float32)
    work[-4:].zero_()
    check(_lstm2(ptr(gates_in0), ptr(w_hh0), ptr(w_ih1), ptr(w_hh1), ptr(bias1), ptr(skip), ptr(y), ptr(work), B, H, T, stream()),
          'acmi_lstm_stack2')
    lstm_check(work[-4:], 'acmi_lstm_stack2')


def lstm_work_floats(B, H) -> int:
    return int(_lstm_work(B, H))


A_ROWMAJOR_F32, A_TILED, A_ROWMAJOR_F32_NORM = 0, 1, 2
OUT_F32, OUT_BF16, OUT_TILED = 0, 1, 2


def _tile_params(dtype: torch.dtype):
    epl = 8 if dtype == torch.bfloat16 else 4
    return epl, 4 * epl


def tile_matrix(m: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """[R, K] -> MFMA fragment order of include/acmi.h ("tiled weight" / "tiled activation"):
    T[rt][kc][lane = kg*16 + r][j] = m[rt*16 + r][kc*KT + kg*e + j], zero padded to 16 rows / KT columns."""
    epl, kt = _tile_params(dtype)
    R, K = m.shape
    Rp, Kp = -(-R // 16) * 16, -(-K // kt) * kt
    p = torch.zeros(Rp, Kp, device=m.device, dtype=dtype)
    p[:R, :K] = m.to(dtype)
    return p.view(Rp // 16, 16, Kp // kt, 4, epl).permute(0, 2, 3, 1, 4).contiguous()


def untile_matrix(t: torch.Tensor, R: int, K: int) -> torch.Tensor:
    """Inverse of tile_matrix -> [R, K] (same dtype)."""
    rt, nkc, kg, r16, epl = t.shape
    return t.permute(0, 3, 1, 2, 4).reshape(rt * 16, nkc * 4 * epl)[:R, :K]


def tile_matrix_half(m: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """[N, K] -> half-tile order (include/acmi.h, acmi_linear_desc.w_half): T[j][u][lane = kg*16 + s*8 + f][e] =
    m[j*8 + f][u*2KT + s*KT + kg*epl + e]; N a multiple of 8, K a multiple of 2 KT."""
    epl, kt = _tile_params(dtype)
    N, K = m.shape
    assert N % 8 == 0 and K % (2 * kt) == 0, (N, K)
    return m.to(dtype).view(N // 8, 8, K // (2 * kt), 2, 4, epl).permute(0, 2, 4, 3, 1, 5).contiguous()


class TiledWeight:
    """nn.Linear weight [N, K] in tiled (B-fragment) order + its logical shape.  half=True: half-tile order."""

    def __init__(self, w: torch.Tensor, dtype: torch.dtype, half: bool = False):
        self.N, self.K = w.shape
        self.dtype = dtype
        self.half = half
        self.data = tile_matrix_half(w.detach(), dtype) if half else tile_matrix(w.detach(), dtype)

    def data_ptr(self):
        return self.data.data_ptr()

    def nbytes(self):
        return self.data.numel() * self.data.element_size()


def tiled_activation_buffer(M: int, K: int, dtype: torch.dtype, device) -> torch.Tensor:
    epl, kt = _tile_params(dtype)
    return torch.zeros(-(-M // 16), -(-K // kt), 4, 16, epl, device=device, dtype=dtype)


def linear(a, w: TiledWeight, out, ln_g=None, ln_b=None, eps=1e-5, bias=None, residual=None, act=0,
           a_tiled=False, out_mode=None, M=None, standardize=False):
    """out[M, N] = act(LN?(a)[M, K] @ W[N, K]^T + bias) + residual, see acmi_linear.
    a: row-major f32 [M, K] (a_tiled=False) or a tiled activation buffer (a_tiled=True, pass M)."""
    if a_tiled:
        assert M is not None and a.dtype == w.dtype
    else:
        M = a.shape[0]
        assert a.dtype == torch.float32 and a.shape[1] == w.K
    if out_mode is None:
        out_mode = OUT_BF16 if out.dtype == torch.bfloat16 else OUT_F32
    a_mode = A_TILED if a_tiled else (A_ROWMAJOR_F32_NORM if standardize else A_ROWMAJOR_F32)
    check(_linear(ptr(a), a_mode, ptr(ln_g), ptr(ln_b), eps, ptr(w.data),
                  dtype_code(w.dtype), ptr(bias), ptr(residual), ptr(out), out_mode, act, M, w.N, w.K, stream()),
          'acmi_linear')
    return out


def linear_desc(a, w: TiledWeight, out, M, a_mode, out_mode, a_stats=None, np_=0, cnt=0, stats_out=None, bias=None,
                residual=None, act=0, eps=1e-5, ksplit=1, a_lo=None, colsum=None, xt_hi=None, xt_lo=None, a_rbs=0,
                a_lo_rbs=0, xt_rbs=0, xt_lo_rbs=0, lo_K=0, K=None, a_shift=None, xt_shift=None, mean_out=None) -> LinearDesc:
    """acmi_linear_desc (include/acmi.h).  K overrides w.K when the weight was tiled with a padded K.
    The descriptor holds raw device pointers: the caller keeps the tensors alive until the launch."""
    d = LinearDesc()
    d.a, d.a_mode, d.eps = ptr(a), a_mode, eps
    d.a_stats, d.a_stats_np, d.a_stats_cnt = ptr(a_stats), np_, cnt
    d.w, d.wdtype, d.bias, d.residual = ptr(w.data), dtype_code(w.dtype), ptr(bias), ptr(residual)
    d.out, d.out_mode, d.act, d.stats_out, d.ksplit = ptr(out), out_mode, act, ptr(stats_out), ksplit
    d.M, d.N, d.K = M, w.N, (w.K if K is None else K)
    d.a_lo, d.colsum, d.xt_hi, d.xt_lo = ptr(a_lo), ptr(colsum), ptr(xt_hi), ptr(xt_lo)
    d.a_rbs, d.a_lo_rbs, d.xt_rbs, d.xt_lo_rbs, d.lo_K = a_rbs, a_lo_rbs, xt_rbs, xt_lo_rbs, lo_K
    d.w_half = 1 if getattr(w, 'half', False) else 0
    d.a_shift, d.xt_shift, d.mean_out = ptr(a_shift), ptr(xt_shift), ptr(mean_out)
    return d


def linear_launch(d: LinearDesc):
    check(_linear_ex(C.byref(d), stream()), 'acmi_linear_ex')


def linear_ex(a, w: TiledWeight, out, M, a_mode, out_mode, **kw):
    """Descriptor form (acmi_linear_ex): statistics hand-off between producer and consumer GEMMs; with
    `colsum` the folded LayerNorm on a raw tiled activation (a [, a_lo]); xt_hi / xt_lo: raw tiled copy of the output."""
    linear_launch(linear_desc(a, w, out, M, a_mode, out_mode, **kw))
    return out


def ffn_engine(d: FfnEngineDesc):
    """cross-out -> linear1 (+ norm2, GELU) -> linear2 of a decode layer as one persistent launch (acmi_ffn_engine)."""
    check(_ffn_engine(C.byref(d), stream()), 'acmi_ffn_engine')


def ffn_engine_supported(M: int, d: int, ffn: int, dtype: torch.dtype) -> bool:
    return bool(_ffn_engine_supported(M, d, ffn, dtype_code(dtype)))


def linear_pair(plain: LinearDesc, xcat: LinearDesc):
    """Two independent tiled GEMMs on the same rows in one launch (acmi_linear_pair)."""
    check(_linear_pair(C.byref(plain), C.byref(xcat), stream()), 'acmi_linear_pair')


def attn_decode(q, k_cache, v_cache, out, length, len_dev=None, len_bias=0, out_tiled=False, out_rbs=0, out_col0=0,
                q_stats=None, q_np=0, q_cnt=0, q_colsum=None, q_bias=None, eps=1e-5, len_rows=None, past_context=0,
                q_shift=None, active_rows=0, start_rows=None):
    """q [Beff, H*hd] f32; out: [Beff, H*hd] f32 or a tiled activation buffer (out_tiled=True; out_rbs / out_col0
    place the head outputs inside a wider buffer).  q_colsum: LayerNorm hook on q (acmi_attn_desc)."""
    cache_rows, H, Tcap, hd = k_cache.shape
    Beff = q.shape[0]   # a multiple of cache_rows: n consecutive positions per cache row (prefill)
    d = AttnDesc()
    d.q, d.k_cache, d.v_cache, d.kvdtype = ptr(q), ptr(k_cache), ptr(v_cache), dtype_code(k_cache.dtype)
    d.out, d.out_mode, d.out_dtype = ptr(out), (OUT_TILED if out_tiled else OUT_F32), dtype_code(out.dtype)
    d.out_rbs, d.out_col0 = out_rbs, out_col0
    d.Beff, d.H, d.hd, d.Tcap, d.len, d.len_dev, d.len_bias = Beff, H, hd, Tcap, length, ptr(len_dev), len_bias
    d.cache_rows = cache_rows
    d.q_stats, d.q_stats_np, d.q_stats_cnt, d.eps = ptr(q_stats), q_np, q_cnt, eps
    d.q_colsum, d.q_bias = ptr(q_colsum), ptr(q_bias)
    d.len_rows = ptr(len_rows)
    d.past_context = int(past_context)
    d.q_shift, d.active_rows = ptr(q_shift), int(active_rows)
    d.start_rows = ptr(start_rows)
    check(_attn_ex(C.byref(d), stream()), 'acmi_attn_decode_ex')
    return out


def cross_fold_fb(d: int, dtype: torch.dtype) -> int:
    """Features per workgroup of acmi_cross_fold (its FB = 0 default): 64, or the largest power of two below it that divides d
    and still holds whole 16-byte vectors per thread.  The U table is laid out in blocks of that many features."""
    vec = 8 if dtype == torch.bfloat16 else 4
    fb = 64
    while fb > vec and d % fb != 0:
        fb >>= 1
    return fb


def cross_fold_u_layout(U: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """U [R, HL, d] -> [R][d / FB][HL][FB] in `dtype` (acmi_cross_fold_desc.u)."""
    R, HL, d = U.shape
    fb = cross_fold_fb(d, dtype)
    return U.to(dtype).view(R, HL, d // fb, fb).permute(0, 2, 1, 3).contiguous()


def cross_fold(s_raw, s_ld, stats, stats_np, stats_cnt, cs, bs, u, x, rows, R, HL, Lc, eps=1e-5, shift=None, bias=None,
               xt=None, xt_nkc=0, xt_shift=None, stats_out=None):
    """acmi_cross_fold (include/acmi.h): x[b] += softmax(folded LayerNorm of row b's raw scores) u[b] (+ bias), in place."""
    dsc = CrossFoldDesc()
    dsc.s_raw, dsc.s_ld, dsc.stats, dsc.stats_np, dsc.stats_cnt = ptr(s_raw), s_ld, ptr(stats), stats_np, stats_cnt
    dsc.shift, dsc.cs, dsc.bs, dsc.u, dsc.wdtype = ptr(shift), ptr(cs), ptr(bs), ptr(u), dtype_code(u.dtype)
    dsc.x, dsc.bias, dsc.xt, dsc.xt_nkc, dsc.xt_shift, dsc.stats_out = ptr(x), ptr(bias), ptr(xt), xt_nkc, ptr(xt_shift), ptr(stats_out)
    dsc.rows, dsc.R, dsc.HL, dsc.Lc, dsc.d, dsc.FB, dsc.eps = rows, R, HL, Lc, x.shape[1], 0, eps
    check(_cross_fold(C.byref(dsc), stream()), 'acmi_cross_fold')
    return x


def ln_tile(x: torch.Tensor, out: torch.Tensor, eps: float = 1e-5):
    """x [M, K] f32 -> standardised rows in the tiled activation buffer `out`."""
    M, K = x.shape
    check(_ln_tile(ptr(x), ptr(out), dtype_code(out.dtype), M, K, eps, stream()), 'acmi_ln_tile')
    return out


def ln_tile_reduce(x: torch.Tensor, slabs: torch.Tensor, out: torch.Tensor, eps: float = 1e-5):
    """x [M, K] += sum of slabs [S, M, K] (in place), then standardised rows into the tiled buffer `out`."""
    M, K = x.shape
    check(_ln_tile_reduce(ptr(x), ptr(slabs), slabs.shape[0], ptr(out), dtype_code(out.dtype), M, K, eps, stream()),
          'acmi_ln_tile_reduce')
    return out


def layer_norm_rows(x: torch.Tensor, gamma: tp.Optional[torch.Tensor], beta: tp.Optional[torch.Tensor], eps: float = 1e-5,
                    out: tp.Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.LayerNorm over the last dimension of x [M, d] f32 (acmi_layer_norm_rows); `out` may be x itself."""
    assert x.dim() == 2 and x.dtype == torch.float32 and x.is_contiguous()
    M, d = x.shape
    out = torch.empty_like(x) if out is None else out
    for t in (gamma, beta):
        assert t is None or (t.dtype == torch.float32 and t.numel() == d and t.is_contiguous())
    check(_layer_norm_rows(ptr(x), None if gamma is None else ptr(gamma), None if beta is None else ptr(beta), ptr(out),
                           M, d, eps, stream()), 'acmi_layer_norm_rows')
    return out


def pos_table(freq: torch.Tensor, T: int, d: int) -> torch.Tensor:
    table = torch.empty(T, d, device=freq.device, dtype=torch.float32)
    check(_pos_table(ptr(freq), ptr(table), T, d, stream()), 'acmi_pos_table')
    return table


def kv_store(src, cache, t0):
    """src [Beff, L, H*hd] f32 -> cache[:, :, t0:t0+L, :]."""
    Beff, H, Tcap, hd = cache.shape
    L = src.shape[1]
    check(_kv_store(ptr(src), ptr(cache), dtype_code(cache.dtype), Beff, H, hd, Tcap, t0, L, stream()), 'acmi_kv_store')


def sample(logits, B, K, card, use_cfg, cfg_coef, use_sampling, temp, top_k, top_p, seed, step, want_mixed=False,
           cfg_coef_beta=0.0):
    """use_cfg: False / CFG_NONE, True / CFG_PAIR ([cond; uncond] rows) or CFG_DOUBLE ([cond; wav; uncond] rows)."""
    tokens = torch.empty(B, K, device=logits.device, dtype=torch.int64)
    mixed = torch.empty(B, K, card, device=logits.device, dtype=torch.float32) if want_mixed else None
    check(_sample(ptr(logits), ptr(tokens), ptr(mixed), B, K, card, int(use_cfg), cfg_coef, cfg_coef_beta,
                  int(use_sampling), temp, top_k, top_p, seed, step, stream()), 'acmi_sample')
    return tokens, mixed


def lm_step(model_desc: LMModelDesc, state: LMState, mode: int):
    check(_lm_step(C.byref(model_desc), C.byref(state), mode, stream()), 'acmi_lm_step')


def chroma(wav: torch.Tensor, radix2_exp: int, twiddle: torch.Tensor, fbanks: torch.Tensor, argmax: bool,
           want_raw: bool = False):
    """wav [B, T] f32 -> chroma [B, frames, n_chroma] f32 (acmi_chroma); with want_raw also the un-normalised values."""
    B, T = wav.shape
    n_chroma = fbanks.shape[0]
    assert fbanks.shape[1] == (1 << radix2_exp) // 2 + 1 and twiddle.shape == ((1 << radix2_exp) // 2, 2)
    frames = _chroma_frames(T, radix2_exp)
    if frames < 0:
        raise AcmiError(f"acmi_chroma: unsupported radix2_exp {radix2_exp}")
    out = torch.empty(B, frames, n_chroma, device=wav.device, dtype=torch.float32)
    raw = torch.empty_like(out) if want_raw else None
    check(_chroma(ptr(wav), B, T, wav.stride(0), radix2_exp, ptr(twiddle), ptr(fbanks), n_chroma, int(argmax), ptr(out),
                  ptr(raw), stream()), 'acmi_chroma')
    return (out, raw) if want_raw else out


def resample_frac(x: torch.Tensor, kernel: torch.Tensor, old_sr: int, new_sr: int, width: int, out_len: int):
    """x [rows, T] f32 -> y [rows, out_len] (acmi_resample_frac); kernel [new_sr, 2 * width + old_sr]."""
    rows, T = x.shape
    assert kernel.shape == (new_sr, 2 * width + old_sr)
    y = torch.empty(rows, out_len, device=x.device, dtype=torch.float32)
    check(_resample(ptr(x), ptr(y), ptr(kernel), rows, T, out_len, old_sr, new_sr, width, stream()), 'acmi_resample_frac')
    return y


def linear_big(a, w: TiledWeight, out, M, bias=None, a_rbs=0, act=0, accumulate=False, out_ld=0):
    """MFMA-tiled GEMM of the prefill (acmi_linear_big): a = tiled activation buffer (M rows, a multiple of 16), out = f32
    row-major [M, N] (accumulate: out += result) or a tiled activation buffer in w.dtype (then act 1 = exact GELU)."""
    out_mode = OUT_F32 if out.dtype == torch.float32 and out.dim() == 2 else OUT_TILED
    check(_linear_big(ptr(a), a_rbs, ptr(w.data), dtype_code(w.dtype), ptr(bias), ptr(out), out_mode, out_ld, act,
                      int(accumulate), M, w.N, w.K, stream()), 'acmi_linear_big')
    return out


def attn_prefill(q, k_cache, vt, out, npos, npos_pad, pos, past_context=0, out_rbs=0):
    """Causal prefill attention (acmi_attn_prefill): q [rows * npos_pad, H * hd] f32 position-minor, k_cache [rows, H, Tcap, hd],
    vt [rows, H, hd, vt_tcap] (V time-minor), pos: device int32 tensor (pos[0] = first position) -> out (tiled buffer)."""
    rows, H, Tcap, hd = k_cache.shape
    assert vt.shape[:3] == (rows, H, hd) and vt.dtype == k_cache.dtype
    check(_attn_prefill(ptr(q), ptr(k_cache), ptr(vt), dtype_code(k_cache.dtype), ptr(out), dtype_code(out.dtype), out_rbs,
                        rows, H, hd, Tcap, vt.shape[3], npos, npos_pad, ptr(pos), int(past_context), stream()),
          'acmi_attn_prefill')
    return out


# ---------------------------------------------------------------------------------- MultiBandDiffusion pieces

def group_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float = 1e-5, relu: bool = False,
               out: torch.Tensor = None) -> torch.Tensor:
    """nn.GroupNorm (+ ReLU) on x [B, C, T] f32 (acmi_group_norm)."""
    B, Cc, T = x.shape
    y = torch.empty_like(x) if out is None else out
    work = torch.empty(int(_gn_work(B, Cc, T, groups)), device=x.device, dtype=torch.float32)
    check(_group_norm(ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(work), B, Cc, T, groups, eps, int(relu), stream()), 'acmi_group_norm')
    return y


def channel_add(z: torch.Tensor, table: torch.Tensor, steps: torch.Tensor):
    B, Cc, T = z.shape
    check(_channel_add(ptr(z), ptr(table), ptr(steps), B, Cc, T, stream()), 'acmi_channel_add')
    return z


def add_cropped(a: torch.Tensor, s: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """a[..., :T] + s for a [B, C, Ta >= T], s [B, C, T]."""
    B, Cc, T = s.shape
    out = torch.empty_like(s) if out is None else out
    check(_add_cropped(ptr(a), a.shape[-1], ptr(s), ptr(out), B * Cc, T, stream()), 'acmi_add_cropped')
    return out


def interp_add(z: torch.Tensor, ce: torch.Tensor):
    B, Cc, T = z.shape
    check(_interp_add(ptr(z), ptr(ce), B * Cc, T, ce.shape[-1], stream()), 'acmi_interp_add')
    return z


def ddpm_step(current, estimate, noise, out, c_est, sqrt_alpha, sigma, clip, est_scale, out_scale):
    check(_ddpm_step(ptr(current), ptr(estimate), ptr(noise), ptr(out), current.numel(), c_est, sqrt_alpha, sigma, clip, est_scale,
                     out_scale, stream()), 'acmi_ddpm_step')
    return out


def fir_bank(x: torch.Tensor, filters: torch.Tensor) -> torch.Tensor:
    """x [rows, T] f32, filters [n, 2 half + 1] -> [n, rows, T] (replicate padding)."""
    rows, T = x.shape
    n, K = filters.shape
    y = torch.empty(n, rows, T, device=x.device, dtype=torch.float32)
    check(_fir_bank(ptr(x), ptr(filters), ptr(y), rows, T, n, (K - 1) // 2, stream()), 'acmi_fir_bank')
    return y


def band_stats(x: torch.Tensor, lows: torch.Tensor, chunks: int = 64) -> torch.Tensor:
    """-> HOST [n_bands, 2] f64 (sum, sum of squares) of the SplitBands bands of x (flattened) given its low-passes
    [n_bands - 1, n]; the kernel leaves `chunks` f64 partials per band, summed here in a fixed order."""
    n_bands, n = lows.shape[0] + 1, x.numel()
    part = torch.empty(n_bands, chunks, 2, device=x.device, dtype=torch.float64)
    check(_band_stats(ptr(x), ptr(lows), ptr(part), n_bands, n, chunks, stream()), 'acmi_band_stats')
    return part.cpu().sum(dim=1)


def band_mix(x: torch.Tensor, lows: torch.Tensor, gains: torch.Tensor, offset: float = 0.0) -> torch.Tensor:
    out = torch.empty_like(x)
    check(_band_mix(ptr(x), ptr(lows) if lows is not None else None, ptr(gains), ptr(out), gains.numel(), x.numel(), float(offset),
                    stream()), 'acmi_band_mix')
    return out
