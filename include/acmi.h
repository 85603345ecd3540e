/*
 * acmi.h -- C ABI of libacmi.so: MI355X (gfx950) kernels for the AudioCraft generation hot path
 * (EnCodec SEANet + RVQ, MusicGen LMModel autoregressive decode).
 *
 * The reference (facebookresearch/audiocraft) is 100 % Python and has no FFI of its own
 * (SURVEY.md section 8b); the drop-in boundary is its Python class API.  This header is the
 * C-level boundary the Python host (`audiocraft_amd/`, ctypes) binds, one entry point per ATen op
 * sequence of the reference's path.  Each declaration cites the reference code it replaces.
 *
 * Conventions
 *   - every function returns 0 on success, a negative ACMI_E* code on failure, and never throws;
 *     `acmi_last_error()` returns a thread-local message for the last failure on this thread;
 *   - all data pointers are DEVICE pointers; the caller owns every buffer (weights, KV cache,
 *     activations, workspaces); the library never allocates or frees device memory;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued asynchronously on it, no
 *     implicit synchronisation; calls are hipGraph-capturable (no sync / malloc inside);
 *   - tensors are dense row-major in the stated shape; codes / tokens are int64 at this boundary
 *     exactly as in the reference's Python API.
 */
#ifndef ACMI_H
#define ACMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACMI_VERSION 100 /* 0.1.0 */

#define ACMI_OK 0
#define ACMI_EINVAL (-1)   /* bad argument / unsupported shape */
#define ACMI_ELAUNCH (-2)  /* HIP launch error */

/* weight / cache element types */
#define ACMI_F32 0
#define ACMI_BF16 1

int acmi_version(void);
const char* acmi_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Residual vector quantizer
 * ------------------------------------------------------------------------------------------ */

/* ||e||^2 per codebook row: `embed.pow(2).sum(0)` of EuclideanCodebook.quantize
 * (audiocraft/quantization/core_vq.py:164-170).  codebooks [K, bins, D] f32 -> norms [K, bins]. */
int acmi_rvq_codebook_norms(const float* codebooks, float* norms, int K, int bins, int D, void* stream);

/* ResidualVectorQuantizer.encode -> ResidualVectorQuantization.encode -> EuclideanCodebook.quantize
 * (audiocraft/quantization/vq.py:87-96, core_vq.py:386-396, :164-172): per level
 * argmax_c -(||x||^2 - 2 x.e_c + ||e_c||^2) (first index on ties), residual -= e_c.
 * latents [B, D, T] f32 (conv layout), codebooks [K, bins, D], norms [K, bins] -> codes [B, K, T] int64. */
int acmi_rvq_encode(const float* latents, const float* codebooks, const float* norms, int64_t* codes,
                    int B, int D, int T, int K, int bins, void* stream);

/* ResidualVectorQuantizer.decode (vq.py:98-103, core_vq.py:398-404, :177-179, :295-298):
 * sum over levels of embedding rows, level order, then "b n d -> b d n".
 * codes [B, K, T] int64 -> out [B, D, T] f32.  Returns ACMI_EINVAL if K/bins/D unsupported; code
 * values outside [0, bins) are clamped (the reference would raise an IndexError). */
int acmi_rvq_decode(const int64_t* codes, const float* codebooks, float* out,
                    int B, int D, int T, int K, int bins, void* stream);

/* ------------------------------------------------------------------------------------------
 * SEANet convolutions
 * ------------------------------------------------------------------------------------------ */

#define ACMI_PAD_ZERO 0
#define ACMI_PAD_REFLECT 1

typedef struct {
    int B, Cin, Tin;       /* input  x  [B, Cin, Tin] f32 */
    int Cout, Tout;        /* output y  [B, Cout, Tout] f32 (for shuffle > 1: Cout/shuffle channels) */
    int ksize, stride, dilation;
    int pad_left;          /* virtual padding before x[.., 0]; right padding is implied by Tout */
    int pad_mode;          /* ACMI_PAD_ZERO | ACMI_PAD_REFLECT */
    int reflect_len;       /* length used for reflection (>= Tin; > Tin only for the short-input case
                              of pad1d, audiocraft/modules/conv.py:79-86; samples in [Tin, reflect_len) are 0) */
    int elu_in;            /* apply ELU(alpha) to the input on load (the nn.ELU preceding the conv) */
    float elu_alpha;
    int shuffle;           /* 1: plain conv.  s > 1: transposed-conv mode -- the Cout GEMM rows are
                              (co, r) pairs, r fastest; output sample index = q * s + r - trim_left,
                              kept when inside [0, Tout) */
    int trim_left;
} acmi_conv_desc;

/* StreamableConv1d.forward (audiocraft/modules/conv.py:185-201: pad1d + F.conv1d + bias) and, with
 * shuffle > 1 and pre-arranged polyphase weights, StreamableConvTranspose1d.forward (:221-243:
 * F.conv_transpose1d + unpad1d).  Weights w [Cout_rows, Cin, ksize] f32 with weight-norm already
 * folded (conv.py:21-30), bias [Cout_rows / shuffle] or NULL, residual [B, Cout, Tout] or NULL
 * (the skip of SEANetResnetBlock.forward, audiocraft/modules/seanet.py:59-60). */
int acmi_conv1d(const acmi_conv_desc* d, const float* x, const float* w, const float* bias,
                const float* residual, float* y, void* stream);

/* nn.LSTM recurrence for one layer (audiocraft/modules/lstm.py:19-25): the input projection
 * gates_in [B, 4H, T] f32 (= W_ih x + b_ih + b_hh, computed with acmi_conv1d, ksize 1) is given;
 * runs T sequential steps  gates = gates_in[:, :, t] + W_hh h;  i,f,g,o;  c,h update, zero initial
 * state, and writes y [B, H, T] (+ skip [B, H, T] if not NULL).  w_hh [4H, H] f32.
 * work: 2 * B * H floats (h double buffer) + B * H floats (c). */
int acmi_lstm_layer(const float* gates_in, const float* w_hh, const float* skip, float* y, float* work,
                    int B, int H, int T, void* stream);
size_t acmi_lstm_work_floats(int B, int H);

/* ------------------------------------------------------------------------------------------
 * MusicGen LM decode step
 * ------------------------------------------------------------------------------------------ */

typedef struct {
    /* StreamingTransformerLayer parameters (audiocraft/modules/transformer.py:454-574); weight
     * matrices are [out_features, in_features] row-major in `wdtype`; norms / biases f32. */
    const void* w_qkv;      /* self_attn.in_proj_weight   [3d, d] */
    const void* w_out;      /* self_attn.out_proj.weight  [d, d] */
    const void* w_cq;       /* cross_attention.in_proj_weight[:d]  [d, d]  (NULL if no cross-attn) */
    const void* w_cout;     /* cross_attention.out_proj.weight     [d, d] */
    const void* w_ff1;      /* linear1.weight [ffn, d] */
    const void* w_ff2;      /* linear2.weight [d, ffn] */
    const float* ln1_g; const float* ln1_b;
    const float* lnc_g; const float* lnc_b;
    const float* ln2_g; const float* ln2_b;
    void* k_cache;          /* [Beff, H, Tmax, hd] in `kvdtype` (past_keys,  transformer.py:266-298) */
    void* v_cache;          /* [Beff, H, Tmax, hd] */
    const void* ck_cache;   /* cross-attention keys   [Beff, H, Lc, hd] in `kvdtype`, projected once */
    const void* cv_cache;   /* cross-attention values [Beff, H, Lc, hd] */
} acmi_lm_layer;

typedef struct {
    int dim, num_heads, num_layers, ffn_dim, n_q, card;
    int wdtype;             /* ACMI_F32 | ACMI_BF16: matrices */
    int kvdtype;            /* ACMI_F32 | ACMI_BF16: KV caches */
    int cross_attention;    /* layers have norm_cross + cross_attention (text models) */
    float eps;              /* LayerNorm eps (1e-5, transformer.py:54-67) */
    float positional_scale; /* StreamingTransformer positional_scale */
    const acmi_lm_layer* layers;    /* host array [num_layers] */
    const void* const* emb;         /* host array [n_q] of device ptrs: emb.k.weight [card+1, d] in wdtype */
    const float* pos_freq;          /* [d/2] f32: max_period ** (i / (d/2 - 1)) (transformer.py:83-88) */
    const float* out_norm_g; const float* out_norm_b;
    const void* w_head;             /* linears.{k}.weight stacked [n_q * card, d] in wdtype */
} acmi_lm_model;

typedef struct {
    int Beff;               /* rows run through the transformer: 2B with CFG ([cond; uncond]), else B */
    int B;                  /* samples */
    int use_cfg;
    int Tmax;               /* KV cache capacity (positions) */
    int Lc;                 /* cross-attention source length (0 if none) */
    int n_prepend;          /* P: rows of `prepend` consumed as inputs before the first token step */
    int S;                  /* pattern sequence length (T + max_delay + 1) */
    int64_t* gen_sequence;  /* [B, K, S] int64; -1 = not generated yet (lm.py:523-534) */
    const uint8_t* seq_mask;/* [K, S] pattern validity mask (codebooks_patterns.py:138-151) */
    const float* prepend;   /* [Beff, P, d] f32 prepended condition rows (conditioners.py:1739-1741) or NULL */
    int* pos;               /* device int[4]: pos[0] = current position index g (advanced by the step) */
    /* activations / scratch, all f32 unless noted; sizes given for M = Beff rows */
    float* x;               /* [M, d]   residual stream */
    float* q;               /* [M, d] */
    float* att;             /* [M, d] */
    void* hidden;           /* [M, ffn] in wdtype-compatible activation type (bf16 when wdtype bf16, else f32) */
    float* logits;          /* [M, n_q * card] */
    float* step_logits;     /* optional [B, n_q, card] copy of the CFG-mixed logits of this step, or NULL */
    /* sampling (lm.py:402-418, utils/utils.py:88-122) */
    int use_sampling; float temp; int top_k; float top_p; float cfg_coef;
    uint64_t seed;
} acmi_lm_state;

#define ACMI_STEP_PREFILL 0 /* run the layers at position g, no head / sampling (prompt + prepend rows) */
#define ACMI_STEP_DECODE 1  /* layers + out_norm + heads + CFG + sampling + pattern write-back */

/* One position of LMModel._sample_next_token / LMModel.forward / StreamingTransformer.forward in
 * streaming mode (audiocraft/models/lm.py:323-418, :221-268; transformer.py:693-713, :550-574,
 * :315-451) followed by the write-back of lm.py:553-562.  Position g is read from state->pos[0] on
 * the device, so a captured hipGraph of this call can be replayed for every step; pos[0] is
 * incremented at the end. */
int acmi_lm_step(const acmi_lm_model* m, const acmi_lm_state* s, int mode, void* stream);

/* Individual operators of the step, exposed for parity tests and for the one-off cross-attention
 * K/V projection (the reference re-projects them every step, transformer.py:344-361).
 *   out[M, N] = act(LN(x)[M, K] @ W[N, K]^T + bias[N]) + residual[M, N]
 *   (ln_g == NULL: no LayerNorm; bias / residual may be NULL)
 * act: 0 none, 1 exact GELU.  out_dtype / a_dtype: ACMI_F32 | ACMI_BF16. */
int acmi_linear(const void* a, int a_dtype, const float* ln_g, const float* ln_b, float eps,
                const void* w, int wdtype, const float* bias, const float* residual, void* out, int out_dtype,
                int act, int M, int N, int K, void* stream);

/* Single-query attention over a [Beff, H, Tcap, hd] cache, positions [0, len): the
 * F.scaled_dot_product_attention call of transformer.py:412-414 for one new step.
 * q [Beff, H*hd] f32 -> out [Beff, H*hd] f32.  len_dev (device int*) overrides len when not NULL
 * (length = *len_dev + len_bias). */
int acmi_attn_decode(const float* q, const void* k_cache, const void* v_cache, int kvdtype, float* out,
                     int Beff, int H, int hd, int Tcap, int len, const int* len_dev, int len_bias,
                     void* stream);

/* Scatter rows [Beff, L, H*hd] f32 into a [Beff, H, Tcap, hd] cache at positions [t0, t0+L). */
int acmi_kv_store(const float* src, void* cache, int kvdtype, int Beff, int H, int hd, int Tcap,
                  int t0, int L, void* stream);

/* CFG mix + sampling on logits [Beff, K*card] (lm.py:391-418): tokens_out [B, K] int64. */
int acmi_sample(const float* logits, int64_t* tokens_out, float* mixed_out, int B, int K, int card,
                int use_cfg, float cfg_coef, int use_sampling, float temp, int top_k, float top_p,
                uint64_t seed, uint64_t step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ACMI_H */
