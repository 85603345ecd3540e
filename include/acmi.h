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

#define ACMI_VERSION 200 /* 0.2.0: acmi_lm_state.qkv_hand / hand_err (the decode step's QKV GEMM and self-attention as one launch with a per-(row, head) hand-off; opt-in per state).  0.1.9: post-norm layers (acmi_lm_model.post_norm, acmi_lm_layer.n1_g .. n2_b; acmi_ln_tile with eps < 0 = raw rows); score-folded cross-attention (acmi_cross_fold, acmi_lm_layer.w_qkvs .., acmi_lm_state.xs_rows).  0.1.8: acmi_ffn_engine (the tail of a decode layer as one persistent launch; measured slower than the launches, not used by
                            acmi_lm_step), the decode step's cross-attention as a kernel of its own (cross_q_kernel, inside acmi_attn_decode_ex).  0.1.7: qk_layer_norm (acmi_lm_layer.q_ln_g .. cq_ln_b, acmi_layer_norm_rows), fuser 'sum' / 'input_interpolate'
                            (acmi_lm_state.input_add).  0.1.6: acmi_lstm_layer_ex / acmi_lstm_layer_work_floats (one recurrence per XCD at H = 1024).  0.1.5: folded LayerNorm with the row statistics taken from the activation fragments (acmi_linear_desc:
                            colsum without a_stats), left-padded streams (acmi_lm_state.row_off, acmi_attn_desc.start_rows: two_step_cfg
                            with prepended conditions of different lengths).  0.1.4: acmi_conv1d takes pre-tiled weights + a work buffer (acmi_conv1d_tile_weights /
                            _weight_floats / _work_floats); MultiBandDiffusion entry points.  0.1.3: single-term raw activations with a per-row shift (acmi_linear_desc.a_shift / xt_shift /
                            mean_out, acmi_lm_state.xshift), cross-attention restricted to the rows with a non-null
                            condition (active_rows), prefill as MFMA-tiled GEMMs + causal prefill attention */

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
 * F.conv_transpose1d + unpad1d); also every nn.Conv1d / nn.ConvTranspose1d of the diffusion U-Net
 * (audiocraft/models/unet.py:32-104).
 *
 * Weights are given PRE-TILED: acmi_conv1d_tile_weights turns w [Cout_rows, Cin, ksize] f32 (weight-norm already folded,
 * conv.py:21-30) into the image the kernel stages (acmi_conv1d_weight_floats(d) floats; depends on Cout, Cin, ksize, stride,
 * dilation, shuffle of the descriptor only -- not on B / Tin / Tout): once per model, not per call.
 * work: acmi_conv1d_work_floats(d) floats of scratch (the padded, ELU'd, stride-phase-split copy of the input; 0 -- and work
 * may be NULL -- for the convolutions with one or two output channels).  bias [Cout_rows / shuffle] or NULL, residual
 * [B, Cout, Tout] or NULL (the skip of SEANetResnetBlock.forward, audiocraft/modules/seanet.py:59-60). */
size_t acmi_conv1d_weight_floats(const acmi_conv_desc* d);
size_t acmi_conv1d_work_floats(const acmi_conv_desc* d);
int acmi_conv1d_tile_weights(const acmi_conv_desc* d, const float* w, float* wt, void* stream);
/* acmi_conv1d_gn: the same convolution on relu?(GroupNorm(x)) -- `norm -> ReLU -> conv` of the diffusion U-Net's ResBlock
 * (audiocraft/models/unet.py:32-53) without materialising the normalised tensor: a statistics pass over x, then the
 * convolution, whose input pack applies (x - mean) rstd gamma + beta (+ ReLU).  gn_work: acmi_group_norm_work_floats(B, Cin,
 * Tin, groups) floats; gamma / beta [Cin]; not for descriptors with elu_in or the one / two output channel kernel. */
int acmi_conv1d_gn(const acmi_conv_desc* d, const float* x, const float* wt, const float* bias, const float* residual, float* y,
                   float* work, float* gn_work, const float* gamma, const float* beta, int groups, float eps, int relu, void* stream);
int acmi_conv1d(const acmi_conv_desc* d, const float* x, const float* wt, const float* bias,
                const float* residual, float* y, float* work, void* stream);

/* nn.LSTM recurrence for one layer (audiocraft/modules/lstm.py:19-25): the input projection
 * gates_in [B, 4H, T] f32 (= W_ih x + b_ih + b_hh, computed with acmi_conv1d, ksize 1) is given;
 * runs T sequential steps  gates = gates_in[:, :, t] + W_hh h;  i,f,g,o;  c,h update, zero initial
 * state, and writes y [B, H, T] (+ skip [B, H, T] if not NULL).  w_hh [4H, H] f32.
 * One persistent launch for all T steps when H % 4 == 0, H <= 1024 and the (H + 3) / 4 workgroups of that launch are
 * all RESIDENT on the current device -- checked per call: multiProcessorCount x the runtime's occupancy answer for the
 * kernel (minus one workgroup per CU of margin) must cover the grid; a partitioned or smaller device takes the fallback --
 * (W_hh slice register resident, hidden state all-gathered between workgroups at every step), else one launch per step.
 * Residency can still be lost to other work sharing the device: every spin is bounded, the first give-up sets the error
 * word, every workgroup polls it and leaves, so the cost is one bounded spin (not one per step) and y is then garbage.
 * work: acmi_lstm_work_floats(B, H) floats = 5 * B * H + 4 (cell state, three hidden-state buffers, and -- word
 * 5 * B * H -- an unsigned count of give-ups).  The CALLER zeroes that word before the first layer of a stack and reads
 * it once after the last: this function never clears it (a give-up in layer 0 must survive layer 1's call). */
int acmi_lstm_layer(const float* gates_in, const float* w_hh, const float* skip, float* y, float* work,
                    int B, int H, int T, void* stream);
size_t acmi_lstm_work_floats(int B, int H);
/* The same layer with a work area whose size the caller states (0.1.6).  With acmi_lstm_layer_work_floats(B, H, T) floats --
 * the layout above (err word at 5 * B * H) followed by a [B][T][H] exchange array -- a layer of H = 1024 runs ONE
 * RECURRENCE PER XCD on a whole MI355X (8 XCDs x 32 CUs): workgroup b on XCD b % 8 holds the W_hh rows of 32 hidden units
 * (registers + LDS) for the batch rows b % 8, b % 8 + 8, ...; a step's all-gather is the 4 KB of one hidden vector between the
 * 32 CUs of one XCD through that XCD's L2 instead of B x 4 KB between all CUs through the memory side.  The placement
 * (workgroup b on XCD b % 8) is verified by the kernel (XCC_ID): a mismatch raises the err word like a lost residency.
 * With a smaller work area, another H or a device that cannot hold the grid the call is acmi_lstm_layer.
 * ACMI_LSTM_XCD=0 switches the form off (2: the same kernel on memory-side stores / loads, placement independent). */
int acmi_lstm_layer_ex(const float* gates_in, const float* w_hh, const float* skip, float* y, float* work,
                       size_t work_floats, int B, int H, int T, void* stream);
size_t acmi_lstm_layer_work_floats(int B, int H, int T);

/* ------------------------------------------------------------------------------------------
 * MusicGen LM decode step
 * ------------------------------------------------------------------------------------------
 *
 * Operand layouts of the skinny GEMM (M = CFG batch rows, weights streamed once per call):
 *
 *   "tiled weight"  W[N, K] (nn.Linear layout, out x in) is stored as 1 KB MFMA B-fragments:
 *       element type E (bf16: 2 B, f32: 4 B), e = 16 / sizeof(E) elements per lane, KT = 4 * e
 *       columns per tile (bf16: 32, f32: 16); N and K zero-padded to multiples of 16 / KT:
 *           Wt[nt][kc][lane][j],  lane = kg * 16 + n,  holds  W[nt*16 + n][kc*KT + kg*e + j]
 *       so that the fragment a wave needs is ONE fully coalesced 64-lane x 16-byte load.
 *   "tiled activation"  A[M, K] in the same element type, as MFMA A-fragments:
 *           At[mt][kc][lane][j],  lane = kg * 16 + m,  holds  A[mt*16 + m][kc*KT + kg*e + j]
 *       (rows / columns beyond M / K must be zero).  Producers on the path (attention output, FFN
 *       hidden) write this layout directly; row-major f32 activations (the residual stream) are
 *       staged through LDS by the consumer, which is also where the LayerNorm is applied.
 */

typedef struct {
    /* StreamingTransformerLayer parameters (audiocraft/modules/transformer.py:454-574); matrices are
     * "tiled weights" in `wdtype`.  The affine part of each LayerNorm is folded on the host into the
     * matrix that consumes it:  LN(x) W^T = ((x - mean) * rstd) (W diag(gamma))^T + W beta,  so
     * w_qkv / w_cq / w_ff1 hold W diag(gamma) and b_qkv / b_cq / b_ff1 hold W beta (+ the layer's own
     * bias if it has one), f32; the kernels only standardise the rows. */
    const void* w_qkv;      /* self_attn.in_proj_weight   [3d, d] */
    const void* w_out;      /* self_attn.out_proj.weight  [d, d] */
    const void* w_cq;       /* cross_attention.in_proj_weight[:d]  [d, d]  (NULL if no cross-attn) */
    const void* w_cout;     /* cross_attention.out_proj.weight     [d, d] */
    const void* w_xcq;      /* legacy ([w_cq | w_cq W_out], one launch with a concatenated activation); unused by acmi_lm_step
                               since 0.1.2, kept for acmi_linear_pair tests */
    const void* w_ff1;      /* linear1.weight [ffn, d] */
    const void* w_ff2;      /* linear2.weight [d, ffn] */
    const float* b_qkv;     /* [3d]  norm1      folded */
    const float* b_cq;      /* [d]   norm_cross folded */
    const float* b_ff1;     /* [ffn] norm2      folded */
    /* column sums of the (dtype-rounded) folded matrices, f32: cs[n] = sum_k W'[n, k].  With them the
     * LayerNorm needs no kernel of its own: LN(x) W'^T = rstd * (x W'^T - mean * cs), the GEMM running on the
     * RAW row (kept next to x in fragment order as a bf16 hi / lo pair) and the epilogue applying the row
     * statistics its producer emitted.  All NULL = separate standardisation kernel (acmi_ln_tile). */
    const float* cs_qkv;    /* [3d] */
    const float* cs_cq;     /* [d] */
    const float* cs_ff1;    /* [ffn] */
    void* k_cache;          /* [Beff, H, Tmax, hd] in `kvdtype` (past_keys,  transformer.py:266-298) */
    void* v_cache;          /* [Beff, H, Tmax, hd] */
    const void* ck_cache;   /* cross-attention keys   [Beff, H, Lc, hd] in `kvdtype`, projected once */
    const void* cv_cache;   /* cross-attention values [Beff, H, Lc, hd] */
    /* Cross-attention query without a launch of its own, and without a dependency edge: with x1 = x0 + att W_out^T,
     *     x1 W_cq'^T = x0 W_cq'^T + att (W_cq' W_out)^T.
     * The first term needs only x0, the INPUT of the QKV launch, so it rides there as a fourth block of output features
     * (w_qkvx = [w_qkv ; w_cq'], [4d, d]; the block is stored raw, without the LayerNorm epilogue, into state->r); the
     * second term rides in the out-projection launch (acmi_linear_pair: w_out and w_mq = w_cq' W_out on the same
     * activation `att`, accumulating onto r).  norm_cross is applied to r by the attention kernel (acmi_attn_desc).
     * All NULL = separate q projection.  b_qkvx / cs_qkvx: [4d], zeros in the last block. */
    const void* w_qkvx; const float* b_qkvx; const float* cs_qkvx;
    const void* w_mq;       /* [d, d] = w_cq' W_out, computed in f32, then rounded */
    const void* w_ff2h;     /* w_ff2 in half-tile order (acmi_linear_desc.w_half), used for calls of <= 32 rows when d / 8 <= 256
                               and the FFN width is a multiple of 2 KT; NULL = always the 16-feature form */
    /* Biases of the layer's OUTPUT projections (transformer.py:190-209 bias_attn, :497-498 bias_ff: on by default in the
     * reference's LM config, off in every released MusicGen), f32 [d] each or NULL; with LayerScale already multiplied by
     * the branch's scale.  b_mq = W_cq' b_out completes the split cross-attention query (r = x1 W_cq'^T with
     * x1 = x0 + att W_out^T + b_out).  (The in_proj / linear1 biases are part of b_qkv / b_cq / b_ff1 above; the cross
     * attention's k / v biases are applied when its caches are filled.) */
    const float* b_out; const float* b_cout; const float* b_ff2; const float* b_mq;
    const void* cvt_cache;  /* cross-attention values TIME-MINOR [Beff, H, hd, cvt_tcap] in kvdtype (acmi_lm_state.cvt_tcap; zero
                               beyond Lc), for the MFMA-tiled prefill's cross-attention; NULL = it runs the decode kernel per row */
    /* qk_layer_norm (transformer.py:216-222, 388-392; config/model/lm/default.yaml:43, off in every release): LayerNorm over
     * the full model dimension of the projected queries and of the projected keys, f32 [d] weight / bias each, applied by a
     * launch of its own right after the QKV GEMM (before the rotary positions): q in place in state->q, k in place in the
     * cache row the GEMM has just appended (a bf16 cache therefore rounds k twice; exact with an f32 cache).  All four NULL =
     * off.  (0.1.8: also through the one-forward prefill.) */
    const float* q_ln_g; const float* q_ln_b; const float* k_ln_g; const float* k_ln_b;
    /* qk_layer_norm_cross (transformer.py:358-360, 526-529): the same on the cross-attention's queries, f32 [d] each or NULL.
     * The step then runs the cross query as a projection of its own (the split of w_qkvx / w_mq needs a query that is linear
     * in x1).  The cross-attention KEYS are normalised by the caller when it fills ck_cache (acmi_layer_norm_rows). */
    const float* cq_ln_g; const float* cq_ln_b;
    /* Post-norm layers (acmi_lm_model.post_norm; transformer.py:567-573, norm_first=False): norm1 / norm_cross / norm2 with
     * their affine parts, f32 [d] each, applied IN PLACE to x after the residual add of the block (acmi_layer_norm_rows); the
     * matrices above are then the plain ones (nothing folded: cs_* NULL, b_* = the projections' own biases or zeros). */
    const float* n1_g; const float* n1_b; const float* nc_g; const float* nc_b; const float* n2_g; const float* n2_b;
    /* Score-folded cross-attention (0.1.9; acmi_lm_state.xs_rows > 0; algebra and table builder:
     * audiocraft_amd/modules/cross_fold.py).  K / V of the cross-attention are constant over a generate, so the contractions
     * over the head dimension live in PER-GENERATE tables of the R = xs_rows conditioned rows, HL = H * Lc entries each:
     *     G[b, hj, :]  = scale * sum_{f in h} K[b, h, j, f] W_cq'[f, :]      G2 = G W_out       U[b, hj, :] = sum_f V[b, h, j, f] W_cout[:, f]
     * w_qkvs = [w_qkv ; G as R*HL more output features] (tiled, [3d + R HL, d]; b_qkvs / cs_qkvs: [3d + R HL], zeros behind
     * 3d): the QKV launch leaves x0 G^T raw in state->r viewed as [Beff, R HL] (row b's own scores at r[b][b HL ..]); w_g2 =
     * G2 (tiled [R HL, d]) completes them in the out-projection launch (+ b_gs = G b_out, [R HL] or NULL); then ONE launch
     * (acmi_cross_fold) applies the query's folded LayerNorm (xs_cs = row sums of G, xs_bs = scale * b_cq . K; [R, HL] f32), the
     * softmax over each head's Lc positions and x2 = x1 + p U[b] (xs_u: [R][d / 64][HL][64] in wdtype) -- the cross-attention
     * launch and the cross-out GEMM of the layer are gone.  All NULL = off. */
    const void* w_qkvs; const float* b_qkvs; const float* cs_qkvs; const void* w_g2; const float* b_gs;
    const void* xs_u; const float* xs_cs; const float* xs_bs;
} acmi_lm_layer;

typedef struct {
    int dim, num_heads, num_layers, ffn_dim, n_q, card;
    int wdtype;             /* ACMI_F32 | ACMI_BF16: matrices (and tiled activations) */
    int kvdtype;            /* ACMI_F32 | ACMI_BF16: KV caches */
    int cross_attention;    /* layers have norm_cross + cross_attention (text models) */
    float eps;              /* LayerNorm eps (1e-5, transformer.py:54-67) */
    float positional_scale; /* StreamingTransformer positional_scale */
    const acmi_lm_layer* layers;    /* host array [num_layers] */
    const void* const* emb;         /* host array [n_q] of device ptrs: emb.k.weight [card+1, d] row-major, wdtype */
    const float* pos_table;         /* [Tmax, d] f32 sinusoidal table from acmi_pos_table */
    const void* w_head;             /* linears.{k}.weight stacked [n_q * card, d] x diag(out_norm.weight), tiled */
    const float* b_head;            /* [n_q * card] = W_head out_norm.bias (+ head biases) */
    const float* cs_head;           /* [n_q * card] column sums of w_head (see acmi_lm_layer.cs_*), or NULL */
    /* Rotary positions on the self-attention q / k (modules/rope.py:75-114, transformer.py:300-313, 394-395;
     * positional_embedding 'rope' | 'sin_rope' -- for 'rope' alone pass positional_scale = 0 so that no sinusoidal
     * embedding is added): pair i = features (2i, 2i+1) of a head, as a complex number, times
     *     (e^{i pos f_i} * decay_i(pos)) * rope_scale + (1 - rope_scale),   decay_i(pos) = rope_decay[i]^(pos / rope_base)
     * (xPos, inverted for keys), applied to the q rows and the freshly appended k rows by a launch of its own right after
     * the QKV GEMM (with a bf16 cache k is therefore rounded twice; exact with an f32 cache).
     * rope_freq NULL = off; rope_decay NULL = no xPos. */
    const float* rope_freq;         /* [hd / 2] f32: max_period^(-2i / hd) */
    const float* rope_decay;        /* [hd / 2] f32: (i / (hd/2) + 0.4) / 1.4, or NULL */
    float rope_scale, rope_base;    /* RotaryEmbedding.scale (the transformer's positional_scale); XPos.base_scale (512) */
    int past_context;               /* self-attention sees keys p - past_context .. p (transformer.py:249-264, 286-293); <= 0: all */
    /* 0.1.9: post-norm layers -- norm_first=False, the CONSTRUCTOR DEFAULT of the reference's LMModel / StreamingTransformer
     * (lm.py:147, config/model/lm/default.yaml:21), which every release overrides with norm_first: true:
     *     x = norm1(x + sa(x));  x = norm_cross(x + ca(q from the LAYER INPUT, transformer.py:569-572));  x = norm2(x + ff(x))
     * and no out_norm in front of the heads (lm.py:171-173).  Correctness path: every GEMM reads x through acmi_ln_tile's raw
     * form, every LayerNorm is a launch of its own (acmi_lm_layer.n1_g ..); w_head / b_head are the plain stacked heads.
     * dim <= 2048.  0 = pre-norm (everything above). */
    int post_norm;
} acmi_lm_model;

typedef struct {
    int Beff;               /* rows run through the transformer: B x (1 | 2 | 3) by use_cfg */
    int B;                  /* samples */
    int use_cfg;            /* ACMI_CFG_NONE: rows = samples;  ACMI_CFG_PAIR: [cond; uncond] (lm.py:391-399; also the
                               two_step_cfg mode of lm.py:377-387, whose two passes are the two row groups with their
                               own cross-attention lengths, see cross_len_rows);  ACMI_CFG_DOUBLE: [text + wav; wav;
                               null], MusicGen-Style double CFG (lm.py:362-376) */
    int Tmax;               /* KV cache capacity (positions) = rows of pos_table */
    int Lc;                 /* cross-attention source length (0 if none) */
    int n_prepend;          /* P: rows of `prepend` consumed as inputs before the first token step */
    int S;                  /* pattern sequence length (T + max_delay + 1) */
    int n_pos;              /* ACMI_STEP_PREFILL only: consecutive positions run by one call (0 / 1 = one).  Every
                               activation buffer below then holds n_pos * Beff rows (row p * Beff + b = position
                               pos[0] + p of CFG row b) and pos[0] advances by n_pos */
    int64_t* gen_sequence;  /* [B, K, S] int64; -1 = not generated yet (lm.py:523-534) */
    const uint8_t* seq_mask;/* [K, S] pattern validity mask (codebooks_patterns.py:138-151) */
    const float* prepend;   /* [Beff, P, d] f32 prepended condition rows (conditioners.py:1739-1741) or NULL */
    int* pos;               /* device int[4]: pos[0] = current position index g (advanced by the step),
                               pos[1] = scratch ticket counter (must be 0 between steps) */
    float* x;               /* [Beff, d] f32 residual stream */
    float* q;               /* [Beff, d] f32 */
    float* stats;           /* [Beff][max(1, d/8)][2] f32: LayerNorm statistics partials of x (see acmi_linear_desc) */
    void* xn;               /* tiled activation [ceil(Beff/16)*16, d_pad] in wdtype, zero-initialised: standardised x
                               (separate LayerNorm kernel) or the raw x / its bf16 high part (folded LayerNorm) */
    void* xlo;              /* bf16 weights + folded LayerNorm: tiled [., d_pad], low part x - bf16(x); else NULL */
    int x_rbs;              /* K tiles per 16-row block of xn / xn2 (0 = d_pad / KT).  2 d_pad / KT or more makes room
                               for the self-attention output next to x ([x | att]), which w_xcq needs */
    void* xn2; void* xlo2;  /* second (xn, xlo) pair: the paired launch reads x0's fragments while writing x1's; or NULL */
    float* r;               /* [Beff, d] f32: cross-attention query before its LayerNorm statistics are applied; or NULL */
    void* att;              /* tiled activation [ceil(Beff/16)*16, d_pad] in wdtype, zero-initialised */
    void* hidden;           /* tiled activation [ceil(Beff/16)*16, ffn_pad] in wdtype, zero-initialised */
    float* logits;          /* [Beff, n_q * card] f32 */
    float* step_logits;     /* optional [B, n_q, card] copy of the CFG-mixed logits of this step, or NULL */
    /* sampling (lm.py:402-418, utils/utils.py:88-122) */
    int use_sampling; float temp; int top_k; float top_p; float cfg_coef;
    uint64_t seed;
    float cfg_coef_beta;    /* ACMI_CFG_DOUBLE: logits = u + cfg_coef * (w + cfg_coef_beta * (c - w) - u) */
    const int* cross_len_rows; /* device int[Beff] or NULL: cross-attention source length of every row (<= Lc; the
                               caches hold Lc positions per row, the tail of a shorter row is never read).  Lets the
                               conditional and the unconditional pass of two_step_cfg keep their own padded lengths
                               (the reference runs them as two forwards with separate streaming states) */
    /* Rotary position of stream position p: p for p < rope_first, p - rope_shift afterwards.  rope_first = length of
     * the FIRST streaming call of the reference (prepended rows + prompt steps), rope_shift = max(0, rope_first -
     * past_context): the reference's dropped-keys counter starts at 0 after that call (transformer.py:294-297), so
     * its later rotary positions lag by the keys the first call dropped.  0 / 0 = positions as they are. */
    int rope_first, rope_shift;
    float* xshift;          /* device float[2 * rows] (rows = Beff * max n_pos) or NULL.  Not NULL with bf16 weights: the raw
                               fragments of x are SINGLE-TERM, bf16(x - c[row]) (acmi_linear_desc.xt_shift), c = the row mean
                               at the layer's input, kept in the two halves of this buffer alternately (the QKV launch
                               of layer l writes the shift layer l's producers use); xlo / xlo2 are then unused.
                               NULL: hi / lo pairs as in 0.1.2 */
    int cross_active_rows;  /* > 0: the cross-attention source of CFG rows >= cross_active_rows is all zero (null
                               conditions, conditioners.py:492-506): their keys / values are 0 and the block adds
                               exactly 0 to x, so the attention launch skips them (acmi_attn_desc.active_rows); the
                               caller zero-initialises `att`.  0 = run every row */
    /* ACMI_STEP_PREFILL through the MFMA-tiled path: pf_xn != NULL and n_pos > 1 (all NULL / 0 = the decode kernels on
     * n_pos * Beff extra rows).  Every activation buffer of this struct (x, q, att, hidden, stats) then holds
     * Beff * npos_pad rows, npos_pad = n_pos rounded up to 16, POSITION-MINOR (row = cache_row * npos_pad + position); att
     * and hidden with exactly ceil(d / KT) resp. ceil(ffn / KT) K tiles per row block; xn / xlo / r / logits unused. */
    void* pf_xn;            /* tiled activation [Beff * npos_pad, d_pad] in wdtype: standardised rows (acmi_ln_tile) */
    void* pf_vt;            /* [Beff, H, hd, pf_tcap] in kvdtype, zero-initialised: V of this call's positions, time-minor */
    int pf_tcap;            /* positions pf_vt holds per (row, head): >= pos[0] + n_pos, a multiple of 32 */
    int cvt_tcap;           /* time extent of acmi_lm_layer.cvt_cache: a multiple of 32, >= Lc (0 = none) */
    const int* row_off;     /* device int32[Beff] or NULL: LEFT PADDING of each cache row's stream.  Row b's own position is
                               (stream position - row_off[b]): its sinusoidal embedding uses that position and its self-attention
                               sees keys [row_off[b], position] only.  For row groups whose streams begin with different numbers of
                               prepended condition rows (two_step_cfg on a prepend fuser, reference lm.py:378-390: the conditional
                               and the unconditional pass keep separate streaming states): the shorter streams are padded on the
                               left -- `prepend` holds zeros there -- so that every row reaches its first token at the same stream
                               position and one sampler launch serves all of them.  With rotary positions (0.1.8) the rotary
                               position of a row is its OWN position as well; not with a bounded context (past_context) then, and
                               not with the one-forward prefill (pf_xn) */
    const float* input_add; /* device f32 [Beff, n_add, d] or NULL: what the fuser's 'sum' / 'input_interpolate' conditions add to
                               the embedded input (conditioners.py:1733-1737: `input += cond` before the positional embedding).
                               Token step t of cache row b (t = stream position - n_prepend; prepended rows take nothing) adds
                               input_add[b, min(t, n_add - 1)]: the host lays out one entry per step of the reference's first
                               call (an interpolated condition is resampled to that call's length) and one last entry for
                               every later single-step call */
    int n_add;              /* entries per row of input_add (>= 1 when input_add is not NULL) */
    int xs_rows;            /* > 0: score-folded cross-attention (acmi_lm_layer.w_qkvs ..) over the first xs_rows CFG rows (the rows
                               with a non-null condition; the others take b_cout only).  Needs the paired launches (w_qkvx / w_mq
                               present), one position per call, single-term or f32 fragments (no hi / lo pair), one source length
                               for all rows (cross_len_rows NULL), H * Lc <= 1024, r with room for Beff * N floats where N = xs_rows * H * Lc
                               rounded up to 16 (the tables' N: w_qkvs / w_g2 hold zero rows behind xs_rows * H * Lc, b_qkvs /
                               cs_qkvs / b_gs zeros); otherwise the step runs the separate launches */
    /* QKV -> self-attention as ONE launch (0.2.0; reference op sequence transformer.py:362-399, 412-414; DESIGN.md sections
     * 5.10 / 5.11).  Non-NULL qkv_hand + hand_err opt a DECODE step in where its geometry allows (bf16 weights and cache, head
     * size 64, <= 16 rows, LayerNorm statistics from the fragments, no rotary positions / qk_layer_norm / bounded context /
     * left-padded streams); otherwise, and in every other case, the step runs the two launches.  Results are bit-identical. */
    void* qkv_hand;         /* [Beff][3 * dim] 32-bit words, every word 0x7fc0dead (the hand-off sentinel) before the first step:
                               the GEMM workgroups store q | k | v of the new position there, the attention workgroups of the same
                               launch consume and re-arm them */
    int* hand_err;          /* [1]: number of hand-off polls that gave up (bounded spin); non-zero = the step's result is not to
                               be trusted (the caller checks it once per generate), and later launches do not spin again */
} acmi_lm_state;

#define ACMI_CFG_NONE 0
#define ACMI_CFG_PAIR 1
#define ACMI_CFG_DOUBLE 2

#define ACMI_STEP_PREFILL 0 /* run the layers at position g (.. g + n_pos - 1), no head / sampling (prompt + prepend rows) */
#define ACMI_STEP_DECODE 1  /* layers + out_norm + heads + CFG + sampling + pattern write-back */

/* One position of LMModel._sample_next_token / LMModel.forward / StreamingTransformer.forward in
 * streaming mode (audiocraft/models/lm.py:323-418, :221-268; transformer.py:693-713, :550-574,
 * :315-451) followed by the write-back of lm.py:553-562.  Position g is read from state->pos[0] on
 * the device, so a captured hipGraph of this call can be replayed for every step; pos[0] is
 * incremented at the end. */
int acmi_lm_step(const acmi_lm_model* m, const acmi_lm_state* s, int mode, void* stream);

/* nn.LayerNorm over the last dimension of a row-major f32 matrix x [M, d] (two-pass statistics, biased variance,
 * y = (x - mean) / sqrt(var + eps) * gamma + beta; gamma / beta f32 [d] or NULL = 1 / 0); y may alias x.  d <= 2048.
 * The kernel behind acmi_lm_layer.q_ln_g ..; exported for the cross-attention keys of qk_layer_norm_cross models, which
 * the caller normalises once per generate before acmi_kv_store (transformer.py:358-360). */
int acmi_layer_norm_rows(const float* x, const float* gamma, const float* beta, float* y, int M, int d, float eps, void* stream);

/* create_sin_embedding (transformer.py:70-89) for positions 0..T-1: table[t, :d/2] = cos(t / f_i),
 * table[t, d/2:] = sin(t / f_i), f_i = freq[i] = max_period ** (i / (d/2 - 1)) supplied by the host
 * exactly as the reference computes it. */
int acmi_pos_table(const float* freq, float* table, int T, int d, void* stream);

/* The score-folded cross-attention block of one decode position as ONE launch (0.1.9; see acmi_lm_layer.w_qkvs and
 * audiocraft_amd/modules/cross_fold.py): replaces cross-attention (transformer.py:344-361) + its output projection + the
 * residual add (:563-566) once the raw scores of every conditioned row are in s_raw.  For row b < R:
 *     mean, rstd  from the `stats_np` equal-count (mean, M2) partials of x1's row (stats [rows][stats_np][2], stats_np * stats_cnt = d)
 *     s[hj]  = rstd * (s_raw[b * s_ld + b * HL + hj] - (mean - shift[b]) * cs[b][hj]) + bs[b][hj]          (shift NULL = 0)
 *     p      = softmax over each group of Lc consecutive hj
 *     x[b]  += p u[b] (+ bias);   rows R <= b < rows: x[b] += bias (launched only with a bias)
 * and, like every producer of x in the step, the raw fragments of the new rows (xt: tiled activation in wdtype with xt_nkc K
 * tiles per 16-row block, values x - xt_shift[b]; NULL = none) and the (mean, M2) partials of their 16-feature groups
 * (stats_out [rows][d / 16][2]; NULL = none).  u: [R][d / FB][HL][FB] in wdtype, FB = features per workgroup (0 = 64 or the
 * largest power-of-two divisor of d below it).  HL <= 1024, d % 16 == 0 with statistics partials. */
typedef struct {
    const float* s_raw; int s_ld;
    const float* stats; int stats_np, stats_cnt;
    const float* shift;
    const float* cs; const float* bs;
    const void* u; int wdtype;
    float* x; const float* bias;
    void* xt; int xt_nkc; const float* xt_shift;
    float* stats_out;
    int rows, R, HL, Lc, d, FB;
    float eps;
} acmi_cross_fold_desc;
int acmi_cross_fold(const acmi_cross_fold_desc* desc, void* stream);

/* Row standardisation ((x - mean) / sqrt(var + eps), two-pass statistics, no affine part) of a
 * row-major f32 matrix x [M, K] into a zero-initialised tiled activation in `wdtype`: the LayerNorm
 * prologue of the decode step's GEMMs (nn.LayerNorm, transformer.py:54-67), whose affine part is
 * folded into the consuming matrix.  K % 4 == 0, K <= 2048.  eps < 0 (0.1.9): no standardisation, the rows are only
 * converted and laid out in fragment order (the GEMM inputs of post-norm layers). */
int acmi_ln_tile(const float* x, void* out, int wdtype, int M, int K, float eps, void* stream);

/* Same, preceded by the deterministic (fixed-order) reduction of a split-K producer:
 * x[M, K] += slabs[0] + ... + slabs[nslabs-1]   (written back to x), then standardise into `out`. */
int acmi_ln_tile_reduce(float* x, const float* slabs, int nslabs, void* out, int wdtype, int M, int K, float eps,
                        void* stream);

/* Operand descriptors of acmi_linear */
#define ACMI_A_ROWMAJOR_F32 0 /* a [M, K] f32 row-major, staged through LDS (+ optional LayerNorm) */
#define ACMI_A_TILED 1        /* a = tiled activation in the weight's element type */
#define ACMI_A_ROWMAJOR_F32_NORM 2 /* as 0, rows standardised ((x - mean) / sqrt(var + eps)) while staging */
#define ACMI_OUT_F32 0        /* out [M, N] f32 row-major */
#define ACMI_OUT_BF16 1       /* out [M, N] bf16 row-major */
#define ACMI_OUT_TILED 2      /* out = tiled activation (element type of w), pad region untouched */

/* The skinny GEMM of the decode step, exposed for parity tests, for the one-off cross-attention K/V
 * projection (the reference re-projects them every step, transformer.py:344-361) and for the
 * conditioners' output_proj (conditioners.py:355-360):
 *   out[M, N] = act(LN?(a)[M, K] @ W[N, K]^T + bias[N]) + residual[M, N]
 * w: tiled weight in wdtype.  ln_g / ln_b (both or neither, only with ACMI_A_ROWMAJOR_F32, K <= 2048): full
 * LayerNorm with affine parameters applied in the kernel (the decode step uses the folded form instead).
 * bias [N] / residual [M, N] f32 row-major or NULL.  act: 0 none, 1 exact (erf) GELU. */
int acmi_linear(const void* a, int a_mode, const float* ln_g, const float* ln_b, float eps,
                const void* w, int wdtype, const float* bias, const float* residual, void* out, int out_mode,
                int act, int M, int N, int K, void* stream);

/* Descriptor form of acmi_linear with the producer/consumer LayerNorm-statistics hand-off:
 *   stats_out (or NULL): this GEMM writes, for every output row m and every workgroup b (16 output
 *     features each, N % 16 == 0), the pair (mean_b, M2_b = sum (v - mean_b)^2) of its final outputs to
 *     stats_out[(m * (N / 16) + b) * 2 .. +1]  ->  N / 16 contiguous partials of 16 elements per row;
 *   a_stats / a_stats_np / a_stats_cnt (with colsum, see below): the consumer combines `np` equal-count
 *     partials (Chan: mean = avg mean_b, M2 = sum M2_b + cnt (mean_b - mean)^2; np * cnt == K, np <= 128)
 *     into mean / rstd per row and applies the LayerNorm in its epilogue. */
typedef struct {
    const void* a; int a_mode;
    const float* ln_g; const float* ln_b; float eps;
    const float* a_stats; int a_stats_np; int a_stats_cnt;
    const void* w; int wdtype;
    const float* bias; const float* residual;
    void* out; int out_mode; int act;
    float* stats_out;
    int ksplit;             /* tiled activation only: > 1 splits K over `ksplit` workgroups per 16-feature tile
                               (K tiles % ksplit == 0, else ignored); `out` then receives RAW partial sums as
                               f32 slabs out[ks][M][N] -- no bias / act / residual -- to be summed by the consumer
                               (acmi_ln_tile_reduce) */
    int M, N, K;
    /* folded LayerNorm (a_mode ACMI_A_TILED, a_stats and colsum given): `a` holds the RAW rows in fragment
     * order (bf16 weights: a = bf16(x), a_lo = bf16(x - a) or NULL for a single-term activation; f32: a = x),
     * colsum[n] = sum_k W[n, k]; the kernel returns act(rstd * (a W^T - mean * colsum) + bias) (+ residual)
     * with mean / rstd per row from the a_stats partials (K elements per row in total).
     * colsum WITHOUT a_stats (a_stats NULL; single-term activation, M <= 32): the row statistics come FROM THE FRAGMENTS --
     * two more MFMAs per activation fragment accumulate the rows' sums and sums of squares (a x ones, a x a^T), so that
     * mean / variance are those of exactly the values the GEMM multiplies (one-pass variance: meant for fragments stored
     * relative to a shift near the row mean, a_shift below; then mean = fragment mean + a_shift).  The consumer loads no
     * partials and the producers of such an activation need no stats_out. */
    const void* a_lo; const float* colsum;
    /* xt_hi (or NULL): the final outputs are ALSO written as a raw tiled activation [., N] in wdtype
     * (bf16: xt_hi = bf16(v), xt_lo = bf16(v - xt_hi); f32: xt_hi = v, xt_lo unused) for such a consumer. */
    void* xt_hi; void* xt_lo;
    /* Tiled activations wider than this GEMM's K / N (e.g. [x | att] side by side, 2d columns): K tiles
     * between consecutive 16-row blocks of a / a_lo / xt_hi / xt_lo.  0 = exactly ceil(K / KT) (a_lo with lo_K:
     * ceil(lo_K / KT)) resp. ceil(N / KT). */
    int a_rbs, a_lo_rbs, xt_rbs, xt_lo_rbs;
    /* a_lo WITHOUT colsum: hi + lo activation and no LayerNorm; only the first lo_K columns (a multiple of
     * KT) have a lo term -- the [x | att] operand of acmi_linear_pair. */
    int lo_K;
    /* w_half != 0: `w` is in HALF-TILE order and the GEMM runs with 8 output features per workgroup (twice the
     * workgroups of the 16-feature form: for a narrow N with a long K, e.g. FFN2, whose N / 16 workgroups would leave
     * most of the 256 CUs idle).  Unit u of half-tile j (8 features j*8 .. j*8+7, K columns u*2KT .. (u+1)*2KT - 1) is
     * 64 lanes x 16 B at ((j * (K / 2KT) + u) * 64 + lane) * 16 B; lane = kg * 16 + s * 8 + f holds
     * w[j*8 + f][u*2KT + s*KT + kg*e .. + e - 1] (e = elements per 16 B, KT = 4 e).  Requires a plain GEMM (no colsum /
     * a_lo / ksplit), N % 8 == 0, K % (2 KT) == 0 and M <= 32.  With stats_out the partials are of 8 elements:
     * stats_out[(m * (N / 8) + j) * 2 ..], and the consumer passes a_stats_np = N / 8 (<= 256), a_stats_cnt = 8. */
    int w_half;
    /* Per-row SHIFT of a single-term raw activation (bf16 weights, folded LayerNorm without a_lo).  A raw row stored as
     * bf16(x) carries a rounding error of 2^-9 |x|; relative to the row's standard deviation -- what the LayerNorm divides
     * by -- that grows with |mean| / std.  Storing bf16(x - c) with c close to the row mean bounds it by 2^-9 |x - c|
     * whatever the mean, and the LayerNorm algebra absorbs c exactly:
     *     LN(x) W'^T = rstd * ((x - c) W'^T - (mean - c) * colsum) + b.
     *   xt_shift (producer, with xt_hi; device float[M] or NULL): the raw tiled copy is xt_hi = bf16(v - xt_shift[row])
     *     (xt_lo, if also given, the remainder of v - xt_shift[row]);
     *   a_shift  (consumer, with colsum; device float[M] or NULL): the shift the rows of `a` were stored with;
     *   mean_out (consumer, with colsum; device float[M] or NULL): the row means this launch combines from a_stats are
     *     also written here (by the workgroup of the first n-tile) -- the shift for the NEXT producers of x.
     * In acmi_lm_step c = the mean of the row at the layer's input (one sub-layer earlier; exact for the embedding). */
    const float* a_shift; const float* xt_shift; float* mean_out;
} acmi_linear_desc;
int acmi_linear_ex(const acmi_linear_desc* desc, void* stream);

/* Two independent GEMMs on the same rows in ONE launch (one dependency edge of the decode chain less):
 * `plain` is an ordinary tiled-activation GEMM (typically x1 = x0 + att W_out^T with stats_out / xt_hi),
 * `xcat` runs on an activation concatenated along K with a lo term for its first lo_K columns.  In the
 * decode step:  r = [x0 | att] [W_cq' | W_cq' W_out]^T = x1 W_cq'^T, the cross-attention query of
 * transformer.py:344-349 before the LayerNorm statistics of x1 are applied (acmi_attn_decode_ex does that),
 * which removes the cross-attention q projection as a launch of its own.  Neither GEMM may use split-K, a
 * folded LayerNorm or the QKV scatter; both must have the same M and wdtype. */
int acmi_linear_pair(const acmi_linear_desc* plain, const acmi_linear_desc* xcat, void* stream);

/* The tail of a decode layer as ONE persistent launch (0.1.8):
 *     x2 = x1 + att W0^T + b0                     (cross-attention out projection,  transformer.py:344-361, 563-566)
 *     h  = gelu(LN(x2) W1'^T + b1)                (norm2 folded + linear1 + GELU,   transformer.py:567-569)
 *     x3 = x2 + h W2^T + b2                       (linear2 + residual,              transformer.py:570-572)
 * i.e. three dependent calls of acmi_linear_ex (half-tile / folded-LayerNorm-from-the-fragments / half-tile forms) whose
 * two inner dependency edges are crossed INSIDE the launch: one workgroup per CU (d / 8 of them, all co-resident) owns 8
 * features of x2 and x3 and 32 features of h; its compute waves stream their weight slices of ALL THREE matrices into a
 * per-wave LDS ring (LDS-DMA, non-temporal) from the start, so that by the time an edge resolves the next GEMM's weights are
 * already on the CU; a control wave reduces the waves' partial tiles in a fixed order (deterministic), applies the epilogues,
 * publishes x2's / h's fragments with write-through stores + one flag per workgroup, and polls the producers' flags.
 * bf16 weights, M <= 16 rows, d % 256 == 0, ffn == 4 d.
 *   w0, w2: half-tile order (acmi_linear_desc.w_half) [d, d] / [d, ffn];  w1: tiled weight [ffn, d] = linear1 diag(gamma);
 *   b0 / b2: f32 [d] or NULL;  b1, cs1: f32 [ffn] (acmi_lm_layer.b_ff1 / cs_ff1);
 *   a0: tiled activation [16, d] (the cross-attention output);  x: f32 [M, d], x1 in, x3 out;
 *   xt_mid / xt_out: raw fragments of x2 / x3, bf16(v - shift[row]) (acmi_linear_desc.xt_hi / xt_shift), xt_rbs K tiles per
 *   row block (0 = d / 32);  hidden: tiled activation [16, ffn] (written and read inside the launch);
 *   flags: ACMI_FFN_ENGINE_FLAG_BYTES device bytes (byte flags, replicated so that no single line is polled by every CU), ALL
 *   ZERO at launch -- the launch leaves them set and zeroes flags_next (same size, a different array) for the next engine
 *   launch of the stream;  err: device uint32, OR-ed with a code when a bounded wait
 *   gave up (a workgroup was not resident: another process on the device) -- the outputs are then garbage;
 *   trace: NULL, or device uint64[(d / 8) * 2 * 16] in-kernel timeline stamps (s_memrealtime) of the control wave and of the
 *   first compute wave of every workgroup.
 * acq_mode: how the consumers read what other workgroups published -- 0 plain loads (each line is read once per launch, after
 *   its flag), 1 one agent-scope acquire by the control wave + plain loads, 2 agent-scope (sc1) loads. */
#define ACMI_FFN_ENGINE_FLAG_BYTES 16384
typedef struct {
    const void* w0; const void* w1; const void* w2;
    const float* b0; const float* b1; const float* cs1; const float* b2;
    const void* a0; float* x;
    void* xt_mid; void* xt_out; int xt_rbs;
    void* hidden;
    const float* shift;
    void* flags; void* flags_next; uint32_t* err;
    int M, d, ffn; float eps;
    int acq_mode;
    int waves;               /* compute waves per workgroup: 0 = default (4), or 4 / 8 */
    int dma_chunk;           /* weight fragments (1 KB) every compute wave may request per flag poll of the control wave (the
                                stream is metered: a CU's memory path is a FIFO, polls wait behind bursts); 0 = default */
    int dma_epi;             /* ... and when an epilogue starts (ahead of its hand-off stores); 0 = none */
    int poll_sleep;          /* pause between two flag polls, in units of 256 clocks; 0 = none */
    uint64_t* trace;
} acmi_ffn_engine_desc;
int acmi_ffn_engine(const acmi_ffn_engine_desc* desc, void* stream);
/* 1 when acmi_ffn_engine supports the geometry (else acmi_lm_step keeps the three launches) */
int acmi_ffn_engine_supported(int M, int d, int ffn, int wdtype);

/* Diagnostic (0.2.0): how many fused QKV + self-attention launches (acmi_lm_state.qkv_hand) this process has enqueued -- lets a
 * caller (tests, bench) see whether a step took the one-launch form or fell back to the two launches. */
long long acmi_qkv_attn_launches(void);

/* Single-query attention over a [Beff, H, Tcap, hd] cache, positions [0, len): the
 * F.scaled_dot_product_attention call of transformer.py:412-414 for one new step.
 * q [Beff, H*hd] f32 -> out: [Beff, H*hd] f32 row-major (out_mode ACMI_OUT_F32) or a tiled activation
 * in `out_dtype` (out_mode ACMI_OUT_TILED).  len_dev (device int*) overrides len when not NULL
 * (length = *len_dev + len_bias). */
int acmi_attn_decode(const float* q, const void* k_cache, const void* v_cache, int kvdtype, void* out,
                     int out_mode, int out_dtype, int Beff, int H, int hd, int Tcap, int len,
                     const int* len_dev, int len_bias, void* stream);

/* Descriptor form: placement of a tiled output inside a wider tiled activation, and a LayerNorm hook on q. */
typedef struct {
    const float* q; const void* k_cache; const void* v_cache; int kvdtype;
    void* out; int out_mode; int out_dtype;
    int out_rbs;            /* tiled output: K tiles between 16-row blocks of `out` (0 = ceil(H*hd / KT)) */
    int out_col0;           /* tiled output: first column (a multiple of KT), e.g. d for the att half of [x | att] */
    int Beff, H, hd, Tcap, len; const int* len_dev; int len_bias;
    int cache_rows;         /* rows of the K / V cache (0 = Beff).  Beff = n * cache_rows query rows: row b attends in cache
                               row b % cache_rows and, with len_dev, over b / cache_rows more positions (n consecutive
                               positions of a prompt in one call) */
    /* q_colsum (or NULL): q holds x W'^T for the RAW row x (acmi_linear_pair); the kernel first applies
     *   q <- rstd (q - mean * q_colsum) + q_bias      (q_colsum, q_bias: [H*hd] f32, both required)
     * with mean / rstd of row b combined from q_stats[b][np][2] ((mean, M2) partials of np * cnt elements): the
     * LayerNorm (norm_cross, transformer.py:559-565) of the cross-attention query. */
    const float* q_stats; int q_stats_np; int q_stats_cnt; float eps;
    const float* q_colsum; const float* q_bias;
    const int* len_rows;    /* device int[cache_rows] or NULL: per-cache-row length (overrides len / len_dev; `len` must
                               still be given: it sizes the launch and bounds every row's length) */
    int past_context;       /* > 0: only the last past_context + 1 positions of the row's length are attended to */
    const float* q_shift;   /* with q_colsum: per-row shift the raw row behind q was stored with (acmi_linear_desc.a_shift):
                               q <- rstd (q - (mean - q_shift[b]) * q_colsum) + q_bias; NULL = 0 */
    int active_rows;        /* > 0: only query rows whose cache row (b % cache_rows) is < active_rows are computed, the
                               output of the others is left untouched -- cross-attention with null conditions at the tail
                               of the batch ([cond; uncond]: K = V = 0 there, their attention output is exactly 0, so the
                               caller zeroes those rows of `out` once and never runs them); 0 = all rows */
    int pos_minor_rows;     /* > 0: the Beff = cache_rows * pos_minor_rows query rows are position-minor (row = cache row *
                               pos_minor_rows + position: the layout of the MFMA-tiled prefill) instead of position-major */
    const int* start_rows;  /* device int32[cache_rows] or NULL: keys before start_rows[cache row] are not attended
                               (acmi_lm_state.row_off) */
} acmi_attn_desc;
/* Limits (the kernels' leading arguments are packed 16-bit words, preloaded into SGPRs): Tcap, cache_rows, pos_minor_rows,
 * active_rows and H at most 65535 -- larger values are rejected with ACMI_EINVAL.  acmi_linear_ex / acmi_linear_pair with a
 * tiled activation likewise: M, K tiles, a_rbs and ksplit at most 65535. */
int acmi_attn_decode_ex(const acmi_attn_desc* desc, void* stream);

/* ------------------------------------------------------------------------------------------
 * Prefill: the prompt / prepended-condition positions through ONE forward (reference lm.py:540-543,
 * transformer.py:233-264, 362-414; every > 30 s window of genmodel.py:233-262 starts with a 600-token prompt)
 * ------------------------------------------------------------------------------------------
 * Activations of a prefill call are POSITION-MINOR and padded: row = cache_row * npos_pad + position, npos_pad = the
 * call's positions rounded up to a multiple of 16 (pad rows are computed and never stored to the caches). */

/* MFMA-tiled GEMM on the decode step's operand layouts: out[M, N] = a[M, K] W[N, K]^T + bias.  a: tiled activation (a_rbs K
 * tiles per 16-row block, 0 = ceil(K / KT)), w: tiled weight, both in `wdtype`; M a multiple of 16, K an even number of KT
 * tiles.  out_mode ACMI_OUT_F32: row-major f32 with leading dimension out_ld (0 = N), `accumulate` != 0 adds to it (the
 * residual stream); ACMI_OUT_TILED: tiled activation in wdtype with out_ld K tiles per row block (0 = ceil(N / KT)),
 * act 1 = exact GELU.  128 x 128 workgroup tiles, fragments staged through LDS. */
int acmi_linear_big(const void* a, int a_rbs, const void* w, int wdtype, const float* bias, void* out, int out_mode,
                    int out_ld, int act, int accumulate, int M, int N, int K, void* stream);

/* Causal attention of npos consecutive positions pos[0] .. pos[0] + npos - 1 of every (cache row, head) over keys
 * [0, position] (past_context > 0: the last past_context + 1 of them), the F.scaled_dot_product_attention(is_causal) of
 * transformer.py:412-414 for a multi-step first call.  q [Beff * npos_pad, H * hd] f32 (position-minor rows), k_cache
 * [Beff, H, Tcap, hd] and vt [Beff, H, hd, vt_tcap] (V time-minor, vt_tcap a multiple of 32, >= pos[0] + npos; entries past
 * the last position must be finite) in `kvdtype` -> out: tiled activation [Beff * npos_pad, H * hd] in out_dtype. */
int acmi_attn_prefill(const float* q, const void* k_cache, const void* vt, int kvdtype, void* out, int out_dtype,
                      int out_rbs, int Beff, int H, int hd, int Tcap, int vt_tcap, int npos, int npos_pad,
                      const int* pos, int past_context, void* stream);

/* Scatter rows [Beff, L, H*hd] f32 into a [Beff, H, Tcap, hd] cache at positions [t0, t0+L). */
int acmi_kv_store(const float* src, void* cache, int kvdtype, int Beff, int H, int hd, int Tcap,
                  int t0, int L, void* stream);

/* CFG mix + sampling on logits [Beff, K*card] (lm.py:362-418): tokens_out [B, K] int64.
 * cfg_mode ACMI_CFG_NONE (Beff = B) | ACMI_CFG_PAIR (2B rows, u + (c - u) * cfg_coef) | ACMI_CFG_DOUBLE (3B rows,
 * u + cfg_coef * (w + cfg_coef_beta * (c - w) - u)); every operation rounded on its own like the reference's
 * tensor expression (no fma). */
int acmi_sample(const float* logits, int64_t* tokens_out, float* mixed_out, int B, int K, int card,
                int cfg_mode, float cfg_coef, float cfg_coef_beta, int use_sampling, float temp, int top_k, float top_p,
                uint64_t seed, uint64_t step, void* stream);

/* ------------------------------------------------------------------------------------------
 * Melody front-end (MusicGen-melody, BASELINE.json config #5)
 * ------------------------------------------------------------------------------------------ */

/* ChromaExtractor.forward (audiocraft/modules/chroma.py:46-66): wav [B, T] f32 (row stride wav_stride) ->
 * power spectrogram (n_fft = win = 2^radix2_exp, hop n_fft / 4, periodic Hann, centre reflect padding, frames divided by
 * sqrt(sum w^2): torchaudio Spectrogram(power=2, center=True, normalized=True)) -> fbanks [n_chroma, n_fft/2 + 1] f32
 * (librosa.filters.chroma, a constant table built by the host) -> x / max(|x|_inf, 1e-6) over the chroma axis ->
 * with `argmax`, the one-hot of the largest class (first index on ties) -> out [B, frames, n_chroma] f32.
 * A row shorter than n_fft is zero padded to n_fft, (n_fft - T) / 2 samples in front (chroma.py:50-54).
 * twiddle: [n_fft / 2] pairs (cos, -sin)(2 pi k / n_fft) f32, computed by the host in double precision.
 * raw_out (or NULL): the un-normalised chroma, same shape (parity tests: near-tie margins of the argmax).
 * frames = acmi_chroma_frames(T, radix2_exp) = 1 + max(T, n_fft) / (n_fft / 4). */
int acmi_chroma_frames(int T, int radix2_exp);
int acmi_chroma(const float* wav, int B, int T, int wav_stride, int radix2_exp, const float* twiddle,
                const float* fbanks, int n_chroma, int argmax, float* out, float* raw_out, void* stream);

/* julius.resample_frac (called by audiocraft/data/audio_utils.py:54-59 `convert_audio`): polyphase windowed-sinc FIR.
 * The rates are already divided by their gcd; kernel [new_sr, 2 * width + old_sr] f32 holds the filter of every output
 * phase (built by the host: sinc(t) * cos^2(t / zeros / 2), zeros = 24, cut-off 0.945 * min(old, new), each row normalised
 * to sum 1); x [rows, T] -> y [rows, Tout], Tout <= ceil(T * new_sr / old_sr), the input replicate-padded at both ends. */
int acmi_resample_frac(const float* x, float* y, const float* kernel, int rows, int T, int Tout, int old_sr, int new_sr,
                       int width, void* stream);

/* A TWO-layer nn.LSTM stack (EnCodec's `lstm=2`, audiocraft/modules/lstm.py:19-25) as one launch: layer 1 runs one step behind
 * layer 0 (T + 1 dependent steps instead of 2 T), its input projection W_ih1 h1_t + bias1 is computed inside the launch.
 * gates_in0 [B, 4H, T] = W_ih0 x + b_ih0 + b_hh0 (acmi_conv1d, ksize 1); w_hh0, w_ih1, w_hh1 [4H, H] f32; bias1 [4H] =
 * b_ih1 + b_hh1; skip [B, H, T] or NULL is added to the output y [B, H, T] of layer 1.
 * acmi_lstm_stack2_supported: 1 when the launch can run on the current device (H % 4 == 0, H <= 1024, the residency rule of
 * acmi_lstm_layer per layer: if only one layer's workgroups fit at a time the launch degrades to layer 0, then layer 1) AND is
 * the faster form (H <= 512 and a shape the XCD-local form of acmi_lstm_layer_ex does not take, by default; ACMI_LSTM_WAVE=2 lifts
 * that, =0 answers 0); otherwise use two acmi_lstm_layer(_ex) calls.  work: acmi_lstm_stack2_work_floats(B, H, T) floats; its LAST four
 * words hold the give-up count: the caller zeroes them before the call and reads word 0 of them after it. */
size_t acmi_lstm_stack2_work_floats(int B, int H, int T);
int acmi_lstm_stack2_supported(int B, int H, int T);
int acmi_lstm_stack2(const float* gates_in0, const float* w_hh0, const float* w_ih1, const float* w_hh1, const float* bias1,
                     const float* skip, float* y, float* work, int B, int H, int T, void* stream);

/* ------------------------------------------------------------------------------------------
 * MultiBandDiffusion decoder option: what its U-Net and reverse process need besides
 * acmi_conv1d (every Conv1d / ConvTranspose1d of audiocraft/models/unet.py) and acmi_lstm_layer (its BiLSTM bottleneck)
 * ------------------------------------------------------------------------------------------ */

/* nn.GroupNorm(groups, C) on x [B, C, T] f32 (statistics over C / groups channels x T per batch item, eps inside the square
 * root, affine gamma / beta [C]) followed, with relu != 0, by the nn.ReLU every GroupNorm of unet.py:32-104 is followed by.
 * work: acmi_group_norm_work_floats(B, C, T, groups) floats of scratch.  y may alias x. */
size_t acmi_group_norm_work_floats(int B, int C, int T, int groups);
int acmi_group_norm(const float* x, const float* gamma, const float* beta, float* y, float* work, int B, int C, int T,
                    int groups, float eps, int relu, void* stream);

/* z[b, c, t] += table[steps[b], c]: the diffusion-step embedding added after an encoder layer (unet.py:176-181).
 * table [num_steps, C] f32, steps [B] int64. */
int acmi_channel_add(float* z, const float* table, const int64_t* steps, int B, int C, int T, void* stream);

/* out[row, t] = a[row, t] + s[row, t], t < T, a with row pitch Ta >= T (the decoder input `z[:, :, :s.shape[2]] + s`,
 * unet.py:209-212).  out may alias s. */
int acmi_add_cropped(const float* a, int Ta, const float* s, float* out, int rows, int T, void* stream);

/* z[row, t] += ce[row, min(floor(t * (Tc / T)), Tc - 1)]: `z += F.interpolate(condition_emb, T)` (nearest, unet.py:191-193) */
int acmi_interp_add(float* z, const float* ce, int rows, int T, int Tc, void* stream);

/* One step of NoiseSchedule.generate / generate_subsampled (diffusion_schedule.py:205-230, 251-268):
 *   out = clamp((current - c_est * (estimate * est_scale)) / sqrt_alpha + sigma * noise, -clip, clip) * out_scale
 * c_est = (1 - alpha) / sqrt(1 - alpha_bar); noise NULL = no noise term; clip <= 0 = no clamp; n elements; out may alias current. */
int acmi_ddpm_step(const float* current, const float* estimate, const float* noise, float* out, size_t n, float c_est,
                   float sqrt_alpha, float sigma, float clip, float est_scale, float out_scale, void* stream);

/* julius.LowPassFilters as a direct FIR bank (replicate padding): y[f, row, t] = sum_k filters[f, k] x[row, clamp(t + k - half)],
 * filters [n_filters, 2 half + 1] built by the host (windowed sinc at mel-spaced cut-offs); x [rows, T] -> y [n_filters, rows, T].
 * The low-passes behind julius.SplitBands (MultiBandProcessor, MultiBandDiffusion.re_eq). */
int acmi_fir_bank(const float* x, const float* filters, float* y, int rows, int T, int n_filters, int half, void* stream);

/* Per-band (sum, sum of squares) of SplitBands' output from x [n] and its n_bands - 1 low-passes lows [n_bands - 1, n] (band 0 =
 * low 0, band i = low i - low i-1, band n_bands-1 = x - low n_bands-2), as f64 chunk partials [n_bands, chunks, 2] the host sums
 * in order: the `.std()` of every band in re_eq (multibanddiffusion.py:160-163). */
int acmi_band_stats(const float* x, const float* lows, double* partials, int n_bands, size_t n, int chunks, void* stream);

/* out = sum_i gains[i] * band_i + offset with the bands as above, i.e. gains[n-1] x + sum_i (gains[i] - gains[i+1]) low_i + offset:
 * MultiBandProcessor.return_sample / project_sample (diffusion_schedule.py:91-109) and the recombination of re_eq. */
int acmi_band_mix(const float* x, const float* lows, const float* gains, float* out, int n_bands, size_t n, float offset,
                  void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ACMI_H */
