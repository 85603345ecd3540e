// Internal interfaces shared by the translation units of the LM decode path (not part of the C ABI).
//   acmi_gemm.hip  skinny GEMMs (tiled / paired / row-major), LayerNorm-as-a-kernel
//   acmi_attn.hip  single-query attention over the KV cache, KV scatter
//   acmi_lm.hip    embedding, sampler, acmi_lm_step
#pragma once
#include "acmi_common.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct LinArgs {
    const void* a; int a_tiled;
    const float* a_stats; int a_np; int a_cnt;  // per-row (mean, M2) partials of the activation's rows, [M][a_np][2]
    float* stats_out;                           // per-row (mean, M2) partials of this GEMM's output, [M][N / 16][2]
    // "folded LayerNorm": a / a_lo hold the RAW activation as hi / lo fragments (x = hi + lo; f32 weights: hi
    // only), colsum[n] = sum_k W'[n,k]; with the row statistics from a_stats the epilogue applies
    //     LN(x) W'^T = rstd * (x W'^T - mean * colsum)
    const void* a_lo; const float* colsum;
    void* xt_hi; void* xt_lo; int xt_nkc, xt_lo_nkc;  // producer side: also write the output as raw hi / lo fragments
                                                // (K tiles per 16-row block of the two buffers)
    int a_rbs, alo_rbs;                         // fragments (x 64 lanes) between 16-row blocks of a / a_lo (0: NKC)
    int lo_split;                               // LN 3: only K fragments < lo_split have a lo term
    int ln_mode; const float* ln_g; const float* ln_b; float eps;
    const void* w;
    const float* bias;
    const float* residual;
    void* out; int out_mode; int act;
    int M, N, K;
    int NKC;      // K tiles
    int NKC_out;  // K tiles of a tiled output (its K is this GEMM's N)
    int RS;       // LDS row pitch (bytes) of the staged activation
    int ksplit;   // tiled path: workgroups per n-tile; > 1 => raw partial sums go to slabs out[ks][M][N] (f32)
    int kcs, fpw; // tiled path: K tiles per split-K slice, fragments every wave owns (kcs / waves), set by the launcher
    int qkv;      // QKV scatter epilogue
    float* q_out; void* k_cache; void* v_cache; int kv_bf16; int H, hd, Tcap, d; const int* pos;
    int w_half;   // the weight is in half-tile order (8 features x 2 K tiles per 1 KB unit): 8-feature workgroups
    float* r_out; // QKV scatter with N = 4d: the fourth block of features is stored raw (no LayerNorm epilogue) to r_out [M, d]
    int rpp;      // QKV scatter: rows per position (row gm = position gm / rpp of the call, cache row gm % rpp)
    // single-term raw activations with a per-row shift (include/acmi.h, acmi_linear_desc.a_shift / xt_shift / mean_out)
    const float* a_shift; const float* xt_shift; float* mean_out;
    float inv_K;  // tiled path: 1 / K (set by the launcher)
    int hd_shift; // QKV scatter: log2(hd) when the specialised epilogue runs (power-of-two head size)
    int epi;      // tiled path: the specialised epilogue this call's flags allow (acmi_gemm.hip, tl_epilogue), set by the launcher
    int r_ld;     // QKV scatter: row pitch of r_out and width of the raw block behind the 3d q / k / v features (0 = d: the cross
                  // query's x0 part; > d: the score-folded cross-attention's [R H Lc] block, acmi_lm_layer.w_qkvs).  Sits in
                  // what was tail padding: sizeof(LinArgs) is unchanged
#ifdef ACMI_TRACE
    unsigned long long* trace;   // this launch's stamps [workgroup][wave][ACMI_TRACE_NSTAMP] (NULL: not recorded)
#endif
};

// ---- in-kernel timeline of the decode step's GEMM launches (libacmi_trace.so only: -DACMI_TRACE, scripts/lin_timeline.py).
// Every wave stamps s_memrealtime (the chip-wide constant-rate clock) at fixed phases of tl_body and lane 0 stores the
// stamps when the wave ends; the production library carries none of this.
//   0 wave entry                      1 every request issued            2 first weight fragment landed
//   3 all weight fragments landed     4 activation / statistics / epilogue operands landed (vmcnt 0)
//   5 MFMAs done, partial sums and row statistics in LDS (before the barrier)        6 barrier passed
//   7 epilogue stores issued          8 stores retired (vmcnt 0)        9 HW_ID | XCC_ID << 32
#ifdef ACMI_TRACE
#define ACMI_TRACE_NSTAMP 10
#define ACMI_TR_PIN() __builtin_amdgcn_sched_barrier(0)
#define ACMI_TR(T, i) do { ACMI_TR_PIN(); (T)[i] = __builtin_amdgcn_s_memrealtime(); ACMI_TR_PIN(); } while (0)
#define ACMI_TR_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n) : "memory")
unsigned long long* acmi_trace_reserve(int kind, int wgs, int waves, int N, int K, int M);   // host: next launch's region or NULL
#else
#define ACMI_TR(T, i) do { } while (0)
#define ACMI_TR_WAIT_VM(n) do { } while (0)
#endif

template <typename WT> struct WTr {
    static constexpr int EPL = 16 / (int)sizeof(WT);  // elements per lane of a fragment
    static constexpr int KT = 4 * EPL;                // K columns per fragment tile
};

// element index of (row, col) inside a tiled activation with `nkc` K tiles
template <typename WT>
__device__ __forceinline__ size_t tiled_index(int row, int col, int nkc) {
    constexpr int EPL = WTr<WT>::EPL, KT = WTr<WT>::KT;
    const int kc = col / KT, r = col - kc * KT;
    return ((((size_t)(row >> 4) * nkc + kc) * 64 + (r / EPL) * 16 + (row & 15)) * EPL) + (r % EPL);
}

// the score-folded cross-attention block (acmi_crossfold.hip)
struct CrossFoldArgs {
    const float* s_raw; int s_ld;                 // raw folded scores: row b's block at s_raw[b * s_ld + b * HL + hj]
    const float* stats; int np, cnt;              // (mean, M2) partials of x1's rows, [rows][np][2]
    const float* shift;                           // [rows] or NULL: S_raw was accumulated on x - shift
    const float* cs; const float* bs;             // [R][HL] f32
    const void* u;                                // [R][d / FB][HL][FB] in WT
    float* x; const float* bias;                  // residual stream [rows][d] (in place), b_cout [d] or NULL
    void* xt; int xt_nkc; const float* xt_shift;  // raw fragments of the new x (WT) or NULL; K tiles per row block; their shift
    float* stats_out;                             // [rows][d / 16][2] partials of the new x, or NULL
    int R, HL, Lc, d, FB; float eps;
};
int acmi_launch_cross_fold(const CrossFoldArgs& a, int wdtype, int rows, hipStream_t st);

// launchers of acmi_gemm.hip used by acmi_lm_step
int acmi_launch_lin(LinArgs& a, int wdtype, hipStream_t st);                 // tiled or row-major activation
int acmi_launch_pair(LinArgs& p0, LinArgs& p1, int wdtype, hipStream_t st);  // two tiled GEMMs, one launch
struct FusedAttnArgs;
// the decode step's QKV GEMM + the self-attention consuming it as ONE launch (bf16; acmi_attn_fused.h)
int acmi_launch_qkv_attn(LinArgs& a, FusedAttnArgs& f, const void* kc, const void* vc, int H, int Tcap, hipStream_t st);
int acmi_launch_ln_tile(float* x, void* out, int wdtype, int M, int K, float eps, const float* slabs, int nslabs,
                        hipStream_t st);

// ---- prefill (acmi_prefill.hip): MFMA-tiled GEMM on tiled operands + causal prefill attention
enum { ACMI_BIG_F32 = 0, ACMI_BIG_RESID = 1, ACMI_BIG_TILED = 2, ACMI_BIG_QKV = 3 };

struct BigArgs {
    const void* a; int a_rbs;   // tiled activation [M / 16][a_rbs][64 lanes][16 B]
    const void* w;              // tiled weight [ceil(N / 16)][NKC][64][16 B]
    const float* bias;          // [N] f32 or NULL
    int M, N, K, NKC;           // M: rows, a multiple of 16 (row blocks past it are never read)
    int epi, act;               // ACMI_BIG_*; act 1 = exact GELU (tiled epilogue)
    int vec;                    // set by acmi_launch_big: f32 output rows take 16-byte stores
    float* out; int ldo;        // F32 / RESID: row-major f32 [M, ldo] (RESID: out += result)
    void* out_t; int out_rbs;   // TILED: tiled activation in the weight's element type, K tiles per 16-row block
    // QKV (N = 3 d): q -> q_out [M, d] f32; K / V -> caches [rows, H, Tcap, hd] at position pos[0] + p of cache row
    // row / npos_pad (p = row % npos_pad < npos); V also time-minor into vt [rows, H, hd, vt_tcap]
    float* q_out; void* k_cache; void* v_cache; void* vt;
    int kv_bf16, H, hd, Tcap, d, npos, npos_pad, vt_tcap; const int* pos;
};
int acmi_launch_big(BigArgs& a, int wdtype, hipStream_t st);

struct PrefillAttnArgs {
    const float* q;             // [rows * npos_pad, H * hd] f32, position-minor rows
    const void* k_cache;        // [rows, H, Tcap, hd]
    const void* vt;             // [rows, H, hd, vt_tcap]
    void* out; int out_bf16, out_rbs;   // tiled activation [rows * npos_pad, H * hd]
    int H, Tcap, vt_tcap, npos, npos_pad; const int* pos; int past_context; float scale;
    // non-causal form (cross-attention of the prefill): every query sees keys [0, klen) of its cache row
    // (klen_rows[row] when given, else klen); causal == 0 selects it
    int causal; int klen; const int* klen_rows;
};
int acmi_launch_prefill_attn(PrefillAttnArgs& a, int kvdtype, int hd, int Beff, hipStream_t st);
