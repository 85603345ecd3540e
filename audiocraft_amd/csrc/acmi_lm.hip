// MusicGen LM decode-step kernels for gfx950 (CDNA4, wave64).
//
//   lin_kernel        skinny GEMM  out[M,N] = LN?(a)[M,K] @ W[N,K]^T : M = CFG batch rows (<= 16 per
//                     MFMA tile), weights streamed once from HBM, one 16-feature n-tile per workgroup,
//                     K split across the waves of the workgroup, deterministic LDS reduction, fused
//                     LayerNorm prologue and bias/GELU/residual/QKV-scatter epilogues.
//   attn_decode_kernel single-query attention over the KV cache (online softmax, KV streamed once).
//   embed_kernel      sum of codebook embeddings (or prepended condition row) + sinusoidal position.
//   sample_kernel     CFG mix + softmax/top-k/top-p/multinomial (or argmax) + delay-pattern write-back.
//
// Reference semantics: audiocraft/models/lm.py:221-268,323-418,536-565;
// audiocraft/modules/transformer.py:70-89,315-451,550-574,693-713; audiocraft/utils/utils.py:88-141.
#include "acmi_common.h"

#include <math.h>
#include <stdlib.h>

// =====================================================================================================
// skinny GEMM   out[M,N] = act(LN?(a)[M,K] @ W[N,K]^T + bias) + residual
//
// One 16-feature n-tile per workgroup, K split across its (up to 16) waves.  Weights are stored as
// 1 KB MFMA B-fragments (include/acmi.h "tiled weight"), so each fragment is ONE fully coalesced
// non-temporal 64 x 16 B load; all of a wave's fragments are in flight before anything else happens.
// Row-major f32 activations (the residual stream) are staged through LDS once per workgroup -- one
// wave per row, which is also where LayerNorm runs (two-pass statistics with wave shuffles only) --
// and read back as A-fragments with ds_read_b128.  Activations produced on the path (attention
// output, FFN hidden) arrive already in A-fragment order and are loaded like the weights.
// Cross-wave reduction through LDS in a fixed order (deterministic), then the fused epilogue.
// =====================================================================================================

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct LinArgs {
    const void* a; int a_tiled;
    const float* a_stats; int a_np; int a_cnt;  // per-row (mean, M2) partials of the row-major activation (stats mode)
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
    int rpp;      // QKV scatter: rows per position (row gm = position gm / rpp of the call, cache row gm % rpp)
};

template <typename WT> struct WTr {
    static constexpr int EPL = 16 / (int)sizeof(WT);  // elements per lane of a fragment
    static constexpr int KT = 4 * EPL;                // K columns per fragment tile
    static constexpr int LPR = KT / 4;                // statistics mode: lanes per activation row (one float4 each)
    static constexpr int RPI = 64 / LPR;              //                  rows covered by one load instruction
    static constexpr int NJ = 16 / RPI;               //                  load instructions per 16-row tile
};

__device__ __forceinline__ u32x4 ld_frag_nt(const u32x4* p) { return __builtin_nontemporal_load(p); }

__device__ __forceinline__ void mma_frag(const u32x4& a, const u32x4& b, f32x4& acc, bf16_t) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_frag(const u32x4& a, const u32x4& b, f32x4& acc, float) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[e]), __uint_as_float(b[e]), acc, 0, 0, 0);
}

__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// element index of (row, col) inside a tiled activation with `nkc` K tiles
template <typename WT>
__device__ __forceinline__ size_t tiled_index(int row, int col, int nkc) {
    constexpr int EPL = WTr<WT>::EPL, KT = WTr<WT>::KT;
    const int kc = col / KT, r = col - kc * KT;
    return ((((size_t)(row >> 4) * nkc + kc) * 64 + (r / EPL) * 16 + (row & 15)) * EPL) + (r % EPL);
}

// Staging of one activation row by one wave: row load (issued by the caller BEFORE the weight fragments:
// vmcnt retires in order, so the row must not queue behind HBM-latency weight loads), then LayerNorm
// (ln_mode 1: standardise only -- the affine part is folded into the weights on the host; 2: affine here)
// and the store to LDS in the weight's element type.
#define ACMI_STAGE_JMAX 8  // Kpad <= 2048

__device__ __forceinline__ void load_row(const float* __restrict__ xrow, int K, int lane, float4 (&v)[ACMI_STAGE_JMAX]) {
#pragma unroll
    for (int j = 0; j < ACMI_STAGE_JMAX; ++j) {
        const int k = (lane + 64 * j) * 4;
        v[j] = (xrow != nullptr && k < K) ? *reinterpret_cast<const float4*>(xrow + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

template <typename WT>
__device__ __forceinline__ void norm_store_row(float4 (&v)[ACMI_STAGE_JMAX], int K, int Kpad, int ln_mode,
                                               const float* __restrict__ g, const float* __restrict__ b, float eps,
                                               unsigned char* dst, int lane) {
    constexpr int JMAX = ACMI_STAGE_JMAX;
    if (ln_mode != 0) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < JMAX; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        const float mean = wave_sum(s) / (float)K;
        float s2 = 0.f;
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            if ((lane + 64 * j) * 4 < K) {
                const float a0 = v[j].x - mean, a1 = v[j].y - mean, a2 = v[j].z - mean, a3 = v[j].w - mean;
                s2 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(s2) / (float)K + eps);
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            const int k = (lane + 64 * j) * 4;
            if (k < K) {
                v[j].x = (v[j].x - mean) * rstd; v[j].y = (v[j].y - mean) * rstd;
                v[j].z = (v[j].z - mean) * rstd; v[j].w = (v[j].w - mean) * rstd;
                if (ln_mode == 2) {
                    const float4 gg = *reinterpret_cast<const float4*>(g + k);
                    const float4 bb = *reinterpret_cast<const float4*>(b + k);
                    v[j].x = v[j].x * gg.x + bb.x; v[j].y = v[j].y * gg.y + bb.y;
                    v[j].z = v[j].z * gg.z + bb.z; v[j].w = v[j].w * gg.w + bb.w;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        const int k = (lane + 64 * j) * 4;
        if (k < Kpad) {
            if (sizeof(WT) == 2)
                *reinterpret_cast<uint2*>(dst + (size_t)k * 2) = make_uint2(pack_bf16x2(v[j].x, v[j].y), pack_bf16x2(v[j].z, v[j].w));
            else
                *reinterpret_cast<float4*>(dst + (size_t)k * 4) = v[j];
        }
    }
}

// Row standardisation as its own tiny kernel (one wave per row): x [M, K] f32 row-major ->
// ((x - mean) * rstd) in A-fragment order, element type WT.  The affine part of the LayerNorm lives in
// the consuming matrix (see acmi_lm_layer).  Doing this once per LayerNorm instead of once per GEMM
// workgroup takes ~5 us of redundant VALU + LDS staging off the critical path of every GEMM workgroup.
template <typename WT>
__global__ __launch_bounds__(64) void ln_tile_kernel(float* __restrict__ x, WT* __restrict__ out, int M, int K, int nkc,
                                                     float eps, const float* __restrict__ slabs, int nslabs) {
    const int m = blockIdx.x, lane = threadIdx.x;
    if (m >= M) return;
    float4 v[ACMI_STAGE_JMAX];
    load_row(x + (size_t)m * K, K, lane, v);
    if (nslabs > 0) {
        // the producer GEMM was split over K: finish it here (fixed order => deterministic) and write the row back
        for (int sidx = 0; sidx < nslabs; ++sidx) {
            float4 t[ACMI_STAGE_JMAX];
            load_row(slabs + ((size_t)sidx * M + m) * K, K, lane, t);
#pragma unroll
            for (int j = 0; j < ACMI_STAGE_JMAX; ++j) { v[j].x += t[j].x; v[j].y += t[j].y; v[j].z += t[j].z; v[j].w += t[j].w; }
        }
#pragma unroll
        for (int j = 0; j < ACMI_STAGE_JMAX; ++j) {
            const int k = (lane + 64 * j) * 4;
            if (k < K) *reinterpret_cast<float4*>(x + (size_t)m * K + k) = v[j];
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < ACMI_STAGE_JMAX; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    const float mean = wave_sum(s) / (float)K;
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < ACMI_STAGE_JMAX; ++j) {
        if ((lane + 64 * j) * 4 < K) {
            const float a0 = v[j].x - mean, a1 = v[j].y - mean, a2 = v[j].z - mean, a3 = v[j].w - mean;
            s2 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(s2) / (float)K + eps);
#pragma unroll
    for (int j = 0; j < ACMI_STAGE_JMAX; ++j) {
        const int k = (lane + 64 * j) * 4;
        if (k < K) {
            const float y0 = (v[j].x - mean) * rstd, y1 = (v[j].y - mean) * rstd;
            const float y2 = (v[j].z - mean) * rstd, y3 = (v[j].w - mean) * rstd;
            WT* dst = out + tiled_index<WT>(m, k, nkc);  // 4 consecutive k stay inside one lane fragment
            if (sizeof(WT) == 2) *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(y0, y1), pack_bf16x2(y2, y3));
            else *reinterpret_cast<float4*>(dst) = make_float4(y0, y1, y2, y3);
        }
    }
}

static int launch_ln_tile(float* x, void* out, int wdtype, int M, int K, float eps, const float* slabs, int nslabs,
                          hipStream_t st) {
    ACMI_REQUIRE(M > 0 && K > 0 && K % 4 == 0 && K <= 2048, "acmi_ln_tile: needs K %% 4 == 0 and K <= 2048 (K=%d)", K);
    if (wdtype == ACMI_BF16)
        hipLaunchKernelGGL(ln_tile_kernel<bf16_t>, dim3(M), dim3(64), 0, st, x, reinterpret_cast<bf16_t*>(out), M, K,
                           (K + 31) / 32, eps, slabs, nslabs);
    else
        hipLaunchKernelGGL(ln_tile_kernel<float>, dim3(M), dim3(64), 0, st, x, reinterpret_cast<float*>(out), M, K,
                           (K + 15) / 16, eps, slabs, nslabs);
    return acmi_check_launch("ln_tile_kernel");
}

extern "C" int acmi_ln_tile_reduce(float* x, const float* slabs, int nslabs, void* out, int wdtype, int M, int K, float eps,
                                   void* stream) {
    ACMI_REQUIRE(nslabs >= 0 && (nslabs == 0 || slabs != nullptr), "acmi_ln_tile_reduce: bad slabs");
    return launch_ln_tile(x, out, wdtype, M, K, eps, slabs, nslabs, (hipStream_t)stream);
}

extern "C" int acmi_ln_tile(const float* x, void* out, int wdtype, int M, int K, float eps, void* stream) {
    return launch_ln_tile(const_cast<float*>(x), out, wdtype, M, K, eps, nullptr, 0, (hipStream_t)stream);
}

// Folded LayerNorm, row statistics: a "group" is 4 rows (16 lanes each); every lane fetches up to 8 of the
// producer's equal-count (mean, M2) partials of its row (np <= 128), combined later with Chan's formula.
// Layout stats[row][np][2]: the partials of a row are contiguous, so a wave's load touches 4 lines, not 64
// (with [np][row][2] the gather cost ~2 us per consuming launch).
__device__ __forceinline__ void rowstat_load(const float* __restrict__ stats, int np, int M, int row0, int lane,
                                             float (&pm)[8], float (&pq)[8]) {
    const int row = min(row0 + (lane >> 4), M - 1), jj = lane & 15;
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // 16 consecutive partials of one row per 16 lanes: one 128-B line
        const float2 t = *reinterpret_cast<const float2*>(stats + ((size_t)row * np + min(jj + 16 * i, np - 1)) * 2);
        pm[i] = t.x; pq[i] = t.y;
    }
}
__device__ __forceinline__ void rowstat_finish(const float (&pm)[8], const float (&pq)[8], int np, int cnt, int K, float eps,
                                               int lane, float* __restrict__ dst /* [4][2] */) {
    const int jj = lane & 15;
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) sm += (jj + 16 * i < np) ? pm[i] : 0.f;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) sm += __shfl_xor(sm, off, 64);
    const float mean = sm / (float)np;
    float q2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float dlt = pm[i] - mean;
        q2 += (jj + 16 * i < np) ? pq[i] + (float)cnt * dlt * dlt : 0.f;
    }
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) q2 += __shfl_xor(q2, off, 64);
    if (jj == 0) {
        dst[(lane >> 4) * 2] = mean;
        dst[(lane >> 4) * 2 + 1] = 1.0f / sqrtf(q2 / (float)K + eps);
    }
}

// -----------------------------------------------------------------------------------------------------
// lin_rowmajor_kernel: row-major f32 activation, staged (and optionally LayerNorm-ed) through LDS.
// The general-purpose form: conditioner projections, the one-off cross-attention K / V projection, tests.
// One workgroup = 16 output features; wave w stages rows w, w + nw, ... of each 16-row block and owns the K
// fragments kc = w, w + nw, ... (<= TMAX of them, held in registers for all row blocks: K <= 2048).
// -----------------------------------------------------------------------------------------------------
template <typename WT>
__global__ __launch_bounds__(1024) void lin_rowmajor_kernel(const LinArgs p) {
    constexpr int KT = WTr<WT>::KT;
    constexpr int TMAX = sizeof(WT) == 2 ? 4 : 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    float* red = reinterpret_cast<float*>(smem);       // [nw][256]
    unsigned char* As = smem + (size_t)nw * 1024;      // [16][RS] staged activation
    const int nl = lane & 15, kg = lane >> 4;
    const int n0 = blockIdx.x * 16, NKC = p.NKC, Kpad = NKC * KT;
    const u32x4* wt = reinterpret_cast<const u32x4*>(p.w) + (size_t)blockIdx.x * NKC * 64 + lane;

    // the first activation row is requested BEFORE the weight fragments: vmcnt retires in order, so the row
    // must not queue behind HBM-latency weight loads
    float4 xv[ACMI_STAGE_JMAX];
    load_row(wave < p.M && wave < 16 ? reinterpret_cast<const float*>(p.a) + (size_t)wave * p.K : nullptr, p.K, lane, xv);
    u32x4 wv[TMAX];
#pragma unroll
    for (int i = 0; i < TMAX; ++i) {
        const int kc = wave + i * nw;
        wv[i] = kc < NKC ? ld_frag_nt(wt + (size_t)kc * 64) : u32x4{0u, 0u, 0u, 0u};
    }

    for (int m0 = 0; m0 < p.M; m0 += 16) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int r = wave; r < 16; r += nw) {
            const int m = m0 + r;
            if (m0 != 0 || r != wave)
                load_row(m < p.M ? reinterpret_cast<const float*>(p.a) + (size_t)m * p.K : nullptr, p.K, lane, xv);
            norm_store_row<WT>(xv, p.K, Kpad, p.ln_mode, p.ln_g, p.ln_b, p.eps, As + (size_t)r * p.RS, lane);
        }
        __syncthreads();
        const unsigned char* arow = As + (size_t)nl * p.RS + (size_t)kg * 16;
#pragma unroll
        for (int i = 0; i < TMAX; ++i) {
            const int kc = wave + i * nw;
            if (kc < NKC) {
                const u32x4 av = *reinterpret_cast<const u32x4*>(arow + (size_t)kc * 64);
                mma_frag(av, wv[i], acc, WT());
            }
        }
        // deterministic cross-wave reduction + epilogue
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave * 256 + lane * 4 + r] = acc[r];
        __syncthreads();
        for (int t = threadIdx.x; t < 256; t += blockDim.x) {
            const int nn = t & 15, mm = t >> 4;
            const int idx = (((mm >> 2) * 16 + nn) << 2) + (mm & 3);
            float v = 0.f;
            for (int w = 0; w < nw; ++w) v += red[w * 256 + idx];
            const int gm = m0 + mm, gn = n0 + nn;
            if (gm >= p.M || gn >= p.N) continue;
            if (p.bias) v += p.bias[gn];
            if (p.act == 1) v = gelu_exact(v);
            const size_t oi = (size_t)gm * p.N + gn;
            if (p.residual) v += p.residual[oi];
            if (p.out_mode == ACMI_OUT_TILED) st_f32(reinterpret_cast<WT*>(p.out) + tiled_index<WT>(gm, gn, p.NKC_out), v);
            else if (p.out_mode == ACMI_OUT_BF16) reinterpret_cast<bf16_t*>(p.out)[oi] = f32_to_bf16(v);
            else reinterpret_cast<float*>(p.out)[oi] = v;
        }
        __syncthreads();
    }
}

template <typename WT>
static int launch_rowmajor(LinArgs& a, hipStream_t st) {
    constexpr int KT = WTr<WT>::KT;
    a.NKC = (a.K + KT - 1) / KT;
    a.NKC_out = (a.N + KT - 1) / KT;
    ACMI_REQUIRE(a.K % 4 == 0 && a.NKC * KT <= 2048, "acmi_linear: row-major activation needs K %% 4 == 0 and K <= 2048 (K=%d)", a.K);
    ACMI_REQUIRE(a.stats_out == nullptr && a.xt_hi == nullptr && a.ksplit <= 1 && a.colsum == nullptr && !a.qkv,
                 "acmi_linear: statistics / raw tiled outputs / split-K / folded LayerNorm need a tiled activation");
    int nw = a.NKC < 16 ? a.NKC : 16;
    if (nw < 4) nw = 4;
    a.RS = a.NKC * KT * (int)sizeof(WT) + 16;
    const size_t lds = (size_t)nw * 1024 + (size_t)16 * a.RS;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&lin_rowmajor_kernel<WT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            acmi_set_error("acmi_linear: cannot raise the dynamic LDS limit");
            return ACMI_ELAUNCH;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL((lin_rowmajor_kernel<WT>), dim3((a.N + 15) / 16), dim3(nw * 64), lds, st, a);
    return acmi_check_launch("lin_rowmajor_kernel");
}

// =====================================================================================================
// lin_tiled_kernel: the decode step's GEMM  (tiled activation x tiled weight)
// =====================================================================================================
// One workgroup = 16 output features (x one K slice with split-K); its nw <= 8 waves own the K fragments
// kc = wave, wave + nw, ...  Every wave requests everything it will ever need up front, in the order in which
// it is consumed -- (weight fragment, activation fragments) pairs, then the LayerNorm row statistics, then the
// epilogue operands of its thread -- because vmcnt retires in order and anything requested later (a cold bias
// vector in the epilogue, say) is a full HBM round trip on the tail of the launch.  Absent operands are
// replaced by the address of the wave's own first weight fragment (already in flight: no extra line, page or
// hot spot), so the prologue is branch free.
//   LN 0: plain   1: folded LayerNorm, single-term activation   2: folded LayerNorm, hi + lo activation
//      3: no LayerNorm, hi + lo activation for the first lo_split K fragments (x | a concatenated along K)
struct TlExtras {
    float pm[8], pq[8];     // LN > 0: (mean, M2) partials of this lane's statistics row
    float bias, colsum, res;  // epilogue operands of this thread's first output element
    int tpos;                 // QKV: the position the new K / V rows are stored at
};

template <typename WT, int MT, int LN, int C>
__device__ __forceinline__ void tl_chunk(const LinArgs& p, const u32x4* __restrict__ wt, const u32x4* __restrict__ at,
                                         const u32x4* __restrict__ al, int mts, int mtl, int mtv, int kc0, int nw,
                                         const float* __restrict__ st_ptr, int np,
                                         const float* __restrict__ pb, const float* __restrict__ pc,
                                         const float* __restrict__ pr, const int* __restrict__ ppos, f32x4 (&acc)[MT],
                                         TlExtras& ex) {
    constexpr bool HL = LN == 2 || LN == 3;
    const int lane = threadIdx.x & 63;
    u32x4 bv[C], av[MT][C], lv[HL ? MT : 1][HL ? C : 1];
#pragma unroll
    for (int i = 0; i < C; ++i) {
        const int ko = (kc0 + i * nw) * 64;  // wave-uniform; 32-bit index math (a matrix has < 2^31 fragments' lanes)
        bv[i] = ld_frag_nt(wt + ko + lane);
#pragma unroll
        for (int u = 0; u < MT; ++u) {  // row blocks beyond M re-read the last valid one (their results are dropped)
            const int ub = min(u, mtv - 1);
            av[u][i] = (at + (ub * mts + ko))[lane];
            if (LN == 2) lv[u][i] = (al + (ub * mtl + ko))[lane];
            if (LN == 3)  // fragments past lo_split have no lo term: re-read the last one (L1 hit), zeroed below
                lv[u][i] = (al + (ub * mtl + min(kc0 + i * nw, p.lo_split - 1) * 64))[lane];
        }
    }
    // (the asm keeps these loop-invariant loads here, behind the weight stream, instead of in front of the K loop)
    int opaque0 = 0;
    asm volatile("" : "+s"(opaque0));
    st_ptr += opaque0; pb += opaque0; pc += opaque0; pr += opaque0; ppos += opaque0;
    if (LN == 1 || LN == 2) {
        const int jj = (int)(threadIdx.x & 15);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float2 t = *reinterpret_cast<const float2*>(st_ptr + min(jj + 16 * i, np - 1) * 2);
            ex.pm[i] = t.x; ex.pq[i] = t.y;
        }
    }
    ex.bias = *pb; ex.colsum = *pc; ex.res = *pr; ex.tpos = *ppos;
    __builtin_amdgcn_sched_barrier(0);  // keep every request in front of the first wait
#pragma unroll
    for (int i = 0; i < C; ++i)
#pragma unroll
        for (int u = 0; u < MT; ++u) {
            mma_frag(av[u][i], bv[i], acc[u], WT());
            if (LN == 3 && kc0 + i * nw >= p.lo_split) lv[u][i] = u32x4{0u, 0u, 0u, 0u};
            if (HL) mma_frag(lv[u][i], bv[i], acc[u], WT());
        }
}

template <typename WT, int MT, int LN>
__device__ __forceinline__ void tl_body(const LinArgs& p, const int ntile, const int kslice, const int ksp) {
    constexpr int D = (LN == 2 || LN == 3) ? 2 : 1;
    constexpr bool FOLD = LN == 1 || LN == 2;
    // fragments per straight-line chunk: (1 + MT D) C fragment registers (4 VGPRs each) must leave the kernel
    // without scratch (a kernel with a private segment starts its waves measurably slower) inside the 256
    // VGPRs of a 2-waves-per-SIMD launch
    constexpr int CQ = (LN > 0 ? 44 : 52) / (1 + MT * D);
    constexpr int CMAX = CQ >= 24 ? 24 : (CQ >= 16 ? 16 : (CQ >= 12 ? 12 : (CQ >= 8 ? 8 : (CQ >= 6 ? 6 : (CQ >= 4 ? 4 : 2)))));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: fragment addresses stay in SGPRs
    float* red = reinterpret_cast<float*>(smem);        // [MT][nw][256] partial accumulators
    float* rowstat = red + (size_t)MT * nw * 256;       // [16 MT][2] mean, rstd
    const int n0 = ntile * 16, NKC = p.NKC;
    const u32x4* wt = reinterpret_cast<const u32x4*>(p.w) + (size_t)ntile * NKC * 64;  // + fragment * 64 + lane
    const int mts = p.a_rbs * 64, mtl = p.alo_rbs * 64;  // fragment lanes between consecutive 16-row blocks (a, a_lo)
    const int kcs = p.kcs, kbeg = kslice * kcs;         // this workgroup's K slice
    const float* own = reinterpret_cast<const float*>(wt + (size_t)(kbeg + min(wave, kcs - 1)) * 64 + lane);

    // one group of MT 16-row blocks per workgroup (grid.z): no loop around the body, so that nothing of the
    // epilogue is hoisted in front of the first load
    {
        const int mg = (int)blockIdx.z * 16 * MT;
        const int mtv = min(MT, (p.M - mg + 15) >> 4);
        f32x4 accs[MT];
#pragma unroll
        for (int u = 0; u < MT; ++u) accs[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        const u32x4* at = reinterpret_cast<const u32x4*>(p.a) + (size_t)((mg >> 4) * mts);
        const u32x4* al = D == 2 ? reinterpret_cast<const u32x4*>(p.a_lo) + (size_t)((mg >> 4) * mtl) : nullptr;
        // statistics: groups of 4 rows (16 lanes each), group g of this wave first = wave
        const int ngroups = 4 * mtv;
        const float* st_ptr = own;
        if (FOLD) st_ptr = p.a_stats + min(mg + min(wave, ngroups - 1) * 4 + (lane >> 4), p.M - 1) * p.a_np * 2;
        // this thread's first epilogue element
        const int e0 = (int)threadIdx.x, eu = e0 >> 8, emm = (e0 >> 4) & 15, enn = e0 & 15;
        const int egn = min(n0 + enn, p.N - 1), egm = min(mg + 16 * eu + emm, p.M - 1);
        const float* pb = p.bias != nullptr ? p.bias + egn : own;
        const float* pc = p.colsum != nullptr ? p.colsum + egn : own;
        const float* pr = p.residual != nullptr ? p.residual + (egm * p.N + egn) : own;
        const int* ppos = p.qkv ? p.pos : reinterpret_cast<const int*>(own);
        TlExtras ex;

        int kc = kbeg + wave, rem = p.fpw;              // fragments every wave owns (+ a ragged tail)
#define ACMI_TL_RUN(Cn)                                                                                                 \
        while (rem >= Cn) {                                                                                            \
            tl_chunk<WT, MT, LN, Cn>(p, wt, at, al, mts, mtl, mtv, kc, nw, st_ptr, p.a_np, pb, pc, pr, ppos, accs, ex); \
            kc += Cn * nw; rem -= Cn;                                                                                  \
        }
        if (CMAX >= 24) { ACMI_TL_RUN(24) }
        if (CMAX >= 16) { ACMI_TL_RUN(16) }
        if (CMAX >= 12) { ACMI_TL_RUN(12) }
        if (CMAX >= 8) { ACMI_TL_RUN(8) }
        if (CMAX >= 6) { ACMI_TL_RUN(6) }
        if (CMAX >= 4) { ACMI_TL_RUN(4) }
        ACMI_TL_RUN(2)
        ACMI_TL_RUN(1)
#undef ACMI_TL_RUN
        if (kc < kbeg + kcs)  // ragged tail: the first kcs % nw waves own one more fragment
            tl_chunk<WT, MT, LN, 1>(p, wt, at, al, mts, mtl, mtv, kc, nw, st_ptr, p.a_np, pb, pc, pr, ppos, accs, ex);

        // ---- deterministic cross-wave reduction through LDS
#pragma unroll
        for (int u = 0; u < MT; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((size_t)u * nw + wave) * 256 + lane * 4 + r] = accs[u][r];
        if (FOLD) {
            // mean / rstd of rows mg .. mg + 16 mtv - 1 from the producer's equal-count partials (Chan)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(ex.pm[i]), "+v"(ex.pq[i]));  // stays behind the K loop
            if (wave < ngroups) rowstat_finish(ex.pm, ex.pq, p.a_np, p.a_cnt, p.K, p.eps, lane, rowstat + wave * 8);
            for (int g = wave + nw; g < ngroups; g += nw) {
                rowstat_load(p.a_stats, p.a_np, p.M, mg + g * 4, lane, ex.pm, ex.pq);
                rowstat_finish(ex.pm, ex.pq, p.a_np, p.a_cnt, p.K, p.eps, lane, rowstat + g * 8);
            }
        }
        __syncthreads();

        // ---- epilogue: one output element per thread and pass
        for (int e = (int)threadIdx.x; e < 256 * mtv; e += (int)blockDim.x) {
            const int u = e >> 8, mm = (e >> 4) & 15, nn = e & 15;
            const bool first = e == (int)threadIdx.x;
            const int idx = (((mm >> 2) * 16 + nn) << 2) + (mm & 3);
            float v = 0.f;
            for (int w = 0; w < nw; ++w) v += red[((size_t)u * nw + w) * 256 + idx];
            const int gm = mg + 16 * u + mm, gn = n0 + nn;
            const bool valid = gm < p.M && gn < p.N;
            if (ksp > 1) {  // split-K: raw partial sums; bias / activation / residual are applied by the reducer
                if (valid) reinterpret_cast<float*>(p.out)[((size_t)kslice * p.M + gm) * p.N + gn] = v;
                continue;
            }
            size_t oi = 0;
            if (valid) {
                if (FOLD) {  // folded LayerNorm: rstd * (x W'^T - mean * colsum)
                    const float* rs = rowstat + (u * 16 + mm) * 2;
                    v = rs[1] * (v - rs[0] * (first ? ex.colsum : p.colsum[gn]));
                }
                if (p.bias) v += first ? ex.bias : p.bias[gn];
                if (!p.qkv) {
                    if (p.act == 1) v = gelu_exact(v);
                    oi = (size_t)gm * p.N + gn;
                    if (p.residual) v += first ? ex.res : p.residual[oi];
                }
            }
            if (p.stats_out != nullptr) {
                // (mean, M2) of this workgroup's 16 output features per row, for the LayerNorm of the consumer
                float sm = valid ? v : 0.f;
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) sm += __shfl_xor(sm, off, 64);
                const float mb = sm * (1.0f / 16.0f);
                float dq = valid ? (v - mb) * (v - mb) : 0.f;
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) dq += __shfl_xor(dq, off, 64);
                if (nn == 0 && gm < p.M)
                    *reinterpret_cast<float2*>(p.stats_out + ((size_t)gm * (p.N >> 4) + ntile) * 2) = make_float2(mb, dq);
            }
            if (!valid) continue;
            if (p.xt_hi != nullptr) {  // the residual stream also raw in A-fragment order for the next GEMM
                const size_t ti = tiled_index<WT>(gm, gn, p.xt_nkc);
                if (sizeof(WT) == 2) {
                    const bf16_t hi = f32_to_bf16(v);
                    reinterpret_cast<bf16_t*>(p.xt_hi)[ti] = hi;
                    if (p.xt_lo != nullptr)
                        reinterpret_cast<bf16_t*>(p.xt_lo)[tiled_index<WT>(gm, gn, p.xt_lo_nkc)] = f32_to_bf16(v - bf16_to_f32(hi));
                } else {
                    reinterpret_cast<float*>(p.xt_hi)[ti] = v;
                }
            }
            if (p.qkv) {
                const int part = gn / p.d, f = gn - part * p.d;
                if (part == 0) {
                    p.q_out[(size_t)gm * p.d + f] = v;
                } else {
                    const int h = f / p.hd, dd = f - h * p.hd;
                    const int pidx = gm / p.rpp, brow = gm - pidx * p.rpp;  // several positions per call (prefill)
                    const size_t ci = (((size_t)brow * p.H + h) * p.Tcap + ex.tpos + pidx) * p.hd + dd;
                    void* cache = part == 1 ? p.k_cache : p.v_cache;
                    if (p.kv_bf16) reinterpret_cast<bf16_t*>(cache)[ci] = f32_to_bf16(v);
                    else reinterpret_cast<float*>(cache)[ci] = v;
                }
            } else if (p.out_mode == ACMI_OUT_TILED) {
                st_f32(reinterpret_cast<WT*>(p.out) + tiled_index<WT>(gm, gn, p.NKC_out), v);
            } else if (p.out_mode == ACMI_OUT_BF16) {
                reinterpret_cast<bf16_t*>(p.out)[oi] = f32_to_bf16(v);
            } else {
                reinterpret_cast<float*>(p.out)[oi] = v;
            }
        }
    }
}

template <typename WT, int MT, int LN>
__global__ __launch_bounds__(512) void lin_tiled_kernel(const LinArgs p) {
    tl_body<WT, MT, LN>(p, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y);
}

// Two independent GEMMs of the chain in ONE launch (one dependency edge less): workgroups [0, tiles0) run p0
// (plain), the rest run p1 (LN 3: x | a concatenated along K, no LayerNorm).  Used for
//   x1 = x0 + att W_out^T   and   r = [x0 | att] [W_cq' | W_cq' W_out]^T  (= x1 W_cq'^T, the cross-attention
// query before its LayerNorm statistics are applied), see acmi_lm_step.
template <typename WT, int MT, int LNB>
__global__ __launch_bounds__(512) void lin_pair_kernel(const LinArgs p0, const LinArgs p1, const int tiles0) {
    if ((int)blockIdx.x < tiles0) tl_body<WT, MT, 0>(p0, (int)blockIdx.x, 0, 1);
    else tl_body<WT, MT, LNB>(p1, (int)blockIdx.x - tiles0, 0, 1);
}

// Workgroup size.  (1) Waves are not free: the dispatcher starts ~1.25 waves / ns (a 288-workgroup x 16-wave
// launch with no loads at all takes 6.2 us), so a wave should own ~12 fragments or more, all of them requested
// before its first wait.  (2) The kernel needs > 128 VGPRs for that, i.e. a CU holds 8 waves: nw in {8, 4, 2, 1}
// packs 1, 2, 4, 8 workgroups per CU exactly, and the grid must fit the 256 CUs in ONE round (a 288-workgroup
// grid of 6-wave workgroups runs 256 + 32: the launch takes twice as long).
static int tiled_waves(int tiles, int frags) {
    int nw = tiles <= 256 ? 8 : (tiles <= 512 ? 4 : (tiles <= 1024 ? 2 : 1));
    while (nw > 1 && frags < 12 * nw) nw >>= 1;
    static const char* e = getenv("ACMI_LIN_NW");
    if (e && atoi(e) > 0 && atoi(e) < nw) nw = atoi(e);
    return nw;
}

template <typename WT>
static int tiled_prepare(LinArgs& a) {
    constexpr int KT = WTr<WT>::KT;
    a.NKC = (a.K + KT - 1) / KT;
    a.NKC_out = (a.N + KT - 1) / KT;
    if (a.ksplit < 1 || a.NKC % a.ksplit != 0) a.ksplit = 1;
    if (a.a_rbs <= 0) a.a_rbs = a.NKC;
    if (a.alo_rbs <= 0) a.alo_rbs = a.lo_split > 0 ? a.lo_split : a.NKC;  // a lo buffer holds only the columns that have one
    ACMI_REQUIRE(a.a_rbs >= a.NKC, "acmi_linear: a_rbs=%d < K tiles %d", a.a_rbs, a.NKC);
    ACMI_REQUIRE(a.stats_out == nullptr || a.N % 16 == 0, "acmi_linear: stats_out needs N %% 16 == 0 (N=%d)", a.N);
    ACMI_REQUIRE(a.ksplit == 1 || (!a.qkv && a.stats_out == nullptr && a.xt_hi == nullptr),
                 "acmi_linear: split-K is incompatible with QKV scatter / stats_out / xt_hi");
    return ACMI_OK;
}

template <typename WT, int MT, int LN>
static int launch_tiled_t(LinArgs& a, hipStream_t st) {
    int rc = tiled_prepare<WT>(a);
    if (rc) return rc;
    const int tiles = ((a.N + 15) / 16) * a.ksplit, frags = a.NKC / a.ksplit;
    const int nw = tiled_waves(tiles, frags);
    a.kcs = frags; a.fpw = frags / nw;
    const size_t lds = (size_t)MT * nw * 1024 + (size_t)MT * 128;
    hipLaunchKernelGGL((lin_tiled_kernel<WT, MT, LN>), dim3((a.N + 15) / 16, a.ksplit, (a.M + 16 * MT - 1) / (16 * MT)), dim3(nw * 64), lds, st, a);
    return acmi_check_launch("lin_tiled_kernel");
}

template <typename WT>
static int launch_tiled(LinArgs& a, hipStream_t st) {
    const int mt = a.M > 32 ? 4 : (a.M > 16 ? 2 : 1);  // 1, 2 or 4 16-row blocks share each weight fragment
    const int ln = a.colsum == nullptr ? (a.lo_split > 0 ? 3 : 0) : (a.a_lo != nullptr ? 2 : 1);
#define ACMI_TL_CASE(MTv, LNv) if (mt == MTv && ln == LNv) return launch_tiled_t<WT, MTv, LNv>(a, st);
    ACMI_TL_CASE(1, 0) ACMI_TL_CASE(1, 1) ACMI_TL_CASE(1, 2) ACMI_TL_CASE(1, 3)
    ACMI_TL_CASE(2, 0) ACMI_TL_CASE(2, 1) ACMI_TL_CASE(2, 2) ACMI_TL_CASE(2, 3)
    ACMI_TL_CASE(4, 0) ACMI_TL_CASE(4, 1) ACMI_TL_CASE(4, 2) ACMI_TL_CASE(4, 3)
#undef ACMI_TL_CASE
    return ACMI_EINVAL;
}

// p0 (plain tiled GEMM) and p1 (x | a concatenated along K, hi + lo for the x part) in one launch
template <typename WT>
static int launch_pair(LinArgs& p0, LinArgs& p1, hipStream_t st) {
    int rc;
    if ((rc = tiled_prepare<WT>(p0)) || (rc = tiled_prepare<WT>(p1))) return rc;
    ACMI_REQUIRE(p0.M == p1.M && p0.ksplit == 1 && p1.ksplit == 1 && !p0.qkv && !p1.qkv && p0.colsum == nullptr &&
                 p1.colsum == nullptr && p0.a_lo == nullptr && (p1.a_lo == nullptr || (p1.lo_split > 0 && p1.lo_split <= p1.NKC)),
                 "acmi_linear_pair: bad operands");
    const bool hl = p1.a_lo != nullptr;  // f32 activations are a single term
    const int t0 = (p0.N + 15) / 16, t1 = (p1.N + 15) / 16;
    const int nw = tiled_waves(t0 + t1, p0.NKC > p1.NKC ? p0.NKC : p1.NKC);  // sized for the longer K
    p0.kcs = p0.NKC; p0.fpw = p0.NKC / nw;
    p1.kcs = p1.NKC; p1.fpw = p1.NKC / nw;
    const int mt = p0.M > 32 ? 4 : (p0.M > 16 ? 2 : 1);
    const size_t lds = (size_t)mt * nw * 1024 + (size_t)mt * 128;
    const dim3 grid(t0 + t1, 1, (p0.M + 16 * mt - 1) / (16 * mt)), block(nw * 64);
#define ACMI_PAIR_CASE(MTv)                                                                              \
    if (mt == MTv) {                                                                                     \
        if (hl) hipLaunchKernelGGL((lin_pair_kernel<WT, MTv, 3>), grid, block, lds, st, p0, p1, t0);      \
        else hipLaunchKernelGGL((lin_pair_kernel<WT, MTv, 0>), grid, block, lds, st, p0, p1, t0);         \
    }
    ACMI_PAIR_CASE(1) ACMI_PAIR_CASE(2) ACMI_PAIR_CASE(4)
#undef ACMI_PAIR_CASE
    return acmi_check_launch("lin_pair_kernel");
}

static int launch_lin(LinArgs& a, int wdtype, hipStream_t st) {
    ACMI_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "acmi_linear: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
    if (a.ksplit < 1) a.ksplit = 1;
    if (a.rpp <= 0) a.rpp = a.M;
    ACMI_REQUIRE(!(a.a_tiled && a.ln_mode), "acmi_linear: LayerNorm needs a row-major activation");
    ACMI_REQUIRE(a.colsum == nullptr || (a.a_tiled && a.a_stats != nullptr && a.a_np >= 1 && a.a_np <= 128 && a.a_np * a.a_cnt == a.K),
                 "acmi_linear: folded LayerNorm needs a tiled activation and row statistics (np=%d cnt=%d K=%d)", a.a_np, a.a_cnt, a.K);
    ACMI_REQUIRE(a.colsum == nullptr || a.ksplit == 1, "acmi_linear: folded LayerNorm cannot be combined with split-K");
    if (a.a_tiled) return wdtype == ACMI_BF16 ? launch_tiled<bf16_t>(a, st) : launch_tiled<float>(a, st);
    return wdtype == ACMI_BF16 ? launch_rowmajor<bf16_t>(a, st) : launch_rowmajor<float>(a, st);
}

extern "C" int acmi_linear(const void* a, int a_mode, const float* ln_g, const float* ln_b, float eps, const void* w,
                           int wdtype, const float* bias, const float* residual, void* out, int out_mode, int act, int M,
                           int N, int K, void* stream) {
    LinArgs p = {};
    p.a = a; p.a_tiled = a_mode == ACMI_A_TILED;
    ACMI_REQUIRE(a_mode >= 0 && a_mode <= 2, "acmi_linear: bad a_mode %d", a_mode);
    ACMI_REQUIRE((ln_g == nullptr) == (ln_b == nullptr), "acmi_linear: ln_g and ln_b go together");
    p.ln_mode = ln_g ? 2 : (a_mode == ACMI_A_ROWMAJOR_F32_NORM ? 1 : 0);
    p.ln_g = ln_g; p.ln_b = ln_b; p.eps = eps;
    p.w = w; p.bias = bias; p.residual = residual; p.out = out; p.out_mode = out_mode; p.act = act;
    p.M = M; p.N = N; p.K = K;
    return launch_lin(p, wdtype, (hipStream_t)stream);
}

static int desc_to_args(const acmi_linear_desc& c, LinArgs& p) {
    ACMI_REQUIRE(c.a_mode >= 0 && c.a_mode <= 2, "acmi_linear: bad a_mode %d", c.a_mode);
    ACMI_REQUIRE((c.ln_g == nullptr) == (c.ln_b == nullptr), "acmi_linear: ln_g and ln_b go together");
    const int kt = c.wdtype == ACMI_BF16 ? 32 : 16;
    p.a = c.a; p.a_tiled = c.a_mode == ACMI_A_TILED;
    p.ln_mode = c.ln_g ? 2 : (c.a_mode == ACMI_A_ROWMAJOR_F32_NORM ? 1 : 0);
    p.ln_g = c.ln_g; p.ln_b = c.ln_b; p.eps = c.eps;
    if (c.colsum != nullptr) {
        ACMI_REQUIRE(c.a_mode == ACMI_A_TILED && c.a_stats != nullptr, "acmi_linear: colsum needs a tiled activation and a_stats");
        p.a_stats = c.a_stats; p.a_np = c.a_stats_np; p.a_cnt = c.a_stats_cnt; p.colsum = c.colsum; p.a_lo = c.a_lo;
    } else if (c.a_lo != nullptr) {  // hi + lo activation without LayerNorm (first lo_K columns)
        ACMI_REQUIRE(c.a_mode == ACMI_A_TILED && c.lo_K > 0 && c.lo_K <= c.K && c.lo_K % kt == 0,
                     "acmi_linear: a_lo without colsum needs a tiled activation and lo_K %% %d == 0 (lo_K=%d)", kt, c.lo_K);
        p.a_lo = c.a_lo; p.lo_split = c.lo_K / kt;
    }
    p.a_rbs = c.a_rbs; p.alo_rbs = c.a_lo_rbs;
    if (c.xt_hi != nullptr) {
        p.xt_hi = c.xt_hi; p.xt_lo = c.xt_lo;
        p.xt_nkc = c.xt_rbs > 0 ? c.xt_rbs : (c.N + kt - 1) / kt;
        p.xt_lo_nkc = c.xt_lo_rbs > 0 ? c.xt_lo_rbs : (c.N + kt - 1) / kt;
        ACMI_REQUIRE(p.xt_nkc >= (c.N + kt - 1) / kt && p.xt_lo_nkc >= (c.N + kt - 1) / kt,
                     "acmi_linear: xt_rbs=%d / xt_lo_rbs=%d too small for N=%d", c.xt_rbs, c.xt_lo_rbs, c.N);
    }
    p.stats_out = c.stats_out; p.ksplit = c.ksplit;
    p.w = c.w; p.bias = c.bias; p.residual = c.residual; p.out = c.out; p.out_mode = c.out_mode; p.act = c.act;
    p.M = c.M; p.N = c.N; p.K = c.K;
    return ACMI_OK;
}

extern "C" int acmi_linear_ex(const acmi_linear_desc* dsc, void* stream) {
    ACMI_REQUIRE(dsc != nullptr, "acmi_linear_ex: null descriptor");
    LinArgs p = {};
    int rc = desc_to_args(*dsc, p);
    if (rc) return rc;
    return launch_lin(p, dsc->wdtype, (hipStream_t)stream);
}

extern "C" int acmi_linear_pair(const acmi_linear_desc* plain, const acmi_linear_desc* xcat, void* stream) {
    ACMI_REQUIRE(plain != nullptr && xcat != nullptr, "acmi_linear_pair: null descriptor");
    ACMI_REQUIRE(plain->wdtype == xcat->wdtype && plain->a_mode == ACMI_A_TILED && xcat->a_mode == ACMI_A_TILED,
                 "acmi_linear_pair: both GEMMs take tiled activations of one element type");
    LinArgs p0 = {}, p1 = {};
    int rc;
    if ((rc = desc_to_args(*plain, p0)) || (rc = desc_to_args(*xcat, p1))) return rc;
    ACMI_REQUIRE(p0.M > 0 && p0.N > 0 && p0.K > 0 && p1.N > 0 && p1.K > 0, "acmi_linear_pair: empty problem");
    return plain->wdtype == ACMI_BF16 ? launch_pair<bf16_t>(p0, p1, (hipStream_t)stream)
                                      : launch_pair<float>(p0, p1, (hipStream_t)stream);
}

// =====================================================================================================
// single-query attention over a KV cache
// =====================================================================================================

__device__ __forceinline__ float raw_to_f32(bf16_t v) { return bf16_to_f32(v); }
__device__ __forceinline__ float raw_to_f32(float v) { return v; }

struct AttnArgs {
    const float* q; const void* kc; const void* vc; void* out;
    int out_tiled, out_bf16, out_rbs, out_col0;  // tiled output: K tiles per 16-row block, first column
    int H, Tcap, len; const int* len_dev; int len_bias; float scale;
    int rpp;      // rows per position: query row b belongs to cache row b % rpp; with len_dev its length grows by b / rpp
    // optional LayerNorm hook on q (the cross-attention query arrives as x W'^T, see acmi_linear_pair):
    //   q <- rstd[b] (q - mean[b] colsum) + bias, mean / rstd of row b from the (mean, M2) partials of x
    const float* q_stats; int q_np, q_cnt, q_K; float q_eps; const float* q_colsum; const float* q_bias;
};

template <typename KT, int HD, bool QN>  // QN: LayerNorm hook on q (separate instantiation: no branch around its loads)
__global__ __launch_bounds__(256) void attn_decode_kernel(const AttnArgs p) {
    const float* __restrict__ q = p.q;
    const KT* __restrict__ kc = reinterpret_cast<const KT*>(p.kc);
    const KT* __restrict__ vc = reinterpret_cast<const KT*>(p.vc);
    const int H = p.H, Tcap = p.Tcap;
    const float scale = p.scale;
    constexpr int DPL = HD >= 8 ? 8 : HD;  // dims per lane
    constexpr int LPP = HD / DPL;          // lanes per position
    constexpr int PPI = 64 / LPP;          // positions covered by one load instruction of a wave
    constexpr int NI = sizeof(KT) == 2 ? 8 : 4;  // positions per lane per chunk (K and V loads in flight: 2 * NI)
    typedef KT rawv __attribute__((ext_vector_type(DPL)));
    constexpr int CH = NI * PPI;
    const int h = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane % LPP, pp = lane / LPP;
    const int b0 = b % p.rpp, pidx = b / p.rpp;   // several positions per call (prefill): cache row, position index
    const int len = p.len_dev ? (*p.len_dev + p.len_bias + pidx) : p.len;

    float qv[DPL];
#pragma unroll
    for (int e = 0; e < DPL; ++e) qv[e] = q[((size_t)b * H + h) * HD + c * DPL + e];
    // LayerNorm hook: everything it needs is requested here, consumed after the first K / V chunk is in flight
    float qcs[DPL], qb[DPL], spm[2], spq[2];
    if (QN) {
#pragma unroll
        for (int e = 0; e < DPL; ++e) {
            qcs[e] = p.q_colsum[h * HD + c * DPL + e];
            qb[e] = p.q_bias[h * HD + c * DPL + e];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {  // np <= 128 equal-count partials of row b, contiguous
            const float2 t = *reinterpret_cast<const float2*>(p.q_stats + ((size_t)b * p.q_np + min(lane + 64 * i, p.q_np - 1)) * 2);
            spm[i] = t.x; spq[i] = t.y;
        }
    }
    const KT* kb = kc + ((size_t)b0 * H + h) * Tcap * HD + c * DPL;
    const KT* vb = vc + ((size_t)b0 * H + h) * Tcap * HD + c * DPL;

    float m = -INFINITY, l = 0.f, o[DPL];
#pragma unroll
    for (int e = 0; e < DPL; ++e) o[e] = 0.f;

    const int nwv = blockDim.x >> 6;  // 1, 2 or 4 waves share the positions of this (row, head)
    // K and V of a whole chunk are requested together (2 * NI wide loads in flight per lane); the first chunk
    // goes out before anything waits on q (its LayerNorm hook needs the fresh statistics of x)
    rawv kr[NI], vr[NI];
    auto load_kv = [&](int t0) {
        // branch free: lanes past the end re-read the last position (their scores are masked below); a per-lane
        // zero fill would write the registers of loads still in flight and make every load wait for the previous
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int t = min(t0 + i * PPI + pp, len - 1);
            kr[i] = __builtin_nontemporal_load(reinterpret_cast<const rawv*>(kb + (size_t)t * HD));
            vr[i] = __builtin_nontemporal_load(reinterpret_cast<const rawv*>(vb + (size_t)t * HD));
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the 2 * NI requests together (the scheduler sinks them to their uses)
    };
    load_kv(wave * CH);
    if (QN) {  // Chan combination of the partials -> mean, rstd of row b; then the affine map of q
        const bool v0 = lane < p.q_np, v1 = lane + 64 < p.q_np;
        const float mean = wave_sum((v0 ? spm[0] : 0.f) + (v1 ? spm[1] : 0.f)) / (float)p.q_np;
        const float d0 = spm[0] - mean, d1 = spm[1] - mean;
        const float q2 = (v0 ? spq[0] + (float)p.q_cnt * d0 * d0 : 0.f) + (v1 ? spq[1] + (float)p.q_cnt * d1 * d1 : 0.f);
        const float rstd = 1.0f / sqrtf(wave_sum(q2) / (float)p.q_K + p.q_eps);
#pragma unroll
        for (int e = 0; e < DPL; ++e) qv[e] = rstd * (qv[e] - mean * qcs[e]) + qb[e];
    }
    for (int t0 = wave * CH; t0 < len; t0 += nwv * CH) {   // kr / vr hold the chunk at t0
        float s[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int t = t0 + i * PPI + pp;
            float part = 0.f;
#pragma unroll
            for (int e = 0; e < DPL; ++e) part = fmaf(qv[e], raw_to_f32(kr[i][e]), part);
#pragma unroll
            for (int off = 1; off < LPP; off <<= 1) part += __shfl_xor(part, off, 64);
            s[i] = (t < len) ? part * scale : -INFINITY;
        }
        float cmax = s[0];
#pragma unroll
        for (int i = 1; i < NI; ++i) cmax = fmaxf(cmax, s[i]);
#pragma unroll
        for (int off = LPP; off < 64; off <<= 1) cmax = fmaxf(cmax, __shfl_xor(cmax, off, 64));
        const float m_new = fmaxf(m, cmax);  // finite: every processed chunk has >= 1 valid position
        const float alpha = expf(m - m_new);
        l *= alpha;
#pragma unroll
        for (int e = 0; e < DPL; ++e) o[e] *= alpha;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int t = t0 + i * PPI + pp;
            const float pr = (t < len) ? expf(s[i] - m_new) : 0.f;
            l += pr;
#pragma unroll
            for (int e = 0; e < DPL; ++e) o[e] = fmaf(pr, raw_to_f32(vr[i][e]), o[e]);
        }
        m = m_new;
        if (t0 + nwv * CH < len) load_kv(t0 + nwv * CH);
    }
#pragma unroll
    for (int off = LPP; off < 64; off <<= 1) {
        l += __shfl_xor(l, off, 64);
#pragma unroll
        for (int e = 0; e < DPL; ++e) o[e] += __shfl_xor(o[e], off, 64);
    }
    __shared__ float sm_o[4][HD];
    __shared__ float sm_m[4], sm_l[4];
    if (lane < LPP) {
#pragma unroll
        for (int e = 0; e < DPL; ++e) sm_o[wave][c * DPL + e] = o[e];
    }
    if (lane == 0) { sm_m[wave] = m; sm_l[wave] = l; }
    __syncthreads();
    if (threadIdx.x < HD) {
        float M = sm_m[0];
        for (int w = 1; w < nwv; ++w) M = fmaxf(M, sm_m[w]);
        float num = 0.f, den = 0.f;
        for (int w = 0; w < nwv; ++w) {
            const float f = (sm_m[w] == -INFINITY) ? 0.f : expf(sm_m[w] - M);
            num += f * sm_o[w][threadIdx.x];
            den += f * sm_l[w];
        }
        const float r = num / den;
        const int f = h * HD + threadIdx.x;
        if (!p.out_tiled) {
            reinterpret_cast<float*>(p.out)[(size_t)b * H * HD + f] = r;
        } else if (p.out_bf16) {  // A-fragment order for the out-projection GEMM (include/acmi.h)
            reinterpret_cast<bf16_t*>(p.out)[tiled_index<bf16_t>(b, p.out_col0 + f, p.out_rbs)] = f32_to_bf16(r);
        } else {
            reinterpret_cast<float*>(p.out)[tiled_index<float>(b, p.out_col0 + f, p.out_rbs)] = r;
        }
    }
}

template <typename KT>
static int launch_attn_t(const AttnArgs& a, int Beff, int hd, hipStream_t st) {
    static int attn_nw = -1;
    if (attn_nw < 0) { const char* e = getenv("ACMI_ATTN_NW"); attn_nw = e ? atoi(e) : 4; if (attn_nw != 1 && attn_nw != 2) attn_nw = 4; }
    // waves per (row, head): 4 by default; a host-known short length (cross-attention) needs no more waves than
    // it has 64-position chunks (bf16 cache, hd 64) -- idle waves still cost dispatch time
    int nwv = attn_nw;
    if (a.len_dev == nullptr) {
        const int dpl = hd >= 8 ? 8 : hd, chunk = (sizeof(KT) == 2 ? 8 : 4) * (64 / (hd / dpl));
        const int need = (a.len + chunk - 1) / chunk;
        while (nwv > 1 && nwv / 2 >= need) nwv /= 2;
    }
    dim3 grid(a.H, Beff), block(64 * nwv);
#define ACMI_ATTN_CASE(HD)                                                                              \
    case HD:                                                                                            \
        if (a.q_colsum != nullptr) hipLaunchKernelGGL((attn_decode_kernel<KT, HD, true>), grid, block, 0, st, a);  \
        else hipLaunchKernelGGL((attn_decode_kernel<KT, HD, false>), grid, block, 0, st, a);            \
        break;
    switch (hd) {
        ACMI_ATTN_CASE(4)
        ACMI_ATTN_CASE(8)
        ACMI_ATTN_CASE(16)
        ACMI_ATTN_CASE(32)
        ACMI_ATTN_CASE(64)
        ACMI_ATTN_CASE(128)
        default:
            acmi_set_error("acmi_attn_decode: head dim %d unsupported (4,8,16,32,64,128)", hd);
            return ACMI_EINVAL;
    }
#undef ACMI_ATTN_CASE
    return acmi_check_launch("attn_decode_kernel");
}

extern "C" int acmi_attn_decode_ex(const acmi_attn_desc* dsc, void* stream) {
    ACMI_REQUIRE(dsc != nullptr, "acmi_attn_decode_ex: null descriptor");
    const acmi_attn_desc& c = *dsc;
    ACMI_REQUIRE(c.out_mode == ACMI_OUT_TILED || c.out_mode == ACMI_OUT_F32, "acmi_attn_decode: bad out_mode %d", c.out_mode);
    ACMI_REQUIRE(c.Beff > 0 && c.H > 0 && c.Tcap > 0 && c.hd > 0, "acmi_attn_decode: bad shape");
    ACMI_REQUIRE(c.len_dev != nullptr || (c.len > 0 && c.len <= c.Tcap), "acmi_attn_decode: len=%d out of (0, %d]", c.len, c.Tcap);
    const int kt = c.out_dtype == ACMI_BF16 ? 32 : 16, nkc = (c.H * c.hd + kt - 1) / kt;
    AttnArgs a = {};
    a.q = c.q; a.kc = c.k_cache; a.vc = c.v_cache; a.out = c.out;
    a.out_tiled = c.out_mode == ACMI_OUT_TILED; a.out_bf16 = c.out_dtype == ACMI_BF16;
    a.out_rbs = c.out_rbs > 0 ? c.out_rbs : nkc; a.out_col0 = c.out_col0;
    ACMI_REQUIRE(c.out_col0 >= 0 && c.out_col0 % kt == 0 && a.out_rbs * kt >= c.out_col0 + c.H * c.hd,
                 "acmi_attn_decode: tiled output placement col0=%d rbs=%d does not hold %d columns", c.out_col0, a.out_rbs, c.H * c.hd);
    a.H = c.H; a.Tcap = c.Tcap; a.len = c.len; a.len_dev = c.len_dev; a.len_bias = c.len_bias;
    a.rpp = c.cache_rows > 0 ? c.cache_rows : c.Beff;
    ACMI_REQUIRE(c.Beff % a.rpp == 0, "acmi_attn_decode: %d query rows are not a multiple of %d cache rows", c.Beff, a.rpp);
    a.scale = 1.0f / sqrtf((float)c.hd);
    if (c.q_colsum != nullptr) {
        ACMI_REQUIRE(c.q_stats != nullptr && c.q_stats_np >= 1 && c.q_stats_np <= 128 && c.q_stats_np * c.q_stats_cnt > 0,
                     "acmi_attn_decode: q LayerNorm hook needs 1..128 statistics partials");
        a.q_stats = c.q_stats; a.q_np = c.q_stats_np; a.q_cnt = c.q_stats_cnt; a.q_K = c.q_stats_np * c.q_stats_cnt;
        a.q_eps = c.eps; a.q_colsum = c.q_colsum;
        a.q_bias = c.q_bias != nullptr ? c.q_bias : nullptr;
        ACMI_REQUIRE(c.q_bias != nullptr, "acmi_attn_decode: q_bias is required with q_colsum (pass zeros for none)");
    }
    return c.kvdtype == ACMI_BF16 ? launch_attn_t<bf16_t>(a, c.Beff, c.hd, (hipStream_t)stream)
                                  : launch_attn_t<float>(a, c.Beff, c.hd, (hipStream_t)stream);
}

extern "C" int acmi_attn_decode(const float* q, const void* k_cache, const void* v_cache, int kvdtype, void* out,
                                int out_mode, int out_dtype, int Beff, int H, int hd, int Tcap, int len,
                                const int* len_dev, int len_bias, void* stream) {
    acmi_attn_desc c = {};
    c.q = q; c.k_cache = k_cache; c.v_cache = v_cache; c.kvdtype = kvdtype; c.out = out; c.out_mode = out_mode;
    c.out_dtype = out_dtype; c.Beff = Beff; c.H = H; c.hd = hd; c.Tcap = Tcap; c.len = len; c.len_dev = len_dev;
    c.len_bias = len_bias;
    return acmi_attn_decode_ex(&c, stream);
}

// scatter [Beff, L, H*hd] f32 rows into a [Beff, H, Tcap, hd] cache
template <typename KT>
__global__ void kv_store_kernel(const float* __restrict__ src, KT* __restrict__ cache, int H, int hd, int Tcap, int t0,
                                int L, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int dd = i % hd;
        size_t r = i / hd;
        const int h = r % H; r /= H;
        const int t = r % L;
        const int b = r / L;
        st_f32(cache + (((size_t)b * H + h) * Tcap + t0 + t) * hd + dd, src[i]);
    }
}

extern "C" int acmi_kv_store(const float* src, void* cache, int kvdtype, int Beff, int H, int hd, int Tcap, int t0,
                             int L, void* stream) {
    ACMI_REQUIRE(t0 >= 0 && L > 0 && t0 + L <= Tcap, "acmi_kv_store: range [%d, %d) outside cache %d", t0, t0 + L, Tcap);
    const size_t total = (size_t)Beff * L * H * hd;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    if (kvdtype == ACMI_BF16)
        hipLaunchKernelGGL(kv_store_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src,
                           reinterpret_cast<bf16_t*>(cache), H, hd, Tcap, t0, L, total);
    else
        hipLaunchKernelGGL(kv_store_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src,
                           reinterpret_cast<float*>(cache), H, hd, Tcap, t0, L, total);
    return acmi_check_launch("kv_store_kernel");
}

// =====================================================================================================
// embedding sum + sinusoidal position
// =====================================================================================================

__device__ __forceinline__ float block_sum(float v, float* sval);

struct EmbedArgs {
    const void* emb[16]; int w_bf16;
    const int64_t* gen_sequence; int B, Beff, K, S, card;
    const float* prepend; int P;
    const float* pos_table; float pos_scale;
    const int* pos;
    float* x; int d;
    float* stats;  // [M][1][2]: (mean, M2) of every produced row (one partial of d elements)
    void* xt_hi; void* xt_lo; int xt_nkc, xt_lo_nkc;  // folded LayerNorm: the raw row in fragment order (hi / lo), or NULL
};

__global__ __launch_bounds__(256) void embed_kernel(const EmbedArgs p) {
    __shared__ float sred[4];
    const int m = blockIdx.x;              // row = position-of-the-call * Beff + CFG row
    const int pidx = m / p.Beff, m0 = m - pidx * p.Beff;
    const int g = *p.pos + pidx;
    const int b = m0 % p.B;
    float loc[8];  // d <= 2048
    float sum = 0.f;
    int cnt = 0;
    // the K tokens of this step, read once (inside the channel loop the stores to x would force a reload and
    // serialise token -> embedding-row round trips: 17 -> 7 us)
    int toks[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        int64_t tok = (g >= p.P && k < p.K) ? p.gen_sequence[((size_t)b * p.K + k) * p.S + (g - p.P)] : 0;
        if (tok < 0 || tok > p.card) tok = p.card;  // never happens for a well-formed sequence
        toks[k] = (int)tok;
    }
    for (int cch = threadIdx.x; cch < p.d; cch += blockDim.x, ++cnt) {
        float v;
        if (g < p.P) {
            v = p.prepend[((size_t)m0 * p.P + g) * p.d + cch];
        } else {
            v = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (k < p.K) {
                    const size_t ei = (size_t)toks[k] * p.d + cch;
                    v += p.w_bf16 ? bf16_to_f32(reinterpret_cast<const bf16_t*>(p.emb[k])[ei])
                                  : reinterpret_cast<const float*>(p.emb[k])[ei];
                }
            }
        }
        v += p.pos_scale * p.pos_table[(size_t)g * p.d + cch];
        p.x[(size_t)m * p.d + cch] = v;
        if (p.xt_hi != nullptr) {
            if (p.w_bf16) {
                const size_t ti = tiled_index<bf16_t>(m, cch, p.xt_nkc);
                const bf16_t hi = f32_to_bf16(v);
                reinterpret_cast<bf16_t*>(p.xt_hi)[ti] = hi;
                if (p.xt_lo != nullptr)
                    reinterpret_cast<bf16_t*>(p.xt_lo)[tiled_index<bf16_t>(m, cch, p.xt_lo_nkc)] = f32_to_bf16(v - bf16_to_f32(hi));
            } else {
                reinterpret_cast<float*>(p.xt_hi)[tiled_index<float>(m, cch, p.xt_nkc)] = v;
            }
        }
        loc[cnt] = v;
        sum += v;
    }
    // two-pass (mean, M2) of the row for the first LayerNorm
    const float mean = block_sum(sum, sred) / (float)p.d;
    float q = 0.f;
    for (int i = 0; i < cnt; ++i) q += (loc[i] - mean) * (loc[i] - mean);
    q = block_sum(q, sred);
    if (threadIdx.x == 0) { p.stats[(size_t)m * 2] = mean; p.stats[(size_t)m * 2 + 1] = q; }
}

// create_sin_embedding (transformer.py:70-89): one block per position, computed once per run geometry
__global__ __launch_bounds__(256) void pos_table_kernel(const float* __restrict__ freq, float* __restrict__ table, int d) {
    const int t = blockIdx.x, half = d / 2;
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        const int i = c < half ? c : c - half;
        const float phase = (float)t / freq[i];
        table[(size_t)t * d + c] = c < half ? cosf(phase) : sinf(phase);
    }
}

extern "C" int acmi_pos_table(const float* freq, float* table, int T, int d, void* stream) {
    ACMI_REQUIRE(T > 0 && d > 0 && d % 2 == 0, "acmi_pos_table: bad shape T=%d d=%d", T, d);
    hipLaunchKernelGGL(pos_table_kernel, dim3(T), dim3(256), 0, (hipStream_t)stream, freq, table, d);
    return acmi_check_launch("pos_table_kernel");
}

// =====================================================================================================
// CFG + sampling + pattern write-back
// =====================================================================================================

#define ACMI_MAX_CARD 4096

struct SampleArgs {
    const float* logits;  // [Beff, K*card]
    int B, K, card, use_cfg;
    float cfg_coef;
    int use_sampling; float temp; int top_k; float top_p;
    uint64_t seed; uint64_t step; int* pos;  // step counter = step + pos[0] when pos != NULL
    int advance;          // last block to finish does pos[0] += 1 (pos[1] is the ticket counter)
    int64_t* tokens_out;  // [B, K] or NULL
    float* mixed_out;     // [B, K, card] or NULL
    // write-back (NULL gen_sequence: skipped)
    int64_t* gen_sequence; const uint8_t* seq_mask; int S, P;
};

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// block-wide argmax with first-index tie break; all threads get the result
__device__ __forceinline__ int block_argmax(float v, int idx, float* sval, int* sidx) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(v, off, 64);
        const int oi = __shfl_xor(idx, off, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) { sval[wave] = v; sidx[wave] = idx; }
    __syncthreads();
    float bv = sval[0]; int bi = sidx[0];
    for (int w = 1; w < nw; ++w)
        if (sval[w] > bv || (sval[w] == bv && sidx[w] < bi)) { bv = sval[w]; bi = sidx[w]; }
    __syncthreads();
    return bi;
}

__device__ __forceinline__ float block_max(float v, float* sval) {
    v = wave_max(v);
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) sval[wave] = v;
    __syncthreads();
    float r = sval[0];
    for (int w = 1; w < nw; ++w) r = fmaxf(r, sval[w]);
    __syncthreads();
    return r;
}
__device__ __forceinline__ float block_sum(float v, float* sval) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) sval[wave] = v;
    __syncthreads();
    float r = 0.f;
    for (int w = 0; w < nw; ++w) r += sval[w];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void sample_kernel(const SampleArgs p) {
    __shared__ float vals[ACMI_MAX_CARD];
    __shared__ float sval[4];
    __shared__ int sidx[4];
    __shared__ unsigned hist[256];
    __shared__ unsigned sel[2];
    __shared__ unsigned wtot[4];
    const int k = blockIdx.x, b = blockIdx.y;
    const int card = p.card;
    const float* cond = p.logits + ((size_t)b * p.K + k) * card;
    const float* uncond = p.logits + ((size_t)(p.B + b) * p.K + k) * card;
    const int gpos = p.pos ? p.pos[0] : 0;
    for (int i = threadIdx.x; i < card; i += blockDim.x) {
        float v = cond[i];
        if (p.use_cfg) {  // uncond + (cond - uncond) * coef, rounded after every op like the reference (no fma)
            const float u = uncond[i];
            v = __fadd_rn(u, __fmul_rn(__fsub_rn(v, u), p.cfg_coef));
        }
        vals[i] = v;
        if (p.mixed_out) p.mixed_out[((size_t)b * p.K + k) * card + i] = v;
    }
    __syncthreads();
    int token;
    if (!(p.use_sampling && p.temp > 0.f)) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int i = threadIdx.x; i < card; i += blockDim.x)
            if (vals[i] > bv) { bv = vals[i]; bi = i; }
        token = block_argmax(bv, bi, sval, sidx);
    } else {
        // softmax(logits / temp)
        float mx = -INFINITY;
        for (int i = threadIdx.x; i < card; i += blockDim.x) { vals[i] = vals[i] / p.temp; mx = fmaxf(mx, vals[i]); }
        mx = block_max(mx, sval);
        float sum = 0.f;
        for (int i = threadIdx.x; i < card; i += blockDim.x) { const float e = expf(vals[i] - mx); vals[i] = e; sum += e; }
        sum = block_sum(sum, sval);
        for (int i = threadIdx.x; i < card; i += blockDim.x) vals[i] = vals[i] / sum;
        __syncthreads();
        float thr = 0.f;
        if (p.top_p > 0.f) {
            // keep i unless the probability mass sorted strictly before it exceeds top_p (utils.py:125-141)
            float keepflag[ACMI_MAX_CARD / 256];
            int cnt = 0;
            for (int i = threadIdx.x; i < card; i += blockDim.x, ++cnt) {
                const float pi = vals[i];
                float before = 0.f;
                for (int j = 0; j < card; ++j) {
                    const float pj = vals[j];
                    before += (pj > pi || (pj == pi && j < i)) ? pj : 0.f;
                }
                keepflag[cnt] = before > p.top_p ? 0.f : 1.f;
            }
            __syncthreads();
            cnt = 0;
            for (int i = threadIdx.x; i < card; i += blockDim.x, ++cnt) vals[i] *= keepflag[cnt];
            __syncthreads();
        } else if (p.top_k > 0 && p.top_k < card) {
            // k-th largest probability by 4-pass radix select on the (non-negative) float bit patterns
            unsigned prefix = 0u, mask = 0u, remaining = (unsigned)p.top_k;
            for (int pass = 3; pass >= 0; --pass) {
                for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0u;
                __syncthreads();
                for (int i = threadIdx.x; i < card; i += blockDim.x) {
                    const unsigned bits = __float_as_uint(vals[i]);
                    if ((bits & mask) == prefix) atomicAdd(&hist[(bits >> (8 * pass)) & 255u], 1u);
                }
                __syncthreads();
                {   // suffix sums S[t] = sum_{j >= t} hist[j] with 256 threads; digit d is selected when
                    // S[d] >= remaining > S[d+1]
                    const unsigned hcnt = hist[threadIdx.x];
                    unsigned sfx = hcnt;
                    const int ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) {
                        const unsigned o = __shfl_down(sfx, off, 64);
                        if (ln + off < 64) sfx += o;
                    }
                    if (ln == 0) wtot[wv] = sfx;
                    __syncthreads();
                    for (int w2 = wv + 1; w2 < 4; ++w2) sfx += wtot[w2];
                    if (sfx >= remaining && sfx - hcnt < remaining) { sel[0] = threadIdx.x; sel[1] = remaining - (sfx - hcnt); }
                }
                __syncthreads();
                prefix |= sel[0] << (8 * pass);
                mask |= 255u << (8 * pass);
                remaining = sel[1];
                __syncthreads();
            }
            thr = __uint_as_float(prefix);
        }
        // multinomial(num_samples=1) as an exponential race: argmax_i p_i / q_i, q_i ~ Exp(1)
        const uint64_t stepc = p.step + (uint64_t)gpos;
        float bv = -1.f; int bi = 0x7fffffff;
        for (int i = threadIdx.x; i < card; i += blockDim.x) {
            const float pi = vals[i];
            if (pi >= thr && pi > 0.f) {
                uint32_t c4[4] = {(uint32_t)i, (uint32_t)(b * p.K + k), (uint32_t)stepc, (uint32_t)(stepc >> 32)};
                philox4x32_10(c4, (uint32_t)p.seed, (uint32_t)(p.seed >> 32));
                const float u = ((float)(c4[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
                const float r = pi / (-logf(u));
                if (r > bv) { bv = r; bi = i; }
            }
        }
        token = block_argmax(bv, bi, sval, sidx);
    }
    if (threadIdx.x == 0) {
        if (p.tokens_out) p.tokens_out[(size_t)b * p.K + k] = token;
        if (p.gen_sequence) {
            const int offset = (gpos - p.P) + 1;  // sequence step being filled (lm.py:540-562)
            if (offset >= 0 && offset < p.S) {
                const int64_t tok = p.seq_mask[(size_t)k * p.S + offset] ? (int64_t)token : (int64_t)p.card;
                int64_t* dst = p.gen_sequence + ((size_t)b * p.K + k) * p.S + offset;
                if (*dst == -1) *dst = tok;
            }
        }
        if (p.advance) {  // every block has read pos[0] before taking its ticket
            __threadfence();
            const int ticket = atomicAdd(&p.pos[1], 1);
            if (ticket == (int)(gridDim.x * gridDim.y) - 1) {
                p.pos[1] = 0;
                p.pos[0] = gpos + 1;
            }
        }
    }
}

static int launch_sample(const SampleArgs& a, hipStream_t st) {
    ACMI_REQUIRE(a.card > 0 && a.card <= ACMI_MAX_CARD, "acmi_sample: card=%d unsupported (max %d)", a.card, ACMI_MAX_CARD);
    ACMI_REQUIRE(a.B > 0 && a.K > 0, "acmi_sample: bad shape");
    hipLaunchKernelGGL(sample_kernel, dim3(a.K, a.B), dim3(256), 0, st, a);
    return acmi_check_launch("sample_kernel");
}

extern "C" int acmi_sample(const float* logits, int64_t* tokens_out, float* mixed_out, int B, int K, int card,
                           int use_cfg, float cfg_coef, int use_sampling, float temp, int top_k, float top_p,
                           uint64_t seed, uint64_t step, void* stream) {
    SampleArgs a = {};
    a.logits = logits; a.B = B; a.K = K; a.card = card; a.use_cfg = use_cfg; a.cfg_coef = cfg_coef;
    a.use_sampling = use_sampling; a.temp = temp; a.top_k = top_k; a.top_p = top_p; a.seed = seed; a.step = step;
    a.tokens_out = tokens_out; a.mixed_out = mixed_out;
    return launch_sample(a, (hipStream_t)stream);
}

__global__ void advance_kernel(int* pos, int n) { if (threadIdx.x == 0 && blockIdx.x == 0) pos[0] += n; }

// =====================================================================================================
// one decode position
// =====================================================================================================

// LayerNorm in front of a GEMM, two interchangeable forms (ACMI_LN_MODE = fold | tile):
//   tile   a separate standardisation kernel writes the tiled `xn` (one extra launch per LayerNorm);
//   fold   the producers of x also emit it raw in fragment order (hi / lo), the consumer runs the plain tiled
//          GEMM on it and applies rstd * (acc - mean * colsum) in its epilogue from the (mean, M2) partials the
//          producer wrote next to x (no launch, no staging).  Default.
// (A third form -- the consumer staging the row-major x through LDS and standardising it there -- measured
// no faster than `tile` and was removed.)
enum { LN_TILE = 0, LN_FOLD = 2 };
enum { LN_X = 100 };  // step_lin: the activation is LayerNorm(x)
static int ln_mode_of(const acmi_lm_model* m, const acmi_lm_state* s) {
    static int want = -1;
    if (want < 0) {
        const char* e = getenv("ACMI_LN_MODE");
        want = (e && e[0] == 't') ? LN_TILE : LN_FOLD;
    }
    if (want == LN_FOLD) {
        const bool have = m->cs_head != nullptr && m->layers[0].cs_qkv != nullptr && m->layers[0].cs_ff1 != nullptr &&
                          (m->wdtype != ACMI_BF16 || s->xlo != nullptr) && m->dim / 16 <= 128 && m->dim % 16 == 0;
        return have ? LN_FOLD : LN_TILE;
    }
    return LN_TILE;
}
static bool fold_uses_lo() {  // experiment switch: ACMI_LN_LO=0 drops the low part of the bf16 pair
    static int v = -1;
    if (v < 0) { const char* e = getenv("ACMI_LN_LO"); v = (e && e[0] == '0') ? 0 : 1; }
    return v == 1;
}

// ---- the step's GEMM chain ---------------------------------------------------------------------------
// The residual stream x lives in four forms (include/acmi.h, acmi_lm_state): f32 row-major `x`, raw fragments
// xh / xl (hi / lo) and the statistics partials.  StepCtx tracks which fragment buffers currently hold x.
struct StepCtx {
    const acmi_lm_model* m; const acmi_lm_state* s; hipStream_t st;
    int lnm, kt, nkc_d, rbs;   // LayerNorm mode, K tile, K tiles of d, K tiles per row block of the xh buffers
    void* xh; void* xl;        // fragments of the current x
    int np, cnt;               // statistics partials of the current x: np partials of cnt elements
    int rows;                  // rows of this call (Beff x positions)
};

// out = act(LayerNorm(x) W'^T + bias): folded into the GEMM, or standardisation kernel + plain GEMM
static int gemm_ln_x(StepCtx& c, LinArgs& p, const void* w, const float* bias, const float* colsum, int N) {
    const acmi_lm_model* m = c.m; const acmi_lm_state* s = c.s;
    p.a_tiled = 1; p.w = w; p.bias = bias; p.M = c.rows; p.N = N; p.K = m->dim;
    if (c.lnm == LN_FOLD) {
        p.a = c.xh; p.a_rbs = c.rbs; p.a_lo = (m->wdtype == ACMI_BF16 && fold_uses_lo()) ? c.xl : nullptr;
        p.a_stats = s->stats; p.a_np = c.np; p.a_cnt = c.cnt; p.eps = m->eps; p.colsum = colsum;
    } else {
        int rc = launch_ln_tile(s->x, c.xh, m->wdtype, c.rows, m->dim, m->eps, nullptr, 0, c.st);
        if (rc) return rc;
        p.a = c.xh;
    }
    return launch_lin(p, m->wdtype, c.st);
}

// describes x <- x + a W^T (in place on the f32 copy), also emitted as fragments into (xh, xl) + statistics
static void gemm_produce_x_args(StepCtx& c, LinArgs& p, const void* a, int a_rbs, const void* w, int K, void* xh, void* xl) {
    const acmi_lm_model* m = c.m; const acmi_lm_state* s = c.s;
    p.a = a; p.a_tiled = 1; p.a_rbs = a_rbs; p.w = w; p.residual = s->x; p.out = s->x; p.out_mode = ACMI_OUT_F32;
    p.M = c.rows; p.N = m->dim; p.K = K;
    if (c.lnm == LN_FOLD) {
        p.stats_out = s->stats;
        p.xt_hi = xh; p.xt_lo = m->wdtype == ACMI_BF16 ? xl : nullptr; p.xt_nkc = c.rbs; p.xt_lo_nkc = c.nkc_d;
    }
}
static int gemm_produce_x(StepCtx& c, const void* a, const void* w, int K) {
    LinArgs p = {};
    gemm_produce_x_args(c, p, a, 0, w, K, c.xh, c.xl);
    int rc = launch_lin(p, c.m->wdtype, c.st);
    c.np = c.m->dim / 16; c.cnt = 16;
    return rc;
}

extern "C" int acmi_lm_step(const acmi_lm_model* m, const acmi_lm_state* s, int mode, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    ACMI_REQUIRE(m && s, "acmi_lm_step: null argument");
    // rows of this call: Beff per position; a PREFILL call may run several consecutive positions at once (row
    // p * Beff + b = position pos[0] + p of CFG row b): K / V of all of them are in the cache before any attends
    const int npos = (mode == ACMI_STEP_PREFILL && s->n_pos > 1) ? s->n_pos : 1;
    const int d = m->dim, H = m->num_heads, hd = d / H, M = s->Beff * npos, F = m->ffn_dim;
    ACMI_REQUIRE(d % H == 0 && d % 16 == 0 && d <= 2048 && F % 4 == 0, "acmi_lm_step: bad dims d=%d H=%d", d, H);
    ACMI_REQUIRE(m->n_q <= 16, "acmi_lm_step: n_q=%d > 16", m->n_q);
    ACMI_REQUIRE(s->Beff == (s->use_cfg ? 2 * s->B : s->B), "acmi_lm_step: Beff/B mismatch");
    ACMI_REQUIRE(mode == ACMI_STEP_PREFILL || s->n_pos <= 1, "acmi_lm_step: n_pos=%d only with ACMI_STEP_PREFILL", s->n_pos);
    const int wbf = m->wdtype == ACMI_BF16, kvbf = m->kvdtype == ACMI_BF16;
    int rc;

    StepCtx c = {};
    c.m = m; c.s = s; c.st = st; c.lnm = ln_mode_of(m, s);
    c.kt = wbf ? 32 : 16; c.nkc_d = (d + c.kt - 1) / c.kt;
    c.rbs = s->x_rbs > 0 ? s->x_rbs : c.nkc_d;
    ACMI_REQUIRE(c.rbs >= c.nkc_d, "acmi_lm_step: x_rbs=%d < %d K tiles of d", c.rbs, c.nkc_d);
    c.xh = s->xn; c.xl = s->xlo; c.np = 1; c.cnt = d; c.rows = M;
    // Cross-attention query without a launch of its own (include/acmi.h, acmi_linear_pair): needs the folded
    // LayerNorm, the [W_cq' | W_cq' W_out] matrices, xh buffers wide enough for [x | att] and a second pair.
    static const bool pair_enabled = !(getenv("ACMI_CROSS_FUSED") != nullptr && getenv("ACMI_CROSS_FUSED")[0] == '0');
    const bool pair = pair_enabled && m->cross_attention && c.lnm == LN_FOLD && m->layers[0].w_xcq != nullptr &&
                      s->xn2 != nullptr && (!wbf || s->xlo2 != nullptr) && s->r != nullptr && c.rbs >= 2 * c.nkc_d;
    void* const xh2[2] = {s->xn, s->xn2};
    void* const xl2[2] = {s->xlo, s->xlo2};
    int cur = 0;

    EmbedArgs e = {};
    for (int k = 0; k < m->n_q; ++k) e.emb[k] = m->emb[k];
    e.w_bf16 = wbf; e.gen_sequence = s->gen_sequence; e.B = s->B; e.Beff = s->Beff; e.K = m->n_q; e.S = s->S; e.card = m->card;
    e.prepend = s->prepend; e.P = s->prepend ? s->n_prepend : 0; e.pos_table = m->pos_table;
    e.pos_scale = m->positional_scale; e.pos = s->pos; e.x = s->x; e.d = d; e.stats = s->stats;
    if (c.lnm == LN_FOLD) { e.xt_hi = c.xh; e.xt_lo = wbf ? c.xl : nullptr; e.xt_nkc = c.rbs; e.xt_lo_nkc = c.nkc_d; }
    hipLaunchKernelGGL(embed_kernel, dim3(M), dim3(256), 0, st, e);
    if ((rc = acmi_check_launch("embed_kernel"))) return rc;

    for (int li = 0; li < m->num_layers; ++li) {
        const acmi_lm_layer& L = m->layers[li];
        // norm1 -> QKV ; K, V appended in place at position g, q to scratch
        {
            LinArgs a = {};
            a.qkv = 1; a.q_out = s->q; a.k_cache = L.k_cache; a.v_cache = L.v_cache; a.kv_bf16 = kvbf;
            a.H = H; a.hd = hd; a.Tcap = s->Tmax; a.d = d; a.pos = s->pos; a.rpp = s->Beff;
            if ((rc = gemm_ln_x(c, a, L.w_qkv, L.b_qkv, L.cs_qkv, 3 * d))) return rc;
        }
        // self attention over positions [0, g]; output in A-fragment order for the out projection: into `att`,
        // or next to x ([x | att], columns d_pad ..) when the out projection is paired with the cross query
        acmi_attn_desc sa = {};
        sa.q = s->q; sa.k_cache = L.k_cache; sa.v_cache = L.v_cache; sa.kvdtype = m->kvdtype;
        sa.out_mode = ACMI_OUT_TILED; sa.out_dtype = m->wdtype; sa.Beff = M; sa.H = H; sa.hd = hd; sa.Tcap = s->Tmax;
        sa.len_dev = s->pos; sa.len_bias = 1; sa.cache_rows = s->Beff;
        if (pair) { sa.out = c.xh; sa.out_rbs = c.rbs; sa.out_col0 = c.nkc_d * c.kt; }
        else sa.out = s->att;
        if ((rc = acmi_attn_decode_ex(&sa, stream))) return rc;

        if (!m->cross_attention) {
            if ((rc = gemm_produce_x(c, s->att, L.w_out, d))) return rc;
        } else {
            ACMI_REQUIRE(s->Lc > 0 && L.ck_cache && L.cv_cache, "acmi_lm_step: cross-attention caches missing");
            acmi_attn_desc ca = {};
            ca.k_cache = L.ck_cache; ca.v_cache = L.cv_cache; ca.kvdtype = m->kvdtype; ca.out = s->att;
            ca.out_mode = ACMI_OUT_TILED; ca.out_dtype = m->wdtype; ca.Beff = M; ca.H = H; ca.hd = hd; ca.Tcap = s->Lc;
            ca.len = s->Lc; ca.cache_rows = s->Beff;
            if (pair) {
                // ONE launch: x1 = x0 + att W_out^T (fragments of x1 into the other buffer pair: this launch
                // still reads x0's) and r = [x0 | att] [W_cq' | W_cq' W_out]^T = x1 W_cq'^T
                LinArgs p0 = {}, p1 = {};
                const void* att_half = reinterpret_cast<const unsigned char*>(c.xh) + (size_t)c.nkc_d * 1024;  // K tile nkc_d
                gemm_produce_x_args(c, p0, att_half, c.rbs, L.w_out, d, xh2[cur ^ 1], xl2[cur ^ 1]);
                p1.a = c.xh; p1.a_tiled = 1; p1.a_rbs = c.rbs; p1.a_lo = wbf ? c.xl : nullptr; p1.alo_rbs = c.nkc_d;
                p1.lo_split = wbf ? c.nkc_d : 0;
                p1.w = L.w_xcq; p1.out = s->r; p1.out_mode = ACMI_OUT_F32; p1.M = M; p1.N = d; p1.K = 2 * c.nkc_d * c.kt;
                if ((rc = wbf ? launch_pair<bf16_t>(p0, p1, st) : launch_pair<float>(p0, p1, st))) return rc;
                cur ^= 1; c.xh = xh2[cur]; c.xl = xl2[cur]; c.np = d / 16; c.cnt = 16;
                // the cross-attention kernel applies norm_cross to r from the statistics of x1
                ca.q = s->r; ca.q_stats = s->stats; ca.q_stats_np = c.np; ca.q_stats_cnt = c.cnt; ca.eps = m->eps;
                ca.q_colsum = L.cs_cq; ca.q_bias = L.b_cq;
            } else {
                if ((rc = gemm_produce_x(c, s->att, L.w_out, d))) return rc;
                LinArgs a = {};
                a.out = s->q; a.out_mode = ACMI_OUT_F32;
                if ((rc = gemm_ln_x(c, a, L.w_cq, L.b_cq, L.cs_cq, d))) return rc;
                ca.q = s->q;
            }
            if ((rc = acmi_attn_decode_ex(&ca, stream))) return rc;
            if ((rc = gemm_produce_x(c, s->att, L.w_cout, d))) return rc;
        }
        {   // norm2 -> linear1 + GELU -> hidden (A-fragment order) ; linear2 -> x
            LinArgs a = {};
            a.out = s->hidden; a.out_mode = ACMI_OUT_TILED; a.act = 1;
            if ((rc = gemm_ln_x(c, a, L.w_ff1, L.b_ff1, L.cs_ff1, F))) return rc;
            if ((rc = gemm_produce_x(c, s->hidden, L.w_ff2, F))) return rc;
        }
    }
    if (mode == ACMI_STEP_DECODE) {
        LinArgs hl = {};
        hl.out = s->logits; hl.out_mode = ACMI_OUT_F32;
        if ((rc = gemm_ln_x(c, hl, m->w_head, m->b_head, m->cs_head, m->n_q * m->card))) return rc;
        SampleArgs a = {};
        a.logits = s->logits; a.B = s->B; a.K = m->n_q; a.card = m->card; a.use_cfg = s->use_cfg;
        a.cfg_coef = s->cfg_coef; a.use_sampling = s->use_sampling; a.temp = s->temp; a.top_k = s->top_k;
        a.top_p = s->top_p; a.seed = s->seed; a.step = 0; a.pos = s->pos; a.advance = 1; a.mixed_out = s->step_logits;
        a.gen_sequence = s->gen_sequence; a.seq_mask = s->seq_mask; a.S = s->S; a.P = s->prepend ? s->n_prepend : 0;
        return launch_sample(a, st);
    }
    hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(64), 0, st, s->pos, npos);
    return acmi_check_launch("advance_kernel");
}
