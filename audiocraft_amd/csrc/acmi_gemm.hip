// Skinny GEMMs of the MusicGen LM decode step for gfx950 (CDNA4, wave64): lin_tiled_kernel / lin_pair_kernel
// (tiled activation x tiled weight, LayerNorm folded into the epilogue), lin_rowmajor_kernel (row-major f32
// activation staged through LDS) and ln_tile_kernel (LayerNorm as a kernel of its own).
// Reference semantics: audiocraft/modules/transformer.py:315-451, 550-574 (every F.linear / nn.LayerNorm of a
// layer), audiocraft/models/lm.py:260-262 (out_norm + heads).
#include "acmi_lm_internal.h"

#include <math.h>
#include <stdlib.h>
#include <atomic>

// Two translation units, one source: the f32-weight instantiations of the launchers below (half of the kernels of this
// file) are compiled by acmi_gemm_f32.hip, which defines ACMI_GEMM_F32_TU and includes this file; everything that is not
// a template on the weight type -- the C entry points, the trace bookkeeping -- is compiled here only.
#ifdef ACMI_GEMM_F32_TU
#define ACMI_GEMM_MAIN 0
#else
#define ACMI_GEMM_MAIN 1
#endif

// =====================================================================================================
// skinny GEMM   out[M,N] = act(LN?(a)[M,K] @ W[N,K]^T + bias) + residual
//
// One 16-feature n-tile per workgroup, K split across its waves.  Weights are stored as 1 KB MFMA
// B-fragments (include/acmi.h "tiled weight"), so each fragment is ONE fully coalesced non-temporal
// 64 x 16 B load.  Activations produced on the path (residual stream as raw hi / lo fragments, attention
// output, FFN hidden) arrive already in A-fragment order and are loaded like the weights
// (lin_tiled_kernel); row-major f32 activations are staged through LDS once per workgroup -- one wave per
// row, which is also where an explicit LayerNorm runs -- and read back with ds_read_b128
// (lin_rowmajor_kernel).  Cross-wave reduction through LDS in a fixed order (deterministic), then the
// fused epilogue.
// =====================================================================================================

#ifdef ACMI_EXP_PLAINW
__device__ __forceinline__ u32x4 ld_frag_nt(const u32x4* p) { return *p; }
#else
__device__ __forceinline__ u32x4 ld_frag_nt(const u32x4* p) { return __builtin_nontemporal_load(p); }
#endif

__device__ __forceinline__ void mma_frag(const u32x4& a, const u32x4& b, f32x4& acc, bf16_t) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_frag(const u32x4& a, const u32x4& b, f32x4& acc, float) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[e]), __uint_as_float(b[e]), acc, 0, 0, 0);
}

__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

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
__global__ __launch_bounds__(256) void ln_tile_kernel(float* __restrict__ x, WT* __restrict__ out, int M, int K, int nkc,
                                                      float eps, const float* __restrict__ slabs, int nslabs) {
    // four rows per workgroup, one wave each (no LDS, no barrier): a one-wave workgroup per row left most of a CU's wave
    // slots empty in the prefill's 9 600-row launches (3.0 TB/s)
    const int m = blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
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
    // eps < 0: no standardisation, the rows go into fragment order as they are (post-norm layers, acmi_lm_model.post_norm:
    // the GEMMs there consume x itself; (v - 0) * 1 is exact)
    float mean = 0.f, rstd = 1.0f;
    if (eps >= 0.f) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < ACMI_STAGE_JMAX; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        mean = wave_sum(s) / (float)K;
        float s2 = 0.f;
#pragma unroll
        for (int j = 0; j < ACMI_STAGE_JMAX; ++j) {
            if ((lane + 64 * j) * 4 < K) {
                const float a0 = v[j].x - mean, a1 = v[j].y - mean, a2 = v[j].z - mean, a3 = v[j].w - mean;
                s2 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
        }
        rstd = 1.0f / sqrtf(wave_sum(s2) / (float)K + eps);
    }
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

#if ACMI_GEMM_MAIN
int acmi_launch_ln_tile(float* x, void* out, int wdtype, int M, int K, float eps, const float* slabs, int nslabs,
                        hipStream_t st) {
    ACMI_REQUIRE(M > 0 && K > 0 && K % 4 == 0 && K <= 2048, "acmi_ln_tile: needs K %% 4 == 0 and K <= 2048 (K=%d)", K);
    if (wdtype == ACMI_BF16)
        hipLaunchKernelGGL(ln_tile_kernel<bf16_t>, dim3((M + 3) / 4), dim3(256), 0, st, x, reinterpret_cast<bf16_t*>(out), M, K,
                           (K + 31) / 32, eps, slabs, nslabs);
    else
        hipLaunchKernelGGL(ln_tile_kernel<float>, dim3((M + 3) / 4), dim3(256), 0, st, x, reinterpret_cast<float*>(out), M, K,
                           (K + 15) / 16, eps, slabs, nslabs);
    return acmi_check_launch("ln_tile_kernel");
}

extern "C" int acmi_ln_tile_reduce(float* x, const float* slabs, int nslabs, void* out, int wdtype, int M, int K, float eps,
                                   void* stream) {
    ACMI_REQUIRE(nslabs >= 0 && (nslabs == 0 || slabs != nullptr), "acmi_ln_tile_reduce: bad slabs");
    return acmi_launch_ln_tile(x, out, wdtype, M, K, eps, slabs, nslabs, (hipStream_t)stream);
}

extern "C" int acmi_ln_tile(const float* x, void* out, int wdtype, int M, int K, float eps, void* stream) {
    return acmi_launch_ln_tile(const_cast<float*>(x), out, wdtype, M, K, eps, nullptr, 0, (hipStream_t)stream);
}
#endif  // ACMI_GEMM_MAIN

// Folded LayerNorm, row statistics: a "group" is 4 rows (16 lanes each); every lane fetches up to NS (8, or 16 when the
// producer ran with 8-feature workgroups) of the producer's equal-count (mean, M2) partials of its row (np <= 16 NS),
// combined later with Chan's formula.
// Layout stats[row][np][2]: the partials of a row are contiguous, so a wave's load touches 4 lines, not 64
// (with [np][row][2] the gather cost ~2 us per consuming launch).
template <int NS>
__device__ __forceinline__ void rowstat_load(const float* __restrict__ stats, int np, int M, int row0, int lane,
                                             float (&pm)[16], float (&pq)[16]) {
    const int row = min(row0 + (lane >> 4), M - 1), jj = lane & 15;
#pragma unroll
    for (int i = 0; i < NS; ++i) {  // 16 consecutive partials of one row per 16 lanes: one 128-B line
        const float2 t = *reinterpret_cast<const float2*>(stats + ((size_t)row * np + min(jj + 16 * i, np - 1)) * 2);
        pm[i] = t.x; pq[i] = t.y;
    }
}
// dst gets (mean - shift, rstd): `shift` is what the row's raw fragments were stored with (0 without one), so the
// epilogue's rstd * (acc - dst[0] * colsum) is the LayerNorm of the unshifted row.  mean_dst (or NULL): the plain mean.
template <int NS>
__device__ __forceinline__ void rowstat_finish(const float (&pm)[16], const float (&pq)[16], int np, int cnt, int K, float eps,
                                               int lane, float* __restrict__ dst /* [4][2] */, float shift = 0.f,
                                               float* __restrict__ mean_dst = nullptr) {
    const int jj = lane & 15;
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < NS; ++i) sm += (jj + 16 * i < np) ? pm[i] : 0.f;
    sm = row16_sum(sm);
    const float mean = sm / (float)np;
    float q2 = 0.f;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const float dlt = pm[i] - mean;
        q2 += (jj + 16 * i < np) ? pq[i] + (float)cnt * dlt * dlt : 0.f;
    }
    q2 = row16_sum(q2);
    if (jj == 0) {
        dst[(lane >> 4) * 2] = mean - shift;
        dst[(lane >> 4) * 2 + 1] = 1.0f / sqrtf(q2 / (float)K + eps);
        if (mean_dst != nullptr) *mean_dst = mean;
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
int launch_rowmajor(LinArgs& a, hipStream_t st) {
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
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&lin_rowmajor_kernel<WT>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;   // once, thread safe
    if (!attr_ok) {
        acmi_set_error("acmi_linear: cannot raise the dynamic LDS limit");
        return ACMI_ELAUNCH;
    }
    hipLaunchKernelGGL((lin_rowmajor_kernel<WT>), dim3((a.N + 15) / 16), dim3(nw * 64), lds, st, a);
    return acmi_check_launch("lin_rowmajor_kernel");
}

// =====================================================================================================
// lin_tiled_kernel: the decode step's GEMM  (tiled activation x tiled weight)
// =====================================================================================================
// One workgroup = 16 output features (x one K slice with split-K); its nw <= 8 waves own contiguous runs of K
// fragments.  Every wave requests everything it will ever need up front -- weight fragments, then activation
// fragments, then the LayerNorm row statistics, then the epilogue operands of its thread -- because vmcnt retires
// in order and anything requested later (a cold bias vector in the epilogue, say) is a full HBM round trip on the
// tail of the launch.  Absent operands are replaced by the address of the wave's own first weight fragment
// (already in flight: no extra line, page or hot spot), so the prologue is branch free.
//   LN 0: plain   1: folded LayerNorm, single-term activation, statistics from the producer's partials
//      2: folded LayerNorm, hi + lo activation   3: no LayerNorm, hi + lo activation for the first lo_split K
//      fragments (x | a concatenated along K)
//      4: folded LayerNorm, single-term activation, statistics FROM THE FRAGMENTS (round 4): two more MFMAs per
//         activation fragment a -- a x ones (row sums) and a x a^T (Gram matrix: its diagonal holds the rows' sums of
//         squares; the A and B operands of the 16x16 MFMA share one register layout, so `a` serves as both) -- give
//         mean and variance of exactly the values the GEMM multiplies, and the consumer loads no partials at all.
//         The in-kernel timeline (profiles/archive/r04_lin_timeline*.csv) priced the partials: 16 dwordx2 requests per lane
//         (24 KB per workgroup, freshly written by the previous launch) through the same per-CU address pipe as the
//         weight stream, plus Chan's combination in front of the barrier.  The fragments are bf16(x - shift) with the
//         shift near the row mean, so the one-pass variance sum(a^2) / K - mean^2 does not cancel.
struct TlExtras {
    float pm[16], pq[16];   // LN 1 / 2: (mean, M2) partials of this lane's statistics row (the first NS of them)
    float bias, colsum, res;  // epilogue operands of this thread's first output element
    int tpos;                 // QKV: the position the new K / V rows are stored at
    float sh, osh;            // shift of this lane's statistics row (consumer) / of this thread's first output row (producer)
};

// What a wave needs to put its weight, activation and statistics requests on their way: the kernels' leading scalar
// arguments.  With -mllvm -amdgpu-kernarg-preload-count=14 (audiocraft_amd/build.py) the command processor hands them
// over in SGPRs at wave start, so a wave's first HBM request is ~30 instructions from its entry.  Everything else -- the
// LinArgs block, by value BEHIND them in the kernarg segment -- is read with scalar loads issued after the weight
// requests (tl_load_args): a decode position is ~340 launches whose kernarg blocks are each read once per replay,
// ~3 GB of traffic after their previous use, i.e. cold in the scalar cache and in L2, and hipcc's own prologue fetched
// the 320-byte block in three DEPENDENT s_load rounds (SGPR pressure) in front of the first weight request of every
// wave (lab/kernarg_lab.hip prices a round; same-box A/B of the two forms: 6.98 -> 6.21 us per launch).
struct TlHot {
    const u32x4* w; const u32x4* a; const float* a_stats; const float* a_shift;
    int NKC, kcs, fpw, nw, ksp, a_rbs, M, a_np;
};
#define ACMI_AS4 __attribute__((address_space(4)))
#define ACMI_TL_ARGS_OFF 48        // byte offset of the LinArgs block in the kernarg segment of lin_tiled_kernel
#define ACMI_TL_ARGS_OFF_PAIR 56   // ... of the first of the two blocks of lin_pair_kernel
static_assert(alignof(LinArgs) == 8 && sizeof(LinArgs) % 8 == 0, "kernarg layout of lin_tiled_kernel / lin_pair_kernel");

__device__ __forceinline__ TlHot tl_unpack(const u32x4* w, const u32x4* a, const float* st, const float* sh, unsigned g0,
                                           unsigned g1, unsigned g2, unsigned g3) {
    TlHot h;
    h.w = w; h.a = a; h.a_stats = st; h.a_shift = sh;
    h.NKC = (int)(g0 & 0xffffu); h.kcs = (int)(g0 >> 16);
    h.fpw = (int)(g1 & 0xfffu); h.nw = (int)((g1 >> 12) & 0xfu); h.ksp = (int)(g1 >> 16);
    h.a_rbs = (int)(g2 & 0xffffu); h.M = (int)(g2 >> 16); h.a_np = (int)g3;
    return h;
}
// the LinArgs block at byte `aoff` of the kernarg segment, read where this is called (`z`: an opaque zero produced behind
// the weight requests -- the loads depend on it and cannot be hoisted in front of them)
__device__ __forceinline__ void tl_load_args(LinArgs& p, int aoff, int z) {
    const char ACMI_AS4* ka = (const char ACMI_AS4*)__builtin_amdgcn_kernarg_segment_ptr();
    const void ACMI_AS4* src = (const void ACMI_AS4*)__builtin_assume_aligned((const void ACMI_AS4*)(ka + (aoff + z)), 8);
    __builtin_memcpy(&p, src, sizeof(LinArgs));
}
// one word of every 64-byte line of that block: later scalar loads of its fields (the epilogue's) hit the scalar cache
__device__ __forceinline__ void tl_touch_args(int aoff, int z) {
    const char ACMI_AS4* ka = (const char ACMI_AS4*)__builtin_amdgcn_kernarg_segment_ptr();
    const int ACMI_AS4* src = (const int ACMI_AS4*)__builtin_assume_aligned((const void ACMI_AS4*)(ka + (aoff + z)), 8);
    int acc = 0;
#pragma unroll
    for (int i = 0; i < (int)sizeof(LinArgs) / 4; i += 16) acc |= src[i];
    acc |= src[(int)sizeof(LinArgs) / 4 - 1];
    asm volatile("" :: "s"(acc));
}

// addresses of a chunk's activation / statistics requests (from TlHot, or from LinArgs for the lo term) and of the
// epilogue operands (from LinArgs); evaluated AFTER the weight requests are out (see tl_chunk)
struct TlLate {
    const u32x4* at; const u32x4* al; int mts, mtl, mtv;
    const float* st_ptr; const float* psh;
};
struct TlEpi { const float* pb; const float* pc; const float* pr; const int* ppos; const float* posh; };

// stamps of the in-kernel timeline (acmi_lm_internal.h); empty in the production build
struct TlTrace {
#ifdef ACMI_TRACE
    unsigned long long t[ACMI_TRACE_NSTAMP];
#endif
};

template <typename WT> __device__ __forceinline__ u32x4 ones_frag();
template <> __device__ __forceinline__ u32x4 ones_frag<bf16_t>() { return u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}; }
template <> __device__ __forceinline__ u32x4 ones_frag<float>() { return u32x4{0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u}; }

template <typename WT, int MT, int LN, int NT, int NS, int C, typename LateFn, typename EpiFn>
__device__ __forceinline__ void tl_chunk(const TlHot& h, LinArgs& p, const int aoff, const u32x4* __restrict__ wt, int kc0,
                                         int np, const LateFn& late_fn, const EpiFn& epi_fn, f32x4 (&acc)[NT * MT],
                                         f32x4 (&accx)[2 * MT], TlExtras& ex, TlTrace& tr) {
    constexpr bool HL = LN == 2 || LN == 3;
    constexpr bool PART = LN == 1 || LN == 2;   // statistics from the producer's partials
    const int lane = threadIdx.x & 63;
    const int wts = h.NKC * 64;  // fragment lanes between the NT adjacent n-tiles of this workgroup
    u32x4 bv[NT][C], av[MT][C], lv[HL ? MT : 1][HL ? C : 1];
    // Request order: all weight fragments first -- they come from HBM, the activation fragments from L2, and the HBM
    // requests should be on their way as early as possible (FFN2 10.5 -> 9.9 us; whole position 2.57 -> 2.50 ms; with
    // (weight, activations) pairs in consumption order only the hi / lo variants in isolation were 0.1-0.2 us faster).
    // ACMI_TL_ORDER (experiment switch; round 4, same box: 5.61 us per launch for 0 against 5.76 / 5.77 / 5.77 for 1 / 2 / 4,
    // RTF 65.0 against 63.3 / 63.8 -- the activation's 0.5 us behind the last weight fragment is cheaper than any delay of
    // the HBM requests): 0 = all weight fragments, then all activation fragments; k > 0 = the first
    // C - C / k weight fragments, then the activation fragments interleaved with the remaining weight fragments, so that
    // the last activation request is out before the last weight request (the activation comes from L2: it is back by then)
#ifndef ACMI_TL_ORDER
#define ACMI_TL_ORDER 0
#endif
    constexpr int CW = (ACMI_TL_ORDER > 0 && LN != 2 && LN != 3) ? C - C / (ACMI_TL_ORDER > 0 ? ACMI_TL_ORDER : 1) : C;   // weight fragments requested up front
#pragma unroll
    for (int i = 0; i < CW; ++i) {
        const int ko = (kc0 + i) * 64;
#pragma unroll
        for (int t = 0; t < NT; ++t) bv[t][i] = ld_frag_nt(wt + (t * wts + ko) + lane);
    }
    // The weight requests need the preloaded arguments (TlHot) and ~20 instructions of address arithmetic; the ~100
    // instructions that set up the activation, statistics and epilogue operand addresses, and every scalar load of
    // the LinArgs block, sit BEHIND them.
    __builtin_amdgcn_sched_barrier(0);
    int opaque0 = 0;
    asm volatile("" : "+s"(opaque0));
    opaque0 = __builtin_amdgcn_readfirstlane(opaque0);   // (an asm result counts as divergent: keep the loads below scalar)
    tl_load_args(p, aoff, opaque0);
    const TlLate L = late_fn(opaque0);
    const u32x4* __restrict__ at = L.at;
    const u32x4* __restrict__ al = L.al;
    const int mts = L.mts, mtl = L.mtl, mtv = L.mtv;
    const float* st_ptr = L.st_ptr;
#pragma unroll
    for (int i = 0; i < C; ++i) {
        const int ko = (kc0 + i) * 64;  // wave-uniform; 32-bit index math (a matrix has < 2^31 fragments' lanes)
#pragma unroll
        for (int u = 0; u < MT; ++u) {  // row blocks beyond M re-read the last valid one (their results are dropped)
            const int ub = min(u, mtv - 1);
            av[u][i] = (at + (ub * mts + ko))[lane];
            if (LN == 2) lv[u][i] = (al + (ub * mtl + ko))[lane];
            if (LN == 3)  // fragments past lo_split have no lo term: re-read the last one (L1 hit), zeroed below
                lv[u][i] = (al + (ub * mtl + min(kc0 + i, p.lo_split - 1) * 64))[lane];
        }
        if constexpr (CW < C) {   // the held-back weight fragments, spread evenly behind the activation fragments
#pragma unroll
            for (int j = CW; j < C; ++j) {
                if ((j - CW) * C / (C - CW) == i) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) bv[t][j] = ld_frag_nt(wt + (t * wts + (kc0 + j) * 64) + lane);
                }
            }
        }
    }
    if (PART) {
        const int jj = (int)(threadIdx.x & 15);
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const float2 t = *reinterpret_cast<const float2*>(st_ptr + min(jj + 16 * i, np - 1) * 2);
            ex.pm[i] = t.x; ex.pq[i] = t.y;
        }
    }
    ex.sh = *L.psh;
    __builtin_amdgcn_sched_barrier(0);  // the requests above need no field of LinArgs (single-term activations): no wait so far
    {
        const TlEpi E = epi_fn(opaque0);   // first use of the block's fields: the scalar loads are waited for here
        ex.bias = *E.pb; ex.colsum = *E.pc; ex.res = *E.pr; ex.tpos = *E.ppos; ex.osh = *E.posh;
        tl_touch_args(aoff, opaque0);
    }
    __builtin_amdgcn_sched_barrier(0);  // keep every request in front of the first wait
#ifdef ACMI_TRACE
    {   // requests in flight, oldest first: NT C weight fragments, then MT C (x 2 with a lo term) activation fragments, the
        // statistics partials and the six epilogue operands
        // (interleaved order: the fragments of both kinds count as "weights", the trailing requests are the rest)
        constexpr bool IL = CW < C;
        constexpr int NWQ = NT * C + (IL ? MT * C : 0), NREST = (IL ? 0 : MT * C * (HL ? 2 : 1)) + (PART ? NS : 0) + 6;
        constexpr int ALL1 = NWQ + NREST - 1 > 63 ? 63 : NWQ + NREST - 1, REST = NREST > 63 ? 63 : NREST;
        ACMI_TR(tr.t, 1);
        ACMI_TR_WAIT_VM(ALL1); ACMI_TR(tr.t, 2);
        ACMI_TR_WAIT_VM(REST); ACMI_TR(tr.t, 3);
        ACMI_TR_WAIT_VM(0); ACMI_TR(tr.t, 4);
    }
#endif
    const u32x4 ones = ones_frag<WT>();
#pragma unroll
    for (int i = 0; i < C; ++i)
#pragma unroll
        for (int u = 0; u < MT; ++u) {
            if (LN == 3 && kc0 + i >= p.lo_split) lv[u][i] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                mma_frag(av[u][i], bv[t][i], acc[t * MT + u], WT());
                if (HL) mma_frag(lv[u][i], bv[t][i], acc[t * MT + u], WT());
            }
            if (LN == 4) {   // row sums and the Gram matrix of the fragment (its diagonal: the rows' sums of squares)
                mma_frag(av[u][i], ones, accx[2 * u], WT());
                mma_frag(av[u][i], av[u][i], accx[2 * u + 1], WT());
            }
        }
}

// Half-tile workgroups (HT): 8 output features per workgroup, for the narrow GEMM with the long K (FFN2: N = d gives
// only d / 16 = 96 workgroups of 16 features, each pulling 196 KB of weights through one CU's TA while 160 CUs idle).
// The weight comes in the "half-tile" order (include/acmi.h, acmi_linear_desc.w_half): a 1 KB unit holds 8 features x
// 2 KT columns, lane (kg, c) = feature c & 7, K fragment c >> 3 of the unit, so a unit is ONE full-width load and is the
// B operand of TWO MFMAs: with the A fragment 2u the accumulator columns 0-7 are feature sums, with A fragment 2u + 1
// the columns 8-15 are (the other columns hold products of mismatched K ranges and are dropped).  Two accumulators per
// row block, merged at the end by a rotation of 8 lanes.  No LayerNorm variants: the producers of x are plain GEMMs.
template <typename WT, int MT, int C, typename LateFn, typename EpiFn>
__device__ __forceinline__ void tl_chunk_ht(LinArgs& p, const int aoff, const u32x4* __restrict__ wt, int ku0, const LateFn& late_fn,
                                            const EpiFn& epi_fn, f32x4 (&acc)[2 * MT], TlExtras& ex, TlTrace& tr) {
    const int lane = threadIdx.x & 63;
    u32x4 bv[C], av[MT][2 * C];
    constexpr int CW = ACMI_TL_ORDER > 0 ? C - C / (ACMI_TL_ORDER > 0 ? ACMI_TL_ORDER : 1) : C;
#pragma unroll
    for (int i = 0; i < CW; ++i) bv[i] = ld_frag_nt(wt + (ku0 + i) * 64 + lane);
    __builtin_amdgcn_sched_barrier(0);
    int opaque0 = 0;
    asm volatile("" : "+s"(opaque0));
    opaque0 = __builtin_amdgcn_readfirstlane(opaque0);
    tl_load_args(p, aoff, opaque0);
    const TlLate L = late_fn(opaque0);
    const u32x4* __restrict__ at = L.at;
    const int mts = L.mts, mtv = L.mtv;
#pragma unroll
    for (int i = 0; i < 2 * C; ++i) {
        const int ko = (2 * ku0 + i) * 64;
#pragma unroll
        for (int u = 0; u < MT; ++u) av[u][i] = (at + (min(u, mtv - 1) * mts + ko))[lane];
        if constexpr (CW < C) {
#pragma unroll
            for (int j = CW; j < C; ++j)
                if ((j - CW) * 2 * C / (C - CW) == i) bv[j] = ld_frag_nt(wt + (ku0 + j) * 64 + lane);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        const TlEpi E = epi_fn(opaque0);
        ex.bias = *E.pb; ex.res = *E.pr; ex.osh = *E.posh;
        tl_touch_args(aoff, opaque0);
    }
    __builtin_amdgcn_sched_barrier(0);
#ifdef ACMI_TRACE
    {
        constexpr bool IL = CW < C;
        constexpr int NWQ = C + (IL ? MT * 2 * C : 0), NREST = (IL ? 0 : MT * 2 * C) + 3;
        constexpr int ALL1 = NWQ + NREST - 1 > 63 ? 63 : NWQ + NREST - 1, REST = NREST > 63 ? 63 : NREST;
        ACMI_TR(tr.t, 1);
        ACMI_TR_WAIT_VM(ALL1); ACMI_TR(tr.t, 2);
        ACMI_TR_WAIT_VM(REST); ACMI_TR(tr.t, 3);
        ACMI_TR_WAIT_VM(0); ACMI_TR(tr.t, 4);
    }
#endif
#pragma unroll
    for (int i = 0; i < C; ++i) {
#pragma unroll
        for (int u = 0; u < MT; ++u) {
            mma_frag(av[u][2 * i], bv[i], acc[2 * u], WT());
            mma_frag(av[u][2 * i + 1], bv[i], acc[2 * u + 1], WT());
        }
    }
}

// ---- the epilogue: one output element per thread and pass.
// EPI fixes, at compile time, what the flags of LinArgs would otherwise decide at run time; the launcher picks the
// specialisation the call's flags allow (tiled_epi) and the kernel jumps to it with ONE uniform branch.  The generic
// form (EPI_GEN) keeps every flag a run-time test; all forms are the same source, so their arithmetic is the same.
// Why: the timeline showed ~1.1-1.3 us between the barrier and the last store of EVERY launch -- ~200 executed
// instructions, but spread over ~30 taken branches of cold code, each a new instruction-cache line fetched from L2.
enum { EPI_GEN = 0, EPI_PRODX = 1, EPI_TILED = 2, EPI_F32 = 3, EPI_QKV = 4, EPI_QKVH = 5 };
//   EPI_PRODX  x <- x + a W^T (+ bias): f32 in place, the raw fragments of the new x (single term, optional shift),
//              optional statistics partials.  No activation, no LayerNorm, no split-K.
//   EPI_TILED  out = act(LN?(a) W^T + bias) in A-fragment order (FFN1)
//   EPI_F32    out = LN?(a) W^T + bias (+ residual), row-major f32 (heads, the paired cross-query GEMM)
//   EPI_QKV    the QKV scatter of a decode step (one position per call, head size and model width multiples of 16, so
//              that a 16-feature tile lies in ONE of q / k / v / r and in ONE head: all index divisions are per tile)
//   EPI_QKVH   the same features handed to the attention workgroups of the SAME launch (qkv_attn_kernel): q | k | v as f32 words
//              with write-through stores into the sentinel-armed hand-off row q_out [M][3 d] (acmi_attn_fused.h); r as in EPI_QKV

// the nw partial sums of one tile element, added in wave order; nw is 1, 2, 4 or 8 (tiled_waves): one uniform branch, then
// straight-line LDS reads (a run-time loop over nw costs a compare + branch per term and hides the reads from each other)
__device__ __forceinline__ float tl_red_sum(const float* __restrict__ base, const int nw) {   // base[w * 256], w < nw
    float v = 0.f;
    if (nw == 8) {
#pragma unroll
        for (int w = 0; w < 8; ++w) v += base[w * 256];
    } else if (nw == 4) {
#pragma unroll
        for (int w = 0; w < 4; ++w) v += base[w * 256];
    } else if (nw == 2) {
        v += base[0];
        v += base[256];
    } else {
        for (int w = 0; w < nw; ++w) v += base[w * 256];
    }
    return v;
}

// NW: the workgroup's wave count as a compile-time constant (8 / 4 / 2: what the decode step's launches run with), or 0 = run
// time.  With it the LDS index arithmetic folds, and a launch whose outputs fit one pass (MT = 1, 256 NT <= 64 NW) has no loop
// and no "later pass" operand loads: same-box, the run-time form of the same source costs +0.37 us per launch (5.98 vs 5.61).
template <typename WT, int MT, int LN, int NT, bool HT, int EPI, int NW>
__device__ __forceinline__ void tl_epilogue(const LinArgs& p, const TlExtras& ex, const float* __restrict__ red,
                                            const float* __restrict__ rowstat, const int nw_rt, const int ksp, const int kslice,
                                            const int mg, const int mtv, const int n0, const int ntile, const int wgtile) {
    const int nw = NW > 0 ? NW : nw_rt;
    constexpr bool ONE = NW > 0 && MT == 1 && 256 * NT <= 64 * NW;   // every thread owns at most one output element
    constexpr bool FOLD = LN == 1 || LN == 2 || LN == 4;
    constexpr bool GRAM = LN == 4;
    constexpr bool G = EPI == EPI_GEN;
    constexpr int XT = (HT ? 1 : NT) * MT;   // first extra tile (row sums / Gram) of the reduction buffer
    const bool split = G ? ksp > 1 : false;
    const bool qkv = G ? p.qkv != 0 : (EPI == EPI_QKV || EPI == EPI_QKVH);
    const bool has_res = G || EPI == EPI_F32 ? p.residual != nullptr : EPI == EPI_PRODX;
    const bool has_stats = G || EPI == EPI_PRODX ? p.stats_out != nullptr : false;
    const bool has_xt = G ? p.xt_hi != nullptr : EPI == EPI_PRODX;
    const bool has_xlo = G ? p.xt_lo != nullptr : false;
    const bool gelu = G || EPI == EPI_TILED ? p.act == 1 : false;
    const int out_mode = G ? p.out_mode : (EPI == EPI_TILED ? ACMI_OUT_TILED : ACMI_OUT_F32);
    for (int e = (int)threadIdx.x; e < 256 * mtv * NT; e += ONE ? 256 * MT * NT : nw * 64) {
        const int t = (e >> 8) % NT, u = (e >> 8) / NT, mm = (e >> 4) & 15, nn = e & 15;
        const bool first = ONE ? true : e == (int)threadIdx.x;
        const int idx = (((mm >> 2) * 16 + nn) << 2) + (mm & 3);
        float v = tl_red_sum(red + (size_t)(t * MT + u) * nw * 256 + idx, nw);
        const int gm = mg + 16 * u + mm, gn = n0 + 16 * t + nn;
        const bool valid = gm < p.M && gn < p.N && (!HT || nn < 8);
        if (split) {  // split-K: raw partial sums; bias / activation / residual are applied by the reducer
            if (valid) reinterpret_cast<float*>(p.out)[((size_t)kslice * p.M + gm) * p.N + gn] = v;
            continue;
        }
        size_t oi = 0;
        // QKV launch carrying the x0 part of the cross-attention query: features >= 3d are stored raw
        const bool rawcol = qkv && gn >= 3 * p.d;
        if (FOLD) {   // folded LayerNorm: rstd * (x W'^T - mean * colsum)
            float mean_s, rstd;
            if (GRAM) {
                // row sum (any column of the row's sums tile) and sum of squares (the diagonal of the Gram tile), summed over
                // the waves' K slices in wave order like the products
                const int idg = (((mm >> 2) * 16 + mm) << 2) + (mm & 3);
                const float s1 = tl_red_sum(red + (size_t)(XT + 2 * u) * nw * 256 + idx, nw);
                const float s2 = tl_red_sum(red + (size_t)(XT + 2 * u + 1) * nw * 256 + idg, nw);
                const float rk = p.inv_K;   // 1 / K, rounded on the host
                mean_s = s1 * rk;   // = mean - shift: the fragments are x - shift
                rstd = __builtin_amdgcn_rsqf(fmaxf(s2 * rk - mean_s * mean_s, 0.f) + p.eps);   // v_rsq_f32: 1 ulp
                // the row means also go to mean_out (the shift of the next producers of x): first n-tile's workgroup only
                if (p.mean_out != nullptr && wgtile == 0 && t == 0 && nn == 0 && gm < p.M)
                    p.mean_out[gm] = mean_s + (p.a_shift != nullptr ? (first ? ex.sh : p.a_shift[gm]) : 0.f);
            } else {
                const float* rs = rowstat + (u * 16 + mm) * 2;
                mean_s = rs[0]; rstd = rs[1];
            }
            if (valid && !rawcol) v = rstd * (v - mean_s * (first ? ex.colsum : p.colsum[gn]));
        }
        if (valid) {
            if (p.bias) v += first ? ex.bias : p.bias[gn];
            if (!qkv) {
                if (gelu) v = gelu_exact(v);
                oi = (size_t)gm * p.N + gn;
                if (has_res) v += first ? ex.res : p.residual[oi];
            }
        }
        if (has_stats) {
            // (mean, M2) of this workgroup's 16 output features per row, for the LayerNorm of the consumer
            // (HT: of its 8 features; lanes 8-15 of a row hold nothing and stay out of lanes 0-7's sums)
            float sm = valid ? v : 0.f;
            sm = HT ? row8_sum(sm) : row16_sum(sm);
            const float mb = sm * (HT ? 0.125f : 0.0625f);
            float dq = valid ? (v - mb) * (v - mb) : 0.f;
            dq = HT ? row8_sum(dq) : row16_sum(dq);
            if (nn == 0 && gm < p.M)
                *reinterpret_cast<float2*>(p.stats_out + ((size_t)gm * (p.N >> (HT ? 3 : 4)) + ntile + t) * 2) = make_float2(mb, dq);
        }
        if (!valid) continue;
        if (has_xt) {  // the residual stream also raw in A-fragment order for the next GEMM
            const size_t ti = tiled_index<WT>(gm, gn, p.xt_nkc);
            // single-term form: relative to the row's shift (include/acmi.h, acmi_linear_desc.xt_shift)
            const float vs = p.xt_shift != nullptr ? v - (first ? ex.osh : p.xt_shift[gm]) : v;
            if (sizeof(WT) == 2) {
                const bf16_t hi = f32_to_bf16(vs);
                reinterpret_cast<bf16_t*>(p.xt_hi)[ti] = hi;
                if (has_xlo)
                    reinterpret_cast<bf16_t*>(p.xt_lo)[tiled_index<WT>(gm, gn, p.xt_lo_nkc)] = f32_to_bf16(vs - bf16_to_f32(hi));
            } else {
                reinterpret_cast<float*>(p.xt_hi)[ti] = vs;
            }
        }
        if (qkv) {
            int part, f, h, dd, pidx, brow;
            if (EPI == EPI_QKV || EPI == EPI_QKVH) {
                // per 16-feature tile (wave-uniform values: scalar arithmetic): the tile lies in one part and one head
                // (no division: four parts at most, and the launcher admits power-of-two head sizes only: p.hd_shift)
                const int fb = __builtin_amdgcn_readfirstlane(n0 + 16 * t);
                part = (fb >= p.d) + (fb >= 2 * p.d) + (fb >= 3 * p.d);
                const int f0 = fb - part * p.d;
                h = f0 >> p.hd_shift;
                f = f0 + nn; dd = f0 - (h << p.hd_shift) + nn;
                pidx = 0; brow = gm;
            } else {
                part = min(gn / p.d, 3); f = gn - part * p.d;   // (the raw block may be wider than d: r_ld)
                h = f / p.hd; dd = f - h * p.hd;
                pidx = gm / p.rpp; brow = gm - pidx * p.rpp;  // several positions per call (prefill)
            }
            if (EPI == EPI_QKVH && part != 3) {
                // agent-scope (write-through) stores: the consumer polls these very words.  The 16 lanes of a row hold 16 consecutive
                // features: every fourth lane collects its three neighbours (DPP row shifts) and issues ONE 16-byte store -- a 4-byte
                // write-through store is a fabric write of its own, ~6 x the time per byte of a 16-byte one (MI355X guide)
                const float v1 = dpp_f32<0x101>(v), v2 = dpp_f32<0x102>(v), v3 = dpp_f32<0x103>(v);   // row_shl:1 .. 3
                if ((nn & 3) == 0) {
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.q_out, 0, -1, 0x00020000);
                    const u32x4 w4 = {__float_as_uint(v), __float_as_uint(v1), __float_as_uint(v2), __float_as_uint(v3)};
                    __builtin_amdgcn_raw_buffer_store_b128(w4, rs, (int)(((size_t)gm * 3 * p.d + part * p.d + f) * 4), 0, 16);   // aux 16 = sc1
                }
            } else if (part == 0) {
                p.q_out[(size_t)gm * p.d + f] = v;
            } else if (part == 3) {
                p.r_out[(size_t)gm * p.r_ld + f] = v;
            } else {
                const size_t ci = (((size_t)brow * p.H + h) * p.Tcap + ex.tpos + pidx) * p.hd + dd;
                void* cache = part == 1 ? p.k_cache : p.v_cache;
                if (p.kv_bf16) reinterpret_cast<bf16_t*>(cache)[ci] = f32_to_bf16(v);
                else reinterpret_cast<float*>(cache)[ci] = v;
            }
        } else if (out_mode == ACMI_OUT_TILED) {
            st_f32(reinterpret_cast<WT*>(p.out) + tiled_index<WT>(gm, gn, p.NKC_out), v);
        } else if (out_mode == ACMI_OUT_BF16) {
            reinterpret_cast<bf16_t*>(p.out)[oi] = f32_to_bf16(v);
        } else {
            reinterpret_cast<float*>(p.out)[oi] = v;
        }
    }
}

// NT = 2: the workgroup owns two adjacent n-tiles (32 features) and every activation fragment feeds both -- for
// the wide GEMMs (N / 16 > 256) whose 16-feature grid would put two workgroups on some CUs.
// CCAP: largest straight-line chunk this instance compiles (its register budget); HAND: the EPI_QKVH epilogue is reachable
template <typename WT, int MT, int LN, int NT = 1, int NS = 8, bool HT = false, int CCAP = 24, bool HAND = false>
__device__ __forceinline__ void tl_body(const TlHot& h, const int aoff, const int wgtile, const int kslice) {
    static_assert(NT == 1 || NT == 2, "one or two n-tiles per workgroup");
    static_assert(!HT || (NT == 1 && LN == 0), "half-tile workgroups: plain GEMM, one (half) n-tile");
    const int ntile = wgtile * NT;   // first n-tile of this workgroup (HT: the half-tile index)
    constexpr int D = (LN == 2 || LN == 3) ? 2 : 1;
    constexpr bool PART = LN == 1 || LN == 2;
    constexpr bool GRAM = LN == 4;
    // fragments per straight-line chunk: (1 + MT D) C fragment registers (4 VGPRs each) must leave the kernel
    // without scratch (a kernel with a private segment starts its waves measurably slower) inside the 256
    // VGPRs of a 2-waves-per-SIMD launch
    constexpr int CQ = HT ? 52 / (1 + 2 * MT) : ((PART ? (NS > 8 ? 36 : 44) : (GRAM ? 48 : 52)) / (NT + MT * D));
    constexpr int CMAX0 = CQ >= 24 ? 24 : (CQ >= 16 ? 16 : (CQ >= 12 ? 12 : (CQ >= 8 ? 8 : (CQ >= 6 ? 6 : (CQ >= 4 ? 4 : 2)))));
    constexpr int CMAX = CMAX0 < CCAP ? CMAX0 : CCAP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TlTrace tr{};
    ACMI_TR(tr.t, 0);
    const int lane = threadIdx.x & 63, nw = h.nw, ksp = h.ksp;   // (blockDim / gridDim would be scalar loads of the kernarg segment)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: fragment addresses stay in SGPRs
    LinArgs p;   // filled by the first chunk, behind its weight requests
    constexpr int NRED = (HT ? 1 : NT) * MT + (GRAM ? 2 * MT : 0);   // tiles of the cross-wave reduction
    float* red = reinterpret_cast<float*>(smem);        // [NRED][nw][256] partial accumulators (+ row sums / Gram tiles)
    float* rowstat = red + (size_t)NRED * nw * 256;     // [16 MT][2] mean, rstd (LN 1 / 2)
    const int n0 = ntile * (HT ? 8 : 16), NKC = HT ? h.NKC >> 1 : h.NKC;   // HT: K counted in 1 KB weight units
    const u32x4* wt = h.w + (size_t)ntile * NKC * 64;  // + fragment * 64 + lane
    const int kcs = h.kcs, kbeg = kslice * kcs;         // this workgroup's K slice
    const float* own = reinterpret_cast<const float*>(wt + (size_t)(kbeg + min(wave, kcs - 1)) * 64 + lane);

    // one group of MT 16-row blocks per workgroup (grid.z): no loop around the body, so that nothing of the
    // epilogue is hoisted in front of the first load
    const int mg = (int)blockIdx.z * 16 * MT;
    const int mtv = min(MT, (h.M - mg + 15) >> 4);
    f32x4 accs[(HT ? 2 : NT) * MT], accx[2 * MT];
#pragma unroll
    for (int u = 0; u < (HT ? 2 : NT) * MT; ++u) accs[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2 * MT; ++u) accx[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ngroups = 4 * mtv;   // statistics: groups of 4 rows (16 lanes each), group g of this wave first = wave
    // addresses of everything but the weights: evaluated by tl_chunk once its weight requests are out
    // (`z` is an opaque zero produced behind the weight requests: added to every index these addresses derive from,
    // it keeps loop-invariant code motion from hoisting the arithmetic back in front of them)
    auto late = [&](const int z) -> TlLate {
        TlLate L;
        const int mgz = mg + z, lz = lane + z, wz = wave + z;
        L.mts = h.a_rbs * 64; L.mtl = D == 2 ? p.alo_rbs * 64 : 0;   // fragment lanes between consecutive 16-row blocks (a, a_lo)
        L.mtv = mtv;
        L.at = h.a + (size_t)((mgz >> 4) * L.mts);
        L.al = D == 2 ? reinterpret_cast<const u32x4*>(p.a_lo) + (size_t)((mgz >> 4) * L.mtl) : nullptr;
        L.st_ptr = own;
        if (PART) L.st_ptr = h.a_stats + min(mgz + min(wz, ngroups - 1) * 4 + (lz >> 4), h.M - 1) * h.a_np * 2;
        L.psh = own;
        if (PART && h.a_shift != nullptr) L.psh = h.a_shift + min(mgz + min(wz, ngroups - 1) * 4 + (lz >> 4), h.M - 1);
        // LN 4: the shift of this thread's first epilogue row (only the workgroup that publishes mean_out adds it back)
        if (GRAM && h.a_shift != nullptr) L.psh = h.a_shift + min(mgz + 16 * ((((int)threadIdx.x + z) >> 8) / NT) + ((lz >> 4) & 3) + 4 * (wz & 3), h.M - 1);
        return L;
    };
    auto epi = [&](const int z) -> TlEpi {
        TlEpi E;
        const int mgz = mg + z, n0z = n0 + z;
        // this thread's first epilogue element
        const int e0 = (int)threadIdx.x + z, eq = e0 >> 8, emm = (e0 >> 4) & 15, enn = e0 & 15;
        const int et = eq % NT, eu = eq / NT;   // (n-tile, row block) of that element: e >> 8 = row block * NT + n-tile
        const int egn = min(n0z + (HT ? (enn & 7) : 16 * et + enn), p.N - 1), egm = min(mgz + 16 * eu + emm, p.M - 1);
        E.pb = p.bias != nullptr ? p.bias + egn : own;
        E.pc = p.colsum != nullptr ? p.colsum + egn : own;
        E.pr = p.residual != nullptr ? p.residual + (egm * p.N + egn) : own;
        E.ppos = p.qkv ? p.pos : reinterpret_cast<const int*>(own);
        E.posh = p.xt_shift != nullptr ? p.xt_shift + egm : own;
        return E;
    };
    TlExtras ex;

    // wave w owns the CONTIGUOUS run of K fragments [w fpw, (w + 1) fpw) (+ a ragged tail): its requests walk
    // 1 KB, 2 KB, ... through one DRAM page instead of striding by nw KB (out-proj 4.81 -> 4.57, FFN2 9.93 -> 9.37 us)
    int kc = kbeg + wave * h.fpw, rem = h.fpw;
    if (rem == 0 && nw * h.fpw + wave >= kcs) tl_load_args(p, aoff, 0);   // a wave without fragments: no chunk fills p
#define ACMI_TL_RUN(Cn)                                                                                                 \
    while (rem >= Cn) {                                                                                                \
        if constexpr (HT) tl_chunk_ht<WT, MT, Cn>(p, aoff, wt, kc, late, epi, accs, ex, tr);                           \
        else tl_chunk<WT, MT, LN, NT, NS, Cn>(h, p, aoff, wt, kc, h.a_np, late, epi, accs, accx, ex, tr);              \
        kc += Cn; rem -= Cn;                                                                                           \
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
    kc = kbeg + nw * h.fpw + wave;
    if (kc < kbeg + kcs) {  // ragged tail: the first kcs % nw waves own one more fragment
        if constexpr (HT) tl_chunk_ht<WT, MT, 1>(p, aoff, wt, kc, late, epi, accs, ex, tr);
        else tl_chunk<WT, MT, LN, NT, NS, 1>(h, p, aoff, wt, kc, h.a_np, late, epi, accs, accx, ex, tr);
    }
    if constexpr (HT) {  // columns 0-7 of the even accumulators + columns 8-15 of the odd ones, rotated onto 0-7
#pragma unroll
        for (int u = 0; u < MT; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) accs[u][r] = accs[2 * u][r] + dpp_f32<0x128>(accs[2 * u + 1][r]);
    }

    // ---- deterministic cross-wave reduction through LDS
#pragma unroll
    for (int u = 0; u < (HT ? 1 : NT) * MT; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((size_t)u * nw + wave) * 256 + lane * 4 + r] = accs[u][r];
    if (GRAM) {
#pragma unroll
        for (int u = 0; u < 2 * MT; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((size_t)((HT ? 1 : NT) * MT + u) * nw + wave) * 256 + lane * 4 + r] = accx[u][r];
    }
    if (PART) {
        // mean / rstd of rows mg .. mg + 16 mtv - 1 from the producer's equal-count partials (Chan)
#pragma unroll
        for (int i = 0; i < NS; ++i) asm volatile("" : "+v"(ex.pm[i]), "+v"(ex.pq[i]));  // stays behind the K loop
        // the row means also go to mean_out (the shift of the next producers of x): first n-tile's workgroup only
        const bool wmean = p.mean_out != nullptr && wgtile == 0;
        if (wave < ngroups) {
            const int row = mg + wave * 4 + (lane >> 4);
            rowstat_finish<NS>(ex.pm, ex.pq, p.a_np, p.a_cnt, p.K, p.eps, lane, rowstat + wave * 8,
                               p.a_shift != nullptr ? ex.sh : 0.f, (wmean && row < p.M) ? p.mean_out + row : nullptr);
        }
        for (int g = wave + nw; g < ngroups; g += nw) {
            const int row = mg + g * 4 + (lane >> 4);
            rowstat_load<NS>(p.a_stats, p.a_np, p.M, mg + g * 4, lane, ex.pm, ex.pq);
            rowstat_finish<NS>(ex.pm, ex.pq, p.a_np, p.a_cnt, p.K, p.eps, lane, rowstat + g * 8,
                               p.a_shift != nullptr ? p.a_shift[min(row, p.M - 1)] : 0.f,
                               (wmean && row < p.M) ? p.mean_out + row : nullptr);
        }
    }
    ACMI_TR(tr.t, 5);
    __syncthreads();
    ACMI_TR(tr.t, 6);

#define ACMI_TL_EPI_W(E, W) tl_epilogue<WT, MT, LN, NT, HT, E, W>(p, ex, red, rowstat, nw, ksp, kslice, mg, mtv, n0, ntile, wgtile)
#define ACMI_TL_EPI(E) do { if (nw == 8) ACMI_TL_EPI_W(E, 8); else if (nw == 4) ACMI_TL_EPI_W(E, 4); else if (nw == 2) ACMI_TL_EPI_W(E, 2); \
                            else ACMI_TL_EPI_W(EPI_GEN, 0); } while (0)
    const int epi_kind = __builtin_amdgcn_readfirstlane(p.epi);
    if (LN == 0 && NT == 1 && epi_kind == EPI_PRODX) ACMI_TL_EPI(EPI_PRODX);
    else if (!HT && LN != 3 && epi_kind == EPI_TILED) ACMI_TL_EPI(EPI_TILED);
    else if (!HT && epi_kind == EPI_F32) ACMI_TL_EPI(EPI_F32);
    else if (!HT && (LN == 1 || LN == 2 || LN == 4) && epi_kind == EPI_QKV) ACMI_TL_EPI(EPI_QKV);
    else if (HAND && epi_kind == EPI_QKVH) ACMI_TL_EPI(EPI_QKVH);
    else ACMI_TL_EPI_W(EPI_GEN, 0);
#undef ACMI_TL_EPI
#undef ACMI_TL_EPI_W
#ifdef ACMI_TRACE
    ACMI_TR(tr.t, 7);
    ACMI_TR_WAIT_VM(0);
    ACMI_TR(tr.t, 8);
    if (p.trace != nullptr && lane == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        tr.t[9] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
        const size_t wg = blockIdx.x + (size_t)gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z);
        unsigned long long* dst = p.trace + (wg * nw + wave) * ACMI_TRACE_NSTAMP;
#pragma unroll
        for (int i = 0; i < ACMI_TRACE_NSTAMP; ++i) dst[i] = tr.t[i];
    }
#endif
}

// Kernarg layout (both kernels): the TlHot words first (<= 14 dwords: preloaded into SGPRs), the LinArgs block(s) at byte
// ACMI_TL_ARGS_OFF(_PAIR) -- never touched through the parameter, only through tl_load_args.
//   g0 = K tiles | K tiles per slice << 16     g1 = fragments per wave | waves << 12 | split-K slices << 16
//   g2 = a_rbs | M << 16                       g3 = statistics partials per row
template <typename WT, int MT, int LN, int NT, int NS = 8, bool HT = false>
__global__ __launch_bounds__(512) void lin_tiled_kernel(const u32x4* hw, const u32x4* ha, const float* hst, const float* hsh,
                                                        unsigned g0, unsigned g1, unsigned g2, unsigned g3, const LinArgs) {
    const TlHot h = tl_unpack(hw, ha, hst, hsh, g0, g1, g2, g3);
    tl_body<WT, MT, LN, NT, NS, HT>(h, ACMI_TL_ARGS_OFF, (int)blockIdx.x, (int)blockIdx.y);
}

// Two independent GEMMs of the chain in ONE launch (one dependency edge less): workgroups [0, tiles0) run p0
// (plain), the rest run p1 (LN 3: x | a concatenated along K, no LayerNorm).  Used for
//   x1 = x0 + att W_out^T   and   r = [x0 | att] [W_cq' | W_cq' W_out]^T  (= x1 W_cq'^T, the cross-attention
// query before its LayerNorm statistics are applied), see acmi_lm_step.
//   g00 / g01 = K tiles | K tiles per slice << 16 of p0 / p1     g1 = fragments per wave of p0 | of p1 << 12 | waves << 24
//   g2 = a_rbs | M << 16 (shared)
template <typename WT, int MT, int LNB>
__global__ __launch_bounds__(512) void lin_pair_kernel(const u32x4* hw0, const u32x4* hw1, const u32x4* ha0, const u32x4* ha1,
                                                       unsigned g00, unsigned g01, unsigned g1, unsigned g2, int tiles0, int,
                                                       const LinArgs, const LinArgs) {
    const bool first = (int)blockIdx.x < tiles0;
    const unsigned nw = g1 >> 24;
    const TlHot h = tl_unpack(first ? hw0 : hw1, first ? ha0 : ha1, nullptr, nullptr, first ? g00 : g01,
                              ((first ? g1 : g1 >> 12) & 0xfffu) | (nw << 12) | (1u << 16), g2, 0u);
    if (first) tl_body<WT, MT, 0, 1, 8, false>(h, ACMI_TL_ARGS_OFF_PAIR, (int)blockIdx.x, 0);
    else tl_body<WT, MT, LNB, 1, 8, false>(h, ACMI_TL_ARGS_OFF_PAIR + (int)sizeof(LinArgs), (int)blockIdx.x - tiles0, 0);
}

#if defined(ACMI_TRACE) && ACMI_GEMM_MAIN
// Host side of the timeline: acmi_trace_config hands over a device buffer and restarts the launch index; every GEMM launch
// of the decode step then reserves [workgroups][waves][ACMI_TRACE_NSTAMP] words of it (in launch order) and is described
// by acmi_trace_info.  Captured into a hipGraph, a launch keeps its region: each replay overwrites the previous one's stamps.
struct TraceRec { long long off; int kind, wgs, waves, N, K, M; };
static unsigned long long* g_trace_buf = nullptr;
static long long g_trace_cap = 0, g_trace_used = 0;
static TraceRec g_trace_rec[1024];
static int g_trace_n = 0;
unsigned long long* acmi_trace_reserve(int kind, int wgs, int waves, int N, int K, int M) {
    const long long need = (long long)wgs * waves * ACMI_TRACE_NSTAMP;
    if (g_trace_buf == nullptr || g_trace_n >= 1024 || g_trace_used + need > g_trace_cap) return nullptr;
    g_trace_rec[g_trace_n++] = TraceRec{g_trace_used, kind, wgs, waves, N, K, M};
    unsigned long long* r = g_trace_buf + g_trace_used;
    g_trace_used += need;
    return r;
}
extern "C" int acmi_trace_config(unsigned long long* buf, long long capacity_words) {
    g_trace_buf = buf; g_trace_cap = capacity_words; g_trace_used = 0; g_trace_n = 0;
    return ACMI_OK;
}
extern "C" int acmi_trace_count() { return g_trace_n; }
// out[0..7] = word offset (lo, hi), kind, workgroups, waves, N, K, M.  kind: bit 0 folded LayerNorm, bit 1 QKV scatter,
// bit 2 half-tile workgroups, bit 3 paired launch, bit 4 produces x (statistics + fragments), bits 8.. features per workgroup
extern "C" int acmi_trace_info(int i, int* out) {
    if (i < 0 || i >= g_trace_n) return ACMI_EINVAL;
    const TraceRec& r = g_trace_rec[i];
    out[0] = (int)(r.off & 0xffffffffll); out[1] = (int)(r.off >> 32); out[2] = r.kind; out[3] = r.wgs; out[4] = r.waves;
    out[5] = r.N; out[6] = r.K; out[7] = r.M;
    return ACMI_OK;
}
#endif

// Workgroup size.  (1) Waves are not free: the dispatcher starts ~1.25 waves / ns (a 288-workgroup x 16-wave
// launch with no loads at all takes 6.2 us), so a wave should own ~12 fragments or more, all of them requested
// before its first wait.  (2) The kernel needs > 128 VGPRs for that, i.e. a CU holds 8 waves: nw in {8, 4, 2, 1}
// packs 1, 2, 4, 8 workgroups per CU exactly, and the grid must fit the 256 CUs in ONE round (a 288-workgroup
// grid of 6-wave workgroups runs 256 + 32: the launch takes twice as long).
static int tiled_waves(int tiles, int frags) {
    int nw = tiles <= 256 ? 8 : (tiles <= 512 ? 4 : (tiles <= 1024 ? 2 : 1));
    static const int min_fpw = getenv("ACMI_LIN_FPW") ? atoi(getenv("ACMI_LIN_FPW")) : 12;
    while (nw > 1 && frags < min_fpw * nw) nw >>= 1;
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
    ACMI_REQUIRE(a.stats_out == nullptr || a.N % (a.w_half ? 8 : 16) == 0, "acmi_linear: stats_out needs N %% %d == 0 (N=%d)",
                 a.w_half ? 8 : 16, a.N);
    ACMI_REQUIRE(a.ksplit == 1 || (!a.qkv && a.stats_out == nullptr && a.xt_hi == nullptr),
                 "acmi_linear: split-K is incompatible with QKV scatter / stats_out / xt_hi");
    return ACMI_OK;
}

// which specialised epilogue the call's flags allow (tl_epilogue); HT / LayerNorm mode are checked in the kernel
static int tiled_epi(const LinArgs& a) {
    if (a.ksplit > 1) return EPI_GEN;
    if (a.qkv)
        return (a.M <= a.rpp && a.d % 16 == 0 && a.hd % 16 == 0 && (a.hd & (a.hd - 1)) == 0 && a.N % 16 == 0 && a.N <= 3 * a.d + a.r_ld &&
                a.stats_out == nullptr && a.xt_hi == nullptr) ? EPI_QKV : EPI_GEN;
    if (a.xt_hi != nullptr)
        return (a.out_mode == ACMI_OUT_F32 && a.residual != nullptr && a.xt_lo == nullptr && a.act == 0) ? EPI_PRODX : EPI_GEN;
    if (a.stats_out != nullptr) return EPI_GEN;
    if (a.out_mode == ACMI_OUT_TILED) return a.residual == nullptr ? EPI_TILED : EPI_GEN;
    if (a.out_mode == ACMI_OUT_F32) return a.act == 0 ? EPI_F32 : EPI_GEN;
    return EPI_GEN;
}

template <typename WT, int MT, int LN, int NT, int NS, bool HT>
static int launch_tiled_k(LinArgs& a, int gx, int nw, size_t lds, hipStream_t st) {
    if (lds > 64 * 1024) {  // 2 n-tiles x 4 row blocks x 8 waves: just above the default dynamic LDS limit
        static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&lin_tiled_kernel<WT, MT, LN, NT, NS, HT>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;   // once, thread safe
        if (!attr_ok) {
            acmi_set_error("acmi_linear: cannot raise the dynamic LDS limit");
            return ACMI_ELAUNCH;
        }
    }
    if (a.r_ld <= 0) a.r_ld = a.d;
    a.epi = tiled_epi(a);
    a.inv_K = 1.0f / (float)a.K;
    a.hd_shift = 0;
    while (a.qkv && (1 << a.hd_shift) < a.hd) ++a.hd_shift;
#ifdef ACMI_TRACE
    a.trace = acmi_trace_reserve((LN == 1 || LN == 2 || LN == 4) | (a.qkv ? 2 : 0) | (HT ? 4 : 0) | (a.xt_hi != nullptr ? 16 : 0) | ((HT ? 8 : 16 * NT) << 8),
                                 gx * a.ksplit * ((a.M + 16 * MT - 1) / (16 * MT)), nw, a.N, a.K, a.M);
#endif
    ACMI_REQUIRE(a.NKC <= 0xffff && a.kcs <= 0xffff && a.fpw <= 0xfff && a.ksplit <= 0xffff && a.a_rbs <= 0xffff && a.M <= 0xffff,
                 "acmi_linear: geometry beyond the packed launch words (K tiles %d, fragments per wave %d, M %d)", a.NKC, a.fpw, a.M);
    const unsigned g0 = (unsigned)a.NKC | ((unsigned)a.kcs << 16), g1 = (unsigned)a.fpw | ((unsigned)nw << 12) | ((unsigned)a.ksplit << 16);
    const unsigned g2 = (unsigned)a.a_rbs | ((unsigned)a.M << 16), g3 = (unsigned)a.a_np;
    constexpr bool PART = LN == 1 || LN == 2;
    hipLaunchKernelGGL((lin_tiled_kernel<WT, MT, LN, NT, NS, HT>), dim3(gx, a.ksplit, (a.M + 16 * MT - 1) / (16 * MT)),
                       dim3(nw * 64), lds, st, reinterpret_cast<const u32x4*>(a.w), reinterpret_cast<const u32x4*>(a.a),
                       PART ? a.a_stats : nullptr, PART || LN == 4 ? a.a_shift : nullptr, g0, g1, g2, g3, a);
    return acmi_check_launch("lin_tiled_kernel");
}

template <typename WT, int MT, int LN, int NT, int NS = 8, bool HT = false>
static int launch_tiled_t(LinArgs& a, hipStream_t st) {
    const int gx = HT ? a.N / 8 : (a.N + 15) / 16 / NT;
    const int wgs = gx * a.ksplit, frags = (HT ? a.NKC / 2 : a.NKC) / a.ksplit;
    const int nw = tiled_waves(wgs, frags * NT);
    a.kcs = frags; a.fpw = frags / nw;
    const size_t lds = (size_t)(NT * MT + (LN == 4 ? 2 * MT : 0)) * nw * 1024 + (size_t)MT * 128;
    return launch_tiled_k<WT, MT, LN, NT, NS, HT>(a, gx, nw, lds, st);
}

template <typename WT>
int launch_tiled(LinArgs& a, hipStream_t st) {
    int rc = tiled_prepare<WT>(a);
    if (rc) return rc;
    const int mt = a.M > 32 ? 4 : (a.M > 16 ? 2 : 1);  // 1, 2 or 4 16-row blocks share each weight fragment
    // folded LayerNorm: statistics from the producer's partials (a_stats; with a lo term: LN 2), or from the fragments (LN 4)
    const int ln = a.colsum == nullptr ? (a.lo_split > 0 ? 3 : 0) : (a.a_stats == nullptr ? 4 : (a.a_lo != nullptr ? 2 : 1));
    // wide (32-feature) workgroups when the 16-feature grid would not fit the 256 CUs in one workgroup each:
    // the activation fragments, re-read by every workgroup, are then shared by two n-tiles
    static const bool wide_ok = !(getenv("ACMI_LIN_WIDE") != nullptr && getenv("ACMI_LIN_WIDE")[0] == '0');
    const int tiles = (a.N + 15) / 16;
    if (a.w_half) {   // 8-feature workgroups over a weight in half-tile order
        ACMI_REQUIRE(ln == 0 && a.ksplit == 1 && !a.qkv && a.NKC % 2 == 0 && a.N % 8 == 0 && mt <= 2,
                     "acmi_linear: w_half needs a plain GEMM (no LayerNorm / split-K / QKV scatter), N %% 8 == 0 (N=%d), "
                     "an even number of K tiles (%d) and M <= 32 (M=%d)", a.N, a.NKC, a.M);
        if (mt == 1) return launch_tiled_t<WT, 1, 0, 1, 8, true>(a, st);
        return launch_tiled_t<WT, 2, 0, 1, 8, true>(a, st);
    }
    if (a.colsum != nullptr && a.a_stats != nullptr && a.a_np > 128) {   // statistics of a half-tile producer: up to 256 partials per row
        ACMI_REQUIRE(a.a_np <= 256 && mt <= 2, "acmi_linear: %d statistics partials per row unsupported (<= 256, and <= 128 "
                     "for M > 32; M=%d)", a.a_np, a.M);
        const bool wide = wide_ok && a.ksplit == 1 && tiles > 256 && tiles % 2 == 0 && a.N % 16 == 0 &&
                          a.stats_out == nullptr && a.xt_hi == nullptr;
#define ACMI_TLS_CASE(MTv, LNv)                                                          \
        if (mt == MTv && ln == LNv) {                                                     \
            if (wide) return launch_tiled_t<WT, MTv, LNv, 2, 16>(a, st);                  \
            return launch_tiled_t<WT, MTv, LNv, 1, 16>(a, st);                            \
        }
        ACMI_TLS_CASE(1, 1) ACMI_TLS_CASE(1, 2) ACMI_TLS_CASE(2, 1) ACMI_TLS_CASE(2, 2)
#undef ACMI_TLS_CASE
        return ACMI_EINVAL;
    }
    if (wide_ok && ln != 3 && a.ksplit == 1 && tiles > 256 && tiles % 2 == 0 && a.N % 16 == 0 &&
        a.stats_out == nullptr && a.xt_hi == nullptr) {
#define ACMI_TLW_CASE(MTv, LNv) if (mt == MTv && ln == LNv) return launch_tiled_t<WT, MTv, LNv, 2>(a, st);
        ACMI_TLW_CASE(1, 0) ACMI_TLW_CASE(1, 1) ACMI_TLW_CASE(1, 2)
        ACMI_TLW_CASE(1, 4) ACMI_TLW_CASE(2, 4)
        ACMI_TLW_CASE(2, 0) ACMI_TLW_CASE(2, 1) ACMI_TLW_CASE(2, 2)
        ACMI_TLW_CASE(4, 0) ACMI_TLW_CASE(4, 1) ACMI_TLW_CASE(4, 2)
#undef ACMI_TLW_CASE
    }
#define ACMI_TL_CASE(MTv, LNv) if (mt == MTv && ln == LNv) return launch_tiled_t<WT, MTv, LNv, 1>(a, st);
    ACMI_TL_CASE(1, 0) ACMI_TL_CASE(1, 1) ACMI_TL_CASE(1, 2) ACMI_TL_CASE(1, 3)
    ACMI_TL_CASE(1, 4) ACMI_TL_CASE(2, 4)
    ACMI_TL_CASE(2, 0) ACMI_TL_CASE(2, 1) ACMI_TL_CASE(2, 2) ACMI_TL_CASE(2, 3)
    ACMI_TL_CASE(4, 0) ACMI_TL_CASE(4, 1) ACMI_TL_CASE(4, 2) ACMI_TL_CASE(4, 3)
#undef ACMI_TL_CASE
    return ACMI_EINVAL;
}

// p0 (plain tiled GEMM) and p1 (x | a concatenated along K, hi + lo for the x part) in one launch
template <typename WT>
int launch_pair(LinArgs& p0, LinArgs& p1, hipStream_t st) {
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
    ACMI_REQUIRE(p0.a_rbs == p1.a_rbs, "acmi_linear_pair: the two activations must share a_rbs (%d vs %d)", p0.a_rbs, p1.a_rbs);
    ACMI_REQUIRE(p0.NKC <= 0xffff && p1.NKC <= 0xffff && p0.fpw <= 0xfff && p1.fpw <= 0xfff && p0.a_rbs <= 0xffff && p0.M <= 0xffff,
                 "acmi_linear_pair: geometry beyond the packed launch words");
    const u32x4* hw0 = reinterpret_cast<const u32x4*>(p0.w); const u32x4* hw1 = reinterpret_cast<const u32x4*>(p1.w);
    const u32x4* ha0 = reinterpret_cast<const u32x4*>(p0.a); const u32x4* ha1 = reinterpret_cast<const u32x4*>(p1.a);
    const unsigned g00 = (unsigned)p0.NKC | ((unsigned)p0.kcs << 16), g01 = (unsigned)p1.NKC | ((unsigned)p1.kcs << 16);
    const unsigned g1 = (unsigned)p0.fpw | ((unsigned)p1.fpw << 12) | ((unsigned)nw << 24), g2 = (unsigned)p0.a_rbs | ((unsigned)p0.M << 16);
#ifdef ACMI_TRACE
    p0.trace = p1.trace = acmi_trace_reserve(8 | 16 | (16 << 8), (int)(grid.x * grid.z), nw, p0.N + p1.N, p0.K, p0.M);
#endif
    p0.epi = tiled_epi(p0); p1.epi = tiled_epi(p1);
    p0.inv_K = 1.0f / (float)p0.K; p1.inv_K = 1.0f / (float)p1.K;
#define ACMI_PAIR_CASE(MTv)                                                                              \
    if (mt == MTv) {                                                                                     \
        if (hl) hipLaunchKernelGGL((lin_pair_kernel<WT, MTv, 3>), grid, block, lds, st, hw0, hw1, ha0, ha1, g00, g01, g1, g2, t0, 0, p0, p1); \
        else hipLaunchKernelGGL((lin_pair_kernel<WT, MTv, 0>), grid, block, lds, st, hw0, hw1, ha0, ha1, g00, g01, g1, g2, t0, 0, p0, p1);    \
    }
    ACMI_PAIR_CASE(1) ACMI_PAIR_CASE(2) ACMI_PAIR_CASE(4)
#undef ACMI_PAIR_CASE
    return acmi_check_launch("lin_pair_kernel");
}

#if ACMI_GEMM_MAIN
// -----------------------------------------------------------------------------------------------------
// qkv_attn_kernel (round 6): the decode step's QKV GEMM and the self-attention that consumes it as ONE launch.  Workgroups
// [0, tiles) run tl_body (16 output features each, 4 waves, folded LayerNorm with the statistics from the fragments) with the
// hand-off epilogue; workgroups [tiles, tiles + rows * H) run attn_fused_role (acmi_attn_fused.h): the K / V stream of a layer
// -- the only large stream of the layer that does not depend on the previous launch's output -- is requested while the weights
// stream, and the 3.6 us a self-attention launch spends before its first byte arrives overlap the GEMM.
// Kernarg: 14 preloaded dwords (weights, activation, its shift, the two caches, four packed words), then the LinArgs block at
// byte ACMI_TL_ARGS_OFF_PAIR and the FusedAttnArgs block behind it.
//   g0 = K tiles | K tiles per slice << 16   g1 = fragments per wave | waves << 12 | heads << 16
//   g2 = a_rbs | M << 16                     g3 = GEMM workgroups | cache capacity << 16
#include "acmi_attn_fused.h"
static_assert(sizeof(FusedAttnArgs) % 8 == 0, "kernarg layout of qkv_attn_kernel");
template <int CCAP, int MT = 1>
__global__ __launch_bounds__(256) void qkv_attn_kernel(const u32x4* hw, const u32x4* ha, const float* hsh, const void* hkc, const void* hvc,
                                                       unsigned g0, unsigned g1, unsigned g2, unsigned g3, const LinArgs, const FusedAttnArgs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tiles = (int)(g3 & 0xffffu);
    if ((int)blockIdx.x < tiles) {
        const TlHot h = tl_unpack(hw, ha, nullptr, hsh, g0, (g1 & 0xffffu) | (1u << 16), g2, 0u);
        tl_body<bf16_t, MT, 4, 1, 8, false, CCAP, true>(h, ACMI_TL_ARGS_OFF_PAIR, (int)blockIdx.x, 0);
    } else {
        attn_fused_role(hkc, hvc, (int)(g1 >> 16), (int)(g3 >> 16), (int)blockIdx.x - tiles,
                        ACMI_TL_ARGS_OFF_PAIR + (int)sizeof(LinArgs), smem);
    }
}

static std::atomic<long long> g_qkv_attn_launches{0};
extern "C" long long acmi_qkv_attn_launches(void) { return g_qkv_attn_launches.load(std::memory_order_relaxed); }

int acmi_launch_qkv_attn(LinArgs& a, FusedAttnArgs& f, const void* kc, const void* vc, int H, int Tcap, hipStream_t st) {
    int rc = tiled_prepare<bf16_t>(a);
    if (rc) return rc;
    const int tiles = (a.N + 15) / 16, nw = 4;
    ACMI_REQUIRE(a.colsum != nullptr && a.a_stats == nullptr && a.a_lo == nullptr && a.M <= 32 && a.N % 16 == 0 && a.qkv &&
                 a.NKC % nw == 0 && a.NKC / nw <= 16 && tiles <= 0xffff && Tcap <= 0xffff && H <= 0xffff && a.hd == 64 &&
                 a.kv_bf16 && a.M == a.rpp,
                 "acmi_qkv_attn: needs the fragment-statistics LayerNorm, <= 32 rows, head size 64, a bf16 cache and K tiles a "
                 "multiple of 4 with at most 16 per wave (M=%d N=%d K=%d hd=%d)", a.M, a.N, a.K, a.hd);
    a.ksplit = 1; a.kcs = a.NKC; a.fpw = a.NKC / nw;
    if (a.r_ld <= 0) a.r_ld = a.d;
    a.epi = EPI_QKVH;
    a.inv_K = 1.0f / (float)a.K;
    a.hd_shift = 6;
#ifdef ACMI_TRACE
    a.trace = nullptr;
#endif
    f.rows = a.M; f.d = a.d;
    const int mt = a.M > 16 ? 2 : 1;   // 16-row blocks sharing each weight fragment
    const size_t lds_g = (size_t)3 * mt * nw * 1024 + (size_t)mt * 128, lds_a = 2048 + (f.stage_k ? (size_t)nw * 8192 : 0);
    const size_t lds = lds_g > lds_a ? lds_g : lds_a;
    const unsigned g0 = (unsigned)a.NKC | ((unsigned)a.kcs << 16), g1 = (unsigned)a.fpw | ((unsigned)nw << 12) | ((unsigned)H << 16);
    const unsigned g2 = (unsigned)a.a_rbs | ((unsigned)a.M << 16), g3 = (unsigned)tiles | ((unsigned)Tcap << 16);
    const dim3 grid(tiles + a.M * H), block(nw * 64);
    if (mt == 2)
        hipLaunchKernelGGL((qkv_attn_kernel<12, 2>), grid, block, lds, st, reinterpret_cast<const u32x4*>(a.w), reinterpret_cast<const u32x4*>(a.a),
                           a.a_shift, kc, vc, g0, g1, g2, g3, a, f);
    else if (a.fpw <= 12)
        hipLaunchKernelGGL(qkv_attn_kernel<12>, grid, block, lds, st, reinterpret_cast<const u32x4*>(a.w), reinterpret_cast<const u32x4*>(a.a),
                           a.a_shift, kc, vc, g0, g1, g2, g3, a, f);
    else
        hipLaunchKernelGGL(qkv_attn_kernel<16>, grid, block, lds, st, reinterpret_cast<const u32x4*>(a.w), reinterpret_cast<const u32x4*>(a.a),
                           a.a_shift, kc, vc, g0, g1, g2, g3, a, f);
    g_qkv_attn_launches.fetch_add(1, std::memory_order_relaxed);
    return acmi_check_launch("qkv_attn_kernel");
}
#endif

#if !ACMI_GEMM_MAIN
template int launch_rowmajor<float>(LinArgs&, hipStream_t);
template int launch_tiled<float>(LinArgs&, hipStream_t);
template int launch_pair<float>(LinArgs&, LinArgs&, hipStream_t);
#else
extern template int launch_rowmajor<float>(LinArgs&, hipStream_t);
extern template int launch_tiled<float>(LinArgs&, hipStream_t);
extern template int launch_pair<float>(LinArgs&, LinArgs&, hipStream_t);

int acmi_launch_lin(LinArgs& a, int wdtype, hipStream_t st) {
    ACMI_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "acmi_linear: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
    if (a.ksplit < 1) a.ksplit = 1;
    if (a.rpp <= 0) a.rpp = a.M;
    ACMI_REQUIRE(!(a.a_tiled && a.ln_mode), "acmi_linear: LayerNorm needs a row-major activation");
    ACMI_REQUIRE(a.colsum == nullptr || (a.a_tiled && (a.a_stats == nullptr || (a.a_np >= 1 && a.a_np <= 256 && a.a_np * a.a_cnt == a.K))),
                 "acmi_linear: folded LayerNorm needs a tiled activation and consistent row statistics (np=%d cnt=%d K=%d)", a.a_np, a.a_cnt, a.K);
    ACMI_REQUIRE(a.colsum == nullptr || a.a_stats != nullptr || (a.a_lo == nullptr && a.M <= 32),
                 "acmi_linear: LayerNorm statistics from the fragments (a_stats NULL) need a single-term activation and M <= 32 (M=%d)", a.M);
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
    return acmi_launch_lin(p, wdtype, (hipStream_t)stream);
}

int acmi_launch_pair(LinArgs& p0, LinArgs& p1, int wdtype, hipStream_t st) {
    return wdtype == ACMI_BF16 ? launch_pair<bf16_t>(p0, p1, st) : launch_pair<float>(p0, p1, st);
}

static int desc_to_args(const acmi_linear_desc& c, LinArgs& p) {
    ACMI_REQUIRE(c.a_mode >= 0 && c.a_mode <= 2, "acmi_linear: bad a_mode %d", c.a_mode);
    ACMI_REQUIRE((c.ln_g == nullptr) == (c.ln_b == nullptr), "acmi_linear: ln_g and ln_b go together");
    const int kt = c.wdtype == ACMI_BF16 ? 32 : 16;
    p.a = c.a; p.a_tiled = c.a_mode == ACMI_A_TILED;
    p.ln_mode = c.ln_g ? 2 : (c.a_mode == ACMI_A_ROWMAJOR_F32_NORM ? 1 : 0);
    p.ln_g = c.ln_g; p.ln_b = c.ln_b; p.eps = c.eps;
    if (c.colsum != nullptr) {
        ACMI_REQUIRE(c.a_mode == ACMI_A_TILED, "acmi_linear: colsum needs a tiled activation");
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
    ACMI_REQUIRE(c.a_shift == nullptr || c.colsum != nullptr, "acmi_linear: a_shift needs the folded LayerNorm (colsum)");
    ACMI_REQUIRE(c.mean_out == nullptr || c.colsum != nullptr, "acmi_linear: mean_out needs the folded LayerNorm (colsum)");
    ACMI_REQUIRE(c.xt_shift == nullptr || c.xt_hi != nullptr, "acmi_linear: xt_shift needs xt_hi");
    p.a_shift = c.a_shift; p.xt_shift = c.xt_shift; p.mean_out = c.mean_out;
    p.stats_out = c.stats_out; p.ksplit = c.ksplit; p.w_half = c.w_half;
    p.w = c.w; p.bias = c.bias; p.residual = c.residual; p.out = c.out; p.out_mode = c.out_mode; p.act = c.act;
    p.M = c.M; p.N = c.N; p.K = c.K;
    return ACMI_OK;
}

extern "C" int acmi_linear_ex(const acmi_linear_desc* dsc, void* stream) {
    ACMI_REQUIRE(dsc != nullptr, "acmi_linear_ex: null descriptor");
    LinArgs p = {};
    int rc = desc_to_args(*dsc, p);
    if (rc) return rc;
    return acmi_launch_lin(p, dsc->wdtype, (hipStream_t)stream);
}

extern "C" int acmi_linear_pair(const acmi_linear_desc* plain, const acmi_linear_desc* xcat, void* stream) {
    ACMI_REQUIRE(plain != nullptr && xcat != nullptr, "acmi_linear_pair: null descriptor");
    ACMI_REQUIRE(plain->wdtype == xcat->wdtype && plain->a_mode == ACMI_A_TILED && xcat->a_mode == ACMI_A_TILED,
                 "acmi_linear_pair: both GEMMs take tiled activations of one element type");
    LinArgs p0 = {}, p1 = {};
    int rc;
    if ((rc = desc_to_args(*plain, p0)) || (rc = desc_to_args(*xcat, p1))) return rc;
    ACMI_REQUIRE(p0.M > 0 && p0.N > 0 && p0.K > 0 && p1.N > 0 && p1.K > 0, "acmi_linear_pair: empty problem");
    return acmi_launch_pair(p0, p1, plain->wdtype, (hipStream_t)stream);
}
#endif  // ACMI_GEMM_MAIN
