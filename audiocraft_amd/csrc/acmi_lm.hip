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

// =====================================================================================================
// skinny GEMM
// =====================================================================================================

struct LinArgs {
    const void* a; int a_bf16;
    const float* ln_g; const float* ln_b; float eps;
    const void* w;
    const float* bias;
    const float* residual;
    void* out; int out_bf16; int act;
    int M, N, K;
    int mode;  // 0 plain, 1 QKV scatter
    float* q_out; void* k_cache; void* v_cache; int kv_bf16; int H, hd, Tcap, d; const int* pos;
};

template <typename WT> struct WTr;
template <> struct WTr<float> { static constexpr int EPL = 16; };    // elements per lane per super-step (64 B)
template <> struct WTr<bf16_t> { static constexpr int EPL = 32; };

template <typename WT>
__device__ __forceinline__ void load_w(const WT* wrow, int k, int K, bool valid, uint4 (&wv)[4]) {
    constexpr int EPV = 16 / (int)sizeof(WT);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int kk = k + j * EPV;
        if (valid && kk < K) wv[j] = *reinterpret_cast<const uint4*>(wrow + kk);
        else wv[j] = make_uint4(0u, 0u, 0u, 0u);
    }
}

template <int EPL>
__device__ __forceinline__ void load_a(const void* a, int a_bf16, size_t row_off, int k, int K, bool valid,
                                       float (&xa)[EPL]) {
#pragma unroll
    for (int g = 0; g < EPL / 8; ++g) {
        const int kk = k + 8 * g;
        float t[8];
        if (valid && kk < K) {
            if (a_bf16) ld8(reinterpret_cast<const bf16_t*>(a) + row_off + kk, t);
            else ld8(reinterpret_cast<const float*>(a) + row_off + kk, t);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) xa[8 * g + e] = t[e];
    }
}

// one super-step of MFMAs: acc[16x16] += A[16 x KSS] * W[16 x KSS]^T with the k-permutation
// "lane (row, kg) owns k0 + kg*EPL + [0, EPL)" applied identically to A and W.
__device__ __forceinline__ void mma_ss(const float (&xa)[32], const uint4 (&wv)[4], f32x4& acc, bf16_t) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        union { uint4 u; bf16x8 v; } A, B;
        A.u = make_uint4(pack_bf16x2(xa[8 * j + 0], xa[8 * j + 1]), pack_bf16x2(xa[8 * j + 2], xa[8 * j + 3]),
                         pack_bf16x2(xa[8 * j + 4], xa[8 * j + 5]), pack_bf16x2(xa[8 * j + 6], xa[8 * j + 7]));
        B.u = wv[j];
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A.v, B.v, acc, 0, 0, 0);
    }
}
__device__ __forceinline__ void mma_ss(const float (&xa)[16], const uint4 (&wv)[4], f32x4& acc, float) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float wf[4] = {__uint_as_float(wv[j].x), __uint_as_float(wv[j].y), __uint_as_float(wv[j].z),
                             __uint_as_float(wv[j].w)};
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[4 * j + e], wf[e], acc, 0, 0, 0);
    }
}

__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <typename WT, bool HAS_LN>
__global__ __launch_bounds__(1024) void lin_kernel(const LinArgs p) {
    constexpr int EPL = WTr<WT>::EPL;
    constexpr int KSS = 4 * EPL;
    constexpr int SSMAX = sizeof(WT) == 2 ? 1 : 2;  // LN path: the wave's slice of the row lives in registers
    __shared__ float smem[16 * 256 + 16 * 16];
    float* red = smem;
    float* stat = smem + 16 * 256;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int nl = lane & 15, kg = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int n = n0 + nl;
    const bool nvalid = n < p.N;
    const int K = p.K;
    const int nss = (K + KSS - 1) / KSS;
    const WT* wrow = reinterpret_cast<const WT*>(p.w) + (size_t)(nvalid ? n : 0) * K;
    const int tpos = (p.mode == 1) ? *p.pos : 0;

    for (int m0 = 0; m0 < p.M; m0 += 16) {
        const int m = m0 + nl;
        const bool mvalid = m < p.M;
        const size_t arow = (size_t)(mvalid ? m : 0) * K;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};

        if (HAS_LN) {
            uint4 wv[SSMAX][4];
            float xr[SSMAX][EPL];
#pragma unroll
            for (int i = 0; i < SSMAX; ++i) {
                const int ss = wave + i * nw;
                const int k = ss * KSS + kg * EPL;
                load_w<WT>(wrow, k, K, nvalid && ss < nss, wv[i]);
                load_a<EPL>(p.a, p.a_bf16, arow, k, K, mvalid && ss < nss, xr[i]);
            }
            // ---- LayerNorm statistics of row m (two pass: mean, then centred second moment)
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < SSMAX; ++i)
#pragma unroll
                for (int e = 0; e < EPL; ++e) s += xr[i][e];
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            if (kg == 0) stat[wave * 16 + nl] = s;
            __syncthreads();
            float tot = 0.f;
            for (int w = 0; w < nw; ++w) tot += stat[w * 16 + nl];
            const float mean = tot / (float)K;
            __syncthreads();
            float s2 = 0.f;
#pragma unroll
            for (int i = 0; i < SSMAX; ++i) {
                const int k = (wave + i * nw) * KSS + kg * EPL;
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    const float dlt = xr[i][e] - mean;
                    s2 += (k + e < K && wave + i * nw < nss) ? dlt * dlt : 0.f;
                }
            }
            s2 += __shfl_xor(s2, 16, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (kg == 0) stat[wave * 16 + nl] = s2;
            __syncthreads();
            float tot2 = 0.f;
            for (int w = 0; w < nw; ++w) tot2 += stat[w * 16 + nl];
            const float rstd = 1.0f / sqrtf(tot2 / (float)K + p.eps);
            __syncthreads();
#pragma unroll
            for (int i = 0; i < SSMAX; ++i) {
                const int ss = wave + i * nw;
                const int k = ss * KSS + kg * EPL;
                if (ss < nss) {
#pragma unroll
                    for (int e = 0; e < EPL; e += 4) {
                        if (k + e < K) {
                            const float4 g = *reinterpret_cast<const float4*>(p.ln_g + k + e);
                            const float4 bb = *reinterpret_cast<const float4*>(p.ln_b + k + e);
                            xr[i][e + 0] = (xr[i][e + 0] - mean) * rstd * g.x + bb.x;
                            xr[i][e + 1] = (xr[i][e + 1] - mean) * rstd * g.y + bb.y;
                            xr[i][e + 2] = (xr[i][e + 2] - mean) * rstd * g.z + bb.z;
                            xr[i][e + 3] = (xr[i][e + 3] - mean) * rstd * g.w + bb.w;
                        } else {
                            xr[i][e + 0] = xr[i][e + 1] = xr[i][e + 2] = xr[i][e + 3] = 0.f;
                        }
                    }
                    if (!mvalid) {
#pragma unroll
                        for (int e = 0; e < EPL; ++e) xr[i][e] = 0.f;
                    }
                    mma_ss(xr[i], wv[i], acc, WT());
                }
            }
        } else {
#pragma unroll 2
            for (int ss = wave; ss < nss; ss += nw) {
                const int k = ss * KSS + kg * EPL;
                uint4 wv[4];
                float xa[EPL];
                load_w<WT>(wrow, k, K, nvalid, wv);
                load_a<EPL>(p.a, p.a_bf16, arow, k, K, mvalid, xa);
                mma_ss(xa, wv, acc, WT());
            }
        }

        // ---- deterministic cross-wave reduction + epilogue
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave * 256 + lane * 4 + r] = acc[r];
        __syncthreads();
        for (int t = threadIdx.x; t < 256; t += blockDim.x) {
            const int nn = t & 15, mm = t >> 4;
            const int idx = (((mm >> 2) * 16 + nn) << 2) + (mm & 3);
            float v = 0.f;
            for (int w = 0; w < nw; ++w) v += red[w * 256 + idx];
            const int gm = m0 + mm, gn = n0 + nn;
            if (gm < p.M && gn < p.N) {
                if (p.bias) v += p.bias[gn];
                if (p.mode == 1) {
                    const int part = gn / p.d, f = gn - part * p.d;
                    if (part == 0) {
                        p.q_out[(size_t)gm * p.d + f] = v;
                    } else {
                        const int h = f / p.hd, dd = f - h * p.hd;
                        const size_t ci = (((size_t)gm * p.H + h) * p.Tcap + tpos) * p.hd + dd;
                        void* cache = part == 1 ? p.k_cache : p.v_cache;
                        if (p.kv_bf16) reinterpret_cast<bf16_t*>(cache)[ci] = f32_to_bf16(v);
                        else reinterpret_cast<float*>(cache)[ci] = v;
                    }
                } else {
                    if (p.act == 1) v = gelu_exact(v);
                    const size_t oi = (size_t)gm * p.N + gn;
                    if (p.residual) v += p.residual[oi];
                    if (p.out_bf16) reinterpret_cast<bf16_t*>(p.out)[oi] = f32_to_bf16(v);
                    else reinterpret_cast<float*>(p.out)[oi] = v;
                }
            }
        }
        __syncthreads();
    }
}

static int launch_lin(const LinArgs& a, int wdtype, hipStream_t st) {
    ACMI_REQUIRE(a.K % 8 == 0, "acmi_linear: K=%d must be a multiple of 8", a.K);
    ACMI_REQUIRE(a.M > 0 && a.N > 0, "acmi_linear: empty problem M=%d N=%d", a.M, a.N);
    const bool ln = a.ln_g != nullptr;
    const int kss = (wdtype == ACMI_BF16) ? 128 : 64;
    const int nss = (a.K + kss - 1) / kss;
    int nw;
    if (ln) {
        const int ssmax = (wdtype == ACMI_BF16) ? 1 : 2;
        ACMI_REQUIRE(nss <= 16 * ssmax, "acmi_linear: LayerNorm-fused K=%d too large (max %d)", a.K, 16 * ssmax * kss);
        nw = nss <= 16 ? nss : (nss + 1) / 2;
    } else {
        nw = nss < 16 ? nss : 16;
    }
    if (nw < 1) nw = 1;
    dim3 grid((a.N + 15) / 16), block(nw * 64);
    if (wdtype == ACMI_BF16) {
        if (ln) hipLaunchKernelGGL((lin_kernel<bf16_t, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((lin_kernel<bf16_t, false>), grid, block, 0, st, a);
    } else {
        if (ln) hipLaunchKernelGGL((lin_kernel<float, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((lin_kernel<float, false>), grid, block, 0, st, a);
    }
    return acmi_check_launch("lin_kernel");
}

extern "C" int acmi_linear(const void* a, int a_dtype, const float* ln_g, const float* ln_b, float eps, const void* w,
                           int wdtype, const float* bias, const float* residual, void* out, int out_dtype, int act, int M,
                           int N, int K, void* stream) {
    LinArgs p = {};
    p.a = a; p.a_bf16 = a_dtype == ACMI_BF16;
    p.ln_g = ln_g; p.ln_b = ln_b; p.eps = eps;
    p.w = w; p.bias = bias; p.residual = residual; p.out = out; p.out_bf16 = out_dtype == ACMI_BF16; p.act = act;
    p.M = M; p.N = N; p.K = K; p.mode = 0;
    return launch_lin(p, wdtype, (hipStream_t)stream);
}

// =====================================================================================================
// single-query attention over a KV cache
// =====================================================================================================

template <typename KT, int HD>
__global__ __launch_bounds__(256) void attn_decode_kernel(const float* __restrict__ q, const KT* __restrict__ kc,
                                                          const KT* __restrict__ vc, float* __restrict__ out, int H,
                                                          int Tcap, int len_arg, const int* len_dev, int len_bias,
                                                          float scale) {
    constexpr int LPP = HD / 8;    // lanes per position (8 dims each)
    constexpr int PPI = 64 / LPP;  // positions covered by one load instruction of a wave
    constexpr int NI = 8;          // load instructions in flight per chunk
    constexpr int CH = NI * PPI;
    const int h = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane % LPP, pp = lane / LPP;
    const int len = len_dev ? (*len_dev + len_bias) : len_arg;

    float qv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[e] = q[((size_t)b * H + h) * HD + c * 8 + e];
    const KT* kb = kc + ((size_t)b * H + h) * Tcap * HD + c * 8;
    const KT* vb = vc + ((size_t)b * H + h) * Tcap * HD + c * 8;

    float m = -INFINITY, l = 0.f, o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;

    for (int t0 = wave * CH; t0 < len; t0 += 4 * CH) {
        float s[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int t = t0 + i * PPI + pp;
            float part = 0.f;
            if (t < len) {
                float kf[8];
                ld8(kb + (size_t)t * HD, kf);
#pragma unroll
                for (int e = 0; e < 8; ++e) part = fmaf(qv[e], kf[e], part);
            }
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
        for (int e = 0; e < 8; ++e) o[e] *= alpha;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int t = t0 + i * PPI + pp;
            if (t < len) {
                const float pr = expf(s[i] - m_new);
                l += pr;
                float vf[8];
                ld8(vb + (size_t)t * HD, vf);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = fmaf(pr, vf[e], o[e]);
            }
        }
        m = m_new;
    }
#pragma unroll
    for (int off = LPP; off < 64; off <<= 1) {
        l += __shfl_xor(l, off, 64);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += __shfl_xor(o[e], off, 64);
    }
    __shared__ float sm_o[4][HD];
    __shared__ float sm_m[4], sm_l[4];
    if (lane < LPP) {
#pragma unroll
        for (int e = 0; e < 8; ++e) sm_o[wave][c * 8 + e] = o[e];
    }
    if (lane == 0) { sm_m[wave] = m; sm_l[wave] = l; }
    __syncthreads();
    if (threadIdx.x < HD) {
        float M = fmaxf(fmaxf(sm_m[0], sm_m[1]), fmaxf(sm_m[2], sm_m[3]));
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float f = (sm_m[w] == -INFINITY) ? 0.f : expf(sm_m[w] - M);
            num += f * sm_o[w][threadIdx.x];
            den += f * sm_l[w];
        }
        out[((size_t)b * H + h) * HD + threadIdx.x] = num / den;
    }
}

template <typename KT>
static int launch_attn_t(const float* q, const void* kc, const void* vc, float* out, int Beff, int H, int hd, int Tcap,
                         int len, const int* len_dev, int len_bias, hipStream_t st) {
    const float scale = 1.0f / sqrtf((float)hd);
    dim3 grid(H, Beff), block(256);
    const KT* k = reinterpret_cast<const KT*>(kc);
    const KT* v = reinterpret_cast<const KT*>(vc);
#define ACMI_ATTN_CASE(HD)                                                                                      \
    case HD:                                                                                                    \
        hipLaunchKernelGGL((attn_decode_kernel<KT, HD>), grid, block, 0, st, q, k, v, out, H, Tcap, len, len_dev, \
                           len_bias, scale);                                                                    \
        break;
    switch (hd) {
        ACMI_ATTN_CASE(8)
        ACMI_ATTN_CASE(16)
        ACMI_ATTN_CASE(32)
        ACMI_ATTN_CASE(64)
        ACMI_ATTN_CASE(128)
        default:
            acmi_set_error("acmi_attn_decode: head dim %d unsupported (8,16,32,64,128)", hd);
            return ACMI_EINVAL;
    }
#undef ACMI_ATTN_CASE
    return acmi_check_launch("attn_decode_kernel");
}

extern "C" int acmi_attn_decode(const float* q, const void* k_cache, const void* v_cache, int kvdtype, float* out,
                                int Beff, int H, int hd, int Tcap, int len, const int* len_dev, int len_bias,
                                void* stream) {
    ACMI_REQUIRE(Beff > 0 && H > 0 && Tcap > 0, "acmi_attn_decode: bad shape");
    ACMI_REQUIRE(len_dev != nullptr || (len > 0 && len <= Tcap), "acmi_attn_decode: len=%d out of (0, %d]", len, Tcap);
    if (kvdtype == ACMI_BF16)
        return launch_attn_t<bf16_t>(q, k_cache, v_cache, out, Beff, H, hd, Tcap, len, len_dev, len_bias,
                                     (hipStream_t)stream);
    return launch_attn_t<float>(q, k_cache, v_cache, out, Beff, H, hd, Tcap, len, len_dev, len_bias,
                                (hipStream_t)stream);
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

struct EmbedArgs {
    const void* emb[16]; int w_bf16;
    const int64_t* gen_sequence; int B, K, S, card;
    const float* prepend; int P;
    const float* pos_freq; float pos_scale;
    const int* pos;
    float* x; int d;
};

__global__ __launch_bounds__(256) void embed_kernel(const EmbedArgs p) {
    const int m = blockIdx.x;
    const int g = *p.pos;
    const int half = p.d / 2;
    const int b = m % p.B;
    for (int cch = threadIdx.x; cch < p.d; cch += blockDim.x) {
        float v;
        if (g < p.P) {
            v = p.prepend[((size_t)m * p.P + g) * p.d + cch];
        } else {
            const int sidx = g - p.P;
            v = 0.f;
            for (int k = 0; k < p.K; ++k) {
                int64_t tok = p.gen_sequence[((size_t)b * p.K + k) * p.S + sidx];
                if (tok < 0 || tok > p.card) tok = p.card;  // never happens for a well-formed sequence
                const size_t ei = (size_t)tok * p.d + cch;
                v += p.w_bf16 ? bf16_to_f32(reinterpret_cast<const bf16_t*>(p.emb[k])[ei])
                              : reinterpret_cast<const float*>(p.emb[k])[ei];
            }
        }
        const int i = cch < half ? cch : cch - half;
        const float phase = (float)g / p.pos_freq[i];
        const float pe = cch < half ? cosf(phase) : sinf(phase);
        p.x[(size_t)m * p.d + cch] = v + p.pos_scale * pe;
    }
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
    uint64_t seed; uint64_t step; const int* pos;  // step counter = step + *pos when pos != NULL
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
    const int k = blockIdx.x, b = blockIdx.y;
    const int card = p.card;
    const float* cond = p.logits + ((size_t)b * p.K + k) * card;
    const float* uncond = p.logits + ((size_t)(p.B + b) * p.K + k) * card;
    for (int i = threadIdx.x; i < card; i += blockDim.x) {
        float v = cond[i];
        if (p.use_cfg) { const float u = uncond[i]; v = u + (v - u) * p.cfg_coef; }
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
                if (threadIdx.x == 0) {
                    unsigned acc = 0u; int dsel = 0;
                    for (int dgt = 255; dgt >= 0; --dgt) {
                        if (acc + hist[dgt] >= remaining) { dsel = dgt; break; }
                        acc += hist[dgt];
                    }
                    sel[0] = (unsigned)dsel; sel[1] = remaining - acc;
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
        const uint64_t stepc = p.step + (p.pos ? (uint64_t)*p.pos : 0ull);
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
            const int offset = (*p.pos - p.P) + 1;  // sequence step being filled (lm.py:540-562)
            if (offset >= 0 && offset < p.S) {
                const int64_t tok = p.seq_mask[(size_t)k * p.S + offset] ? (int64_t)token : (int64_t)p.card;
                int64_t* dst = p.gen_sequence + ((size_t)b * p.K + k) * p.S + offset;
                if (*dst == -1) *dst = tok;
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

__global__ void advance_kernel(int* pos) { if (threadIdx.x == 0 && blockIdx.x == 0) pos[0] += 1; }

// =====================================================================================================
// one decode position
// =====================================================================================================

extern "C" int acmi_lm_step(const acmi_lm_model* m, const acmi_lm_state* s, int mode, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    ACMI_REQUIRE(m && s, "acmi_lm_step: null argument");
    const int d = m->dim, H = m->num_heads, hd = d / H, M = s->Beff;
    ACMI_REQUIRE(d % H == 0 && d % 8 == 0 && m->ffn_dim % 8 == 0, "acmi_lm_step: bad dims d=%d H=%d", d, H);
    ACMI_REQUIRE(m->n_q <= 16, "acmi_lm_step: n_q=%d > 16", m->n_q);
    ACMI_REQUIRE(s->Beff == (s->use_cfg ? 2 * s->B : s->B), "acmi_lm_step: Beff/B mismatch");
    const int wbf = m->wdtype == ACMI_BF16, kvbf = m->kvdtype == ACMI_BF16;
    int rc;

    EmbedArgs e = {};
    for (int k = 0; k < m->n_q; ++k) e.emb[k] = m->emb[k];
    e.w_bf16 = wbf; e.gen_sequence = s->gen_sequence; e.B = s->B; e.K = m->n_q; e.S = s->S; e.card = m->card;
    e.prepend = s->prepend; e.P = s->prepend ? s->n_prepend : 0; e.pos_freq = m->pos_freq;
    e.pos_scale = m->positional_scale; e.pos = s->pos; e.x = s->x; e.d = d;
    hipLaunchKernelGGL(embed_kernel, dim3(M), dim3(256), 0, st, e);
    if ((rc = acmi_check_launch("embed_kernel"))) return rc;

    for (int li = 0; li < m->num_layers; ++li) {
        const acmi_lm_layer& L = m->layers[li];
        // x -> LN1 -> QKV ; K,V appended in place at position g, q to scratch
        LinArgs a = {};
        a.a = s->x; a.ln_g = L.ln1_g; a.ln_b = L.ln1_b; a.eps = m->eps; a.w = L.w_qkv;
        a.M = M; a.N = 3 * d; a.K = d; a.mode = 1;
        a.q_out = s->q; a.k_cache = L.k_cache; a.v_cache = L.v_cache; a.kv_bf16 = kvbf;
        a.H = H; a.hd = hd; a.Tcap = s->Tmax; a.d = d; a.pos = s->pos;
        if ((rc = launch_lin(a, m->wdtype, st))) return rc;
        if ((rc = acmi_attn_decode(s->q, L.k_cache, L.v_cache, m->kvdtype, s->att, M, H, hd, s->Tmax, 0, s->pos, 1,
                                   stream)))
            return rc;
        if ((rc = acmi_linear(s->att, ACMI_F32, nullptr, nullptr, 0.f, L.w_out, m->wdtype, nullptr, s->x, s->x, ACMI_F32, 0, M,
                              d, d, stream)))
            return rc;
        if (m->cross_attention) {
            ACMI_REQUIRE(s->Lc > 0 && L.ck_cache && L.cv_cache, "acmi_lm_step: cross-attention caches missing");
            if ((rc = acmi_linear(s->x, ACMI_F32, L.lnc_g, L.lnc_b, m->eps, L.w_cq, m->wdtype, nullptr, nullptr, s->q, ACMI_F32,
                                  0, M, d, d, stream)))
                return rc;
            if ((rc = acmi_attn_decode(s->q, L.ck_cache, L.cv_cache, m->kvdtype, s->att, M, H, hd, s->Lc, s->Lc,
                                       nullptr, 0, stream)))
                return rc;
            if ((rc = acmi_linear(s->att, ACMI_F32, nullptr, nullptr, 0.f, L.w_cout, m->wdtype, nullptr, s->x, s->x, ACMI_F32, 0,
                                  M, d, d, stream)))
                return rc;
        }
        const int hdt = wbf ? ACMI_BF16 : ACMI_F32;
        if ((rc = acmi_linear(s->x, ACMI_F32, L.ln2_g, L.ln2_b, m->eps, L.w_ff1, m->wdtype, nullptr, nullptr, s->hidden, hdt, 1,
                              M, m->ffn_dim, d, stream)))
            return rc;
        if ((rc = acmi_linear(s->hidden, hdt, nullptr, nullptr, 0.f, L.w_ff2, m->wdtype, nullptr, s->x, s->x, ACMI_F32, 0, M, d,
                              m->ffn_dim, stream)))
            return rc;
    }
    if (mode == ACMI_STEP_DECODE) {
        if ((rc = acmi_linear(s->x, ACMI_F32, m->out_norm_g, m->out_norm_b, m->eps, m->w_head, m->wdtype, nullptr, nullptr,
                              s->logits, ACMI_F32, 0, M, m->n_q * m->card, d, stream)))
            return rc;
        SampleArgs a = {};
        a.logits = s->logits; a.B = s->B; a.K = m->n_q; a.card = m->card; a.use_cfg = s->use_cfg;
        a.cfg_coef = s->cfg_coef; a.use_sampling = s->use_sampling; a.temp = s->temp; a.top_k = s->top_k;
        a.top_p = s->top_p; a.seed = s->seed; a.step = 0; a.pos = s->pos; a.mixed_out = s->step_logits;
        a.gen_sequence = s->gen_sequence; a.seq_mask = s->seq_mask; a.S = s->S; a.P = s->prepend ? s->n_prepend : 0;
        if ((rc = launch_sample(a, st))) return rc;
    }
    hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(64), 0, st, s->pos);
    return acmi_check_launch("advance_kernel");
}
