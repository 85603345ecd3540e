// Residual vector quantizer kernels for gfx950.
//
//   rvq_encode_kernel: nearest-codebook search, all levels in one launch.  One wave owns 16 latent
//   rows for the whole cascade: the rows live in registers in MFMA A-operand layout, every level
//   streams the codebook (L2-resident, 1 MB) through v_mfma_f32_16x16x4_f32 (exact f32 FMA chain),
//   keeps a running first-index argmax of  -(||x||^2 - 2 x.e + ||e||^2)  per lane, reduces it across
//   the 16 lanes of a row group, then subtracts the chosen code from the registers.
//   Reference: audiocraft/quantization/core_vq.py:164-172 (quantize), :386-396 (cascade).
//   rvq_decode_kernel: sum of embedding rows in level order + transpose to conv layout
//   (core_vq.py:398-404, :177-179, :295-298).
#include "acmi_common.h"

#include <math.h>
#include <stdlib.h>

__global__ void rvq_norms_kernel(const float* __restrict__ cb, float* __restrict__ norms, int rows, int D) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* e = cb + (size_t)r * D;
    float s = 0.f;
    for (int d = 0; d < D; ++d) s += e[d] * e[d];
    norms[r] = s;
}

extern "C" int acmi_rvq_codebook_norms(const float* codebooks, float* norms, int K, int bins, int D, void* stream) {
    ACMI_REQUIRE(K > 0 && bins > 0 && D > 0, "acmi_rvq_codebook_norms: bad shape");
    const int rows = K * bins;
    hipLaunchKernelGGL(rvq_norms_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, codebooks, norms,
                       rows, D);
    return acmi_check_launch("rvq_norms_kernel");
}

// SP = 1: every wave owns 16 rows and scans the whole codebook of a level (long inputs: thousands of waves hide the codebook
// reads behind each other).  SP = 4 (few rows, e.g. one 10 s clip = 750 frames = 47 waves): the four waves of a workgroup share
// 16 rows and scan a quarter of the codebook each, then merge their (best, index) pairs in code order -- a level costs a
// quarter of the dependent codebook reads (65 -> ~20 us per level at 32 levels x 1024 codes x 128 dims, B T = 750).
template <int D, int SP>
__global__ __launch_bounds__(256) void rvq_encode_kernel(const float* __restrict__ latents,
                                                         const float* __restrict__ codebooks,
                                                         const float* __restrict__ norms, int64_t* __restrict__ codes,
                                                         int B, int T, int K, int bins) {
    constexpr int NKK = D / 16;  // 16-wide k groups; lane (row, kg) owns k = kk*16 + kg*4 + j
    __shared__ float s_x2[4][16];
    __shared__ int s_idx[4][16];
    __shared__ float s_best[4][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nl = lane & 15, kg = lane >> 4;
    const long long R = (SP == 1 ? ((long long)blockIdx.x * 4 + wave) : (long long)blockIdx.x) * 16 + nl;  // A-layout row of this lane
    const long long NR = (long long)B * T;
    const bool rvalid = R < NR;
    const int rb = rvalid ? (int)(R / T) : 0, rt = rvalid ? (int)(R % T) : 0;
    // this wave's slice of every codebook
    const int span = SP == 1 ? bins : ((bins + 16 * SP - 1) / (16 * SP)) * 16;
    const int c_lo = SP == 1 ? 0 : min(bins, wave * span), c_hi = SP == 1 ? bins : min(bins, c_lo + span);

    float xr[NKK * 4];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = kk * 16 + kg * 4 + j;
            xr[kk * 4 + j] = rvalid ? latents[((size_t)rb * D + k) * T + rt] : 0.f;
        }

    for (int q = 0; q < K; ++q) {
        const float* cb = codebooks + (size_t)q * bins * D;
        const float* nq = norms + (size_t)q * bins;
        // ||x||^2 of the (residual) rows, redistributed to the C layout (row = (lane>>4)*4 + i)
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NKK * 4; ++i) s += xr[i] * xr[i];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (kg == 0) s_x2[wave][nl] = s;
        __syncthreads();
        float x2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) x2[i] = s_x2[wave][kg * 4 + i];

        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bidx[4] = {0, 0, 0, 0};
        // the 16 codes of the next iteration are requested before the MFMAs of the current one
        float4 ev[NKK], en[NKK];
        float e2 = 0.f, e2n = 0.f;
        auto fetch = [&](int c0, float4 (&dst)[NKK], float& n2) {
            const int c = c0 + nl;
            const bool ok = c < c_hi;
            const float* erow = cb + (size_t)(ok ? c : 0) * D + kg * 4;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) dst[kk] = *reinterpret_cast<const float4*>(erow + kk * 16);
            n2 = ok ? nq[c] : 0.f;
        };
        if (c_lo < c_hi) fetch(c_lo, ev, e2);
        for (int c0 = c_lo; c0 < c_hi; c0 += 16) {
            if (c0 + 16 < c_hi) fetch(c0 + 16, en, e2n);
            const int c = c0 + nl;
            const bool cvalid = c < c_hi;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xr[kk * 4 + 0], ev[kk].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xr[kk * 4 + 1], ev[kk].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xr[kk * 4 + 2], ev[kk].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xr[kk * 4 + 3], ev[kk].w, acc, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // -(x2 - 2*dot + e2), evaluated left to right without contraction like the reference
                const float t1 = __fsub_rn(x2[i], __fmul_rn(2.0f, acc[i]));
                const float dist = -__fadd_rn(t1, e2);
                if (cvalid && dist > best[i]) { best[i] = dist; bidx[i] = c; }
            }
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) ev[kk] = en[kk];
            e2 = e2n;
        }
        // first-index argmax across the 16 lanes (codes c = nl mod 16) of each row group
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float ov = __shfl_xor(best[i], off, 64);
                const int oi = __shfl_xor(bidx[i], off, 64);
                if (ov > best[i] || (ov == best[i] && oi < bidx[i])) { best[i] = ov; bidx[i] = oi; }
            }
        }
        if (nl == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { s_idx[wave][kg * 4 + i] = bidx[i]; s_best[wave][kg * 4 + i] = best[i]; }
        }
        __syncthreads();
        int my = s_idx[wave][nl];
        if (SP > 1) {   // merge the slices in code order: a later slice wins only when strictly better (first index on ties)
            float bv = s_best[0][nl];
            my = s_idx[0][nl];
#pragma unroll
            for (int w = 1; w < SP; ++w) {
                const float ov = s_best[w][nl];
                if (ov > bv) { bv = ov; my = s_idx[w][nl]; }
            }
        }
        if (rvalid && kg == 0 && (SP == 1 || wave == 0)) codes[((size_t)rb * K + q) * T + rt] = (int64_t)my;
        const float* erow = cb + (size_t)my * D + kg * 4;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            const float4 e = *reinterpret_cast<const float4*>(erow + kk * 16);
            xr[kk * 4 + 0] -= e.x; xr[kk * 4 + 1] -= e.y; xr[kk * 4 + 2] -= e.z; xr[kk * 4 + 3] -= e.w;
        }
        __syncthreads();
    }
}

extern "C" int acmi_rvq_encode(const float* latents, const float* codebooks, const float* norms, int64_t* codes, int B,
                               int D, int T, int K, int bins, void* stream) {
    ACMI_REQUIRE(B >= 0 && T >= 0 && K > 0 && bins > 0, "acmi_rvq_encode: bad shape");
    if ((long long)B * T == 0) return ACMI_OK;
    const long long rows = (long long)B * T;
    // few rows: four waves per 16 rows, each scanning a quarter of the codebook (ACMI_RVQ_SPLIT=0/1 forces a form)
    static const int force = getenv("ACMI_RVQ_SPLIT") ? atoi(getenv("ACMI_RVQ_SPLIT")) : -1;
    const bool split = force >= 0 ? force != 0 : rows <= 8192;
    dim3 grid((unsigned)(split ? (rows + 15) / 16 : (rows + 63) / 64)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define ACMI_RVQ_CASE(DD)                                                                                          \
    case DD:                                                                                                       \
        if (split) hipLaunchKernelGGL((rvq_encode_kernel<DD, 4>), grid, block, 0, st, latents, codebooks, norms, codes, B, T, K, bins); \
        else hipLaunchKernelGGL((rvq_encode_kernel<DD, 1>), grid, block, 0, st, latents, codebooks, norms, codes, B, T, K, bins); \
        break;
    switch (D) {
        ACMI_RVQ_CASE(16)
        ACMI_RVQ_CASE(32)
        ACMI_RVQ_CASE(64)
        ACMI_RVQ_CASE(128)
        ACMI_RVQ_CASE(256)
        default:
            acmi_set_error("acmi_rvq_encode: dimension %d unsupported (16,32,64,128,256)", D);
            return ACMI_EINVAL;
    }
#undef ACMI_RVQ_CASE
    return acmi_check_launch("rvq_encode_kernel");
}

__global__ __launch_bounds__(256) void rvq_decode_kernel(const int64_t* __restrict__ codes,
                                                         const float* __restrict__ codebooks, float* __restrict__ out,
                                                         int D, int T, int K, int bins) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dg = threadIdx.x >> 6;
    if (t >= T) return;
    int idx[32];
    for (int q = 0; q < K; ++q) {
        long long c = codes[((size_t)b * K + q) * T + t];
        c = c < 0 ? 0 : (c >= bins ? bins - 1 : c);
        idx[q] = (int)c;
    }
    for (int d = dg; d < D; d += 4) {
        float v = 0.f;
        for (int q = 0; q < K; ++q) v += codebooks[((size_t)q * bins + idx[q]) * D + d];
        out[((size_t)b * D + d) * T + t] = v;
    }
}

extern "C" int acmi_rvq_decode(const int64_t* codes, const float* codebooks, float* out, int B, int D, int T, int K,
                               int bins, void* stream) {
    ACMI_REQUIRE(K > 0 && K <= 32, "acmi_rvq_decode: K=%d unsupported (1..32)", K);
    ACMI_REQUIRE(B >= 0 && T >= 0 && D > 0 && bins > 0, "acmi_rvq_decode: bad shape");
    if ((long long)B * T == 0) return ACMI_OK;
    hipLaunchKernelGGL(rvq_decode_kernel, dim3((T + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, codes, codebooks,
                       out, D, T, K, bins);
    return acmi_check_launch("rvq_decode_kernel");
}
