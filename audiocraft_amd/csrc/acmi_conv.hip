// SEANet convolution kernels for gfx950: implicit-GEMM Conv1d / ConvTranspose1d in exact f32 on the
// matrix cores (v_mfma_f32_32x32x2_f32 == k-ordered fmaf chain), plus the LSTM recurrence.
//
// conv_mfma_kernel: GEMM rows = output channels (for transposed convs: (channel, phase) pairs of the
// polyphase decomposition), GEMM cols = output time steps, K = Cin * ksize.  A 64 x 64 output tile per
// 256-thread workgroup (4 waves, 2 x 2 of 32 x 32); per K chunk the weight tile and the input span
// (with halo) are staged through LDS with coalesced reads; padding (zero / reflect, asymmetric,
// "extra" right padding), the ELU that precedes every SEANet conv, bias, the resnet skip add and the
// transposed-conv trim + phase interleave are all folded into the load / store index math.
// The input span is stored de-interleaved by stride phase so that the MFMA B-operand reads of a
// strided conv hit 32 consecutive LDS banks.
// Reference: audiocraft/modules/conv.py:47-88,185-243; audiocraft/modules/seanet.py:16-60.
#include "acmi_common.h"

#include <math.h>

struct ConvArgs {
    acmi_conv_desc d;
    const float* x; const float* w; const float* bias; const float* res; float* y;
    int Tq;    // GEMM columns per batch item
    int CIC;   // input channels per K chunk
    int KCE;   // CIC * ksize rounded up to even
    int KCP;   // LDS row pitch of the weight tile (odd)
    int LP;    // LDS row pitch of one (channel, phase) input row
};

__device__ __forceinline__ float conv_fetch(const ConvArgs& a, const float* xrow, int pos) {
    const acmi_conv_desc& d = a.d;
    int src = pos;
    if (d.pad_mode == ACMI_PAD_REFLECT) {
        if (src < 0) src = -src;
        if (src >= d.reflect_len) src = 2 * (d.reflect_len - 1) - src;
    }
    if (src < 0 || src >= d.Tin) return 0.f;
    float v = xrow[src];
    if (d.elu_in) v = v > 0.f ? v : d.elu_alpha * expm1f(v);
    return v;
}

__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const acmi_conv_desc& d = a.d;
    const int s = d.stride, ks = d.ksize;
    float* Ws = smem;                                   // [64][KCP]
    float* Xs = Ws + 64 * a.KCP;                        // [CIC][s][LP]
    int* koff = reinterpret_cast<int*>(Xs + a.CIC * s * a.LP);  // [KCE]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, kk = lane >> 5;
    const int q0 = blockIdx.x * 64, m0 = blockIdx.y * 64, b = blockIdx.z;
    const int span = 63 * s + (ks - 1) * d.dilation + 1;
    const int base_in = q0 * s - d.pad_left;
    const int KC = a.CIC * ks;

    for (int kl = tid; kl < a.KCE; kl += 256) {
        int off = 0;
        if (kl < KC) {
            const int ci = kl / ks, j = kl - ci * ks;
            const int jd = j * d.dilation;
            off = (ci * s + jd % s) * a.LP + jd / s;
        }
        koff[kl] = off;
    }

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const size_t wpitch = (size_t)d.Cin * ks;
    for (int ci0 = 0; ci0 < d.Cin; ci0 += a.CIC) {
        const int cic = min(a.CIC, d.Cin - ci0);
        const int kvalid = cic * ks;
        __syncthreads();
        // ---- weight tile: 16 rows per wave, 64 consecutive k per pass
        for (int r = wave; r < 64; r += 4) {
            const int mrow = m0 + r;
            const float* wsrc = a.w + (size_t)mrow * wpitch + (size_t)ci0 * ks;
            for (int kl = lane; kl < a.KCE; kl += 64)
                Ws[r * a.KCP + kl] = (mrow < d.Cout && kl < kvalid) ? wsrc[kl] : 0.f;
        }
        // ---- input span, phase de-interleaved
        for (int ci = wave; ci < a.CIC; ci += 4) {
            const float* xrow = a.x + ((size_t)b * d.Cin + ci0 + ci) * d.Tin;
            float* dst = Xs + (size_t)ci * s * a.LP;
            if (s == 1) {
                for (int rel = lane; rel < span; rel += 64)
                    dst[rel] = ci < cic ? conv_fetch(a, xrow, base_in + rel) : 0.f;
            } else {
                for (int rel = lane; rel < span; rel += 64) {
                    const int qq = rel / s, ph = rel - qq * s;
                    dst[ph * a.LP + qq] = ci < cic ? conv_fetch(a, xrow, base_in + rel) : 0.f;
                }
            }
        }
        __syncthreads();
        const float* wp = Ws + (wr * 32 + li) * a.KCP + kk;
        const float* xp = Xs + wc * 32 + li;
        for (int k2 = 0; k2 < a.KCE; k2 += 2) {
            const float av = wp[k2];
            const float bv = xp[koff[k2 + kk]];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
    }

    // ---- epilogue
    const int q = q0 + wc * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
        const int mrow = m0 + wr * 32 + row;
        if (mrow >= d.Cout) continue;
        float v = acc[r];
        if (d.shuffle <= 1) {
            if (q < d.Tout) {
                const size_t oi = ((size_t)b * d.Cout + mrow) * d.Tout + q;
                if (a.bias) v += a.bias[mrow];
                if (a.res) v += a.res[oi];
                a.y[oi] = v;
            }
        } else {
            const int co = mrow / d.shuffle, ph = mrow - co * d.shuffle;
            const long long o = (long long)q * d.shuffle + ph - d.trim_left;
            if (o >= 0 && o < d.Tout && q < a.Tq) {
                const size_t oi = ((size_t)b * (d.Cout / d.shuffle) + co) * d.Tout + (size_t)o;
                if (a.bias) v += a.bias[co];
                if (a.res) v += a.res[oi];
                a.y[oi] = v;
            }
        }
    }
}

extern "C" int acmi_conv1d(const acmi_conv_desc* dp, const float* x, const float* w, const float* bias,
                           const float* residual, float* y, void* stream) {
    ACMI_REQUIRE(dp != nullptr, "acmi_conv1d: null descriptor");
    const acmi_conv_desc& d = *dp;
    ACMI_REQUIRE(d.B >= 0 && d.Cin > 0 && d.Cout > 0 && d.ksize > 0 && d.stride > 0 && d.dilation > 0,
                 "acmi_conv1d: bad shape");
    ACMI_REQUIRE(d.ksize <= 128, "acmi_conv1d: ksize=%d unsupported (max 128)", d.ksize);
    ACMI_REQUIRE(d.shuffle >= 1 && d.Cout % d.shuffle == 0, "acmi_conv1d: Cout=%d not divisible by shuffle=%d", d.Cout,
                 d.shuffle);
    ACMI_REQUIRE(d.shuffle == 1 || (d.stride == 1 && d.dilation == 1), "acmi_conv1d: shuffle needs stride=dilation=1");
    ACMI_REQUIRE(d.pad_mode == ACMI_PAD_ZERO || d.reflect_len >= d.Tin, "acmi_conv1d: reflect_len < Tin");
    if (d.B == 0 || d.Tout <= 0) return ACMI_OK;
    ConvArgs a;
    a.d = d; a.x = x; a.w = w; a.bias = bias; a.res = residual; a.y = y;
    a.Tq = d.shuffle > 1 ? (int)(((long long)d.Tout + d.trim_left + d.shuffle - 1) / d.shuffle) : d.Tout;
    int cic = 128 / d.ksize;
    if (cic < 1) cic = 1;
    if (cic > d.Cin) cic = d.Cin;
    // keep the staged input span within the LDS budget
    const int lp = 64 + ((d.ksize - 1) * d.dilation) / d.stride + 2;
    while (cic > 1 && (size_t)cic * d.stride * lp * 4 > 24 * 1024) cic >>= 1;
    a.CIC = cic;
    a.KCE = (cic * d.ksize + 1) & ~1;
    a.KCP = a.KCE | 1;
    a.LP = lp;
    const size_t lds = ((size_t)64 * a.KCP + (size_t)a.CIC * d.stride * a.LP + a.KCE) * sizeof(float);
    ACMI_REQUIRE(lds <= 64 * 1024, "acmi_conv1d: LDS budget exceeded (%zu B)", lds);
    dim3 grid((a.Tq + 63) / 64, (d.Cout + 63) / 64, d.B), block(256);
    hipLaunchKernelGGL(conv_mfma_kernel, grid, block, lds, (hipStream_t)stream, a);
    return acmi_check_launch("conv_mfma_kernel");
}

// =====================================================================================================
// LSTM recurrence (audiocraft/modules/lstm.py:19-25 -> nn.LSTM, gate order i, f, g, o)
// =====================================================================================================

#define LSTM_BB 8  // batch rows per pass

__global__ __launch_bounds__(256) void lstm_step_kernel(const float* __restrict__ gates_in,
                                                        const float* __restrict__ w_hh,
                                                        const float* __restrict__ h_prev, float* __restrict__ h_next,
                                                        float* __restrict__ cst, const float* __restrict__ skip,
                                                        float* __restrict__ y, int B, int H, int T, int t) {
    extern __shared__ __attribute__((aligned(16))) float sh[];
    float* hs = sh;                 // [LSTM_BB][H]
    float* gs = sh + LSTM_BB * H;   // [16][LSTM_BB]
    const int tid = threadIdx.x;
    const int r = tid >> 4, ksl = tid & 15;
    const int gate = r >> 2, u = r & 3;
    const int j0 = blockIdx.x * 4;
    const bool jvalid = j0 + u < H;
    const float* wrow = w_hh + ((size_t)gate * H + (jvalid ? j0 + u : 0)) * H;
    for (int b0 = 0; b0 < B; b0 += LSTM_BB) {
        const int nb = min(LSTM_BB, B - b0);
        for (int idx = tid; idx < LSTM_BB * H; idx += 256) hs[idx] = idx < nb * H ? h_prev[(size_t)b0 * H + idx] : 0.f;
        __syncthreads();
        float acc[LSTM_BB];
#pragma unroll
        for (int bb = 0; bb < LSTM_BB; ++bb) acc[bb] = 0.f;
        for (int k = ksl; k < H; k += 16) {
            const float wv = wrow[k];
#pragma unroll
            for (int bb = 0; bb < LSTM_BB; ++bb) acc[bb] = fmaf(wv, hs[bb * H + k], acc[bb]);
        }
#pragma unroll
        for (int off = 1; off < 16; off <<= 1)
#pragma unroll
            for (int bb = 0; bb < LSTM_BB; ++bb) acc[bb] += __shfl_xor(acc[bb], off, 64);
        if (ksl == 0) {
#pragma unroll
            for (int bb = 0; bb < LSTM_BB; ++bb) gs[r * LSTM_BB + bb] = acc[bb];
        }
        __syncthreads();
        if (tid < 4 * LSTM_BB) {
            const int uu = tid & 3, bb = tid >> 2;
            const int j = j0 + uu, bidx = b0 + bb;
            if (bb < nb && j < H) {
                const size_t gbase = ((size_t)bidx * 4 * H + j) * T + t;
                const float gi = gs[(0 * 4 + uu) * LSTM_BB + bb] + gates_in[gbase];
                const float gf = gs[(1 * 4 + uu) * LSTM_BB + bb] + gates_in[gbase + (size_t)H * T];
                const float gg = gs[(2 * 4 + uu) * LSTM_BB + bb] + gates_in[gbase + (size_t)2 * H * T];
                const float go = gs[(3 * 4 + uu) * LSTM_BB + bb] + gates_in[gbase + (size_t)3 * H * T];
                const float ig = 1.f / (1.f + expf(-gi));
                const float fg = 1.f / (1.f + expf(-gf));
                const float og = 1.f / (1.f + expf(-go));
                const size_t si = (size_t)bidx * H + j;
                const float cn = fg * cst[si] + ig * tanhf(gg);
                const float hn = og * tanhf(cn);
                cst[si] = cn;
                h_next[si] = hn;
                const size_t yi = ((size_t)bidx * H + j) * T + t;
                y[yi] = skip ? hn + skip[yi] : hn;
            }
        }
        __syncthreads();
    }
}

extern "C" size_t acmi_lstm_work_floats(int B, int H) { return (size_t)3 * B * H; }

extern "C" int acmi_lstm_layer(const float* gates_in, const float* w_hh, const float* skip, float* y, float* work, int B,
                               int H, int T, void* stream) {
    ACMI_REQUIRE(B > 0 && H > 0 && T >= 0, "acmi_lstm_layer: bad shape");
    ACMI_REQUIRE((size_t)(LSTM_BB * H + 16 * LSTM_BB) * 4 <= 64 * 1024, "acmi_lstm_layer: H=%d too large", H);
    hipStream_t st = (hipStream_t)stream;
    float* h0 = work;
    float* h1 = work + (size_t)B * H;
    float* c = work + (size_t)2 * B * H;
    if (hipMemsetAsync(work, 0, sizeof(float) * 3 * B * H, st) != hipSuccess) {
        acmi_set_error("acmi_lstm_layer: hipMemsetAsync failed");
        return ACMI_ELAUNCH;
    }
    const size_t lds = (size_t)(LSTM_BB * H + 16 * LSTM_BB) * sizeof(float);
    dim3 grid((H + 3) / 4), block(256);
    for (int t = 0; t < T; ++t) {
        hipLaunchKernelGGL(lstm_step_kernel, grid, block, lds, st, gates_in, w_hh, (t & 1) ? h1 : h0, (t & 1) ? h0 : h1, c,
                           skip, y, B, H, T, t);
    }
    return acmi_check_launch("lstm_step_kernel");
}
