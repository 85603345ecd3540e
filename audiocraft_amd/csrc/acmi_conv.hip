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
#include <stdlib.h>

struct ConvArgs {
    acmi_conv_desc d;
    const float* x; const float* w; const float* bias; const float* res; float* y;
    int Tq;    // GEMM columns per batch item
    int CIC;   // input channels per K chunk
    int KCE;   // CIC * ksize rounded up to a multiple of 16 (one block of the inner loop = 8 MFMAs = 16 k)
    int KCP;   // LDS row pitch (floats) of one parity plane of the weight tile: KCE / 2 rounded up to 4 (mod 64)
    int LP;    // LDS row pitch of one (channel, phase) input row
    int XSZ;   // floats of the staged input span, rounded up to a multiple of 4 (the offset table behind it is read 16 B at a time)
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

// NTQ: 64-column (time) tiles per workgroup.  The weight tile of a K chunk is staged ONCE and multiplied with NTQ input
// spans in turn (NTQ accumulators): with one tile per workgroup the 32 KB weight tile was re-staged from L2 for every 64
// output samples and that traffic (~35 KB per 1.7 us of MFMA work per workgroup) co-bounded the kernel.
// PK: the parity-plane LDS layout + 8-MFMA blocks below (default); !PK: weights [64][KCP odd], one offset table, one
// (weight, offset, input) read triple per MFMA (ACMI_CONV_PARITY=0, the round-2 inner loop, kept for A/B).
template <int NTQ, bool PK>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const acmi_conv_desc& d = a.d;
    const int s = d.stride, ks = d.ksize;
    // The MFMA's two k slots are the two halves of the wave (kk = lane >> 5): half kk consumes k = 2 u + kk.  Weights and
    // the k -> LDS offset table are therefore stored DE-INTERLEAVED BY PARITY, so that the 8 values a lane needs for a block
    // of 8 MFMAs (16 k) are contiguous: two ds_read_b128 each instead of 8 + 8 dependent ds_read_b32 (the table lookup in
    // front of every input read was a second LDS latency on the critical path of every MFMA).
    float* Ws = smem;                                   // PK: [2 parities][64][KCP]; else [64][KCP]
    float* Xs = Ws + (PK ? 2 : 1) * 64 * a.KCP;         // [CIC][s][LP]
    int* koff = reinterpret_cast<int*>(Xs + a.XSZ);     // PK: [2 parities][KCE / 2]; else [KCE]
    const int KH = a.KCE >> 1;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, kk = lane >> 5;
    const int qb = blockIdx.x * 64 * NTQ, m0 = blockIdx.y * 64, b = blockIdx.z;
    const int span = 63 * s + (ks - 1) * d.dilation + 1;
    const int KC = a.CIC * ks;

    for (int kl = tid; kl < a.KCE; kl += 256) {
        int off = 0;
        if (kl < KC) {
            const int ci = kl / ks, j = kl - ci * ks;
            const int jd = j * d.dilation;
            off = (ci * s + jd % s) * a.LP + jd / s;
        }
        koff[PK ? (kl & 1) * KH + (kl >> 1) : kl] = off;
    }

    f32x16 acc[NTQ];
#pragma unroll
    for (int t = 0; t < NTQ; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

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
                Ws[PK ? ((kl & 1) * 64 + r) * a.KCP + (kl >> 1) : r * a.KCP + kl] = (mrow < d.Cout && kl < kvalid) ? wsrc[kl] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < NTQ; ++t) {
            const int q0 = qb + t * 64;
            if (t > 0) {
                if (q0 >= a.Tq) break;     // block uniform: no column of this tile exists
                __syncthreads();           // every wave is done with the previous tile's span
            }
            const int base_in = q0 * s - d.pad_left;
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
            const float* xp = Xs + wc * 32 + li;
            if constexpr (!PK) {
                const float* wq = Ws + (wr * 32 + li) * a.KCP + kk;
                for (int k2 = 0; k2 < a.KCE; k2 += 2) {
                    const float av = wq[k2];
                    const float bv = xp[koff[k2 + kk]];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
                }
            }
            const float* wp = Ws + (kk * 64 + wr * 32 + li) * a.KCP;   // this lane's row of its parity plane
            const int* kp = koff + kk * KH;
            for (int kb = 0; PK && kb < KH; kb += 8) {   // 8 MFMAs: k = 2 (kb + u) + kk, u = 0 .. 7, ascending (the fmaf chain of a scalar loop)
                const float4 wa = *reinterpret_cast<const float4*>(wp + kb), wb = *reinterpret_cast<const float4*>(wp + kb + 4);
                const int4 oa = *reinterpret_cast<const int4*>(kp + kb), ob = *reinterpret_cast<const int4*>(kp + kb + 4);
                const float x0 = xp[oa.x], x1 = xp[oa.y], x2 = xp[oa.z], x3 = xp[oa.w];
                const float x4 = xp[ob.x], x5 = xp[ob.y], x6 = xp[ob.z], x7 = xp[ob.w];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.x, x0, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.y, x1, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.z, x2, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.w, x3, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb.x, x4, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb.y, x5, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb.z, x6, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb.w, x7, acc[t], 0, 0, 0);
            }
        }
    }

    // ---- epilogue
#pragma unroll
    for (int t = 0; t < NTQ; ++t) {
        const int q = qb + t * 64 + wc * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
            const int mrow = m0 + wr * 32 + row;
            if (mrow >= d.Cout) continue;
            float v = acc[t][r];
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
}

// -----------------------------------------------------------------------------------------------------
// Convolutions with ONE or TWO output channels (the decoder's last conv: 64 -> 1 channel, k = 7, on the full-rate
// signal).  As a 64-row MFMA tile 63 of 64 rows are padding -- that launch alone was ~1/5 of the EnCodec-32k decode.
// Here: 256 threads x 4 consecutive outputs; per chunk of 8 input channels the span (ELU, padding applied on load, like
// conv_fetch everywhere) sits in LDS and a thread slides a 12-float window (three ds_read_b128) over its 4 outputs x KS
// taps; the weights are wave-uniform (scalar loads).  stride = dilation = 1.
// -----------------------------------------------------------------------------------------------------
template <int CO, int KS>
__global__ __launch_bounds__(256) void conv_fewout_kernel(const ConvArgs a) {
    constexpr int TO = 4, CIC = 8, NOUT = 256 * TO, SP = NOUT + 12;   // row pitch: a multiple of 4 floats
    __shared__ __attribute__((aligned(16))) float xs[CIC * SP];
    const acmi_conv_desc& d = a.d;
    const int tid = threadIdx.x, b = blockIdx.z;
    const int q0 = blockIdx.x * NOUT, base_in = q0 - d.pad_left;
    float acc[CO][TO];
#pragma unroll
    for (int c = 0; c < CO; ++c)
#pragma unroll
        for (int o = 0; o < TO; ++o) acc[c][o] = 0.f;
    for (int ci0 = 0; ci0 < d.Cin; ci0 += CIC) {
        __syncthreads();
        for (int idx = tid; idx < CIC * (NOUT + KS - 1); idx += 256) {
            const int ci = idx / (NOUT + KS - 1), rel = idx - ci * (NOUT + KS - 1);
            const bool live = ci0 + ci < d.Cin;
            const float* xrow = a.x + ((size_t)b * d.Cin + min(ci0 + ci, d.Cin - 1)) * d.Tin;
            xs[ci * SP + rel] = live ? conv_fetch(a, xrow, base_in + rel) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int ci = 0; ci < CIC; ++ci) {
            const float4 w0 = *reinterpret_cast<const float4*>(xs + ci * SP + tid * TO);
            const float4 w1 = *reinterpret_cast<const float4*>(xs + ci * SP + tid * TO + 4);
            const float4 w2 = *reinterpret_cast<const float4*>(xs + ci * SP + tid * TO + 8);
            const float win[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
            const int cig = min(ci0 + ci, d.Cin - 1);   // rows past Cin hold zeros: any finite weight will do
#pragma unroll
            for (int c = 0; c < CO; ++c) {
                const float* wr = a.w + ((size_t)c * d.Cin + cig) * KS;   // wave uniform -> scalar loads
#pragma unroll
                for (int j = 0; j < KS; ++j) {
                    const float wv = wr[j];
#pragma unroll
                    for (int o = 0; o < TO; ++o) acc[c][o] = fmaf(wv, win[o + j], acc[c][o]);
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CO; ++c) {
        const float bias = a.bias ? a.bias[c] : 0.f;
#pragma unroll
        for (int o = 0; o < TO; ++o) {
            const int q = q0 + tid * TO + o;
            if (q < d.Tout) {
                const size_t oi = ((size_t)b * CO + c) * d.Tout + q;
                float v = acc[c][o] + bias;
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
    static const bool fewout_ok = !(getenv("ACMI_CONV_FEWOUT") != nullptr && getenv("ACMI_CONV_FEWOUT")[0] == '0');
    if (fewout_ok && d.Cout <= 2 && d.ksize == 7 && d.stride == 1 && d.dilation == 1 && d.shuffle == 1) {
        a.Tq = d.Tout; a.CIC = a.KCE = a.KCP = a.LP = a.XSZ = 0;
        dim3 grid((d.Tout + 1023) / 1024, 1, d.B), block(256);
        if (d.Cout == 1) hipLaunchKernelGGL((conv_fewout_kernel<1, 7>), grid, block, 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((conv_fewout_kernel<2, 7>), grid, block, 0, (hipStream_t)stream, a);
        return acmi_check_launch("conv_fewout_kernel");
    }
    a.Tq = d.shuffle > 1 ? (int)(((long long)d.Tout + d.trim_left + d.shuffle - 1) / d.shuffle) : d.Tout;
    int cic = 128 / d.ksize;
    if (cic < 1) cic = 1;
    if (cic > d.Cin) cic = d.Cin;
    // keep the staged input span within the LDS budget
    const int lp = 64 + ((d.ksize - 1) * d.dilation) / d.stride + 2;
    while (cic > 1 && (size_t)cic * d.stride * lp * 4 > 24 * 1024) cic >>= 1;
    // inner-loop form: parity planes + 8-MFMA blocks (default), or one read triple per MFMA (ACMI_CONV_PARITY=0, A/B).  With
    // one column tile per workgroup the plain form was 4 % faster; with the weight tile reused over 4 column tiles the
    // block form is (EnCodec-32k, 8 x 30 s: encode 73.1 vs 75.6 ms, decode 75.4 vs 77.8 ms; profiles/r03_codec_bench_*.jsonl)
    static const bool parity = !(getenv("ACMI_CONV_PARITY") != nullptr && getenv("ACMI_CONV_PARITY")[0] == '0');
    a.CIC = cic;
    a.LP = lp;
    a.XSZ = (a.CIC * d.stride * a.LP + 3) & ~3;
    size_t lds;
    if (parity) {
        a.KCE = (cic * d.ksize + 15) & ~15;
        a.KCP = (a.KCE / 2 + 63) / 64 * 64 + 4;   // = 4 (mod 64): the 16 lanes of a ds_read_b128 group hit 16 distinct 16-byte bank slots
        lds = ((size_t)2 * 64 * a.KCP + (size_t)a.XSZ + a.KCE) * sizeof(float);
    } else {
        a.KCE = (cic * d.ksize + 1) & ~1;
        a.KCP = a.KCE | 1;
        lds = ((size_t)64 * a.KCP + (size_t)a.XSZ + a.KCE) * sizeof(float);
    }
    ACMI_REQUIRE(lds <= 64 * 1024, "acmi_conv1d: LDS budget exceeded (%zu B)", lds);
    // column tiles per workgroup: as many as keep >= 2 workgroups per CU in flight (4, 2 or 1); ACMI_CONV_NTQ forces one
    const int tq = (a.Tq + 63) / 64, my = (d.Cout + 63) / 64;
    static const int want = getenv("ACMI_CONV_NTQ") ? atoi(getenv("ACMI_CONV_NTQ")) : 0;
    int ntq = want == 1 || want == 2 || want == 4 ? want : ((long)tq * my * d.B >= 4 * 512 ? 4 : ((long)tq * my * d.B >= 2 * 512 ? 2 : 1));
    dim3 grid((tq + ntq - 1) / ntq, my, d.B), block(256);
#define ACMI_CONV_LAUNCH(N)                                                                                   \
    if (parity) hipLaunchKernelGGL((conv_mfma_kernel<N, true>), grid, block, lds, (hipStream_t)stream, a);    \
    else hipLaunchKernelGGL((conv_mfma_kernel<N, false>), grid, block, lds, (hipStream_t)stream, a);
    if (ntq == 4) { ACMI_CONV_LAUNCH(4) } else if (ntq == 2) { ACMI_CONV_LAUNCH(2) } else { ACMI_CONV_LAUNCH(1) }
#undef ACMI_CONV_LAUNCH
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
        for (int bb = 0; bb < LSTM_BB; ++bb) acc[bb] = row16_sum(acc[bb]);
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

// -----------------------------------------------------------------------------------------------------
// Persistent form: ONE launch per LSTM layer.  Workgroup g owns hidden units 4g .. 4g+3 (16 gate rows of W_hh) for all T
// steps and keeps its 16 x H slice of W_hh in REGISTERS (64 floats per thread at H = 1024; the step kernel above
// re-streams all 16.8 MB of W_hh from L2 at every step).  What remains per step is the all-gather of h_{t-1}: 32 KB per
// workgroup at B = 8.  The data is its own flag (MI355X guide, Guideline 16 form R2, here with a sentinel instead of a
// tag): hidden values are never NaN, so a slot holding the quiet-NaN pattern LSTM_EMPTY means "not written yet".
//   * three [B, H] f32 buffers rotate by step (t mod 3); all slots start EMPTY;
//   * the owner of a slot writes h_t with a write-through (agent-scope) store; readers sweep the previous step's buffer
//     with agent-scope 16-byte loads (the 4 units of one owner and one batch row) until no value is EMPTY;
//   * right after its own gather of step t - 1 a workgroup re-arms its slots in the buffer of step t + 1 (it holds step
//     t - 2, which every workgroup finished reading before it published step t - 1 -- and all of those have just been
//     seen); the publish of step t + 1 into that buffer is one full step (and an s_waitcnt vmcnt(0)) later.
// Spins are bounded (err[0] counts give-ups; the host checks it).
// -----------------------------------------------------------------------------------------------------
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
#define LSTM_EMPTY 0x7fc00001u

template <int KI>   // KI = ceil(H / 16) <= 64: W_hh values per thread
__global__ __launch_bounds__(256) void lstm_persistent_kernel(const float* __restrict__ gates_in,
                                                              const float* __restrict__ w_hh, float* __restrict__ cst,
                                                              const float* __restrict__ skip, float* __restrict__ y,
                                                              unsigned* hbuf, unsigned* err, int B, int H, int T) {
    extern __shared__ __attribute__((aligned(16))) float sh[];
    // h_{t-1} in LDS, k-major: hs[k][q], pitch 12 floats (two conflict-free ds_read_b128 fetch the 8 batch values of a k)
    constexpr int HP = 12;
    float* hs = sh;                    // [H][HP]
    float* gs = sh + (size_t)H * HP;   // [16][LSTM_BB]
    int* s_abort = reinterpret_cast<int*>(gs + 16 * LSTM_BB);   // workgroup-wide "give up" flag (below)
    const int tid = threadIdx.x;
    // A workgroup that starts only after another one has given up (it was not resident while the others spun) leaves at
    // once, and so does every workgroup that sees the error word set while it waits: a residency failure costs ONE bounded
    // spin, not one per remaining step.
    if (tid == 0) *s_abort = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    __syncthreads();
    if (*s_abort) return;
    const int r = tid >> 4, ksl = tid & 15;
    const int gate = r >> 2, u = r & 3;
    const int j0 = blockIdx.x * 4;
    const bool jvalid = j0 + u < H;
    const float* wrow = w_hh + ((size_t)gate * H + (jvalid ? j0 + u : 0)) * H;
    float wv[KI];
#pragma unroll
    for (int i = 0; i < KI; ++i) wv[i] = (ksl + 16 * i < H) ? wrow[ksl + 16 * i] : 0.f;
    const size_t BH = (size_t)B * H;
    const int H4 = H >> 2;                       // 16-byte groups per row (H % 4 == 0 is required by the launcher)
    const bool one_pass = B <= LSTM_BB;
    float c_reg = 0.f;                           // cell state of this thread's (unit, row) when B fits one pass
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(hbuf, 0, (int)(3 * BH * 4), 0x00020000);

    for (int t = 0; t < T; ++t) {
        const unsigned prev_off = (unsigned)(((t + 2) % 3) * BH);   // buffer of step t - 1
        unsigned* hnext = hbuf + (size_t)(t % 3) * BH;              // buffer of step t
        unsigned* hrearm = hbuf + (size_t)((t + 1) % 3) * BH;       // buffer of step t + 1 (still holds step t - 2)
        for (int b0 = 0; b0 < B; b0 += LSTM_BB) {
            const int nb = min(LSTM_BB, B - b0);
            // this thread's gate inputs of the step: requested before the sweep (they do not depend on h)
            float gin[4] = {0.f, 0.f, 0.f, 0.f};
            const int uu = tid & 3, bb = tid >> 2, j = j0 + uu, bidx = b0 + bb;
            const bool owner = tid < 4 * LSTM_BB && bb < nb && j < H;
            if (owner) {
                const size_t gbase = ((size_t)bidx * 4 * H + j) * T + t;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) gin[g4] = gates_in[gbase + (size_t)g4 * H * T];
            }
            // ---- all-gather of h_{t-1}[b0 .. b0 + nb): 16-byte groups (row q, units 4 k4 .. 4 k4 + 3), 8 per thread in flight
            if (t == 0) {
                for (int idx = tid; idx < H * HP; idx += 256) hs[idx] = 0.f;
            } else {
                const int ngr = nb * H4;
                for (int base = 0; base < LSTM_BB * H4; base += 256 * 8) {
                    u32x4_t v[8];
                    unsigned pending = 0;
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (base + i * 256 + tid < ngr) pending |= 1u << i;
                    unsigned spins = 0;
                    while (pending) {
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            if (pending & (1u << i)) {
                                const int idx = base + i * 256 + tid;    // = q * H4 + k4
                                v[i] = __builtin_amdgcn_raw_buffer_load_b128(
                                    rs, (prev_off + (unsigned)(b0 * H) + (unsigned)idx * 4u) * 4u, 0, 16 /* sc1 */);
                            }
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            if ((pending & (1u << i)) && v[i][0] != LSTM_EMPTY && v[i][1] != LSTM_EMPTY &&
                                v[i][2] != LSTM_EMPTY && v[i][3] != LSTM_EMPTY) {
                                const int idx = base + i * 256 + tid, q = idx / H4, k = (idx - q * H4) * 4;
#pragma unroll
                                for (int e = 0; e < 4; ++e) hs[(k + e) * HP + q] = __uint_as_float(v[i][e]);
                                pending &= ~(1u << i);
                            }
                        if (pending && (++spins & 0x3ffu) == 0u) {   // every 1024 polls: bounded wait, global abort flag
                            if (spins > 1000000u) { atomicAdd(err, 1u); *s_abort = 1; break; }
                            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { *s_abort = 1; break; }
                        }
                    }
                }
                for (int idx = tid; idx < (LSTM_BB - nb) * H; idx += 256)   // rows past nb of this pass: zeros
                    hs[(idx % H) * HP + nb + idx / H] = 0.f;
            }
            __syncthreads();
            if (*s_abort) return;   // workgroup-uniform (written before the barrier): the host finds err != 0 and raises
            // the WHOLE gather is complete (barrier above): re-arm this workgroup's slots of step t + 1 (header comment).
            // The 4 units of a row are 4 consecutive lanes (uu = tid & 3): ONE 16-byte write-through store by the first of
            // them instead of four 4-byte ones (a scalar sc1 store is a fabric write of its own, ~6x the cost per byte).
            const unsigned slot_off = (unsigned)(((size_t)bidx * H + j0) * 4u);   // byte offset of (row, units j0 .. j0 + 3) in a buffer
            if (owner && uu == 0)
                __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{LSTM_EMPTY, LSTM_EMPTY, LSTM_EMPTY, LSTM_EMPTY}, rs,
                                                       (unsigned)(((t + 1) % 3) * BH * 4) + slot_off, 0, 16 /* sc1 */);
            (void)hrearm;
            // ---- 16 rows x H against nb hidden vectors: W from registers, h from LDS
            float acc[LSTM_BB];
#pragma unroll
            for (int q = 0; q < LSTM_BB; ++q) acc[q] = 0.f;
#pragma unroll
            for (int i = 0; i < KI; ++i) {
                const int k = min(ksl + 16 * i, H - 1);
                const float4 h0 = *reinterpret_cast<const float4*>(hs + k * HP);
                const float4 h1 = *reinterpret_cast<const float4*>(hs + k * HP + 4);
                acc[0] = fmaf(wv[i], h0.x, acc[0]); acc[1] = fmaf(wv[i], h0.y, acc[1]);
                acc[2] = fmaf(wv[i], h0.z, acc[2]); acc[3] = fmaf(wv[i], h0.w, acc[3]);
                acc[4] = fmaf(wv[i], h1.x, acc[4]); acc[5] = fmaf(wv[i], h1.y, acc[5]);
                acc[6] = fmaf(wv[i], h1.z, acc[6]); acc[7] = fmaf(wv[i], h1.w, acc[7]);
                if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);   // keep the LDS reads from being hoisted en bloc
            }
#pragma unroll
            for (int q = 0; q < LSTM_BB; ++q) acc[q] = row16_sum(acc[q]);
            if (ksl == 0) {
#pragma unroll
                for (int q = 0; q < LSTM_BB; ++q) gs[r * LSTM_BB + q] = acc[q];
            }
            __syncthreads();
            float c_hn = 0.f;   // this thread's new hidden value (owners)
            if (owner) {
                const float gi = gs[(0 * 4 + uu) * LSTM_BB + bb] + gin[0];
                const float gf = gs[(1 * 4 + uu) * LSTM_BB + bb] + gin[1];
                const float gg = gs[(2 * 4 + uu) * LSTM_BB + bb] + gin[2];
                const float go = gs[(3 * 4 + uu) * LSTM_BB + bb] + gin[3];
                const float ig = 1.f / (1.f + expf(-gi));
                const float fg = 1.f / (1.f + expf(-gf));
                const float og = 1.f / (1.f + expf(-go));
                const size_t si = (size_t)bidx * H + j;
                const float cp = t == 0 ? 0.f : (one_pass ? c_reg : cst[si]);
                const float cn = fg * cp + ig * tanhf(gg);
                const float hn = og * tanhf(cn);
                c_reg = cn;
                if (!one_pass) cst[si] = cn;   // private to this workgroup
                c_hn = hn;
            }
            {   // publish h_t: the quad's 4 units as one 16-byte write-through store (all lanes take part in the exchange)
                const unsigned h0 = __float_as_uint(dpp_f32<0x00>(c_hn)), h1 = __float_as_uint(dpp_f32<0x55>(c_hn));
                const unsigned h2 = __float_as_uint(dpp_f32<0xAA>(c_hn)), h3 = __float_as_uint(dpp_f32<0xFF>(c_hn));
                if (owner && uu == 0) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the re-arm store of the previous step has landed
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{h0, h1, h2, h3}, rs, (unsigned)((t % 3) * BH * 4) + slot_off, 0,
                                                           16 /* sc1 */);
                }
                (void)hnext;
            }
            if (owner) {   // the output row after the publish: nothing on the recurrence's critical path waits for this store
                const size_t yi = ((size_t)bidx * H + j) * T + t;
                y[yi] = skip ? c_hn + skip[yi] : c_hn;
            }
            __syncthreads();
        }
    }
}

// work: 3 * B * H floats for the step form (h double buffer + c); the persistent form needs c (B * H floats), three
// hidden-state buffers (3 * B * H floats) and an error word: 5 * B * H + 4 floats cover both
extern "C" size_t acmi_lstm_work_floats(int B, int H) { return (size_t)5 * B * H + 4; }

// Every workgroup of the persistent form must be RESIDENT for its all-gather to complete.  The grid ((H + 3) / 4
// workgroups) is checked against what the device this call runs on can hold: CUs (hipDeviceProp_t.multiProcessorCount:
// 256 on a whole MI355X, fewer on a partitioned one) x the occupancy the runtime reports for this kernel with its LDS,
// minus one workgroup per CU of margin when more than one fits (the API answers one too many at some SGPR counts,
// MI355X_MICROARCH.md "Residency and cooperative launch").  Anything that does not fit runs the per-step kernel.
// Residency can still be lost to OTHER work on the device (another stream or process): the kernel's spins are bounded,
// the first give-up raises a device-wide abort flag (the err word) that every workgroup polls, and the host raises.
template <typename KernelT>
static bool lstm_grid_resident(KernelT kernel, int grid, size_t lds_bytes) {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return false;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, lds_bytes) != hipSuccess || per_cu <= 0) return false;
    if (per_cu > 1) per_cu -= 1;
    return (long)grid <= (long)cus * per_cu;
}

static int lstm_persistent_ok(int B, int H) {
    static int want = -1;
    if (want < 0) { const char* e = getenv("ACMI_LSTM_PERSISTENT"); want = (e && e[0] == '0') ? 0 : 1; }
    return want && H <= 1024 && H % 4 == 0 && B >= 1;
}



extern "C" int acmi_lstm_layer(const float* gates_in, const float* w_hh, const float* skip, float* y, float* work, int B,
                               int H, int T, void* stream) {
    ACMI_REQUIRE(B > 0 && H > 0 && T >= 0, "acmi_lstm_layer: bad shape");
    ACMI_REQUIRE((size_t)(LSTM_BB * H + 16 * LSTM_BB) * 4 <= 64 * 1024, "acmi_lstm_layer: H=%d too large", H);
    hipStream_t st = (hipStream_t)stream;
    float* h0 = work;
    float* h1 = work + (size_t)B * H;
    float* c = work + (size_t)2 * B * H;
    // (the err word at work[5 B H] is NOT cleared here: it accumulates over the layers of a stack and the caller, who
    // zeroed it, reads it once at the end -- a give-up in layer 0 must not be erased by layer 1's call)
    if (hipMemsetAsync(work, 0, sizeof(float) * 3 * B * H, st) != hipSuccess) {
        acmi_set_error("acmi_lstm_layer: hipMemsetAsync failed");
        return ACMI_ELAUNCH;
    }
    const size_t lds = (size_t)(LSTM_BB * H + 16 * LSTM_BB) * sizeof(float);
    dim3 grid((H + 3) / 4), block(256);
    if (T > 0 && lstm_persistent_ok(B, H)) {
        const size_t lds_p = (size_t)(12 * H + 16 * LSTM_BB) * sizeof(float) + 16;   // + the abort flag
        // layout of `work` for this form: [c: B H floats][h buffers: 3 B H words, all EMPTY][...][err: 1 word]
        unsigned* hbuf = reinterpret_cast<unsigned*>(work + (size_t)B * H);
        unsigned* err = reinterpret_cast<unsigned*>(work + (size_t)5 * B * H);
        const int ki = (H + 15) / 16;
#define ACMI_LSTM_CASE(KIv)                                                                                                  \
        if (ki <= KIv) {                                                                                                     \
            if (lstm_grid_resident(lstm_persistent_kernel<KIv>, (int)grid.x, lds_p)) {                                       \
                if (hipMemsetAsync(work, 0, sizeof(float) * (size_t)5 * B * H, st) != hipSuccess ||                          \
                    hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(hbuf), (int)LSTM_EMPTY, (size_t)3 * B * H, st) != hipSuccess) { \
                    acmi_set_error("acmi_lstm_layer: hipMemsetAsync failed");                                                \
                    return ACMI_ELAUNCH;                                                                                     \
                }                                                                                                            \
                hipLaunchKernelGGL(lstm_persistent_kernel<KIv>, grid, block, lds_p, st, gates_in, w_hh, work, skip, y, hbuf, \
                                   err, B, H, T);                                                                            \
                return acmi_check_launch("lstm_persistent_kernel");                                                          \
            }                                                                                                                \
        } else
        ACMI_LSTM_CASE(8) ACMI_LSTM_CASE(16) ACMI_LSTM_CASE(32) ACMI_LSTM_CASE(64) {}
#undef ACMI_LSTM_CASE
    }
    for (int t = 0; t < T; ++t) {
        hipLaunchKernelGGL(lstm_step_kernel, grid, block, lds, st, gates_in, w_hh, (t & 1) ? h1 : h0, (t & 1) ? h0 : h1, c,
                           skip, y, B, H, T, t);
    }
    return acmi_check_launch("lstm_step_kernel");
}
