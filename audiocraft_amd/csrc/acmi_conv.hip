// SEANet / diffusion-U-Net convolution kernels for gfx950: implicit-GEMM Conv1d / ConvTranspose1d in exact f32 on the
// matrix cores (v_mfma_f32_32x32x2_f32 == k-ordered fmaf chain), plus the LSTM recurrence.
//
// Three kernels per convolution, the first once per model:
//   conv_tile_weights_kernel   weights -> the LDS image the main kernel stages: [row tile of 64 output rows][K chunk]
//                              [k parity][64][KCP], zero padded (rows past Cout, k past the chunk): staging a weight tile is
//                              a linear 16-byte copy.  Done when the model is loaded (acmi_conv1d_tile_weights).
//   conv_pack_kernel           input [B, Cin, Tin] -> Xp [B, Cin_pad, stride, Qp]: padding (zero / reflect, asymmetric,
//                              "extra" right padding) materialised, the ELU that precedes every SEANet conv applied ONCE per
//                              element, time de-interleaved by stride phase, rows 16-byte aligned: staging an input span is
//                              a copy of contiguous 16-byte pieces.  (Round 2/3 did padding + ELU + phase split per element
//                              inside the main kernel, once per 64-row output tile: the staging arithmetic -- ~40 VALU
//                              instructions per element incl. expm1f, 43 elements per thread per K chunk -- took longer than
//                              the 64 MFMAs it fed; the convolutions ran at 20-29 % of the f32 MFMA rate.)
//   conv_mfma_kernel           GEMM rows = output channels (transposed convs: (channel, phase) pairs of the polyphase
//                              decomposition), cols = output time steps, K = Cin * ksize.  64 x (64 NTQ) output tile per
//                              256-thread workgroup (4 waves, 2 x 2 of 32 x 32); per K chunk the weight tile and per
//                              (chunk, column tile) the input span go HBM/L2 -> registers -> LDS one phase AHEAD of the MFMA
//                              loop that consumes them; bias, the resnet skip add and the transposed-conv trim + phase
//                              interleave are folded into the store index math.
// Reference: audiocraft/modules/conv.py:47-88,185-243; audiocraft/modules/seanet.py:16-60; audiocraft/models/unet.py:32-104.
#include "acmi_common.h"
#include <type_traits>

#include <math.h>
#include <stdlib.h>

struct ConvArgs {
    acmi_conv_desc d;
    const float* x;    // main kernel: the packed input Xp; few-output kernel: the raw input
    const float* w;    // main kernel: the tiled weight image; few-output kernel: the raw weights
    const float* bias; const float* res; float* y;
    int Tq;    // GEMM columns per batch item
    int CIC;   // input channels per K chunk
    int KCE;   // CIC * ksize rounded up to a multiple of 16 (one block of the inner loop = 8 MFMAs = 16 k)
    int KCP;   // LDS row pitch (floats) of one parity plane of the weight tile: KCE / 2 rounded up to 4 (mod 64)
    int LP;    // LDS / Xp piece length (floats, a multiple of 4) of one (channel, phase) input row: 64 + halo
    int XSZ;   // floats of the staged input span = CIC * stride * LP
    int nchunks, cin_pad;   // K chunks; Cin rounded up to whole chunks (rows of Xp per batch item)
    long long Qp;           // row pitch of Xp (floats, a multiple of 4)
    unsigned lp4_magic;     // ceil(2^32 / (LP / 4))
    int ksplit;             // > 1: blockIdx.z = b * ksplit + part; part p multiplies the chunks [p cps, (p + 1) cps) into `part`
    int cps;                // chunks per part
    float* part;            // [ksplit][B][Cout rows][tq * 64] partial sums (GEMM layout)
    // GroupNorm (+ ReLU) of the INPUT applied by the pack pass (acmi_conv1d_gn): gn_part = the (mean, M2) partials of
    // acmi_diffusion.hip's gn_partial_kernel, [B groups][gn_nchunks][2], chunks of gn_chunk elements; NULL: none
    const float* gn_part; const float* gn_gamma; const float* gn_beta;
    int gn_groups, gn_nchunks, gn_chunk, gn_relu; float gn_eps;
};

// geometry shared by the weight tiler, the packer and the main kernel
struct ConvGeom {
    bool fewout;
    bool pw;               // pointwise (k = 1) channel-doubling conv of a SEANet resnet block on a long signal: conv_pw_kernel
    int cic, nchunks, KCE, KCP, LP, XSZ, Tq, tq, my;
    long long Qp;
    size_t wt_floats, work_floats, lds;
    int ksplit;            // K parts of a launch that would not fill the chip otherwise
    size_t pack_floats;    // work = [packed input: pack_floats][partial sums: ksplit * B * Cout * tq * 64 when ksplit > 1]
};

static int conv_geometry(const acmi_conv_desc& d, ConvGeom& g) {
    ACMI_REQUIRE(d.B >= 0 && d.Cin > 0 && d.Cout > 0 && d.ksize > 0 && d.stride > 0 && d.dilation > 0, "acmi_conv1d: bad shape");
    ACMI_REQUIRE(d.ksize <= 128, "acmi_conv1d: ksize=%d unsupported (max 128)", d.ksize);
    ACMI_REQUIRE(d.shuffle >= 1 && d.Cout % d.shuffle == 0, "acmi_conv1d: Cout=%d not divisible by shuffle=%d", d.Cout, d.shuffle);
    ACMI_REQUIRE(d.shuffle == 1 || (d.stride == 1 && d.dilation == 1), "acmi_conv1d: shuffle needs stride=dilation=1");
    ACMI_REQUIRE(d.pad_mode == ACMI_PAD_ZERO || d.reflect_len >= d.Tin, "acmi_conv1d: reflect_len < Tin");
    static const bool fewout_ok = !(getenv("ACMI_CONV_FEWOUT") != nullptr && getenv("ACMI_CONV_FEWOUT")[0] == '0');
    g.fewout = fewout_ok && d.Cout <= 2 && d.ksize == 7 && d.stride == 1 && d.dilation == 1 && d.shuffle == 1;
    g.Tq = d.shuffle > 1 ? (int)(((long long)d.Tout + d.trim_left + d.shuffle - 1) / d.shuffle) : d.Tout;
    // the second conv of a SEANet resnet block (seanet.py:16-60: ELU -> k = 1 conv C / 2 -> C, + skip) in the narrow, long stages:
    // as a 64-row MFMA tile with K = 32 or 64 it is all staging, barriers and a pack pass (7 TF/s, 1.1 TB/s); see conv_pw_kernel
    static const bool pw_ok = !(getenv("ACMI_CONV_PW") != nullptr && getenv("ACMI_CONV_PW")[0] == '0');
    g.pw = pw_ok && !g.fewout && d.ksize == 1 && d.stride == 1 && d.dilation == 1 && d.shuffle == 1 && d.pad_left == 0 && d.elu_in &&
           d.Tout == d.Tin && d.Tout % 4 == 0 && ((d.Cin == 32 && d.Cout == 64) || (d.Cin == 64 && d.Cout == 128)) &&
           (long long)d.B * d.Tout >= 65536;
    if (g.fewout || g.pw) {
        g.cic = g.nchunks = g.KCE = g.KCP = g.LP = g.XSZ = g.tq = g.my = 0; g.Qp = 0; g.lds = 0;
        g.wt_floats = (size_t)d.Cout * d.Cin * d.ksize;   // the raw weights
        g.work_floats = g.pack_floats = 0;
        g.ksplit = 1;
        return ACMI_OK;
    }
    int cic = 128 / d.ksize;
    if (cic < 1) cic = 1;
    if (cic > d.Cin) cic = d.Cin;
    // one (channel, phase) piece of the staged span: 64 outputs + the halo of the taps, rounded up to 16 bytes
    const int lp = (64 + ((d.ksize - 1) * d.dilation) / d.stride + 2 + 3) & ~3;
    while (cic > 1 && (size_t)cic * d.stride * lp * 4 > 24 * 1024) cic >>= 1;   // LDS budget == the 6 x 16 B staging registers per thread
    ACMI_REQUIRE((size_t)cic * d.stride * lp * 4 <= 24 * 1024, "acmi_conv1d: ksize %d x dilation %d x stride %d exceeds the staged span",
                 d.ksize, d.dilation, d.stride);
    g.cic = cic;
    g.nchunks = (d.Cin + cic - 1) / cic;
    g.LP = lp;
    g.XSZ = cic * d.stride * lp;
    g.KCE = (cic * d.ksize + 15) & ~15;
    g.KCP = (g.KCE / 2 + 63) / 64 * 64 + 4;   // = 4 (mod 64): the 16 lanes of a ds_read_b128 group hit 16 distinct 16-byte bank slots
    g.lds = ((size_t)2 * 64 * g.KCP + (size_t)g.XSZ + g.KCE) * sizeof(float);
    ACMI_REQUIRE(g.lds <= 64 * 1024, "acmi_conv1d: LDS budget exceeded (%zu B)", g.lds);
    g.tq = (g.Tq + 63) / 64;
    g.my = (d.Cout + 63) / 64;
    g.Qp = (long long)((g.tq + 3) & ~3) * 64 + lp;   // every column tile a workgroup may touch, plus the halo
    g.wt_floats = (size_t)g.my * g.nchunks * 2 * 64 * g.KCP;
    g.pack_floats = (size_t)d.B * g.nchunks * cic * d.stride * (size_t)g.Qp;
    // Few output tiles, long K (the 3072-wide U-Net convolutions on 1 s of audio: 96 workgroups x 74 chunks, 2.6 MB of weights
    // each through one-phase-ahead staging = 0.25 ms): split K over up to 8 workgroups per tile, partial sums in GEMM layout,
    // summed in a fixed order by conv_splitk_reduce_kernel (which also does the bias / skip / phase-interleave epilogue).
    static const int ks_max = getenv("ACMI_CONV_KSPLIT") ? atoi(getenv("ACMI_CONV_KSPLIT")) : 8;
    const long wgs1 = (long)g.tq * g.my * d.B;
    g.ksplit = 1;
    if (wgs1 > 0 && wgs1 < 640 && g.nchunks >= 8 && ks_max > 1) {
        int ksp = (int)((1024 + wgs1 - 1) / wgs1);
        if (ksp > ks_max) ksp = ks_max;
        if (ksp > g.nchunks / 4) ksp = g.nchunks / 4;
        if (ksp > 1) g.ksplit = ksp;
    }
    g.work_floats = g.pack_floats + (g.ksplit > 1 ? (size_t)g.ksplit * d.B * d.Cout * g.tq * 64 : 0);
    return ACMI_OK;
}

struct GnCh { float mu, rstd, gamma, beta; };   // GroupNorm of the input channel a pack block works on

__device__ __forceinline__ float conv_fetch(const ConvArgs& a, const float* xrow, int pos, const GnCh& gn) {
    const acmi_conv_desc& d = a.d;
    int src = pos;
    if (d.pad_mode == ACMI_PAD_REFLECT) {
        if (src < 0) src = -src;
        if (src >= d.reflect_len) src = 2 * (d.reflect_len - 1) - src;
    }
    if (src < 0 || src >= d.Tin) return 0.f;        // (the padding of a GroupNorm'd input is zeros of the NORMALISED signal)
    float v = xrow[src];
    if (d.elu_in) v = v > 0.f ? v : d.elu_alpha * expm1f(v);
    if (a.gn_part != nullptr) {                     // same expression as gn_apply_kernel (acmi_diffusion.hip)
        v = (v - gn.mu) * gn.rstd * gn.gamma + gn.beta;
        if (a.gn_relu) v = fmaxf(v, 0.f);
    }
    return v;
}

// ---- weights [Cout, Cin, ks] -> [my][nchunks][2 parities][64][KCP]
struct TileWArgs { const float* w; float* wt; int Cout, Cin, ks, cic, nchunks, KCE, KCP; size_t total; };

__global__ __launch_bounds__(256) void conv_tile_weights_kernel(const TileWArgs p) {
    const size_t wpitch = (size_t)p.Cin * p.ks;
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < p.total; o += (size_t)gridDim.x * 256) {
        const int col = (int)(o % p.KCP);
        size_t t = o / p.KCP;
        const int r = (int)(t % 64); t /= 64;
        const int plane = (int)(t % 2); t /= 2;
        const int chunk = (int)(t % p.nchunks);
        const int rt = (int)(t / p.nchunks);
        const int kl = 2 * col + plane, ci0 = chunk * p.cic, mrow = rt * 64 + r;
        const int kvalid = min(p.cic, p.Cin - ci0) * p.ks;
        p.wt[o] = (col < p.KCE / 2 && kl < kvalid && mrow < p.Cout) ? p.w[(size_t)mrow * wpitch + (size_t)ci0 * p.ks + kl] : 0.f;
    }
}

// ---- input [B, Cin, Tin] -> Xp [B, cin_pad, stride, Qp]:  Xp[b][ci][u % s][u / s] = ELU(pad(x))[b][ci][u - pad_left]
__global__ __launch_bounds__(256) void conv_pack_kernel(const ConvArgs a, float* xp) {
    const acmi_conv_desc& d = a.d;
    const int ci = blockIdx.y, b = blockIdx.z, s = d.stride;
    GnCh gn = {0.f, 1.f, 1.f, 0.f};
    if (a.gn_part != nullptr) {
        // mean / rstd of this block's (batch item, group) from the partials, combined in the order gn_apply_kernel uses (Chan;
        // counts are gn_chunk except for the last chunk): every block of a group arrives at the same two numbers
        __shared__ float s4[4];
        __shared__ float stat[2];
        const int cpg = d.Cin / a.gn_groups, g = min(ci, d.Cin - 1) / cpg;
        const size_t n = (size_t)cpg * d.Tin;
        const float* part = a.gn_part + ((size_t)b * a.gn_groups + g) * a.gn_nchunks * 2;
        auto bsum = [&](float v) {
            v = wave_sum(v);
            if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
            __syncthreads();
            const float r = (s4[0] + s4[1]) + (s4[2] + s4[3]);
            __syncthreads();
            return r;
        };
        float ws = 0.f;
        for (int c = threadIdx.x; c < a.gn_nchunks; c += 256) {
            const float cntc = (float)min((size_t)a.gn_chunk, n - (size_t)c * a.gn_chunk);
            ws += part[(size_t)c * 2] * cntc;
        }
        const float mean = bsum(ws) / (float)n;
        float m2 = 0.f;
        for (int c = threadIdx.x; c < a.gn_nchunks; c += 256) {
            const float cntc = (float)min((size_t)a.gn_chunk, n - (size_t)c * a.gn_chunk);
            const float dlt = part[(size_t)c * 2] - mean;
            m2 += part[(size_t)c * 2 + 1] + cntc * dlt * dlt;
        }
        m2 = bsum(m2);
        if (threadIdx.x == 0) { stat[0] = mean; stat[1] = 1.0f / sqrtf(m2 / (float)n + a.gn_eps); }
        __syncthreads();
        gn.mu = stat[0]; gn.rstd = stat[1];
        gn.gamma = a.gn_gamma[min(ci, d.Cin - 1)]; gn.beta = a.gn_beta[min(ci, d.Cin - 1)];
    }
    const long long n = (long long)s * a.Qp;
    const long long u0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    float* dst = xp + ((size_t)b * a.cin_pad + ci) * (size_t)n;
    const float* xrow = a.x + ((size_t)b * d.Cin + min(ci, d.Cin - 1)) * d.Tin;
    if (s == 1) {
        if (u0 >= n) return;     // n is a multiple of 4
        float4 v;
        const int pos = (int)u0 - d.pad_left;
        if (ci < d.Cin) {
            v.x = conv_fetch(a, xrow, pos, gn); v.y = conv_fetch(a, xrow, pos + 1, gn);
            v.z = conv_fetch(a, xrow, pos + 2, gn); v.w = conv_fetch(a, xrow, pos + 3, gn);
        } else v = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(dst + u0) = v;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long u = (long long)blockIdx.x * 1024 + i * 256 + threadIdx.x;   // coalesced reads, phase-scattered writes
            if (u < n) {
                const long long qq = u / s;
                dst[(u - qq * s) * a.Qp + qq] = ci < d.Cin ? conv_fetch(a, xrow, (int)u - d.pad_left, gn) : 0.f;
            }
        }
    }
}

// NTQ: 64-column (time) tiles per workgroup: the weight tile of a K chunk is staged ONCE and multiplied with NTQ input spans
// in turn (NTQ accumulators).  XV: 16-byte input staging slots per thread (2, 4 or 6: CIC * stride * LP / 1024 rounded up).
//
// LDS: Ws [2 parities][64][KCP] | Xs [CIC * stride][LP] | koff [2 parities][KCE / 2].  The MFMA's two k slots are the two
// halves of the wave (kk = lane >> 5): half kk consumes k = 2 u + kk.  Weights and the k -> LDS offset table are therefore
// stored DE-INTERLEAVED BY PARITY, so that the 8 values a lane needs for a block of 8 MFMAs (16 k) are contiguous: two
// ds_read_b128 each instead of 8 + 8 dependent ds_read_b32.
template <int NTQ, int XV>
__global__ __launch_bounds__(256, NTQ == 4 ? 2 : 3) void conv_mfma_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const acmi_conv_desc& d = a.d;
    const int s = d.stride, ks = d.ksize;
    float* Ws = smem;
    float* Xs = Ws + 2 * 64 * a.KCP;
    int* koff = reinterpret_cast<int*>(Xs + a.XSZ);
    const int KH = a.KCE >> 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, kk = lane >> 5;
    const int qb = blockIdx.x * 64 * NTQ, m0 = blockIdx.y * 64;
    const int b = a.ksplit > 1 ? blockIdx.z / a.ksplit : blockIdx.z, kpart = a.ksplit > 1 ? blockIdx.z - b * a.ksplit : 0;
    const int chunk_lo = kpart * a.cps, chunk_hi = min(a.nchunks, chunk_lo + a.cps);   // a.cps = nchunks when ksplit == 1
    const int KC = a.CIC * ks;

    for (int kl = tid; kl < a.KCE; kl += 256) {
        int off = 0;
        if (kl < KC) {
            const int ci = kl / ks, j = kl - ci * ks;
            const int jd = j * d.dilation;
            off = (ci * s + jd % s) * a.LP + jd / s;
        }
        koff[(kl & 1) * KH + (kl >> 1)] = off;
    }

    f32x16 acc[NTQ];
#pragma unroll
    for (int t = 0; t < NTQ; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ---- staging through registers, one phase ahead (phase = (K chunk, column tile)); everything is a 16-byte copy
    constexpr int WV = 9;                        // 2 * 64 * KCP / 4 = 2176 pieces for KCP = 68: 8.5 per thread
    const int nw4 = 2 * 64 * a.KCP / 4, nx4 = a.XSZ / 4;
    f32x4 wreg[WV], xreg[XV];   // native vectors: arrays of HIP's float4 struct stay in scratch memory
    unsigned xsrc[XV];                           // piece j of this thread: element offset inside the chunk's rows of Xp
#pragma unroll
    for (int j = 0; j < XV; ++j) {
        const unsigned f = min((unsigned)(j * 256 + tid), (unsigned)(nx4 - 1));
        const unsigned row = __umulhi(f, a.lp4_magic);
        xsrc[j] = row * (unsigned)a.Qp + (f - row * (unsigned)(a.LP >> 2)) * 4u;
    }
    const float* wt = a.w + (size_t)blockIdx.y * a.nchunks * (size_t)(nw4 * 4);
    const float* xp = a.x + (size_t)b * a.cin_pad * s * (size_t)a.Qp + qb;
    const unsigned wlast = (unsigned)min(8 * 256 + tid, nw4 - 1) * 4u;
    const int ntl = min(NTQ, (a.Tq - qb + 63) / 64);   // live column tiles of this workgroup (block uniform)
    const size_t xchunk = (size_t)a.CIC * s * (size_t)a.Qp;

    auto issue_w = [&](int chunk) {
        const float* src = wt + (size_t)chunk * (size_t)(nw4 * 4);
#pragma unroll
        for (int j = 0; j < WV - 1; ++j) wreg[j] = *reinterpret_cast<const f32x4*>(src + (j * 256 + tid) * 4);
        wreg[WV - 1] = *reinterpret_cast<const f32x4*>(src + wlast);
    };
    auto store_w = [&]() {
#pragma unroll
        for (int j = 0; j < WV - 1; ++j) *reinterpret_cast<f32x4*>(Ws + (j * 256 + tid) * 4) = wreg[j];
        if ((WV - 1) * 256 + tid < nw4) *reinterpret_cast<f32x4*>(Ws + ((WV - 1) * 256 + tid) * 4) = wreg[WV - 1];
    };
    auto issue_x = [&](int chunk, int t) {
        const float* src = xp + (size_t)chunk * xchunk + t * 64;
#pragma unroll
        for (int j = 0; j < XV; ++j) xreg[j] = *reinterpret_cast<const f32x4*>(src + xsrc[j]);
    };
    auto store_x = [&]() {
#pragma unroll
        for (int j = 0; j < XV; ++j)
            if (j * 256 + tid < nx4) *reinterpret_cast<f32x4*>(Xs + (j * 256 + tid) * 4) = xreg[j];
    };

    issue_w(min(chunk_lo, a.nchunks - 1));
    issue_x(min(chunk_lo, a.nchunks - 1), 0);
    for (int chunk = chunk_lo; chunk < chunk_hi; ++chunk) {
        const int nchunk = min(chunk + 1, chunk_hi - 1);   // past the last chunk: a redundant (unused) prefetch of the last one
#pragma unroll
        for (int t = 0; t < NTQ; ++t) {
            if (t >= ntl) break;       // block uniform: no column of this tile exists
            __syncthreads();           // every wave is done with the previous phase's tiles
            if (t == 0) store_w();
            store_x();
            __syncthreads();
            if (t == 0) issue_w(nchunk);
            if (t + 1 < ntl) issue_x(chunk, t + 1);
            else issue_x(nchunk, 0);
            const float* xq = Xs + wc * 32 + li;
            const float* wp = Ws + (kk * 64 + wr * 32 + li) * a.KCP;   // this lane's row of its parity plane
            const int* kp = koff + kk * KH;
            for (int kb = 0; kb < KH; kb += 8) {   // 8 MFMAs: k = 2 (kb + u) + kk, u = 0 .. 7, ascending (the fmaf chain of a scalar loop)
                const float4 wa = *reinterpret_cast<const float4*>(wp + kb), wb = *reinterpret_cast<const float4*>(wp + kb + 4);
                const int4 oa = *reinterpret_cast<const int4*>(kp + kb), ob = *reinterpret_cast<const int4*>(kp + kb + 4);
                const float x0 = xq[oa.x], x1 = xq[oa.y], x2 = xq[oa.z], x3 = xq[oa.w];
                const float x4 = xq[ob.x], x5 = xq[ob.y], x6 = xq[ob.z], x7 = xq[ob.w];
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
            if (a.ksplit > 1) {   // partial sum of this K part; q < tq * 64 always
                a.part[(((size_t)kpart * d.B + b) * d.Cout + mrow) * ((size_t)gridDim.x * 64 * NTQ) + q] = v;
                continue;
            }
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

// sum of the K parts in part order + the epilogue of conv_mfma_kernel (bias, skip, transposed-conv trim + phase interleave)
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const ConvArgs a, int tqp) {
    const acmi_conv_desc& d = a.d;
    const int q = blockIdx.x * 256 + threadIdx.x, mrow = blockIdx.y, b = blockIdx.z;
    if (q >= a.Tq) return;
    float v = 0.f;
    for (int p = 0; p < a.ksplit; ++p) v += a.part[(((size_t)p * d.B + b) * d.Cout + mrow) * (size_t)tqp + q];
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
        if (o >= 0 && o < d.Tout) {
            const size_t oi = ((size_t)b * (d.Cout / d.shuffle) + co) * d.Tout + (size_t)o;
            if (a.bias) v += a.bias[co];
            if (a.res) v += a.res[oi];
            a.y[oi] = v;
        }
    }
}

// -----------------------------------------------------------------------------------------------------
// Pointwise channel-doubling convolution of a resnet block (k = 1, CI -> 2 CI, ELU on the input, bias, skip add) on a long signal.
// One thread = four (CI = 32) or two (CI = 64) consecutive time steps: its inputs are loaded once (16- / 8-byte loads, coalesced
// along time), ELU'd once, and held in registers; the weights sit in LDS and are read as wave-uniform 16-byte broadcasts; one output
// channel per loop trip, each an ascending-k fmaf chain from 0 (the order of the MFMA kernel's accumulation), then + bias, + skip like its epilogue.
// Memory-bound by construction (EnCodec-32k, 8 x 30 s, 32 -> 64 at T = 960 000: 4.9 GB once, no pack pass).
// -----------------------------------------------------------------------------------------------------
template <int CI>
__global__ __launch_bounds__(256) void conv_pw_kernel(const ConvArgs a) {
    constexpr int CO = 2 * CI, TT = CI <= 32 ? 4 : 2;   // time steps per thread: CI x TT input registers
    typedef float vec_t __attribute__((ext_vector_type(TT)));
    __shared__ __attribute__((aligned(16))) float ws[CO * CI];
    const acmi_conv_desc& d = a.d;
    const int tid = threadIdx.x, b = blockIdx.z;
    for (int i = tid; i < CO * CI / 4; i += 256) reinterpret_cast<float4*>(ws)[i] = reinterpret_cast<const float4*>(a.w)[i];
    const long long t = ((long long)blockIdx.x * 256 + tid) * TT;
    const bool live = t < d.Tout;   // Tout is a multiple of TT: every step of a live thread exists
    const long long tc = live ? t : 0;
    float xin[CI][TT];
    const float* xb = a.x + (size_t)b * CI * d.Tin + tc;
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) {
        const vec_t v = *reinterpret_cast<const vec_t*>(xb + (size_t)ci * d.Tin);
#pragma unroll
        for (int u = 0; u < TT; ++u) xin[ci][u] = v[u];
    }
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
        for (int u = 0; u < TT; ++u) { const float v = xin[ci][u]; xin[ci][u] = v > 0.f ? v : d.elu_alpha * expm1f(v); }
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < CO; ++c) {   // one output channel per trip: TT independent fmaf chains, nothing but the inputs stays live
        float acc[TT];
#pragma unroll
        for (int u = 0; u < TT; ++u) acc[u] = 0.f;
        const float* wr = ws + c * CI;
#pragma unroll
        for (int ci = 0; ci < CI; ci += 4) {
            const float4 w4 = *reinterpret_cast<const float4*>(wr + ci);   // wave uniform: one LDS broadcast
#pragma unroll
            for (int u = 0; u < TT; ++u) {
                acc[u] = fmaf(w4.x, xin[ci][u], acc[u]);
                acc[u] = fmaf(w4.y, xin[ci + 1][u], acc[u]);
                acc[u] = fmaf(w4.z, xin[ci + 2][u], acc[u]);
                acc[u] = fmaf(w4.w, xin[ci + 3][u], acc[u]);
            }
        }
        if (live) {
            const size_t oi = ((size_t)b * CO + c) * d.Tout + (size_t)t;
            const float bv = a.bias ? a.bias[c] : 0.f;
            vec_t v;
#pragma unroll
            for (int u = 0; u < TT; ++u) v[u] = acc[u] + bv;
            if (a.res) {
                const vec_t r = *reinterpret_cast<const vec_t*>(a.res + oi);
#pragma unroll
                for (int u = 0; u < TT; ++u) v[u] += r[u];
            }
            *reinterpret_cast<vec_t*>(a.y + oi) = v;
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
            xs[ci * SP + rel] = live ? conv_fetch(a, xrow, base_in + rel, GnCh{0.f, 1.f, 1.f, 0.f}) : 0.f;
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

extern "C" size_t acmi_conv1d_weight_floats(const acmi_conv_desc* dp) {
    ConvGeom g;
    if (dp == nullptr || conv_geometry(*dp, g) != ACMI_OK) return 0;
    return g.wt_floats;
}

extern "C" size_t acmi_conv1d_work_floats(const acmi_conv_desc* dp) {
    ConvGeom g;
    if (dp == nullptr || conv_geometry(*dp, g) != ACMI_OK) return 0;
    return g.work_floats;
}

extern "C" int acmi_conv1d_tile_weights(const acmi_conv_desc* dp, const float* w, float* wt, void* stream) {
    ACMI_REQUIRE(dp != nullptr && w != nullptr && wt != nullptr, "acmi_conv1d_tile_weights: null argument");
    ConvGeom g;
    if (int rc = conv_geometry(*dp, g)) return rc;
    if (g.fewout || g.pw) {   // the few-output and the pointwise kernels read the weights as they are
        if (hipMemcpyAsync(wt, w, g.wt_floats * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
            acmi_set_error("acmi_conv1d_tile_weights: hipMemcpyAsync failed");
            return ACMI_ELAUNCH;
        }
        return ACMI_OK;
    }
    TileWArgs p = {w, wt, dp->Cout, dp->Cin, dp->ksize, g.cic, g.nchunks, g.KCE, g.KCP, g.wt_floats};
    const int blocks = (int)min((g.wt_floats + 255) / 256, (size_t)16384);
    hipLaunchKernelGGL(conv_tile_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    return acmi_check_launch("conv_tile_weights_kernel");
}

struct ConvGn { const float* part; const float* gamma; const float* beta; int groups, nchunks, chunk, relu; float eps; };

static int conv1d_impl(const acmi_conv_desc* dp, const float* x, const float* wt, const float* bias, const float* residual,
                       float* y, float* work, const ConvGn* gn, void* stream) {
    ACMI_REQUIRE(dp != nullptr, "acmi_conv1d: null descriptor");
    const acmi_conv_desc& d = *dp;
    ConvGeom g;
    if (int rc = conv_geometry(d, g)) return rc;
    if (d.B == 0 || d.Tout <= 0) return ACMI_OK;
    ConvArgs a;
    a.d = d; a.x = x; a.w = wt; a.bias = bias; a.res = residual; a.y = y;
    a.gn_part = nullptr; a.gn_gamma = a.gn_beta = nullptr; a.gn_groups = a.gn_nchunks = a.gn_chunk = a.gn_relu = 0; a.gn_eps = 0.f;
    if (gn != nullptr) {
        ACMI_REQUIRE(!g.fewout && !d.elu_in, "acmi_conv1d_gn: not for the few-output kernel / together with an input ELU");
        a.gn_part = gn->part; a.gn_gamma = gn->gamma; a.gn_beta = gn->beta; a.gn_groups = gn->groups; a.gn_nchunks = gn->nchunks;
        a.gn_chunk = gn->chunk; a.gn_relu = gn->relu; a.gn_eps = gn->eps;
    }
    a.Tq = g.Tq; a.CIC = g.cic; a.KCE = g.KCE; a.KCP = g.KCP; a.LP = g.LP; a.XSZ = g.XSZ;
    a.nchunks = g.nchunks; a.cin_pad = g.nchunks * g.cic; a.Qp = g.Qp; a.lp4_magic = 0;
    a.ksplit = 1; a.cps = g.nchunks; a.part = nullptr;
    if (g.fewout) {
        dim3 grid((d.Tout + 1023) / 1024, 1, d.B), block(256);
        if (d.Cout == 1) hipLaunchKernelGGL((conv_fewout_kernel<1, 7>), grid, block, 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((conv_fewout_kernel<2, 7>), grid, block, 0, (hipStream_t)stream, a);
        return acmi_check_launch("conv_fewout_kernel");
    }
    if (g.pw) {
        ACMI_REQUIRE(gn == nullptr, "acmi_conv1d_gn: not for the pointwise kernel");
        const int tt = d.Cin == 32 ? 4 : 2;   // time steps per thread (conv_pw_kernel)
        dim3 grid((unsigned)((d.Tout / tt + 255) / 256), 1, d.B), block(256);
        if (d.Cin == 32) hipLaunchKernelGGL(conv_pw_kernel<32>, grid, block, 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL(conv_pw_kernel<64>, grid, block, 0, (hipStream_t)stream, a);
        return acmi_check_launch("conv_pw_kernel");
    }
    ACMI_REQUIRE(work != nullptr, "acmi_conv1d: work buffer (acmi_conv1d_work_floats) missing");
    ACMI_REQUIRE(a.cin_pad <= 65535 && d.B <= 65535, "acmi_conv1d: Cin=%d / B=%d exceed the grid", d.Cin, d.B);
    ACMI_REQUIRE((unsigned long long)a.cin_pad * d.stride * (unsigned long long)g.Qp < (1ULL << 32),
                 "acmi_conv1d: one batch item of the packed input exceeds 2^32 elements");
    {   // pack: padding + ELU + stride-phase split, once per element
        const long long n = (long long)d.stride * g.Qp;
        const long long per_block = 1024;
        dim3 grid((unsigned)((n + per_block - 1) / per_block), a.cin_pad, d.B);
        hipLaunchKernelGGL(conv_pack_kernel, grid, dim3(256), 0, (hipStream_t)stream, a, work);
    }
    a.x = work;
    a.lp4_magic = (unsigned)((0x100000000ULL + (unsigned)(g.LP / 4) - 1) / (unsigned)(g.LP / 4));
    // column tiles per workgroup: as many as keep >= 2 workgroups per CU in flight (4, 2 or 1); ACMI_CONV_NTQ forces one
    static const int want = getenv("ACMI_CONV_NTQ") ? atoi(getenv("ACMI_CONV_NTQ")) : 0;
    const long wgs = (long)g.tq * g.my * d.B;
    int ntq = want == 1 || want == 2 || want == 4 ? want : (wgs >= 4 * 512 ? 4 : (wgs >= 2 * 512 ? 2 : 1));
    if (g.ksplit > 1) {
        ntq = 1;
        a.ksplit = g.ksplit; a.cps = (g.nchunks + g.ksplit - 1) / g.ksplit; a.part = work + g.pack_floats;
        ACMI_REQUIRE((long)d.B * g.ksplit <= 65535 && d.Cout <= 65535, "acmi_conv1d: split-K grid too large");
    }
    const int xv = (g.XSZ / 4 + 255) / 256;   // <= 6 by the 24 KB span budget
    dim3 grid((g.tq + ntq - 1) / ntq, g.my, d.B * a.ksplit), block(256);
#define ACMI_CONV_LAUNCH(N, X) hipLaunchKernelGGL((conv_mfma_kernel<N, X>), grid, block, g.lds, (hipStream_t)stream, a)
#define ACMI_CONV_LAUNCH_X(N)                                     \
    if (xv <= 2) ACMI_CONV_LAUNCH(N, 2); else if (xv <= 4) ACMI_CONV_LAUNCH(N, 4); else ACMI_CONV_LAUNCH(N, 6)
    if (ntq == 4) { ACMI_CONV_LAUNCH_X(4); } else if (ntq == 2) { ACMI_CONV_LAUNCH_X(2); } else { ACMI_CONV_LAUNCH_X(1); }
#undef ACMI_CONV_LAUNCH_X
#undef ACMI_CONV_LAUNCH
    if (a.ksplit > 1) {
        if (int rc = acmi_check_launch("conv_mfma_kernel")) return rc;
        hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((g.Tq + 255) / 256, d.Cout, d.B), dim3(256), 0, (hipStream_t)stream, a,
                           g.tq * 64);
        return acmi_check_launch("conv_splitk_reduce_kernel");
    }
    return acmi_check_launch("conv_mfma_kernel");
}

extern "C" int acmi_conv1d(const acmi_conv_desc* dp, const float* x, const float* wt, const float* bias, const float* residual,
                           float* y, float* work, void* stream) {
    return conv1d_impl(dp, x, wt, bias, residual, y, work, nullptr, stream);
}

// conv(relu?(GroupNorm(x))): the normalisation's statistics pass (gn_partial_kernel), then the convolution whose pack pass applies
// (x - mean) rstd gamma + beta (+ ReLU) on the way into the packed layout -- three launches and ONE extra read of x instead of
// statistics + apply (read + write) + pack (read + write) + convolution (models/unet.py:32-53: norm -> ReLU -> conv of a ResBlock).
extern "C" int acmi_conv1d_gn(const acmi_conv_desc* dp, const float* x, const float* wt, const float* bias, const float* residual,
                              float* y, float* work, float* gn_work, const float* gamma, const float* beta, int groups, float eps,
                              int relu, void* stream) {
    ACMI_REQUIRE(dp != nullptr && gn_work != nullptr && gamma != nullptr && beta != nullptr, "acmi_conv1d_gn: null argument");
    ACMI_REQUIRE(groups > 0 && dp->Cin % groups == 0, "acmi_conv1d_gn: Cin=%d not divisible into %d groups", dp->Cin, groups);
    if (dp->B == 0 || dp->Tout <= 0) return ACMI_OK;
    ConvGn gn = {gn_work, gamma, beta, groups, 0, 0, relu, eps};
    if (int rc = acmi_launch_gn_partial(x, gn_work, dp->B, dp->Cin, dp->Tin, groups, &gn.nchunks, &gn.chunk, (hipStream_t)stream)) return rc;
    return conv1d_impl(dp, x, wt, bias, residual, y, work, &gn, stream);
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
__global__ __launch_bounds__(256, KI > 32 ? 2 : 1) void lstm_persistent_kernel(const float* __restrict__ gates_in,
                                                              const float* __restrict__ w_hh, float* __restrict__ cst,
                                                              const float* __restrict__ skip, float* __restrict__ y,
                                                              unsigned* hbuf, unsigned* err, int B, int H, int T, int nsplit) {
    extern __shared__ __attribute__((aligned(16))) float sh[];
    // h_{t-1} in LDS, k-major: hs[k][q], pitch 12 floats (two conflict-free ds_read_b128 fetch the 8 batch values of a k)
    constexpr int HP = 12;
    float* hs = sh;                           // [16 KI >= H][HP]; rows H .. 16 KI stay zero: the k loop needs no clamp and its LDS
                                              // addresses are one register + immediates
    float* gs = sh + (size_t)(16 * KI) * HP;  // [16][LSTM_BB]
    int* s_abort = reinterpret_cast<int*>(gs + 16 * LSTM_BB);   // workgroup-wide "give up" flag (below)
    const int tid = threadIdx.x;
    // A workgroup that starts only after another one has given up (it was not resident while the others spun) leaves at
    // once, and so does every workgroup that sees the error word set while it waits: a residency failure costs ONE bounded
    // spin, not one per remaining step.
    if (tid == 0) *s_abort = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    for (int idx = tid; idx < 16 * KI * HP; idx += 256) hs[idx] = 0.f;
    __syncthreads();
    if (*s_abort) return;
    // The batch rows are INDEPENDENT recurrences: with nsplit > 1 the grid holds nsplit copies of the (H + 3) / 4 unit
    // groups, copy s owns the rows [s Bs, (s + 1) Bs) -- half the hidden-state bytes to gather per workgroup and step, two
    // workgroups per CU pulling them.  Copies never wait for each other: a copy that is not resident simply runs later.
    const int nwg = gridDim.x / nsplit;
    const int split = blockIdx.x / nwg, Bs = (B + nsplit - 1) / nsplit;
    const int blo = split * Bs, bhi = min(B, blo + Bs);
    if (blo >= bhi) return;
    const int r = tid >> 4, ksl = tid & 15;
    const int gate = r >> 2, u = r & 3;
    const int j0 = (blockIdx.x - split * nwg) * 4;
    const bool jvalid = j0 + u < H;
    const float* wrow = w_hh + ((size_t)gate * H + (jvalid ? j0 + u : 0)) * H;
    float wv[KI];
#pragma unroll
    for (int i = 0; i < KI; ++i) wv[i] = (ksl + 16 * i < H) ? wrow[ksl + 16 * i] : 0.f;
    const size_t BH = (size_t)B * H;
    const int H4 = H >> 2;                       // 16-byte groups per row (H % 4 == 0 is required by the launcher)
    const bool one_pass = bhi - blo <= LSTM_BB;
    float c_reg = 0.f;                           // cell state of this thread's (unit, row) when B fits one pass
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(hbuf, 0, (int)(3 * BH * 4), 0x00020000);

    for (int t = 0; t < T; ++t) {
        const unsigned prev_off = (unsigned)(((t + 2) % 3) * BH);   // buffer of step t - 1
        unsigned* hnext = hbuf + (size_t)(t % 3) * BH;              // buffer of step t
        unsigned* hrearm = hbuf + (size_t)((t + 1) % 3) * BH;       // buffer of step t + 1 (still holds step t - 2)
        for (int b0 = blo; b0 < bhi; b0 += LSTM_BB) {
            const int nb = min(LSTM_BB, bhi - b0);
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
                if (!one_pass)   // one pass: rows past nb were zeroed before the first step and are never written
                    for (int idx = tid; idx < (LSTM_BB - nb) * H; idx += 256) hs[(idx % H) * HP + nb + idx / H] = 0.f;
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
            const float* hk = hs + ksl * HP;
#pragma unroll
            for (int i = 0; i < KI; ++i) {
                const float4 h0 = *reinterpret_cast<const float4*>(hk + i * 16 * HP);
                const float4 h1 = *reinterpret_cast<const float4*>(hk + i * 16 * HP + 4);
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

// -----------------------------------------------------------------------------------------------------
// XCD-local form (H = 512, 768, 1024; the numbers below are H = 1024's): ONE RECURRENCE PER XCD.  The batch rows are independent recurrences and an MI355X is 8 XCDs
// of 32 CUs, each with its own L2: workgroup b (512 threads, one per CU) runs on XCD b % 8 and owns the 32 hidden units
// [32 (b / 8), +32) -- 128 gate rows of W_hh, 512 KB: of every thread's 8 rows x 8 column chunks 45 chunks in registers
// (180 floats), 19 in LDS (152 KB) -- for the batch rows r = b % 8, b % 8 + 8, ...  So the 32 workgroups of an XCD hold ALL of W_hh and a step's
// all-gather moves the 4 KB of ONE hidden vector between 32 CUs through the XCD's own L2 (plain stores; `nt` polling
// loads), instead of 32 KB for 8 rows between 256 CUs through the memory side
// (lstm_persistent_kernel: 5.1-5.4 us per step; lab/xcd_local_lab.hip: 0.97 us per hand-off inside an XCD, 1.22-1.33
// through the memory side).  Every (row, step) has its OWN 4 KB slot in the exchange array [B][T][H], pre-filled with the
// LSTM_EMPTY pattern (the data is its own flag, as above): no re-arming, no rotation.
// The XCD of a workgroup is not an API promise: every workgroup records its XCC_ID in its group's word (first writer
// wins) and a workgroup that finds another id there raises the error word -- every spin is bounded and polls that word.
// MEM = 1: write-through stores and agent-scope loads (correct for any placement; A/B).
// -----------------------------------------------------------------------------------------------------
// geometry of the form for hidden size HH (512, 768, 1024): 32 workgroups per XCD, HH / 32 hidden units (HH / 8 gate rows) each;
// a thread = 8 gate rows x one of 32 column slices: NI 16-byte column chunks per row, 8 NI chunks in all, the first REGCH of
// them (row-major) in registers, the rest in LDS
template <int HH> struct LxCfg {
    static constexpr int UNITS = HH / 32;
    static constexpr int THREADS = HH / 2;                    // (4 UNITS / 8) row groups x 32 column slices
    static constexpr int NI = HH / 128;
    static constexpr int REGCH = HH == 1024 ? 45 : HH == 768 ? 40 : 8 * NI;
    static constexpr int SLABS = 8 * NI - REGCH;
    static constexpr size_t LDS = (size_t)(SLABS * THREADS * 4 + HH + 8 * UNITS) * sizeof(float) + 16;
};
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
// compile-time loop: the body sees its index as a constant expression (`if constexpr` on it: no dead out-of-range access ever
// reaches the optimiser, which otherwise leaves the weight array in scratch memory)
template <int B_, int E_, typename F>
__device__ __forceinline__ void lx_static_for(F&& f) {
    if constexpr (B_ < E_) {
        f(std::integral_constant<int, B_>{});
        lx_static_for<B_ + 1, E_>(f);
    }
}

template <int HH, int MEM>
__global__ __launch_bounds__(LxCfg<HH>::THREADS, 1) void lstm_xcd_kernel(const float* __restrict__ gates_in, const float* __restrict__ w_hh,
                                                                         const float* __restrict__ skip, float* __restrict__ y, unsigned* hx,
                                                                         unsigned* xcc_of_group, unsigned* err, int B, int T) {
    using CF = LxCfg<HH>;
    constexpr int H = HH, UNITS = CF::UNITS, THREADS = CF::THREADS, NI = CF::NI, REGCH = CF::REGCH;
    extern __shared__ __attribute__((aligned(16))) float sh[];
    float4* wl = reinterpret_cast<float4*>(sh);                 // [SLABS][THREADS]: the weight chunks that do not fit the registers
    float* hs = sh + CF::SLABS * THREADS * 4;                   // h_{t-1}: [H]
    float* gs = hs + H;                                         // gate sums: [4 gates][UNITS]
    float* ga = gs + 4 * UNITS;                                 // gate activations: [4 gates][UNITS]
    int* s_abort = reinterpret_cast<int*>(ga + 4 * UNITS);
    const int tid = threadIdx.x;
    const int grp = blockIdx.x & 7, idx = blockIdx.x >> 3;
    if (grp >= B) return;                                        // no batch row for this XCD
    if (tid == 0) {
        int bad = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        if (MEM != 1 && !bad) {
            const unsigned me = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xfu;   // XCC_ID
            const unsigned seen = atomicCAS(xcc_of_group + grp, 0xffffffffu, me);
            if (seen != 0xffffffffu && seen != me) { atomicAdd(err, 0x10000u); bad = 1; }   // (high half: placement, low half: give-ups)
        }
        *s_abort = bad;
    }
    const int rg = tid >> 5, ksl = tid & 31;                     // UNITS / 2 groups of 8 local rows x 32 column slices
    const int gate = rg / (UNITS / 8), u0 = (rg % (UNITS / 8)) * 8, j0 = idx * UNITS;
    // this thread's weights: rows gate H + j0 + u0 + r (r < 8), columns (i 32 + ksl) 4 .. + 3 (i < NI); chunk q = r NI + i
    float4 wv[REGCH];
    {
        const float* wbase = w_hh + ((size_t)gate * H + j0 + u0) * H + ksl * 4;
        lx_static_for<0, 8>([&](auto rc) {
            lx_static_for<0, NI>([&](auto ic) {
                constexpr int r = decltype(rc)::value, i = decltype(ic)::value, q = r * NI + i;
                const float4 w = *reinterpret_cast<const float4*>(wbase + (size_t)r * H + i * 128);
                if constexpr (q < REGCH) wv[q] = w;
                else wl[(q - REGCH) * THREADS + tid] = w;
            });
        });
    }
    __syncthreads();
    if (*s_abort) return;
    const bool owner = tid < UNITS;                              // thread u: state of hidden unit j0 + u
    const int j = j0 + tid;
    // the four gate activations of a unit are computed side by side: lanes 0 .. UNITS - 1 of wave g (one wave per SIMD) take gate g
    const int ag = tid >> 6, au = tid & 63;
    const bool act = ag < 4 && au < UNITS;
    const size_t hx_bytes = (size_t)B * T * H * 4;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(hx, 0, (int)hx_bytes, 0x00020000);
    // MEM 0: plain stores (write-through L1 -> the XCD's L2) and `nt` polling loads, which are served by the L2 every time
    // (measured, profiles/archive/r04_session33_lstm.log: an `sc0` load keeps returning the CU's stale L1 line; `buffer_inv sc1` before a plain
    // load works at 29 us per step); MEM 1: write-through stores, agent-scope loads (memory side)
    constexpr int LD_AUX = MEM == 1 ? 16 : 2, ST_AUX = MEM == 1 ? 16 : 0;

    for (int row = grp; row < B; row += 8) {
        float c_reg = 0.f;
        for (int t = 0; t < T; ++t) {
            float gin = 0.f;                                     // requested before the wait for h_{t-1}
            if (act) gin = gates_in[((size_t)row * 4 * H + (size_t)ag * H + j0 + au) * T + t];
            if (t == 0) {
                hs[2 * tid] = 0.f; hs[2 * tid + 1] = 0.f;
            } else {
                const unsigned off = (unsigned)((((size_t)row * T + (t - 1)) * H + 2 * tid) * 4);
                u32x2_t v;
                unsigned spins = 0;
                for (;;) {
                    v = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, LD_AUX);
                    if (v[0] != LSTM_EMPTY && v[1] != LSTM_EMPTY) break;
                    if ((++spins & 0xffu) == 0u) {               // bounded wait, device-wide abort word
                        if (spins > 400000u) { atomicAdd(err, 1u); *s_abort = 1; break; }
                        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { *s_abort = 1; break; }
                    }
                }
                *reinterpret_cast<float2*>(hs + 2 * tid) = make_float2(__uint_as_float(v[0]), __uint_as_float(v[1]));
            }
            __syncthreads();
            if (*s_abort) return;
            // ---- 8 rows x NI chunks per thread: weights from registers / LDS, h from LDS
            float acc[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] = 0.f;
            const float4* h4 = reinterpret_cast<const float4*>(hs) + ksl;
            lx_static_for<0, NI>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const float4 h = h4[i * 32];
                lx_static_for<0, 8>([&](auto rc) {
                    constexpr int r = decltype(rc)::value, q = r * NI + i;
                    float4 w;
                    if constexpr (q < REGCH) w = wv[q];
                    else w = wl[(q - REGCH) * THREADS + tid];
                    acc[r] = fmaf(w.x, h.x, acc[r]); acc[r] = fmaf(w.y, h.y, acc[r]);
                    acc[r] = fmaf(w.z, h.z, acc[r]); acc[r] = fmaf(w.w, h.w, acc[r]);
                });
                if (CF::SLABS > 0) __builtin_amdgcn_sched_barrier(0);   // one chunk's LDS operands in flight: the registers are full of weights
            });
            // sum over the 32 column slices: the 16 lanes of a row (DPP butterflies), then row_bcast:15 adds the lower row's total to
            // every lane of the upper one: lanes 16 .. 31 / 48 .. 63 hold the totals of the wave's two row groups
#pragma unroll
            for (int r = 0; r < 8; ++r) { acc[r] = row16_sum(acc[r]); acc[r] += dpp_f32<0x142>(acc[r]); }
            if (ksl == 31) {
                *reinterpret_cast<float4*>(gs + rg * 8) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                *reinterpret_cast<float4*>(gs + rg * 8 + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
            }
            __syncthreads();
            if (act) {                                           // i, f, o: sigmoid; g: tanh (nn.LSTM's gate order i, f, g, o)
                const float x = gs[ag * UNITS + au] + gin;
                ga[ag * UNITS + au] = ag == 2 ? tanhf(x) : 1.f / (1.f + expf(-x));
            }
            __syncthreads();
            if (tid < 64) {                                      // wave 0: lanes 0 .. UNITS - 1 own a unit each (the DPP exchange needs the whole wave)
                float hn = 0.f;
                if (owner) {
                    const float cn = ga[UNITS + tid] * c_reg + ga[tid] * ga[2 * UNITS + tid];
                    hn = ga[3 * UNITS + tid] * tanhf(cn);
                    c_reg = cn;
                }
                // publish h_t: a quad's 4 units as one 16-byte store
                const unsigned q0 = __float_as_uint(dpp_f32<0x00>(hn)), q1 = __float_as_uint(dpp_f32<0x55>(hn));
                const unsigned q2 = __float_as_uint(dpp_f32<0xAA>(hn)), q3 = __float_as_uint(dpp_f32<0xFF>(hn));
                if (owner && (tid & 3) == 0)
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{q0, q1, q2, q3}, rs,
                                                           (unsigned)((((size_t)row * T + t) * H + j) * 4), 0, ST_AUX);
                if (owner) {
                    const size_t yi = ((size_t)row * H + j) * T + t;
                    y[yi] = skip ? hn + skip[yi] : hn;
                }
            }
        }
    }
}

// -----------------------------------------------------------------------------------------------------
// Two-layer wavefront: ONE launch for a 2-layer nn.LSTM stack (EnCodec's `lstm=2`).  2 x ceil(H / 4) workgroups: the first half
// runs layer 0 exactly like lstm_persistent_kernel and additionally leaves every h1_t in a [T][B, H] exchange array (the
// data is its own flag again: all slots start EMPTY, nothing is ever re-armed, so layer 1 may lag by any number of steps);
// the second half runs layer 1 one step behind: per step it gathers h1_t (long since there), multiplies it with its register
// resident slice of W_ih1 (the input projection that used to be a k = 1 convolution over all T between the two launches),
// then does the recurrence step on its own rotating buffers.  2 T dependent all-gathers become T + 1.
// Layer 0 never waits for layer 1 and is dispatched first: if only half of the grid is resident the launch degrades to
// "layer 0, then layer 1" instead of dead-locking; the per-layer residency check is the same as for the single-layer form.
// -----------------------------------------------------------------------------------------------------
#define LSTM_WAVE2_WL 28   // W_ih1 values per thread kept in LDS at H > 512

template <int KI>
__global__ __launch_bounds__(256, KI > 32 ? 2 : 1) void lstm_wave2_kernel(const float* __restrict__ gates_in0, const float* __restrict__ w_hh0,
                                                         const float* __restrict__ w_ih1, const float* __restrict__ w_hh1,
                                                         const float* __restrict__ bias1, float* __restrict__ cst,
                                                         const float* __restrict__ skip, float* __restrict__ y, unsigned* hbuf,
                                                         unsigned* x01, unsigned* err, int B, int H, int T) {
    extern __shared__ __attribute__((aligned(16))) float sh[];
    constexpr int HP = 12;
    constexpr int GV = KI > 32 ? 4 : 8;   // 16-byte groups a thread polls at once / k steps per LDS read batch: two weight slices
    constexpr int MB = KI > 32 ? 4 : 8;   // (128 VGPRs at H = 1024) must leave room for two workgroups per CU (256 VGPRs)
    float* hs = sh;                                // [16 KI >= H][HP]: the hidden vector being multiplied (h1_t, then h_{t-1} of this layer);
                                                   // rows H .. 16 KI stay zero, so the k loop needs no clamp and its LDS addresses are
                                                   // one register + immediates (64 clamped addresses were 64 more VGPRs)
    float* gs = sh + (size_t)(16 * KI) * HP;       // [16][LSTM_BB] recurrent part of the gates
    float* gsi = gs + 16 * LSTM_BB;                // [16][LSTM_BB] input part of the gates (layer 1)
    int* s_abort = reinterpret_cast<int*>(gsi + 16 * LSTM_BB);
    // Two weight slices at H = 1024 are 128 VGPRs per thread; with the ~145 the rest of the step needs that does not fit the 256
    // a thread gets at two workgroups per CU (the compiler spilled ~150 of them to scratch and re-read 120 per step).  The last
    // WL values of the W_ih1 slice live in LDS instead ([WL][256] floats behind the abort flag: conflict-free ds_read_b32).
    constexpr int WL = KI > 32 ? LSTM_WAVE2_WL : 0;
    float* wl = reinterpret_cast<float*>(s_abort + 4);
    const int tid = threadIdx.x;
    if (tid == 0) *s_abort = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    for (int idx = tid; idx < 16 * KI * HP; idx += 256) hs[idx] = 0.f;
    __syncthreads();
    if (*s_abort) return;
    const int nwg = gridDim.x >> 1;
    const int role = blockIdx.x >= nwg ? 1 : 0, g = blockIdx.x - role * nwg;   // block uniform
    const int r = tid >> 4, ksl = tid & 15;
    const int gate = r >> 2, u = r & 3;
    const int j0 = g * 4;
    const bool jvalid = j0 + u < H;
    const size_t wrow_off = ((size_t)gate * H + (jvalid ? j0 + u : 0)) * H;
    const float* wrow = (role ? w_hh1 : w_hh0) + wrow_off;
    float wv[KI], wi[KI - WL];
#pragma unroll
    for (int i = 0; i < KI; ++i) wv[i] = (ksl + 16 * i < H) ? wrow[ksl + 16 * i] : 0.f;
#pragma unroll
    for (int i = 0; i < KI - WL; ++i) wi[i] = (role && ksl + 16 * i < H) ? w_ih1[wrow_off + ksl + 16 * i] : 0.f;
#pragma unroll
    for (int i = KI - WL; i < KI; ++i) wl[(i - (KI - WL)) * 256 + tid] = (role && ksl + 16 * i < H) ? w_ih1[wrow_off + ksl + 16 * i] : 0.f;
    const size_t BH = (size_t)B * H;
    const int H4 = H >> 2;
    const bool one_pass = B <= LSTM_BB;
    float c_reg = 0.f;
    float* cmy = cst + (size_t)role * BH;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(hbuf + (size_t)role * 3 * BH, 0, (int)(3 * BH * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(x01, 0, (int)((size_t)T * BH * 4), 0x00020000);

    // all-gather of nb rows x H values starting at element `off` of `src` into hs (k-major), rows past nb zeroed
    auto gather = [&](const __amdgpu_buffer_rsrc_t src, unsigned off, int nb) {
        const int ngr = nb * H4;
        for (int base = 0; base < LSTM_BB * H4; base += 256 * GV) {
            u32x4_t v[GV];
            unsigned pending = 0;
#pragma unroll
            for (int i = 0; i < GV; ++i)
                if (base + i * 256 + tid < ngr) pending |= 1u << i;
            unsigned spins = 0;
            while (pending) {
#pragma unroll
                for (int i = 0; i < GV; ++i)
                    if (pending & (1u << i))
                        v[i] = __builtin_amdgcn_raw_buffer_load_b128(src, (off + (unsigned)(base + i * 256 + tid) * 4u) * 4u, 0, 16 /* sc1 */);
#pragma unroll
                for (int i = 0; i < GV; ++i)
                    if ((pending & (1u << i)) && v[i][0] != LSTM_EMPTY && v[i][1] != LSTM_EMPTY && v[i][2] != LSTM_EMPTY &&
                        v[i][3] != LSTM_EMPTY) {
                        const int idx = base + i * 256 + tid, q = idx / H4, k = (idx - q * H4) * 4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) hs[(k + e) * HP + q] = __uint_as_float(v[i][e]);
                        pending &= ~(1u << i);
                    }
                if (pending && (++spins & 0x3ffu) == 0u) {
                    if (spins > 4000000u) { atomicAdd(err, 1u); *s_abort = 1; break; }
                    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { *s_abort = 1; break; }
                }
            }
        }
        if (!one_pass)   // one pass: rows past nb were zeroed before the first step and are never written
            for (int idx = tid; idx < (LSTM_BB - nb) * H; idx += 256) hs[(idx % H) * HP + nb + idx / H] = 0.f;
    };
    // Layer 1, one batch pass, H <= 512 (4 groups per thread): the gather of h1_{t+1} is ISSUED during step t, as soon as h1_t
    // has been consumed -- layer 0 runs ahead, the data is there -- and completed at the start of step t + 1: its round trip to
    // the memory side (~1.2 us) runs under the recurrence step instead of in front of the next one.
    constexpr bool PF = KI <= 32;
    constexpr int GP = 4;
    u32x4_t pv[GP];
    unsigned ppend = 0;
    auto prefetch_issue = [&](unsigned off, int nb) {
        const int ngr = nb * H4;
        ppend = 0;
#pragma unroll
        for (int i = 0; i < GP; ++i)
            if (i * 256 + tid < ngr) {
                ppend |= 1u << i;
                pv[i] = __builtin_amdgcn_raw_buffer_load_b128(rsx, (off + (unsigned)(i * 256 + tid) * 4u) * 4u, 0, 16 /* sc1 */);
            }
    };
    auto prefetch_finish = [&](unsigned off, int nb) {
        unsigned spins = 0;
        bool reload = false;
        while (ppend) {
            if (reload) {
#pragma unroll
                for (int i = 0; i < GP; ++i)
                    if (ppend & (1u << i))
                        pv[i] = __builtin_amdgcn_raw_buffer_load_b128(rsx, (off + (unsigned)(i * 256 + tid) * 4u) * 4u, 0, 16 /* sc1 */);
            }
            reload = true;
#pragma unroll
            for (int i = 0; i < GP; ++i)
                if ((ppend & (1u << i)) && pv[i][0] != LSTM_EMPTY && pv[i][1] != LSTM_EMPTY && pv[i][2] != LSTM_EMPTY &&
                    pv[i][3] != LSTM_EMPTY) {
                    const int idx = i * 256 + tid, q = idx / H4, kk = (idx - q * H4) * 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) hs[(kk + e) * HP + q] = __uint_as_float(pv[i][e]);
                    ppend &= ~(1u << i);
                }
            if (ppend && (++spins & 0x3ffu) == 0u) {
                if (spins > 4000000u) { atomicAdd(err, 1u); *s_abort = 1; break; }
                if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { *s_abort = 1; break; }
            }
        }
    };
    // 16 gate rows x H against the LSTM_BB vectors in hs -> dst [16][LSTM_BB]
    auto matvec = [&](const auto& w, const int nreg, float* dst) {   // w[i] for i < nreg, the LDS tail behind it
        float acc[LSTM_BB];
#pragma unroll
        for (int q = 0; q < LSTM_BB; ++q) acc[q] = 0.f;
        const float* hk = hs + ksl * HP;
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            const float4 h0 = *reinterpret_cast<const float4*>(hk + i * 16 * HP);
            const float4 h1 = *reinterpret_cast<const float4*>(hk + i * 16 * HP + 4);
            const float wk = i < nreg ? w[i < nreg ? i : 0] : wl[(i - nreg) * 256 + tid];
            acc[0] = fmaf(wk, h0.x, acc[0]); acc[1] = fmaf(wk, h0.y, acc[1]);
            acc[2] = fmaf(wk, h0.z, acc[2]); acc[3] = fmaf(wk, h0.w, acc[3]);
            acc[4] = fmaf(wk, h1.x, acc[4]); acc[5] = fmaf(wk, h1.y, acc[5]);
            acc[6] = fmaf(wk, h1.z, acc[6]); acc[7] = fmaf(wk, h1.w, acc[7]);
            if ((i & (MB - 1)) == MB - 1) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < LSTM_BB; ++q) acc[q] = row16_sum(acc[q]);
        if (ksl == 0) {
#pragma unroll
            for (int q = 0; q < LSTM_BB; ++q) dst[r * LSTM_BB + q] = acc[q];
        }
    };

    float b1[4] = {0.f, 0.f, 0.f, 0.f};   // layer 1: this thread's four gate biases (one pass: fixed unit for the whole run)
    if (role == 1 && (tid & 3) + j0 < H) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) b1[g4] = bias1[(size_t)g4 * H + j0 + (tid & 3)];
    }
    for (int t = 0; t < T; ++t) {
        const unsigned prev_off = (unsigned)(((t + 2) % 3) * BH);
        for (int b0 = 0; b0 < B; b0 += LSTM_BB) {
            const int nb = min(LSTM_BB, B - b0);
            const int uu = tid & 3, bb = tid >> 2, j = j0 + uu, bidx = b0 + bb;
            const bool owner = tid < 4 * LSTM_BB && bb < nb && j < H;
            float gin[4] = {0.f, 0.f, 0.f, 0.f};
            if (role == 0) {
                if (owner) {
                    const size_t gbase = ((size_t)bidx * 4 * H + j) * T + t;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) gin[g4] = gates_in0[gbase + (size_t)g4 * H * T];
                }
            } else {
                // input part of layer 1's gates: W_ih1 h1_t (+ bias); h1_t was published by layer 0 at least a step ago
                if (PF && one_pass && t > 0) prefetch_finish((unsigned)((size_t)t * BH), nb);
                else gather(rsx, (unsigned)((size_t)t * BH) + (unsigned)(b0 * H), nb);
                __syncthreads();
                if (*s_abort) return;
                matvec(wi, KI - WL, gsi);
                __syncthreads();       // gsi complete, every wave done with hs
                // h1_{t+1} (layer 0 runs ahead: it is there) requested NOW: its round trip hides under this whole step
                if (PF && one_pass && t + 1 < T) prefetch_issue((unsigned)((size_t)(t + 1) * BH), nb);
                if (owner) {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) gin[g4] = gsi[(g4 * 4 + uu) * LSTM_BB + bb] + b1[g4];
                }
            }
            // ---- the recurrence step of this layer (lstm_persistent_kernel's, see there)
            if (t == 0) {
                for (int idx = tid; idx < H * HP; idx += 256) hs[idx] = 0.f;
            } else {
                gather(rs, prev_off + (unsigned)(b0 * H), nb);
            }
            __syncthreads();
            if (*s_abort) return;
            const unsigned slot_off = (unsigned)(((size_t)bidx * H + j0) * 4u);
            if (owner && uu == 0)
                __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{LSTM_EMPTY, LSTM_EMPTY, LSTM_EMPTY, LSTM_EMPTY}, rs,
                                                       (unsigned)(((t + 1) % 3) * BH * 4) + slot_off, 0, 16 /* sc1 */);
            matvec(wv, KI, gs);
            __syncthreads();
            float c_hn = 0.f;
            if (owner) {
                const float gi = gs[(0 * 4 + uu) * LSTM_BB + bb] + gin[0];
                const float gf = gs[(1 * 4 + uu) * LSTM_BB + bb] + gin[1];
                const float gg = gs[(2 * 4 + uu) * LSTM_BB + bb] + gin[2];
                const float go = gs[(3 * 4 + uu) * LSTM_BB + bb] + gin[3];
                const float ig = 1.f / (1.f + expf(-gi));
                const float fg = 1.f / (1.f + expf(-gf));
                const float og = 1.f / (1.f + expf(-go));
                const size_t si = (size_t)bidx * H + j;
                const float cp = t == 0 ? 0.f : (one_pass ? c_reg : cmy[si]);
                const float cn = fg * cp + ig * tanhf(gg);
                const float hn = og * tanhf(cn);
                c_reg = cn;
                if (!one_pass) cmy[si] = cn;
                c_hn = hn;
            }
            {
                const unsigned h0 = __float_as_uint(dpp_f32<0x00>(c_hn)), h1 = __float_as_uint(dpp_f32<0x55>(c_hn));
                const unsigned h2 = __float_as_uint(dpp_f32<0xAA>(c_hn)), h3 = __float_as_uint(dpp_f32<0xFF>(c_hn));
                if (owner && uu == 0) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the re-arm store of this step has landed
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{h0, h1, h2, h3}, rs, (unsigned)((t % 3) * BH * 4) + slot_off, 0,
                                                           16 /* sc1 */);
                    if (role == 0)   // the copy layer 1 reads, after the one this layer's own recurrence waits for
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{h0, h1, h2, h3}, rsx,
                                                               (unsigned)((size_t)t * BH * 4) + slot_off, 0, 16 /* sc1 */);
                }
            }
            if (owner && role == 1) {
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

// The recurrence kernels' work areas are armed by a KERNEL, not by hipMemset*Async: inside a replayed hipGraph a large memset
// node was seen to take effect late (replay >= 1 of a captured 8 x 1024 x 200 layer found the previous replay's words,
// profiles/archive/r04_session36_lstm_graph.log); a kernel node is ordered like every other launch of the pass.
__global__ __launch_bounds__(256) void lstm_fill_kernel(unsigned* a, size_t na, unsigned va, unsigned* b, size_t nb, unsigned vb) {
    const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    for (size_t i = i0; i < na; i += step) a[i] = va;
    for (size_t i = i0; i < nb; i += step) b[i] = vb;
}
static int lstm_fill(unsigned* a, size_t na, unsigned va, unsigned* b, size_t nb, unsigned vb, hipStream_t st) {
    const size_t n = na > nb ? na : nb;
    hipLaunchKernelGGL(lstm_fill_kernel, dim3((unsigned)min((size_t)2048, (n + 255) / 256)), dim3(256), 0, st, a, na, va, b, nb, vb);
    return acmi_check_launch("lstm_fill_kernel");
}

// Every workgroup of the persistent form must be RESIDENT for its all-gather to complete.  The grid ((H + 3) / 4
// workgroups) is checked against what the device this call runs on can hold: CUs (hipDeviceProp_t.multiProcessorCount:
// 256 on a whole MI355X, fewer on a partitioned one) x the occupancy the runtime reports for this kernel with its LDS,
// minus one workgroup per CU of margin when more than one fits (the API answers one too many at some SGPR counts,
// MI355X_MICROARCH.md "Residency and cooperative launch").  Anything that does not fit runs the per-step kernel.
// Residency can still be lost to OTHER work on the device (another stream or process): the kernel's spins are bounded,
// the first give-up raises a device-wide abort flag (the err word) that every workgroup polls, and the host raises.
template <typename KernelT>
static bool lstm_grid_resident(KernelT kernel, int grid, size_t lds_bytes, int threads = 256) {
    // answers are remembered per (kernel, grid, LDS, device): the queries are not free, and a call that arrives while its
    // stream is being captured into a hipGraph must not need them
    struct Seen { const void* k; int grid, dev; size_t lds; bool ok; };
    static thread_local Seen seen[16];
    static thread_local int n_seen = 0;
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const void* kp = reinterpret_cast<const void*>(kernel);
    for (int i = 0; i < n_seen; ++i)
        if (seen[i].k == kp && seen[i].grid == grid && seen[i].dev == dev && seen[i].lds == lds_bytes) return seen[i].ok;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return false;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds_bytes) != hipSuccess || per_cu <= 0) return false;
    if (per_cu > 1) per_cu -= 1;
    const bool ok = (long)grid <= (long)cus * per_cu;
    if (n_seen < 16) seen[n_seen++] = Seen{kp, grid, dev, lds_bytes, ok};
    return ok;
}

static int lstm_persistent_ok(int B, int H) {
    static const int want = [] { const char* e = getenv("ACMI_LSTM_PERSISTENT"); return (e && e[0] == '0') ? 0 : 1; }();
    return want && H <= 1024 && H % 4 == 0 && B >= 1;
}



// work of the XCD-local form: the legacy layout (5 B H + 4 floats: its err word stays at 5 B H), 12 words (XCC id per group),
// then the exchange array [B][T][H]
static size_t lstm_xcd_work_floats(int B, int H, int T) { return (size_t)5 * B * H + 4 + 12 + (size_t)B * T * H; }
// every XCD reads ALL of W_hh at the start (8 x 16.8 MB at H = 1024): a handful of steps is cheaper on the all-CU form
static bool lstm_xcd_shape(int B, int H, int T) {
    return (H == 512 || H == 768 || H == 1024) && B > 0 && T >= 16 && (size_t)B * T * H * 4 <= ((size_t)1 << 30);
}
extern "C" size_t acmi_lstm_layer_work_floats(int B, int H, int T) {
    const size_t legacy = (size_t)5 * B * H + 4;
    return (lstm_xcd_shape(B, H, T)) ? lstm_xcd_work_floats(B, H, T) : legacy;
}

// arms a launch of lstm_xcd_kernel: the XCC words of the 8 groups and every slot of the exchange array (a kernel, like
// lstm_fill: replay >= 1 of a captured 8 x 1024 x 200 layer armed by memset nodes had all 256 workgroups find the previous replay's
// XCC words)
__global__ __launch_bounds__(256) void lstm_xcd_arm_kernel(u32x4_t* hx4, size_t n4, unsigned* xcc_of_group) {
    if (blockIdx.x == 0 && threadIdx.x < 12) xcc_of_group[threadIdx.x] = 0xffffffffu;
    const u32x4_t e = {LSTM_EMPTY, LSTM_EMPTY, LSTM_EMPTY, LSTM_EMPTY};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) hx4[i] = e;
}

template <int HH>
static int lstm_launch_xcd(int mode, const float* gates_in, const float* w_hh, const float* skip, float* y, unsigned* hx,
                           unsigned* xcc_of_group, unsigned* err, int B, int T, hipStream_t st, bool* launched) {
    constexpr size_t lds = LxCfg<HH>::LDS;
    const int grid = 8 * 32;
    static const bool attr_ok =   // once per instantiation, thread safe
        hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_xcd_kernel<HH, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_xcd_kernel<HH, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
    if (!attr_ok) return ACMI_OK;   // the other forms remain
    if (!lstm_grid_resident(lstm_xcd_kernel<HH, 0>, grid, lds, LxCfg<HH>::THREADS)) return ACMI_OK;
    {
        const size_t n4 = (size_t)B * T * HH / 4;   // (hx is 16-byte aligned: 5 B H + 16 words into a 256-byte aligned area, H % 4 == 0)
        hipLaunchKernelGGL(lstm_xcd_arm_kernel, dim3((unsigned)min((size_t)2048, (n4 + 255) / 256)), dim3(256), 0, st,
                           reinterpret_cast<u32x4_t*>(hx), n4, xcc_of_group);
    }
    if (mode == 2) hipLaunchKernelGGL((lstm_xcd_kernel<HH, 1>), dim3(grid), dim3(LxCfg<HH>::THREADS), lds, st, gates_in, w_hh, skip, y, hx, xcc_of_group, err, B, T);
    else hipLaunchKernelGGL((lstm_xcd_kernel<HH, 0>), dim3(grid), dim3(LxCfg<HH>::THREADS), lds, st, gates_in, w_hh, skip, y, hx, xcc_of_group, err, B, T);
    *launched = true;
    return acmi_check_launch("lstm_xcd_kernel");
}

// the XCD-local form when it applies: H = 512 / 768 / 1024, the caller's work area holds the exchange array, 8 XCDs x 32 CUs
// that can each hold one workgroup of the form.  ACMI_LSTM_XCD = 0: off; 2: memory-side stores / loads (A/B)
static int lstm_try_xcd(const float* gates_in, const float* w_hh, const float* skip, float* y, float* work, size_t work_floats,
                        int B, int H, int T, hipStream_t st, bool* launched) {
    *launched = false;
    const char* env = getenv("ACMI_LSTM_XCD");   // read per call: tests switch it
    const int mode = env != nullptr ? atoi(env) : 1;
    if (mode == 0 || !lstm_xcd_shape(B, H, T) || work_floats < lstm_xcd_work_floats(B, H, T)) return ACMI_OK;
    unsigned* err = reinterpret_cast<unsigned*>(work + (size_t)5 * B * H);
    unsigned* xcc_of_group = err + 4;
    unsigned* hx = err + 16;
    if ((reinterpret_cast<size_t>(hx) & 15) != 0) return ACMI_OK;   // 16-byte stores into the exchange array
    if (H == 1024) return lstm_launch_xcd<1024>(mode, gates_in, w_hh, skip, y, hx, xcc_of_group, err, B, T, st, launched);
    if (H == 768) return lstm_launch_xcd<768>(mode, gates_in, w_hh, skip, y, hx, xcc_of_group, err, B, T, st, launched);
    return lstm_launch_xcd<512>(mode, gates_in, w_hh, skip, y, hx, xcc_of_group, err, B, T, st, launched);
}

// would a layer of this shape (given a large enough work area) run the XCD-local form on the current device?
static bool lstm_xcd_would_run(int B, int H, int T) {
    const char* env = getenv("ACMI_LSTM_XCD");
    if ((env != nullptr && atoi(env) == 0) || !lstm_xcd_shape(B, H, T)) return false;
    if (H == 1024) return lstm_grid_resident(lstm_xcd_kernel<1024, 0>, 256, LxCfg<1024>::LDS, LxCfg<1024>::THREADS);
    if (H == 768) return lstm_grid_resident(lstm_xcd_kernel<768, 0>, 256, LxCfg<768>::LDS, LxCfg<768>::THREADS);
    return lstm_grid_resident(lstm_xcd_kernel<512, 0>, 256, LxCfg<512>::LDS, LxCfg<512>::THREADS);
}

static int lstm_layer_impl(const float* gates_in, const float* w_hh, const float* skip, float* y, float* work, size_t work_floats,
                           int B, int H, int T, void* stream);

extern "C" int acmi_lstm_layer(const float* gates_in, const float* w_hh, const float* skip, float* y, float* work, int B,
                               int H, int T, void* stream) {
    return lstm_layer_impl(gates_in, w_hh, skip, y, work, (size_t)5 * B * H + 4, B, H, T, stream);
}
extern "C" int acmi_lstm_layer_ex(const float* gates_in, const float* w_hh, const float* skip, float* y, float* work,
                                  size_t work_floats, int B, int H, int T, void* stream) {
    ACMI_REQUIRE(B > 0 && H > 0 && work_floats >= (size_t)5 * B * H + 4, "acmi_lstm_layer_ex: work area of %zu floats is too small", work_floats);
    return lstm_layer_impl(gates_in, w_hh, skip, y, work, work_floats, B, H, T, stream);
}

static int lstm_layer_impl(const float* gates_in, const float* w_hh, const float* skip, float* y, float* work, size_t work_floats,
                           int B, int H, int T, void* stream) {
    ACMI_REQUIRE(B > 0 && H > 0 && T >= 0, "acmi_lstm_layer: bad shape");
    ACMI_REQUIRE((size_t)(LSTM_BB * H + 16 * LSTM_BB) * 4 <= 64 * 1024, "acmi_lstm_layer: H=%d too large", H);
    hipStream_t st = (hipStream_t)stream;
    {
        bool launched = false;
        if (int rc = lstm_try_xcd(gates_in, w_hh, skip, y, work, work_floats, B, H, T, st, &launched)) return rc;
        if (launched) return ACMI_OK;
    }
    float* h0 = work;
    float* h1 = work + (size_t)B * H;
    float* c = work + (size_t)2 * B * H;
    // (the err word at work[5 B H] is NOT cleared here: it accumulates over the layers of a stack and the caller, who
    // zeroed it, reads it once at the end -- a give-up in layer 0 must not be erased by layer 1's call)
    if (int rc = lstm_fill(reinterpret_cast<unsigned*>(work), (size_t)3 * B * H, 0u, nullptr, 0, 0u, st)) return rc;
    const size_t lds = (size_t)(LSTM_BB * H + 16 * LSTM_BB) * sizeof(float);
    dim3 grid((H + 3) / 4), block(256);
    if (T > 0 && lstm_persistent_ok(B, H)) {
        const int ki0 = (H + 15) / 16, kit = ki0 <= 8 ? 8 : ki0 <= 16 ? 16 : ki0 <= 32 ? 32 : 64;
        const size_t lds_p = (size_t)(12 * 16 * kit + 16 * LSTM_BB) * sizeof(float) + 16;   // hs padded to 16 KI rows, + the abort flag
        // More than LSTM_BB rows: the passes of one workgroup become copies of the grid, one pass each (the rows are
        // independent recurrences).  16 x 10 s at H = 512: encode 23.5 -> 17.7 ms, decode 21.4 -> 15.7 ms.  Splitting 8 rows into
        // 4 + 4 at H = 1024 LOSES 3 ms per 1500 steps (47.7 -> 50.6 ms decode): not done.  ACMI_LSTM_SPLIT caps the copies.
        static const int max_split = getenv("ACMI_LSTM_SPLIT") ? atoi(getenv("ACMI_LSTM_SPLIT")) : 4;
        const int nsplit = max(1, min((B + LSTM_BB - 1) / LSTM_BB, max_split));
        dim3 pgrid(grid.x * nsplit);
        // layout of `work` for this form: [c: B H floats][h buffers: 3 B H words, all EMPTY][...][err: 1 word]
        unsigned* hbuf = reinterpret_cast<unsigned*>(work + (size_t)B * H);
        unsigned* err = reinterpret_cast<unsigned*>(work + (size_t)5 * B * H);
        const int ki = (H + 15) / 16;
#define ACMI_LSTM_CASE(KIv)                                                                                                  \
        if (ki <= KIv) {                                                                                                     \
            if (lstm_grid_resident(lstm_persistent_kernel<KIv>, (int)grid.x, lds_p)) {                              \
                /* cell state zero, the three hidden-state buffers EMPTY, the tail of the 5 B H area zero */                 \
                if (int rc = lstm_fill(reinterpret_cast<unsigned*>(work), (size_t)5 * B * H, 0u, nullptr, 0, 0u, st)) return rc; \
                if (int rc = lstm_fill(hbuf, (size_t)3 * B * H, LSTM_EMPTY, nullptr, 0, 0u, st)) return rc;                  \
                hipLaunchKernelGGL(lstm_persistent_kernel<KIv>, pgrid, block, lds_p, st, gates_in, w_hh, work, skip, y, hbuf, \
                                   err, B, H, T, nsplit);                                                                    \
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


// work of the two-layer form: [c: 2 B H][h buffers: 2 x 3 B H words][exchange: T B H words][err: 1 word (+ 3 pad)]
extern "C" size_t acmi_lstm_stack2_work_floats(int B, int H, int T) {
    if (B <= 0 || H <= 0 || T < 0) return 0;
    return (size_t)8 * B * H + (size_t)T * B * H + 4;
}

static size_t lstm_wave2_lds(int H) {   // hs rows padded to the kernel's 16 KI (KI in {8, 16, 32, 64})
    const int ki = (H + 15) / 16, kit = ki <= 8 ? 8 : ki <= 16 ? 16 : ki <= 32 ? 32 : 64;
    return (size_t)(12 * 16 * kit + 32 * LSTM_BB) * sizeof(float) + 16 + (kit > 32 ? (size_t)LSTM_WAVE2_WL * 256 * sizeof(float) : 0);
}

template <int KI>
static bool lstm_wave2_resident(int nwg, size_t lds) { return lstm_grid_resident(lstm_wave2_kernel<KI>, nwg, lds); }

// can the two-layer launch run here at all (shape limits, per-layer residency)
static bool lstm_wave2_can(int B, int H, int T, size_t lds) {
    if (!lstm_persistent_ok(B, H) || T <= 0 || lds > 80 * 1024) return false;
    if ((size_t)T * B * H * 4 >= (1ull << 31)) return false;   // the exchange array is addressed through one buffer descriptor
    const int ki = (H + 15) / 16, nwg = (H + 3) / 4;
    return ki <= 8 ? lstm_wave2_resident<8>(nwg, lds) : ki <= 16 ? lstm_wave2_resident<16>(nwg, lds)
         : ki <= 32 ? lstm_wave2_resident<32>(nwg, lds) : lstm_wave2_resident<64>(nwg, lds);
}

// ... and should it (the advice acmi_lstm_stack2_supported gives the host)
extern "C" int acmi_lstm_stack2_supported(int B, int H, int T) {
    // ACMI_LSTM_WAVE: 0 never, 1 (default) where it was measured to win (H <= 512), 2 wherever it can run
    static const int want = [] { const char* e = getenv("ACMI_LSTM_WAVE"); return e ? (e[0] == '0' ? 0 : e[0] == '2' ? 2 : 1) : 1; }();
    if (!want || B <= 0 || H <= 0) return 0;
    // more than LSTM_BB rows: one launch per layer with the passes spread over copies of the grid wins (16 x 10 s, H = 512:
    // encode 17.7 vs 22.5 ms)
    if (want == 1 && B > LSTM_BB) return 0;
    // H = 1024, B = 8: 15.5 us per wavefront step against 2 x 6.4 + the input projection (23.2 vs 20.3 ms per 1500 steps): two
    // 32 KB gathers per layer-1 step, each in two rounds (register budget of two workgroups per CU), do not pay
    if (want == 1 && H > 512) return 0;
    // two launches of the XCD-local form (2 T steps of 1.4 us at H = 512) beat the wavefront's T + 1 steps: EnCodec-24k geometry,
    // 1 x 10 s: decode 4.58 -> 3.29 ms, encode 4.95 -> 3.63 ms (profiles/archive/r04_session37_lstm.log)
    if (want == 1 && lstm_xcd_would_run(B, H, T)) return 0;
    return lstm_wave2_can(B, H, T, lstm_wave2_lds(H)) ? 1 : 0;
}

extern "C" int acmi_lstm_stack2(const float* gates_in0, const float* w_hh0, const float* w_ih1, const float* w_hh1,
                                const float* bias1, const float* skip, float* y, float* work, int B, int H, int T, void* stream) {
    ACMI_REQUIRE(B > 0 && H > 0 && T > 0, "acmi_lstm_stack2: bad shape");
    const size_t lds = lstm_wave2_lds(H);
    ACMI_REQUIRE(lds <= 80 * 1024, "acmi_lstm_stack2: H=%d too large", H);
    ACMI_REQUIRE(lstm_wave2_can(B, H, T, lds), "acmi_lstm_stack2: not runnable here (ask acmi_lstm_stack2_supported first)");
    hipStream_t st = (hipStream_t)stream;
    const size_t BH = (size_t)B * H;
    unsigned* hbuf = reinterpret_cast<unsigned*>(work + 2 * BH);
    unsigned* x01 = hbuf + 6 * BH;
    unsigned* err = x01 + (size_t)T * BH;
    if (int rc = lstm_fill(reinterpret_cast<unsigned*>(work), 2 * BH, 0u, hbuf, 6 * BH + (size_t)T * BH, LSTM_EMPTY, st)) return rc;
    dim3 grid(2 * ((H + 3) / 4)), block(256);
    const int ki = (H + 15) / 16;
#define ACMI_LSTM2_CASE(KIv)                                                                                              \
    if (ki <= KIv) {                                                                                                      \
        if (lds > 64 * 1024)   /* more dynamic LDS than the default per-workgroup limit: opt in */                      \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_wave2_kernel<KIv>),                             \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                             \
        hipLaunchKernelGGL(lstm_wave2_kernel<KIv>, grid, block, lds, st, gates_in0, w_hh0, w_ih1, w_hh1, bias1, work, skip, y, \
                           hbuf, x01, err, B, H, T);                                                                      \
    } else
    ACMI_LSTM2_CASE(8) ACMI_LSTM2_CASE(16) ACMI_LSTM2_CASE(32) ACMI_LSTM2_CASE(64) {}
#undef ACMI_LSTM2_CASE
    return acmi_check_launch("lstm_wave2_kernel");
}
