// libacmi -- the score-folded cross-attention block of a decode step (include/acmi.h: acmi_cross_fold, acmi_lm_layer.w_qkvs;
// audiocraft_amd/modules/cross_fold.py holds the algebra and the table builder).
//
// The reference's layer (transformer.py:344-361, 563-566) runs, per position, norm_cross -> q projection -> q.K^T -> softmax
// -> p.V -> out projection -> residual add.  With K and V constant over a generate both head contractions are moved into
// per-generate tables, the raw scores S_raw[b] = (x1[b] - shift[b]) G[b]^T ride in the QKV and out-projection launches as
// extra output features, and ONE launch replaces the attention launch AND the output-projection GEMM:
//
//     s  = rstd (S_raw - (mean - shift) CS) + BS        the query's LayerNorm, folded (statistics of x1 from its producer)
//     p  = softmax over the Lc source positions of each head
//     x2 = x1 + p U[b] (+ b_cout)                       a [H Lc] x [H Lc, d] row-vector product per conditioned row
//
// One workgroup per (row, block of FB output features): 256 threads, thread = (vector of 16 bytes of features, slice of the
// H Lc table rows).  Request order: what the previous launch has just written first (statistics partials, raw scores, the
// residual), then this workgroup's slice of U (R H Lc d elements per layer in all: 49 KB per workgroup at d = 1536, 8 rows x
// 24 heads x 16 positions, bf16) -- everything before the first wait.  The dependent chain behind the data is short:
// statistics -> scores -> softmax (through LDS) -> 12 multiply-adds per feature -> a fixed-order sum over the table slices ->
// the x-producer's epilogue (f32 in place, raw fragments of the new x, optional statistics partials).
#include "acmi_lm_internal.h"


__device__ __forceinline__ float cf_val(bf16_t v) { return bf16_to_f32(v); }
__device__ __forceinline__ float cf_val(float v) { return v; }

#define ACMI_CF_HLMAX 1024   // table rows (H Lc) per conditioned row: LDS score / probability arrays, 4 slots per thread
#define ACMI_CF_NKPRE 16     // table rows per thread requested before the first wait (the rest after the softmax)

template <typename WT>
__global__ __launch_bounds__(256) void cross_fold_kernel(const CrossFoldArgs p) {
    constexpr int VEC = 16 / (int)sizeof(WT);     // features per thread: 8 (bf16) / 4 (f32)
    typedef WT rawv __attribute__((ext_vector_type(VEC)));
    __shared__ float s_sc[ACMI_CF_HLMAX];
    __shared__ float s_p[ACMI_CF_HLMAX];
    __shared__ float s_red[256 * VEC];            // [KG][FB], KG FB = 256 VEC
    const int nb = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int FB = p.FB, HL = p.HL, d = p.d;
    const int TPB = FB / VEC, KG = 256 / TPB;     // threads per table row of this block; table-row slices
    const int fv = tid % TPB, kg = tid / TPB;
    const bool live = b < p.R;                    // rows beyond R (null conditions) only take the bias
    // 1. what the previous launch wrote: statistics partials of row b, its raw scores, the residual
    float spm[2], spq[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float2 t = *reinterpret_cast<const float2*>(p.stats + ((size_t)b * p.np + min(lane + 64 * i, p.np - 1)) * 2);
        spm[i] = t.x; spq[i] = t.y;
    }
    float sr[4], csv[4], bsv[4];
    const size_t sbase = (size_t)b * p.s_ld + (size_t)(live ? b : 0) * HL, tbase = (size_t)(live ? b : 0) * HL;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int hj = min(tid + 256 * i, HL - 1);
        sr[i] = p.s_raw[sbase + hj];
        csv[i] = p.cs[tbase + hj];
        bsv[i] = p.bs[tbase + hj];
    }
    const int n = nb * FB + tid;                  // the feature a thread of the first wave(s) finishes (tid < FB)
    const float xres = tid < FB ? p.x[(size_t)b * d + n] : 0.f;
    const float qsh = p.shift != nullptr ? p.shift[b] : 0.f;
    const float osh = p.xt_shift != nullptr ? p.xt_shift[b] : 0.f;
    const float bia = (p.bias != nullptr && tid < FB) ? p.bias[n] : 0.f;
    // 2. this workgroup's slice of U: table rows kg, kg + KG, ...; 16 bytes of features per thread and row
    const int nk = kg < HL ? (HL - kg + KG - 1) / KG : 0;   // rows of this thread's slice (none when H Lc < the slice count)
    const WT* ub = reinterpret_cast<const WT*>(p.u) + (((size_t)(live ? b : 0) * (d / FB) + nb) * HL) * FB + fv * VEC;
    rawv ur[ACMI_CF_NKPRE];
#pragma unroll
    for (int i = 0; i < ACMI_CF_NKPRE; ++i) {
        const int hj = min(kg + KG * i, HL - 1);
        ur[i] = __builtin_nontemporal_load(reinterpret_cast<const rawv*>(ub + (size_t)hj * FB));
    }
    __builtin_amdgcn_sched_barrier(0);
    // 3. mean / rstd of x1's row b (Chan combination of the equal-count partials, as the attention kernel's query hook does)
    float rstd, meff;
    {
        const bool v0 = lane < p.np, v1 = lane + 64 < p.np;
        const float mean = wave_sum((v0 ? spm[0] : 0.f) + (v1 ? spm[1] : 0.f)) / (float)p.np;
        const float d0 = spm[0] - mean, d1 = spm[1] - mean;
        const float q2 = (v0 ? spq[0] + (float)p.cnt * d0 * d0 : 0.f) + (v1 ? spq[1] + (float)p.cnt * d1 * d1 : 0.f);
        rstd = 1.0f / sqrtf(wave_sum(q2) / (float)d + p.eps);
        meff = mean - qsh;
    }
    // 4. scores -> LDS -> softmax over the Lc positions of each head -> probabilities in LDS
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int hj = tid + 256 * i;
        if (hj < HL) s_sc[hj] = rstd * (sr[i] - meff * csv[i]) + bsv[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int hj = tid + 256 * i;
        if (hj < HL) {
            const int h0 = (hj / p.Lc) * p.Lc;
            float m = -INFINITY;
            for (int j = 0; j < p.Lc; ++j) m = fmaxf(m, s_sc[h0 + j]);
            float l = 0.f;
            for (int j = 0; j < p.Lc; ++j) l += expf(s_sc[h0 + j] - m);
            s_p[hj] = expf(s_sc[hj] - m) / l;
        }
    }
    __syncthreads();
    // 5. p U over this thread's table rows (f32 accumulation, rows in increasing order)
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    if (live) {
#pragma unroll
        for (int i = 0; i < ACMI_CF_NKPRE; ++i) {
            if (i < nk) {
                const float pr = s_p[kg + KG * i];
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = fmaf(pr, cf_val(ur[i][e]), acc[e]);
            }
        }
        for (int i = ACMI_CF_NKPRE; i < nk; ++i) {   // slices longer than the prefetch window (H Lc > 16 KG)
            const int hj = kg + KG * i;
            const rawv uv = *reinterpret_cast<const rawv*>(ub + (size_t)hj * FB);
            const float pr = s_p[hj];
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = fmaf(pr, cf_val(uv[e]), acc[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) s_red[kg * FB + fv * VEC + e] = acc[e];
    __syncthreads();
    // 6. the x-producer's epilogue, one feature per thread: slices summed in slice order (deterministic), bias, residual;
    //    f32 in place, raw fragment of the new x, statistics partial of its 16 features
    if (tid < FB) {
        float v = 0.f;
        for (int g = 0; g < KG; ++g) v += s_red[g * FB + tid];
        v += bia;
        v += xres;
        p.x[(size_t)b * d + n] = v;
        if (p.xt != nullptr) st_f32<WT>(reinterpret_cast<WT*>(p.xt) + tiled_index<WT>(b, n, p.xt_nkc), v - osh);
        if (p.stats_out != nullptr) {   // FB % 16 == 0 (launcher): a row of 16 lanes = 16 consecutive features
            const float mb = row16_sum(v) * 0.0625f;
            const float dq = row16_sum((v - mb) * (v - mb));
            if ((tid & 15) == 0) *reinterpret_cast<float2*>(p.stats_out + ((size_t)b * (d >> 4) + (n >> 4)) * 2) = make_float2(mb, dq);
        }
    }
}

int acmi_launch_cross_fold(const CrossFoldArgs& a_in, int wdtype, int rows, hipStream_t st) {
    CrossFoldArgs a = a_in;
    const int vec = wdtype == ACMI_BF16 ? 8 : 4;
    ACMI_REQUIRE(a.R > 0 && rows >= a.R && a.HL > 0 && a.HL <= ACMI_CF_HLMAX && a.Lc > 0 && a.HL % a.Lc == 0 && a.d > 0 && a.d % vec == 0,
                 "acmi_cross_fold: bad geometry R=%d rows=%d H Lc=%d Lc=%d d=%d", a.R, rows, a.HL, a.Lc, a.d);
    ACMI_REQUIRE(a.np >= 1 && a.np <= 128 && a.np * a.cnt == a.d, "acmi_cross_fold: %d statistics partials of %d elements for d=%d",
                 a.np, a.cnt, a.d);
    // features per workgroup: 64 where the width allows, else the largest divisor that keeps whole 16-byte vectors per thread
    // (and whole groups of 16 features when statistics partials are written)
    if (a.FB <= 0) {
        a.FB = 64;
        while (a.FB > vec && a.d % a.FB != 0) a.FB >>= 1;
    }
    ACMI_REQUIRE(a.FB >= vec && a.FB <= 64 && a.FB % vec == 0 && a.d % a.FB == 0 && (256 * vec) % a.FB == 0,
                 "acmi_cross_fold: %d features per workgroup (d=%d)", a.FB, a.d);
    ACMI_REQUIRE(a.stats_out == nullptr || (a.FB % 16 == 0 && a.d % 16 == 0), "acmi_cross_fold: statistics partials need 16-feature groups");
    // rows without a condition only pass through (bias, statistics partials of the unchanged row): launched when either exists
    const dim3 grid(a.d / a.FB, (a.bias != nullptr || a.stats_out != nullptr) ? rows : a.R);
    if (wdtype == ACMI_BF16) hipLaunchKernelGGL(cross_fold_kernel<bf16_t>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(cross_fold_kernel<float>, grid, dim3(256), 0, st, a);
    return acmi_check_launch("cross_fold_kernel");
}

extern "C" int acmi_cross_fold(const acmi_cross_fold_desc* c, void* stream) {
    ACMI_REQUIRE(c != nullptr && c->s_raw && c->stats && c->cs && c->bs && c->u && c->x, "acmi_cross_fold: null argument");
    ACMI_REQUIRE(c->wdtype == ACMI_BF16 || c->wdtype == ACMI_F32, "acmi_cross_fold: bad wdtype %d", c->wdtype);
    CrossFoldArgs a = {};
    a.s_raw = c->s_raw; a.s_ld = c->s_ld; a.stats = c->stats; a.np = c->stats_np; a.cnt = c->stats_cnt; a.shift = c->shift;
    a.cs = c->cs; a.bs = c->bs; a.u = c->u; a.x = c->x; a.bias = c->bias; a.xt = c->xt; a.xt_nkc = c->xt_nkc; a.xt_shift = c->xt_shift;
    a.stats_out = c->stats_out; a.R = c->R; a.HL = c->HL; a.Lc = c->Lc; a.d = c->d; a.FB = c->FB; a.eps = c->eps;
    ACMI_REQUIRE(a.s_ld >= a.R * a.HL || a.s_ld >= a.HL, "acmi_cross_fold: s_ld=%d", a.s_ld);
    return acmi_launch_cross_fold(a, c->wdtype, c->rows, (hipStream_t)stream);
}
