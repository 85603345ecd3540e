// Single-query attention over the KV cache (self and cross) and the KV scatter, gfx950 (CDNA4, wave64).
// Reference semantics: audiocraft/modules/transformer.py:266-298 (_complete_kv), :412-414 (SDPA for one new
// step), :559-565 (norm_cross, as the optional LayerNorm hook on the query).
#include "acmi_lm_internal.h"

#include <math.h>
#include <stdlib.h>

// =====================================================================================================
// single-query attention over a KV cache
// =====================================================================================================

// (dpp_f32<CTRL>: acmi_common.h -- cross-lane exchange through the DPP path of the VALU instead of ds_bpermute)
template <int LPP>   // sum over groups of LPP consecutive lanes (LPP = 1, 2, 4, 8, 16), result in every lane of the group
__device__ __forceinline__ float group_sum(float v) {
    if (LPP >= 2) v += dpp_f32<0xB1>(v);
    if (LPP >= 4) v += dpp_f32<0x4E>(v);
    if (LPP >= 8) v += dpp_f32<0x141>(v);
    if (LPP >= 16) v += dpp_f32<0x140>(v);
    return v;
}

__device__ __forceinline__ float raw_to_f32(bf16_t v) { return bf16_to_f32(v); }
__device__ __forceinline__ float raw_to_f32(float v) { return v; }

struct AttnArgs {
    const float* q; const void* kc; const void* vc; void* out;
    int out_tiled, out_bf16, out_rbs, out_col0;  // tiled output: K tiles per 16-row block, first column
    int H, Tcap, len; const int* len_dev; int len_bias; float scale;
    const int* len_rows;  // per-cache-row length (two_step_cfg: the two passes keep their own condition length), or NULL
    int past_context;  // > 0: positions [len - 1 - past_context, len) only
    int rpp;      // rows per position: query row b belongs to cache row b % rpp; with len_dev its length grows by b / rpp
    // optional LayerNorm hook on q (the cross-attention query arrives as x W'^T, see acmi_linear_pair):
    //   q <- rstd[b] (q - mean[b] colsum) + bias, mean / rstd of row b from the (mean, M2) partials of x
    const float* q_stats; int q_np, q_cnt, q_K; float q_eps; const float* q_colsum; const float* q_bias;
    const float* q_shift;  // shift of the raw row behind q (acmi_attn_desc.q_shift), or NULL
    int active_rows;       // > 0: cache rows >= active_rows are skipped (their output stays as the caller left it)
    int pm_n;              // > 0: position-minor query rows (row = cache row * pm_n + position), else row = position * rpp + cache row
    const int* start_rows; // per cache row: first key position attended (acmi_attn_desc.start_rows), or NULL
};

// Leading arguments = what the first K / V requests need; with kernarg preload (audiocraft_amd/build.py) they arrive in
// SGPRs at wave start.  The AttnArgs block behind them (byte ACMI_ATTN_ARGS_OFF of the kernarg segment) is read with scalar
// loads issued AFTER those requests, and so is the device-side length word: a decode position runs 96 of these launches,
// every one of which used to spend two dependent scalar round trips (kernarg block, then *len_dev -- a word the previous
// position's sampler wrote, cold in this CU's scalar cache) in front of its first HBM request.  The first chunk is now
// requested SPECULATIVELY, clamped to the cache capacity instead of the length (any address below Tcap is readable), and
// masked once the length is known.
//   g0 = H | Tcap << 16   g1 = rpp | pm_n << 16   g2 = len (host value / bound)   g3 = active_rows | speculate << 16 | waves << 24
#define ACMI_ATTN_ARGS_OFF 48
#define ACMI_AS4 __attribute__((address_space(4)))
__device__ __forceinline__ void attn_load_args(AttnArgs& p, int z) {   // z: an opaque zero produced behind the first requests
    const char ACMI_AS4* ka = (const char ACMI_AS4*)__builtin_amdgcn_kernarg_segment_ptr();
    __builtin_memcpy(&p, (const void ACMI_AS4*)__builtin_assume_aligned((const void ACMI_AS4*)(ka + (ACMI_ATTN_ARGS_OFF + z)), 8),
                     sizeof(AttnArgs));
}
template <typename KT, int HD, bool QN>  // QN: LayerNorm hook on q (separate instantiation: no branch around its loads)
__global__ __launch_bounds__(256) void attn_decode_kernel(const float* hq, const void* hkc, const void* hvc, const int* hlen_dev,
                                                          unsigned g0, unsigned g1, unsigned g2, unsigned g3, const AttnArgs) {
    const float* __restrict__ q = hq;
    const KT* __restrict__ kc = reinterpret_cast<const KT*>(hkc);
    const KT* __restrict__ vc = reinterpret_cast<const KT*>(hvc);
    const int H = (int)(g0 & 0xffffu), Tcap = (int)(g0 >> 16);
    const int h_rpp = (int)(g1 & 0xffffu), h_pm_n = (int)(g1 >> 16), h_len = (int)g2, h_active = (int)(g3 & 0xffffu);
    const bool spec = ((g3 >> 16) & 0xffu) != 0;   // first chunk before the length is known (no window, no per-row lengths)
    constexpr int DPL = HD >= 8 ? 8 : HD;  // dims per lane
    constexpr int LPP = HD / DPL;          // lanes per position
    constexpr int PPI = 64 / LPP;          // positions covered by one load instruction of a wave
#ifndef ACMI_ATTN_NI
#define ACMI_ATTN_NI 8   // experiment switch (round 5): 16 with 2 waves per (row, head) = the same bytes in flight from half the waves
#endif
    constexpr int NI = sizeof(KT) == 2 ? ACMI_ATTN_NI : 4;  // positions per lane per chunk (K and V loads in flight: 2 * NI)
    typedef KT rawv __attribute__((ext_vector_type(DPL)));
    constexpr int CH = NI * PPI;
    const int h = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane % LPP, pp = lane / LPP;
    // several positions per call (prefill): cache row b0 and position index pidx of query row b (one division either way)
    const int dv = h_pm_n > 0 ? h_pm_n : h_rpp, qd = b / dv, rm = b - qd * dv;
    const int b0 = h_pm_n > 0 ? qd : rm, pidx = h_pm_n > 0 ? rm : qd;
    if (h_active > 0 && b0 >= h_active) return;   // null condition: K = V = 0, the output is exactly 0 (workgroup uniform)
    const KT* kb = kc + ((size_t)b0 * H + h) * Tcap * HD + c * DPL;
    const KT* vb = vc + ((size_t)b0 * H + h) * Tcap * HD + c * DPL;
    const int nwv = __builtin_amdgcn_readfirstlane((int)(g3 >> 24));  // 1, 2 or 4 waves share the positions of this (row, head)
    rawv kr[NI], vr[NI];
    int lim = spec ? (hlen_dev != nullptr ? Tcap : h_len) : 0;   // addresses are clamped to lim - 1: the length once it is known
    auto load_kv = [&](int t0) {
        // branch free: lanes past the end re-read the last position (their scores are masked below); a per-lane
        // zero fill would write the registers of loads still in flight and make every load wait for the previous
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int t = min(t0 + i * PPI + pp, lim - 1);
            kr[i] = __builtin_nontemporal_load(reinterpret_cast<const rawv*>(kb + (size_t)t * HD));
            vr[i] = __builtin_nontemporal_load(reinterpret_cast<const rawv*>(vb + (size_t)t * HD));
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the 2 * NI requests together (the scheduler sinks them to their uses)
    };
    // the device-side length word goes out FIRST (vmcnt retires in order: it is back before the chunk behind it)
    int len_word = 0;
    if (hlen_dev != nullptr) len_word = __builtin_nontemporal_load(hlen_dev);
    if (spec) load_kv(wave * CH);
    // everything else comes from the AttnArgs block, read here (behind the first chunk's requests)
    int opaque0 = 0;
    asm volatile("" : "+s"(opaque0));
    opaque0 = __builtin_amdgcn_readfirstlane(opaque0);
    AttnArgs p;
    attn_load_args(p, opaque0);
    const float scale = p.scale;
    float qv[DPL];
#pragma unroll
    for (int e = 0; e < DPL; ++e) qv[e] = q[((size_t)b * H + h) * HD + c * DPL + e];
    // LayerNorm hook: everything it needs is requested here, consumed after the first K / V chunk is in flight
    float qcs[DPL], qb[DPL], spm[2], spq[2], qsh = 0.f;
    if (QN) {
        qsh = *(p.q_shift != nullptr ? p.q_shift + b : p.q_stats);   // unconditional request; dropped below without a shift
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
    float m = -INFINITY, l = 0.f, o[DPL];
#pragma unroll
    for (int e = 0; e < DPL; ++e) o[e] = 0.f;
    __builtin_amdgcn_sched_barrier(0);   // every request above is out before the length is waited for
    const int len = p.len_rows ? max(1, min(p.len_rows[b0], p.len))
                               : (hlen_dev != nullptr ? (__builtin_amdgcn_readfirstlane(len_word) + p.len_bias + pidx) : p.len);

    // K and V of a whole chunk are requested together (2 * NI wide loads in flight per lane); the first chunk
    // goes out before anything waits on q (its LayerNorm hook needs the fresh statistics of x).
    // (Round 3 tried a second register set -- the next chunk in flight while one is multiplied, with exact vmcnt counts in
    // the steady state: 27.4 us at t = 1500 either way, +0.4 ... 0.8 us at every length from the longer prologue and the
    // 214 VGPRs.  The slope of this kernel, 0.0159 us per position = 6.2 TB/s, IS the copy bandwidth of the chip: what is
    // left is the fixed 3.6 us, 1.55 of them the kernel boundary.  profiles/archive/r03_attn_microbench.log)
    int start = p.past_context > 0 ? max(0, len - 1 - p.past_context) : 0;   // bounded receptive field
    if (p.start_rows != nullptr) start = max(start, __builtin_amdgcn_readfirstlane(p.start_rows[b0]));   // left-padded stream
    lim = len;
    if (!spec) load_kv(start + wave * CH);
    if (QN) {  // Chan combination of the partials -> mean, rstd of row b; then the affine map of q
        const bool v0 = lane < p.q_np, v1 = lane + 64 < p.q_np;
        const float mean = wave_sum((v0 ? spm[0] : 0.f) + (v1 ? spm[1] : 0.f)) / (float)p.q_np;
        const float d0 = spm[0] - mean, d1 = spm[1] - mean;
        const float q2 = (v0 ? spq[0] + (float)p.q_cnt * d0 * d0 : 0.f) + (v1 ? spq[1] + (float)p.q_cnt * d1 * d1 : 0.f);
        const float rstd = 1.0f / sqrtf(wave_sum(q2) / (float)p.q_K + p.q_eps);
        const float meff = mean - (p.q_shift != nullptr ? qsh : 0.f);   // q was built on the row minus its shift
#pragma unroll
        for (int e = 0; e < DPL; ++e) qv[e] = rstd * (qv[e] - meff * qcs[e]) + qb[e];
    }
    for (int t0 = start + wave * CH; t0 < len; t0 += nwv * CH) {   // kr / vr hold the chunk at t0
        if (t0 + CH > len) {   // (wave-uniform) the chunk runs past the end: whatever the clamped / speculative requests fetched
#pragma unroll                  // there gets weight 0 below -- make it a finite 0 as well (0 x NaN would poison the sum)
            for (int i = 0; i < NI; ++i)
                if (t0 + i * PPI + pp >= len) {
#pragma unroll
                    for (int e = 0; e < DPL; ++e) vr[i][e] = (KT)0;
                }
        }
        float s[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int t = t0 + i * PPI + pp;
            float part = 0.f;
#pragma unroll
            for (int e = 0; e < DPL; ++e) part = fmaf(qv[e], raw_to_f32(kr[i][e]), part);
            part = group_sum<LPP>(part);
            s[i] = (t < len) ? part * scale : -INFINITY;
        }
        float cmax = s[0];
#pragma unroll
        for (int i = 1; i < NI; ++i) cmax = fmaxf(cmax, s[i]);
#pragma unroll
        for (int off = LPP; off < 64; off <<= 1)   // across the position groups of the wave (xor 8 = row_ror:8, DPP)
            cmax = fmaxf(cmax, (LPP == 8 && off == 8) ? dpp_f32<0x128>(cmax) : __shfl_xor(cmax, off, 64));
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
        const bool dpp8 = LPP == 8 && off == 8;
        l += dpp8 ? dpp_f32<0x128>(l) : __shfl_xor(l, off, 64);
#pragma unroll
        for (int e = 0; e < DPL; ++e) o[e] += dpp8 ? dpp_f32<0x128>(o[e]) : __shfl_xor(o[e], off, 64);   // same dims, other position group
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
        const float r = den > 0.f ? num / den : 0.f;   // (no key at all: a left-padding position of its stream, acmi_lm_state.row_off)
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

// -----------------------------------------------------------------------------------------------------
// cross_q_kernel: the decode step's cross-attention (transformer.py:344-361) as its own kernel -- one wave per (row, head),
// a host-known source length of at most NI * 8 positions (one chunk), head size 64, the LayerNorm hook on the query.
// Same arithmetic, in the same order, as attn_decode_kernel<KT, 64, true> run with one wave (bit-identical results); what
// differs is WHEN things are requested.  This launch is pure latency (16 keys x 8 active rows at the headline configuration:
// 5.2 us per layer, 10 % of a decode position, for ~0.4 MB): its critical path is the round trip of what the PREVIOUS launch
// has just written -- the statistics partials of x1 and the raw query r -- so those requests go out first, from preloaded
// arguments (q, caches, q_stats, out + four packed words = 14 dwords), before any scalar load; the constant operands
// (column sums, bias, shift: L2-warm) follow once the argument block is in; only NI (2 for <= 16 keys) K / V positions per lane
// are requested instead of the generic kernel's 8 + 8 (clamped re-reads of the last position otherwise); one wave needs no LDS
// combine, no barrier: the eight lanes of position group 0 hold the 64 outputs and store them as 16-byte fragments.
//   g0 = H | Tcap << 16   g1 = len | q_np << 16   g2 = q_cnt | out_rbs << 16   g3 = flags: bit 0 tiled, bit 1 bf16 output
struct CrossQArgs {
    const float* q_colsum; const float* q_bias; const float* q_shift; const int* len_rows;
    float q_eps, scale; int out_col0, q_K;
};
#define ACMI_CROSSQ_ARGS_OFF 56
__device__ __forceinline__ void crossq_load_args(CrossQArgs& p, int z) {
    const char ACMI_AS4* ka = (const char ACMI_AS4*)__builtin_amdgcn_kernarg_segment_ptr();
    __builtin_memcpy(&p, (const void ACMI_AS4*)__builtin_assume_aligned((const void ACMI_AS4*)(ka + (ACMI_CROSSQ_ARGS_OFF + z)), 8),
                     sizeof(CrossQArgs));
}
template <typename KT, int NI>
__global__ __launch_bounds__(64) void cross_q_kernel(const float* hq, const void* hkc, const void* hvc, const float* hstats, void* hout,
                                                     unsigned g0, unsigned g1, unsigned g2, unsigned g3, const CrossQArgs) {
    constexpr int HD = 64, DPL = 8, LPP = 8, PPI = 8;
    typedef KT rawv __attribute__((ext_vector_type(DPL)));
    const int H = (int)(g0 & 0xffffu), Tcap = (int)(g0 >> 16), hlen = (int)(g1 & 0xffffu), q_np = (int)(g1 >> 16);
    const int q_cnt = (int)(g2 & 0xffffu), out_rbs = (int)(g2 >> 16);
    const int h = blockIdx.x, b = blockIdx.y, lane = threadIdx.x, c = lane % LPP, pp = lane / LPP;
    // 1. what the previous launch wrote: statistics partials of row b, the raw query
    float spm[2], spq[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float2 t = *reinterpret_cast<const float2*>(hstats + ((size_t)b * q_np + min(lane + 64 * i, q_np - 1)) * 2);
        spm[i] = t.x; spq[i] = t.y;
    }
    float qv[DPL];
#pragma unroll
    for (int e = 0; e < DPL; ++e) qv[e] = hq[((size_t)b * H + h) * HD + c * DPL + e];
    // 2. keys / values of the (constant) cross-attention cache: NI positions per lane, clamped to the host-side length bound
    const KT* kb = reinterpret_cast<const KT*>(hkc) + ((size_t)b * H + h) * Tcap * HD + c * DPL;
    const KT* vb = reinterpret_cast<const KT*>(hvc) + ((size_t)b * H + h) * Tcap * HD + c * DPL;
    rawv kr[NI], vr[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int t = min(i * PPI + pp, hlen - 1);
        kr[i] = __builtin_nontemporal_load(reinterpret_cast<const rawv*>(kb + (size_t)t * HD));
        vr[i] = __builtin_nontemporal_load(reinterpret_cast<const rawv*>(vb + (size_t)t * HD));
    }
    __builtin_amdgcn_sched_barrier(0);
    // 3. the argument block (scalar loads behind the requests above), then the constant operands
    int opaque0 = 0;
    asm volatile("" : "+s"(opaque0));
    opaque0 = __builtin_amdgcn_readfirstlane(opaque0);
    CrossQArgs p;
    crossq_load_args(p, opaque0);
    float qcs[DPL], qb[DPL];
    const float qsh = *(p.q_shift != nullptr ? p.q_shift + b : hstats);
#pragma unroll
    for (int e = 0; e < DPL; ++e) {
        qcs[e] = p.q_colsum[h * HD + c * DPL + e];
        qb[e] = p.q_bias[h * HD + c * DPL + e];
    }
    const int len = p.len_rows ? max(1, min(p.len_rows[b], hlen)) : hlen;
    __builtin_amdgcn_sched_barrier(0);
    {   // Chan combination of the partials -> mean, rstd of row b; then the affine map of q (as attn_decode_kernel<.., true>)
        const bool v0 = lane < q_np, v1 = lane + 64 < q_np;
        const float mean = wave_sum((v0 ? spm[0] : 0.f) + (v1 ? spm[1] : 0.f)) / (float)q_np;
        const float d0 = spm[0] - mean, d1 = spm[1] - mean;
        const float q2 = (v0 ? spq[0] + (float)q_cnt * d0 * d0 : 0.f) + (v1 ? spq[1] + (float)q_cnt * d1 * d1 : 0.f);
        const float rstd = 1.0f / sqrtf(wave_sum(q2) / (float)p.q_K + p.q_eps);
        const float meff = mean - (p.q_shift != nullptr ? qsh : 0.f);
#pragma unroll
        for (int e = 0; e < DPL; ++e) qv[e] = rstd * (qv[e] - meff * qcs[e]) + qb[e];
    }
    float s[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int t = i * PPI + pp;
        if (t >= len) {
#pragma unroll
            for (int e = 0; e < DPL; ++e) vr[i][e] = (KT)0;
        }
        float part = 0.f;
#pragma unroll
        for (int e = 0; e < DPL; ++e) part = fmaf(qv[e], raw_to_f32(kr[i][e]), part);
        part = group_sum<LPP>(part);
        s[i] = (t < len) ? part * p.scale : -INFINITY;
    }
    float cmax = s[0];
#pragma unroll
    for (int i = 1; i < NI; ++i) cmax = fmaxf(cmax, s[i]);
#pragma unroll
    for (int off = LPP; off < 64; off <<= 1) cmax = fmaxf(cmax, off == 8 ? dpp_f32<0x128>(cmax) : __shfl_xor(cmax, off, 64));
    float l = 0.f, o[DPL];
#pragma unroll
    for (int e = 0; e < DPL; ++e) o[e] = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int t = i * PPI + pp;
        const float pr = (t < len) ? expf(s[i] - cmax) : 0.f;
        l += pr;
#pragma unroll
        for (int e = 0; e < DPL; ++e) o[e] = fmaf(pr, raw_to_f32(vr[i][e]), o[e]);
    }
#pragma unroll
    for (int off = LPP; off < 64; off <<= 1) {
        const bool dpp8 = off == 8;
        l += dpp8 ? dpp_f32<0x128>(l) : __shfl_xor(l, off, 64);
#pragma unroll
        for (int e = 0; e < DPL; ++e) o[e] += dpp8 ? dpp_f32<0x128>(o[e]) : __shfl_xor(o[e], off, 64);
    }
    if (lane < LPP) {   // position group 0: lane c holds output dims [8 c, 8 c + 8) of head h
        float r[DPL];
#pragma unroll
        for (int e = 0; e < DPL; ++e) r[e] = l > 0.f ? (1.0f * o[e]) / (1.0f * l) : 0.f;
        const int f = h * HD + c * DPL;
        if (!(g3 & 1u)) {
            float* dst = reinterpret_cast<float*>(hout) + (size_t)b * H * HD + f;
#pragma unroll
            for (int e = 0; e < DPL; ++e) dst[e] = r[e];
        } else if (g3 & 2u) {   // 8 consecutive features of one row = one lane's 16 bytes of a bf16 A-fragment
            u32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = pack_bf16x2(r[2 * e], r[2 * e + 1]);
            *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(hout) + tiled_index<bf16_t>(b, p.out_col0 + f, out_rbs)) = v;
        } else {
#pragma unroll
            for (int e = 0; e < DPL; ++e) reinterpret_cast<float*>(hout)[tiled_index<float>(b, p.out_col0 + f + e, out_rbs)] = r[e];
        }
    }
}

template <typename KT, int NI>
static int launch_cross_q(const AttnArgs& a, int rows, hipStream_t st) {
    CrossQArgs c = {};
    c.q_colsum = a.q_colsum; c.q_bias = a.q_bias; c.q_shift = a.q_shift; c.len_rows = a.len_rows;
    c.q_eps = a.q_eps; c.scale = a.scale; c.out_col0 = a.out_col0; c.q_K = a.q_K;
    const unsigned g0 = (unsigned)a.H | ((unsigned)a.Tcap << 16), g1 = (unsigned)a.len | ((unsigned)a.q_np << 16);
    const unsigned g2 = (unsigned)a.q_cnt | ((unsigned)a.out_rbs << 16), g3 = (a.out_tiled ? 1u : 0u) | (a.out_bf16 ? 2u : 0u);
    hipLaunchKernelGGL((cross_q_kernel<KT, NI>), dim3(a.H, rows), dim3(64), 0, st, a.q, a.kc, a.vc, a.q_stats, a.out, g0, g1, g2, g3, c);
    return acmi_check_launch("cross_q_kernel");
}

template <typename KT>
static int launch_attn_t(const AttnArgs& a, int Beff, int hd, hipStream_t st) {
    // (function-local statics with an initialiser are set once, thread safe: no check-then-write on a plain static)
    static const int attn_nw = [] { const char* e = getenv("ACMI_ATTN_NW"); const int v = e ? atoi(e) : 4; return (v == 1 || v == 2) ? v : 4; }();
    // waves per (row, head): 4 by default; a host-known short length (cross-attention) needs no more waves than
    // it has 64-position chunks (bf16 cache, hd 64) -- idle waves still cost dispatch time
    int nwv = attn_nw;
    if (a.len_dev == nullptr) {
        const int dpl = hd >= 8 ? 8 : hd, chunk = (sizeof(KT) == 2 ? ACMI_ATTN_NI : 4) * (64 / (hd / dpl));
        const int need = (a.len + chunk - 1) / chunk;
        while (nwv > 1 && nwv / 2 >= need) nwv /= 2;
    }
    // rows past active_rows do nothing: with one position per call (query row == cache row) they are not even launched
    const int rows = (a.active_rows > 0 && Beff == a.rpp) ? a.active_rows : Beff;
    // the decode step's cross-attention: a kernel of its own (see cross_q_kernel); ACMI_CROSSQ=0 keeps the generic one (A/B)
    static const bool crossq_ok = !(getenv("ACMI_CROSSQ") != nullptr && getenv("ACMI_CROSSQ")[0] == '0');
    constexpr int PPC = 8;   // positions per load instruction at head size 64
    if (crossq_ok && a.q_colsum != nullptr && a.len_dev == nullptr && hd == 64 && a.past_context <= 0 && a.start_rows == nullptr &&
        a.pm_n == 0 && Beff == a.rpp && a.len <= PPC * (sizeof(KT) == 2 ? 8 : 4) && a.Tcap <= 0xffff && a.q_np <= 0xffff &&
        a.q_cnt <= 0xffff && a.out_rbs <= 0xffff && a.H <= 0xffff) {
        if (a.len <= 2 * PPC) return launch_cross_q<KT, 2>(a, rows, st);
        if (a.len <= 4 * PPC) return launch_cross_q<KT, 4>(a, rows, st);
        if (sizeof(KT) == 2) return launch_cross_q<KT, 8>(a, rows, st);
    }
    dim3 grid(a.H, rows), block(64 * nwv);
    ACMI_REQUIRE(a.H <= 0xffff && a.Tcap <= 0xffff && a.rpp <= 0xffff && a.pm_n <= 0xffff && a.active_rows <= 0xffff,
                 "acmi_attn_decode: geometry beyond the packed launch words (H %d, Tcap %d, rows %d)", a.H, a.Tcap, a.rpp);
    // the first chunk may be requested before the length is known unless the window start or a per-row length decides it
    const unsigned spec = (a.past_context <= 0 && a.len_rows == nullptr && a.start_rows == nullptr) ? 1u : 0u;
    const unsigned g0 = (unsigned)a.H | ((unsigned)a.Tcap << 16), g1 = (unsigned)a.rpp | ((unsigned)a.pm_n << 16);
    const unsigned g2 = (unsigned)a.len, g3 = (unsigned)a.active_rows | (spec << 16) | ((unsigned)nwv << 24);
#define ACMI_ATTN_CASE(HD)                                                                              \
    case HD:                                                                                            \
        if (a.q_colsum != nullptr)                                                                      \
            hipLaunchKernelGGL((attn_decode_kernel<KT, HD, true>), grid, block, 0, st, a.q, a.kc, a.vc, a.len_dev, g0, g1, g2, g3, a);  \
        else hipLaunchKernelGGL((attn_decode_kernel<KT, HD, false>), grid, block, 0, st, a.q, a.kc, a.vc, a.len_dev, g0, g1, g2, g3, a); \
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
    a.H = c.H; a.Tcap = c.Tcap; a.len = c.len; a.len_dev = c.len_dev; a.len_bias = c.len_bias; a.len_rows = c.len_rows;
    ACMI_REQUIRE(c.len_rows == nullptr || (c.len_dev == nullptr && c.len > 0), "acmi_attn_decode: len_rows needs a host `len` bound");
    a.past_context = c.past_context;
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
        a.q_shift = c.q_shift;
    }
    ACMI_REQUIRE(c.active_rows >= 0 && c.active_rows <= a.rpp, "acmi_attn_decode: active_rows=%d outside [0, %d]", c.active_rows, a.rpp);
    a.active_rows = c.active_rows;
    ACMI_REQUIRE(c.pos_minor_rows >= 0 && (c.pos_minor_rows == 0 || c.Beff == a.rpp * c.pos_minor_rows),
                 "acmi_attn_decode: pos_minor_rows=%d does not match %d query rows over %d cache rows", c.pos_minor_rows, c.Beff, a.rpp);
    a.pm_n = c.pos_minor_rows;
    a.start_rows = c.start_rows;
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
