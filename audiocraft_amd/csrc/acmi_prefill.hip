// Prefill of the MusicGen LM for gfx950 (CDNA4, wave64): the prompt / prepended-condition positions of a generate run
// through ONE forward, like the reference does (audiocraft/models/lm.py:540-543, modules/transformer.py:233-264, 362-414,
// models/genmodel.py:233-262), instead of position by position through the decode kernels.
//
//   lin_big_kernel        MFMA-tiled GEMM  out[M, N] = a[M, K] W[N, K]^T  on the SAME tiled operands the decode step uses
//                         (1 KB A / B fragments, include/acmi.h): 128 x 128 workgroup tiles, 4 waves of 64 x 64, K in steps
//                         of two fragments, fragments staged through LDS in fragment order (every ds_read_b128 is lane
//                         linear: conflict free), epilogues: f32 / residual add / tiled (+ GELU) / QKV scatter into the KV
//                         cache (+ V time-minor for the prefill attention).
//   attn_prefill_kernel   causal attention of all prompt positions of a (cache row, head): flash-style over the K cache,
//                         S^T = K Q^T so that a query's scores sit in one lane column, P feeds the second MFMA straight
//                         from registers (its k slots are matched by the load pattern of the time-minor V), online softmax.
//
// Row layout of every prefill activation: POSITION-MINOR and padded, row = cache_row * npos_pad + position (npos_pad =
// positions rounded up to 16), so that a 16-row MFMA block is 16 consecutive positions of ONE cache row: the QKV
// epilogue then stores 4 consecutive positions of V^T with one 8-byte store and never straddles cache rows.
#include "acmi_lm_internal.h"

#include <math.h>
#include <stdlib.h>

__device__ __forceinline__ void big_mma(const u32x4& a, const u32x4& b, f32x4& acc, bf16_t) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
__device__ __forceinline__ void big_mma(const u32x4& a, const u32x4& b, f32x4& acc, float) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[e]), __uint_as_float(b[e]), acc, 0, 0, 0);
}
__device__ __forceinline__ float big_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// =====================================================================================================
// MFMA-tiled GEMM on tiled operands
// =====================================================================================================
// DMA (round 4): the fragments of a stage go global -> LDS directly (`global_load_lds_dwordx4`: one op per 1 KB fragment, the
// wave's lanes land at consecutive 16-byte slots -- exactly the fragment order), one stage ahead of the MFMAs, instead of
// global -> VGPR -> ds_write_b128 two stages ahead.  Cycle budget of a K step per CU (two workgroups): 64 KB of ds_write_b128
// at ~79 B/clk = ~830 cycles + 128 KB of ds_read_b128 at 256 B/clk = 512 cycles on ONE LDS pipe against 1024 cycles of MFMA
// issue: the register staging made the LDS pipe, not the matrix pipe, the longer one.
//   DMA 1: stages of two K fragments (32 KB), two LDS buffers, the next stage requested one K step ahead;
//   DMA 2: stages of ONE K fragment (16 KB), a ring of four, requests three stages ahead (counted vmcnt: two stages stay in
//          flight across every barrier) -- the same 64 KB of LDS, three times the latency tolerance.
template <typename WT, int EPI, int DMA>
__global__ __launch_bounds__(256, 2) void lin_big_kernel(const BigArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 stages x 32 fragments x 1 KB
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + 127) >> 7, tiles_m = (p.M + 127) >> 7, nb = tiles_m * tiles_n;
    // Workgroup b runs on XCD b % 8 and every XCD has its own 4 MB L2.  Round 4: XCD x owns tiles_n / 8 ADJACENT COLUMN tiles
    // and sweeps the row tiles: its weight panels (6 x 393 KB at N = 6144, K = 1536) stay in its L2 and every activation
    // panel is read once per XCD.  A remainder of 1, 2 or 4 column tiles (N = 4608: 36 = 8 x 4 + 4, N = 1536: 12 = 8 + 4) is
    // shared by 8 / remainder XCDs each, which take its row tiles in turn (big_grid below sizes the launch: workgroups past
    // an XCD's list leave at once).  (Round 3 gave each XCD a contiguous run of ROW-major tiles: the 64 tiles it works on at
    // a time then touch all N / 128 weight panels -- 18.9 MB, far beyond its L2 -- once per row tile: 1.4 GB from the memory
    // side per FFN1 GEMM of a 600-position prefill, 6.7 TB/s for the 213-280 us it took.)
    int bid = blockIdx.x, tm, tn;
    const int P = tiles_n >> 3, R = tiles_n & 7;
    if (R == 0) {
        const int x = bid & 7, l = bid >> 3;
        tm = l / P;
        tn = x * P + (l - tm * P);
    } else if (R == 1 || R == 2 || R == 4) {
        const int share = 8 / R, G = share * P + 1;   // a group = `share` row tiles: their primary tiles + ONE tile of the shared column
        const int x = bid & 7, l = bid >> 3, g = l / G, r = l - g * G;
        if (r < share * P) { tm = g * share + r / P; tn = x * P + r % P; }
        else { tm = g * share + x % share; tn = 8 * P + x / share; }
        if (tm >= tiles_m) return;
    } else {
        if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);   // one contiguous run of row-major tiles per L2
        tm = bid / tiles_n;
        tn = bid - tm * tiles_n;
    }
    const int MT16 = p.M >> 4, NT16 = (p.N + 15) >> 4;
    const u32x4* __restrict__ A = reinterpret_cast<const u32x4*>(p.a);
    const u32x4* __restrict__ W = reinterpret_cast<const u32x4*>(p.w);

    // the 8 fragments of a stage this wave moves: waves 0, 1 the activation's (row block wave * 4 + i / 2, K fragment i & 1),
    // waves 2, 3 the weight's; LDS fragment index = wave * 8 + i in both cases
    size_t base[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        if (wave < 2) base[h] = (size_t)min(tm * 8 + wave * 4 + h, MT16 - 1) * p.a_rbs * 64 + lane;
        else base[h] = (size_t)min(tn * 8 + (wave - 2) * 4 + h, NT16 - 1) * p.NKC * 64 + lane;
    }
    const u32x4* __restrict__ src = wave < 2 ? A : W;
    u32x4* lds = reinterpret_cast<u32x4*>(smem);
    const int nks = p.NKC >> 1;
    // Two register sets: the fragments of stage ks + 2 are requested while stage ks is multiplied and stage ks + 1 (requested
    // one step earlier) is written to the other LDS buffer -- a request has two steps (~2 x 500 MFMA cycles) to come back
    // instead of one.  The steady-state loop is entered only with its prefetch in flight and its body is straight line, so
    // that the compiler's vmcnt counts stay exact (cf. attn_decode_kernel); short K and the last stages are peeled.
    u32x4 sa[8], sb[8];
    auto fetch = [&](int ks, u32x4 (&st)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) st[i] = src[base[i >> 1] + (size_t)(2 * ks + (i & 1)) * 64];
        __builtin_amdgcn_sched_barrier(0);   // the requests stay HERE (the scheduler otherwise sinks them below the MFMAs, next to their use)
    };
    auto stash = [&](int stage, const u32x4 (&st)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) lds[((stage & 1) * 32 + wave * 8 + i) * 64 + lane] = st[i];
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int ks) {
        const u32x4* sb_ = lds + (ks & 1) * 32 * 64 + lane;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            u32x4 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = sb_[((wm * 4 + i) * 2 + kc) * 64];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = sb_[(16 + (wn * 4 + j) * 2 + kc) * 64];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) big_mma(a[i], b[j], acc[i][j], WT());
        }
        __builtin_amdgcn_sched_barrier(0);   // the LDS writes of the next stage (and their vmcnt waits) stay behind the MFMAs
    };

    if constexpr (DMA == 2) {
        // The DMA ops are inline asm (hipcc would otherwise drain vmcnt to 0 in front of every barrier and every ds_read that
        // may alias the DMA's destination -- no request would survive a K step); their completion is counted here by hand:
        // each wave issues 4 per stage, `vmcnt(8)` = everything but the two youngest stages has landed.  The kernel's only
        // other VMEM operations are the epilogue's, behind a vmcnt(0).
        const int nst = p.NKC;
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
        auto dma = [&](int st) {   // this wave's 4 fragments of stage st (K fragment st) into ring slot st & 3
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x4* g = src + base[i] + (size_t)st * 64;
                const unsigned dst = lds0 + (unsigned)(((st & 3) * 16 + wave * 4 + i) * 1024);   // wave-uniform: M0
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
            }
        };
        auto bar = [&]() {         // workgroup barrier WITHOUT the fence of __syncthreads (that fence is a vmcnt(0))
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        auto comp1 = [&](int st) {
            const u32x4* sb_ = lds + (st & 3) * 16 * 64 + lane;
            u32x4 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = sb_[(wm * 4 + i) * 64];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = sb_[(8 + wn * 4 + j) * 64];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) big_mma(a[i], b[j], acc[i][j], WT());
            __builtin_amdgcn_sched_barrier(0);
        };
        dma(0);
        if (nst > 1) dma(1);
        if (nst > 2) dma(2);
        if (nst > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bar();
        int st = 0;
        for (; st + 3 < nst; ++st) {          // stage st is in LDS; st + 1, st + 2 in flight
            dma(st + 3);                      // slot (st - 1) & 3: every wave finished reading it before the last barrier
            comp1(st);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // stage st + 1 (this wave's part) has landed; two stages stay in flight
            bar();
        }
        for (; st < nst; ++st) {              // the last (up to three) stages: nothing left to request
            comp1(st);
            if (st + 2 < nst) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bar();
        }
    } else if constexpr (DMA == 1) {
        auto dma = [&](int ks) {   // this wave's 8 fragments of stage ks, straight into their slots of buffer ks & 1
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(src + base[i >> 1] + (size_t)(2 * ks + (i & 1)) * 64),
                    (__attribute__((address_space(3))) void*)(lds + ((ks & 1) * 32 + wave * 8 + i) * 64), 16, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        dma(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int ks = 0; ks < nks; ++ks) {
            if (ks + 1 < nks) dma(ks + 1);   // the other buffer: every wave finished reading it before the last barrier
            compute(ks);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's writes into LDS have landed ...
            __syncthreads();                                    // ... and so have everybody's
        }
    } else {
    fetch(0, sa);
    stash(0, sa);
    if (nks >= 4) {
        fetch(1, sa);
        __syncthreads();
        int ks = 0;
        do {                                  // sa = stage ks + 1 in flight
            fetch(ks + 2, sb);
            compute(ks);
            stash(ks + 1, sa);                // the other buffer: every wave left it at the previous barrier
            __syncthreads();
            fetch(ks + 3, sa);
            compute(ks + 1);
            stash(ks + 2, sb);
            __syncthreads();
            ks += 2;
        } while (ks + 3 < nks);
        const bool three = nks - ks == 3;     // 2 or 3 stages left: ks in LDS, ks + 1 in sa
        if (three) fetch(ks + 2, sb);
        compute(ks);
        stash(ks + 1, sa);
        __syncthreads();
        compute(ks + 1);
        if (three) {
            stash(ks + 2, sb);
            __syncthreads();
            compute(ks + 2);
        }
    } else {
        __syncthreads();
        for (int ks = 0; ks < nks; ++ks) {
            const bool more = ks + 1 < nks;
            if (more) fetch(ks + 1, sa);
            compute(ks);
            if (more) stash(ks + 1, sa);
            __syncthreads();
        }
    }
    }

    // ---- epilogue: lane (kg, n) of tile (i, j) holds rows kg * 4 + r, column n
    const int nl = lane & 15, kg = lane >> 4;
    const int row0 = tm * 128 + wm * 64 + kg * 4, col0 = tn * 128 + wn * 64 + nl;
    // every operand of the epilogue is requested before the first one is used (a load left inside the store loops is a
    // memory round trip per element: 64 of them per lane)
    int pos0 = 0;
    if (EPI == ACMI_BIG_QKV) pos0 = *p.pos;
    float biasv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) biasv[j] = p.bias != nullptr ? p.bias[min(col0 + j * 16, p.N - 1)] : 0.f;
    if (EPI == ACMI_BIG_RESID) {   // out += acc: the old values in two batches of 32 loads in flight (register budget: two
#pragma unroll
        for (int jh = 0; jh < 4; jh += 2) {   // workgroups per CU)
            float old[4][2][4];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        old[i][j][r] = p.out[(size_t)min(row0 + i * 16 + r, p.M - 1) * p.ldo + min(col0 + (jh + j) * 16, p.N - 1)];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][jh + j][r] += old[i][j][r];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = col0 + j * 16;
        if (col >= p.N) continue;
        const float bias = biasv[j];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rb = row0 + i * 16;           // first of this lane's 4 rows
            if (rb >= p.M) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + bias;
            if (EPI == ACMI_BIG_F32 || EPI == ACMI_BIG_RESID) {
#pragma unroll
                for (int r = 0; r < 4; ++r) p.out[(size_t)(rb + r) * p.ldo + col] = v[r];
            } else if (EPI == ACMI_BIG_TILED) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float y = p.act == 1 ? big_gelu(v[r]) : v[r];
                    st_f32(reinterpret_cast<WT*>(p.out_t) + tiled_index<WT>(rb + r, col, p.out_rbs), y);
                }
            } else {   // QKV: q to scratch, K / V appended to the cache, V also time-minor (4 positions = one store)
                const int part = col / p.d, f = col - part * p.d;
                if (part == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) p.q_out[(size_t)(rb + r) * p.d + f] = v[r];
                } else {
                    const int h = f / p.hd, dd = f - h * p.hd;
                    const int brow = rb / p.npos_pad, pidx = rb - brow * p.npos_pad;   // 4 rows = positions pidx .. pidx + 3 of one cache row
                    void* cache = part == 1 ? p.k_cache : p.v_cache;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (pidx + r >= p.npos) continue;
                        const size_t ci = (((size_t)brow * p.H + h) * p.Tcap + pos0 + pidx + r) * p.hd + dd;
                        if (p.kv_bf16) reinterpret_cast<bf16_t*>(cache)[ci] = f32_to_bf16(v[r]);
                        else reinterpret_cast<float*>(cache)[ci] = v[r];
                    }
                    if (part == 2 && p.vt != nullptr) {   // pad positions get the (finite) values of the pad rows: never attended to
                        const size_t vi = (((size_t)brow * p.H + h) * p.hd + dd) * p.vt_tcap + pos0 + pidx;
                        if (p.kv_bf16) {
                            bf16_t* dst = reinterpret_cast<bf16_t*>(p.vt) + vi;
                            if (((pos0 + pidx) & 3) == 0) {
                                *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                            } else {
#pragma unroll
                                for (int r = 0; r < 4; ++r) dst[r] = f32_to_bf16(v[r]);
                            }
                        } else {
                            float* dst = reinterpret_cast<float*>(p.vt) + vi;
                            if (((pos0 + pidx) & 3) == 0) {
                                *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                            } else {
#pragma unroll
                                for (int r = 0; r < 4; ++r) dst[r] = v[r];
                            }
                        }
                    }
                }
            }
        }
    }
}

template <typename WT, int E, int DMA>
static int launch_big_k(const BigArgs& a, int tiles, size_t lds, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&lin_big_kernel<WT, E, DMA>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess) {
            acmi_set_error("acmi_linear_big: cannot set the dynamic LDS limit");
            return ACMI_ELAUNCH;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL((lin_big_kernel<WT, E, DMA>), dim3(tiles), dim3(256), lds, st, a);
    return ACMI_OK;
}

// workgroups of a launch under lin_big_kernel's tile -> XCD mapping
static int big_grid(int M, int N) {
    const int tiles_m = (M + 127) / 128, tiles_n = (N + 127) / 128, P = tiles_n >> 3, R = tiles_n & 7;
    if (R == 1 || R == 2 || R == 4) {
        const int share = 8 / R;
        return 8 * ((tiles_m + share - 1) / share) * (share * P + 1);
    }
    return tiles_m * tiles_n;
}

template <typename WT>
static int launch_big_t(const BigArgs& a, hipStream_t st) {
    const int tiles = big_grid(a.M, a.N);
    const size_t lds = 2 * 32 * 1024;
    // A/B switch: ACMI_BIG_DMA = 0 (register staging of round 3), 1 (two-buffer DMA), 2 (four-slot DMA ring; default)
    static const int dma = getenv("ACMI_BIG_DMA") != nullptr ? atoi(getenv("ACMI_BIG_DMA")) : 2;
#define ACMI_BIG_CASE(E)                                                                                                  \
    case E: {                                                                                                             \
        const int rc = dma == 2 ? launch_big_k<WT, E, 2>(a, tiles, lds, st)                                               \
                                : (dma == 1 ? launch_big_k<WT, E, 1>(a, tiles, lds, st) : launch_big_k<WT, E, 0>(a, tiles, lds, st)); \
        if (rc) return rc;                                                                                                \
        break;                                                                                                            \
    }
    switch (a.epi) {
        ACMI_BIG_CASE(ACMI_BIG_F32)
        ACMI_BIG_CASE(ACMI_BIG_RESID)
        ACMI_BIG_CASE(ACMI_BIG_TILED)
        ACMI_BIG_CASE(ACMI_BIG_QKV)
        default:
            acmi_set_error("acmi_linear_big: bad epilogue %d", a.epi);
            return ACMI_EINVAL;
    }
#undef ACMI_BIG_CASE
    return acmi_check_launch("lin_big_kernel");
}

int acmi_launch_big(BigArgs& a, int wdtype, hipStream_t st) {
    const int kt = wdtype == ACMI_BF16 ? 32 : 16;
    ACMI_REQUIRE(a.M > 0 && a.M % 16 == 0 && a.N > 0 && a.K > 0, "acmi_linear_big: M=%d must be a positive multiple of 16 (N=%d K=%d)",
                 a.M, a.N, a.K);
    a.NKC = (a.K + kt - 1) / kt;
    ACMI_REQUIRE(a.NKC % 2 == 0, "acmi_linear_big: K=%d must span an even number of %d-column tiles", a.K, kt);
    if (a.a_rbs <= 0) a.a_rbs = a.NKC;
    ACMI_REQUIRE(a.a_rbs >= a.NKC, "acmi_linear_big: a_rbs=%d < %d K tiles", a.a_rbs, a.NKC);
    return wdtype == ACMI_BF16 ? launch_big_t<bf16_t>(a, st) : launch_big_t<float>(a, st);
}

extern "C" int acmi_linear_big(const void* a, int a_rbs, const void* w, int wdtype, const float* bias, void* out, int out_mode,
                               int out_ld, int act, int accumulate, int M, int N, int K, void* stream) {
    BigArgs p = {};
    p.a = a; p.a_rbs = a_rbs; p.w = w; p.bias = bias; p.M = M; p.N = N; p.K = K; p.act = act;
    ACMI_REQUIRE(out_mode == ACMI_OUT_F32 || out_mode == ACMI_OUT_TILED, "acmi_linear_big: out_mode %d unsupported", out_mode);
    ACMI_REQUIRE(!(accumulate && out_mode != ACMI_OUT_F32), "acmi_linear_big: accumulate needs an f32 row-major output");
    ACMI_REQUIRE(act == 0 || out_mode == ACMI_OUT_TILED, "acmi_linear_big: the activation is fused into the tiled epilogue only");
    if (out_mode == ACMI_OUT_F32) {
        p.epi = accumulate ? ACMI_BIG_RESID : ACMI_BIG_F32;
        p.out = reinterpret_cast<float*>(out); p.ldo = out_ld > 0 ? out_ld : N;
    } else {
        const int kt = wdtype == ACMI_BF16 ? 32 : 16;
        p.epi = ACMI_BIG_TILED; p.out_t = out; p.out_rbs = out_ld > 0 ? out_ld : (N + kt - 1) / kt;
    }
    return acmi_launch_big(p, wdtype, (hipStream_t)stream);
}

// =====================================================================================================
// causal prefill attention
// =====================================================================================================
// One wave = 16 consecutive query positions of one (cache row, head); a workgroup = 4 such waves (no LDS, no barrier).
// S^T[t, q] = sum_d K[t, d] Q[q, d]: MFMA A operand = K rows straight from the cache (lane (kg, m = t): 16 bytes of row t),
// B operand = the wave's queries (lane (kg, n = q)).  In the C layout lane (kg, q) then holds the scores of keys
// t0 + tt * 16 + kg * 4 + r (tt < TT, r < 4) of ITS query: softmax statistics are per lane column (4 lanes, xor 16 / 32),
// and the probabilities are already the B operand of the second MFMA  O^T[d, q] = sum_t V^T[d, t] P[q, t]  if that MFMA's
// k slot (kg, j = tt * 4 + r) means key t0 + tt * 16 + kg * 4 + r -- which is how the A operand is fetched from the
// time-minor V: lane (kg, m = d) reads keys t0 + tt * 16 + kg * 4 .. + 3 of row d (8 bytes per tt in bf16).
template <typename KT, int HD>
__global__ __launch_bounds__(256) void attn_prefill_kernel(const PrefillAttnArgs p) {
    constexpr int E = 16 / (int)sizeof(KT);       // elements per lane of a fragment: 8 (bf16) / 4 (f32)
    constexpr int KTILE = 4 * E;                  // k columns per fragment: 32 / 16
    constexpr int TT = KTILE / 16;                // 16-key score tiles per probability fragment: 2 / 1
    constexpr int NKD = (HD + KTILE - 1) / KTILE; // fragments along the head dimension
    constexpr int ND = (HD + 15) / 16;            // 16-row tiles of O^T
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nl = lane & 15, kg = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = (blockIdx.x * 4 + wave) * 16;   // first query position (index inside this call) of the wave
    if (q0 >= p.npos) return;
    const int pos0 = *p.pos;
    const int d = p.H * HD;
    const KT* __restrict__ kc = reinterpret_cast<const KT*>(p.k_cache) + ((size_t)b * p.H + h) * p.Tcap * HD;
    const KT* __restrict__ vt = reinterpret_cast<const KT*>(p.vt) + ((size_t)b * p.H + h) * HD * p.vt_tcap;

    // queries as B fragments: lane (kg, n) holds Q[q0 + n][kc * KTILE + kg * E .. + E), scaled, in the cache's element type
    const int qi = min(q0 + nl, p.npos - 1);
    const float* qrow = p.q + ((size_t)b * p.npos_pad + qi) * d + h * HD;
    u32x4 qf[NKD];
#pragma unroll
    for (int c = 0; c < NKD; ++c) {
        float t[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int f = c * KTILE + kg * E + e;
            t[e] = f < HD ? qrow[min(f, HD - 1)] : 0.f;
        }
        if (sizeof(KT) == 2) {
            qf[c] = u32x4{pack_bf16x2(t[0], t[1]), pack_bf16x2(t[2 % E], t[3 % E]), pack_bf16x2(t[4 % E], t[5 % E]),
                          pack_bf16x2(t[6 % E], t[7 % E])};
        } else {
            qf[c] = u32x4{__float_as_uint(t[0]), __float_as_uint(t[1]), __float_as_uint(t[2 % E]), __float_as_uint(t[3 % E])};
        }
    }
    // causal: keys <= the query's absolute position are visible; non-causal (cross-attention): keys [0, klen) of the row
    const int klen = p.causal ? 0 : (p.klen_rows != nullptr ? max(1, min(p.klen_rows[b], p.klen)) : p.klen);
    const int tq = p.causal ? pos0 + q0 + nl : klen - 1;
    const int tq_last = p.causal ? pos0 + min(q0 + 15, p.npos - 1) : klen - 1;
    int t_first = 0;
    if (p.causal && p.past_context > 0) t_first = max(0, pos0 + q0 - p.past_context) / KTILE * KTILE;
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 o[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int t0 = t_first; t0 <= tq_last; t0 += KTILE) {
        // K fragments (A operand of S^T) and V^T fragments (A operand of O^T) of this key block: all requested up front
        u32x4 kf[TT][NKD];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const int t = min(t0 + tt * 16 + nl, p.Tcap - 1);
#pragma unroll
            for (int c = 0; c < NKD; ++c) {
                const int f = c * KTILE + kg * E;   // first head feature of this lane's 16 bytes
                kf[tt][c] = f < HD ? *reinterpret_cast<const u32x4*>(kc + (size_t)t * HD + f) : u32x4{0u, 0u, 0u, 0u};
            }
        }
        u32x4 vf[ND];
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int dd = min(i * 16 + nl, HD - 1);
            const KT* vrow = vt + (size_t)dd * p.vt_tcap + t0 + kg * 4;   // keys t0 + tt * 16 + kg * 4 .. + 3
            if (sizeof(KT) == 2) {
                const uint2 lo = *reinterpret_cast<const uint2*>(vrow);
                const uint2 hi = *reinterpret_cast<const uint2*>(vrow + 16);
                vf[i] = u32x4{lo.x, lo.y, hi.x, hi.y};
            } else {
                vf[i] = *reinterpret_cast<const u32x4*>(vrow);
            }
        }
        // scores of this lane's query against keys t0 + tt * 16 + kg * 4 + r
        f32x4 s[TT];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            s[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NKD; ++c) big_mma(kf[tt][c], qf[c], s[tt], KT());
        }
        float cmax = -INFINITY;
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = t0 + tt * 16 + kg * 4 + r;
                const bool vis = t <= tq && (!p.causal || p.past_context <= 0 || t >= tq - p.past_context);
                s[tt][r] = vis ? s[tt][r] * p.scale : -INFINITY;
                cmax = fmaxf(cmax, s[tt][r]);
            }
        cmax = fmaxf(cmax, __shfl_xor(cmax, 16, 64));
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
        const float m_new = fmaxf(m_run, cmax);
        // a query with no visible key so far (its window starts in a later block) keeps m = -inf: alpha = 1, p = 0
        const float alpha = m_new == -INFINITY ? 1.f : expf(m_run - m_new);
        float psum = 0.f, pr[TT * 4];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = s[tt][r] == -INFINITY ? 0.f : expf(s[tt][r] - m_new);
                pr[tt * 4 + r] = e;
                psum += e;
            }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        u32x4 pf;
        if (sizeof(KT) == 2) {
            pf = u32x4{pack_bf16x2(pr[0], pr[1]), pack_bf16x2(pr[2], pr[3]), pack_bf16x2(pr[4 % (TT * 4)], pr[5 % (TT * 4)]),
                       pack_bf16x2(pr[6 % (TT * 4)], pr[7 % (TT * 4)])};
        } else {
            pf = u32x4{__float_as_uint(pr[0]), __float_as_uint(pr[1]), __float_as_uint(pr[2]), __float_as_uint(pr[3])};
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[i][r] *= alpha;
            big_mma(vf[i], pf, o[i], KT());
        }
    }
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    // O^T: lane (kg, n = q) holds head features i * 16 + kg * 4 + r -> 4 consecutive columns of the tiled output row
    const int q = q0 + nl;
    if (q >= p.npos) return;
    const int row = b * p.npos_pad + q;
    const float inv = 1.0f / l_run;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const int f = i * 16 + kg * 4;
        if (f >= HD) continue;
        const int col = h * HD + f;
        if (p.out_bf16) {
            bf16_t* dst = reinterpret_cast<bf16_t*>(p.out) + tiled_index<bf16_t>(row, col, p.out_rbs);
            *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(o[i][0] * inv, o[i][1] * inv), pack_bf16x2(o[i][2] * inv, o[i][3] * inv));
        } else {
            float* dst = reinterpret_cast<float*>(p.out) + tiled_index<float>(row, col, p.out_rbs);
            *reinterpret_cast<float4*>(dst) = make_float4(o[i][0] * inv, o[i][1] * inv, o[i][2] * inv, o[i][3] * inv);
        }
    }
}

int acmi_launch_prefill_attn(PrefillAttnArgs& a, int kvdtype, int hd, int Beff, hipStream_t st) {
    ACMI_REQUIRE(a.npos > 0 && a.npos_pad >= a.npos && a.npos_pad % 16 == 0, "acmi_attn_prefill: bad npos=%d npos_pad=%d", a.npos, a.npos_pad);
    ACMI_REQUIRE(a.vt_tcap % 32 == 0 && a.vt_tcap > 0, "acmi_attn_prefill: vt_tcap=%d must be a positive multiple of 32", a.vt_tcap);
    ACMI_REQUIRE(a.causal || (a.klen > 0 && a.klen <= a.Tcap && a.klen <= a.vt_tcap), "acmi_attn_prefill: klen=%d outside the caches", a.klen);
    ACMI_REQUIRE(hd % 4 == 0, "acmi_attn_prefill: head dim %d", hd);
    a.scale = 1.0f / sqrtf((float)hd);
    dim3 grid((a.npos + 63) / 64, a.H, Beff), block(256);
#define ACMI_PFA_CASE(HDv)                                                                                              \
    case HDv:                                                                                                           \
        if (kvdtype == ACMI_BF16) hipLaunchKernelGGL((attn_prefill_kernel<bf16_t, HDv>), grid, block, 0, st, a);        \
        else hipLaunchKernelGGL((attn_prefill_kernel<float, HDv>), grid, block, 0, st, a);                              \
        break;
    switch (hd) {
        ACMI_PFA_CASE(8)
        ACMI_PFA_CASE(16)
        ACMI_PFA_CASE(32)
        ACMI_PFA_CASE(64)
        ACMI_PFA_CASE(128)
        default:
            acmi_set_error("acmi_attn_prefill: head dim %d unsupported (8, 16, 32, 64, 128)", hd);
            return ACMI_EINVAL;
    }
#undef ACMI_PFA_CASE
    return acmi_check_launch("attn_prefill_kernel");
}

extern "C" int acmi_attn_prefill(const float* q, const void* k_cache, const void* vt, int kvdtype, void* out, int out_dtype,
                                 int out_rbs, int Beff, int H, int hd, int Tcap, int vt_tcap, int npos, int npos_pad,
                                 const int* pos, int past_context, void* stream) {
    PrefillAttnArgs a = {};
    a.q = q; a.k_cache = k_cache; a.vt = vt; a.out = out; a.out_bf16 = out_dtype == ACMI_BF16;
    const int kt = out_dtype == ACMI_BF16 ? 32 : 16;
    a.out_rbs = out_rbs > 0 ? out_rbs : (H * hd + kt - 1) / kt;
    a.H = H; a.Tcap = Tcap; a.vt_tcap = vt_tcap; a.npos = npos; a.npos_pad = npos_pad; a.pos = pos; a.past_context = past_context;
    a.causal = 1;
    ACMI_REQUIRE(pos != nullptr && Beff > 0 && H > 0 && Tcap > 0, "acmi_attn_prefill: bad arguments");
    return acmi_launch_prefill_attn(a, kvdtype, hd, Beff, (hipStream_t)stream);
}
