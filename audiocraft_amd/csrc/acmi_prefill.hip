// Prefill of the MusicGen LM for gfx950 (CDNA4, wave64): the prompt / prepended-condition positions of a generate run
// through ONE forward, like the reference does (audiocraft/models/lm.py:540-543, modules/transformer.py:233-264, 362-414,
// models/genmodel.py:233-262), instead of position by position through the decode kernels.
//
//   lin_big_kernel        MFMA-tiled GEMM  out[M, N] = a[M, K] W[N, K]^T  on the SAME tiled operands the decode step uses
//                         (1 KB A / B fragments, include/acmi.h): 256 x 256 workgroup tiles of 8 waves (128 x 128 of 4
//                         for small launches), fragments DMA'd into an LDS ring in fragment order (every ds_read_b128 is
//                         lane linear: conflict free), epilogues: f32 / residual add / tiled (+ GELU) / QKV scatter into the KV
//                         cache (+ V time-minor for the prefill attention).
//   attn_prefill_kernel   causal attention of all prompt positions of a (cache row, head): flash-style over the K cache,
//                         S^T = K Q^T so that a query's scores sit in one lane column, P feeds the second MFMA straight
//                         from registers (its k slots are matched by the load pattern of the time-minor V), online softmax.
//
// Row layout of every prefill activation: POSITION-MINOR and padded, row = cache_row * npos_pad + position (npos_pad =
// positions rounded up to 16), so that a 16-row MFMA block is 16 consecutive positions of ONE cache row: the 16 lanes that
// hold one feature of such a block in the QKV epilogue store 16 consecutive entries of a V^T row and never straddle
// cache rows.
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
// GELU for a bf16 result: erfc(|x| / sqrt 2) by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7: one v_rcp_f32, one v_exp_f32
// and six FMAs against erff's ~45 instructions -- the exact form cost the FFN1 epilogue more than its K loop's MFMAs).
// |gelu - exact| <= 4.3e-7 everywhere, 1e-7 relative in norm; after rounding to bf16 9e-5 of N(0, 1) inputs land one ulp
// off (never more above |y| = 1e-3): below what the order of the K sum already moves.  f32 results keep erff.
__device__ __forceinline__ float big_gelu_bf16(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
    float q = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    q = __builtin_fmaf(q, t, 1.421413741f);
    q = __builtin_fmaf(q, t, -0.284496736f);
    q = __builtin_fmaf(q, t, 0.254829592f);
    const float c = q * t * __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);   // erfc(z)
    return 0.5f * x * (x >= 0.f ? 2.0f - c : c);                                      // 1 + erf(x / sqrt 2)
}

// =====================================================================================================
// MFMA-tiled GEMM on tiled operands
// =====================================================================================================
// Tiles (BIG): 0 = 128 x 128, 4 waves (2 x 2) of 64 x 64, two workgroups per CU -- small launches, which a 256 x 256 grid
// would not spread over the chip; 1 = 256 x 256, 8 waves (2 x 4) of 128 x 64, one workgroup per CU (128 KB of LDS), 128
// accumulator VGPRs per lane: half the fragment bytes per flop through the CU's memory path, the LDS and the registers.
//
// Staging: the fragments of a stage (ONE K fragment of every 16-row / 16-feature block of the tile) go global -> LDS
// directly (`global_load_lds_dwordx4`: one op per 1 KB fragment, the wave's lanes land at consecutive 16-byte slots --
// exactly the fragment order, so every ds_read_b128 is lane linear: conflict free) into a ring of four stages.  Every wave
// moves 4 fragments per stage as ONE block of inline asm: M0 (the LDS destination) written once, the fragments addressed
// by an SGPR base + one VGPR offset + the instruction's immediate, which moves the LDS and the global address alike (the
// bases are pre-decremented by it).  Inline asm because hipcc drains vmcnt to 0 in front of every barrier and every
// ds_read that may alias a DMA destination -- no request would survive a K step; completion is counted by hand
// (`vmcnt(8)` = all but this wave's two youngest stages have landed; the kernel's only other VMEM operations are the
// epilogue's, behind a vmcnt(0)).
//
// K loop (the second half of round 4; timelines by scripts/big_gemm_bench.py --trace with a -DACMI_BIG_TRACE build,
// profiles/archive/r04_session14 .. 22_prefill_gemm.log; cycles per K step of the 256 x 256 tile, ideal = 1024: the SIMD's 64
// MFMAs of 16 cycles):
//   1735  every wave requests, then reads its 12 fragments from LDS, then multiplies: 96 KB of reads per CU at the top
//         of the step starve the matrix pipes, the waves meet again at the barrier (no wait for memory at all: 24 cycles)
//   1623  requests spread between the MFMAs
//   1495  fragments read HALF A STEP AHEAD into registers (below)
//   1400  + the requests as one block (a separate M0 write + 64-bit VGPR address per request cost ~25 cycles each)
//   1374  the same loop without requests, 1192 without LDS reads either (timing only): what is left is the matrix pipe --
//         of a SIMD's two waves the older one issues ALL its MFMAs first (17.8 cycles each), the younger one follows --
//         and ~190 cycles in which LDS reads and MFMAs do not overlap.
// A stage s is multiplied in two halves: half 0 takes the lower row blocks (`b`, `lo`: read during the previous half) while
// the upper ones (`hi`) arrive; the barrier sits between the halves; half 1 takes the upper row blocks while `b`, `lo` of
// stage s + 1 arrive, and requests stage s + 4 into the slot of stage s, which every wave has finished reading
// (lgkmcnt(0) in front of the barrier).  The cycle counts are data independent; the clock is not: all-zero operands run
// the same cycles 1.2-1.4x faster (r04_session17 / 21: 1.8-1.9 GHz under random operands) -- 2.5 PFLOP/s is 2.4 GHz.
//
// The MFMAs run TRANSPOSED (A operand = weight fragment, B operand = activation fragment; the two share one register
// layout): lane (kg, nl) of accumulator (i, j) then holds 4 CONSECUTIVE FEATURES j * 16 + kg * 4 + r of activation row
// i * 16 + nl, so every epilogue stores 16 bytes (f32) / 8 bytes (bf16) per lane -- a tiled bf16 output lands as contiguous
// 512-byte halves of its fragments -- instead of four scattered scalars.
template <int BIG> struct BigTile {
    static constexpr int TS = BIG ? 8 : 7;     // log2 of the tile edge
    static constexpr int T = 1 << TS;
    static constexpr int FR = T / 16;          // fragments per operand per stage
    static constexpr int NW = BIG ? 8 : 4;     // waves: 2 along m x (NW / 2) along n
    static constexpr int WI = BIG ? 8 : 4;     // 16-row blocks of a wave's tile (its width is 64 features)
    static constexpr int WNS = BIG ? 2 : 1;    // log2 of the waves along n
    static constexpr int SLOT = 2 * FR;        // fragments (KB) per ring slot
};

#ifdef ACMI_BIG_TRACE
// Lab build only (libacmi_bigtrace.so, scripts/big_gemm_bench.py --trace): per wave, the shader-clock cycles spent in each
// part of the kernel: [0] prologue (start -> first barrier passed), [1] half 0, [2] half 1, [3] lgkmcnt / vmcnt wait,
// [4] barrier wait (1-4 summed over the steady-state stages), [5] drain stages, [6] epilogue, [7] steady-state stages.
__device__ unsigned long long g_big_trace[2048 * 8 * 8];
extern "C" int acmi_big_trace_read(unsigned long long* out, int words) {
    // read and clear (the next launch may have fewer workgroups)
    void* sym = nullptr;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_big_trace), (size_t)words * 8) != hipSuccess) return ACMI_ELAUNCH;
    if (hipGetSymbolAddress(&sym, HIP_SYMBOL(g_big_trace)) != hipSuccess) return ACMI_ELAUNCH;
    return hipMemset(sym, 0, sizeof(g_big_trace)) == hipSuccess ? ACMI_OK : ACMI_ELAUNCH;
}
#define ACMI_BT(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define ACMI_BT_SET(var) var = __builtin_amdgcn_s_memtime()
#define ACMI_BT_ADD(k, a, b) tr[k] += (b) - (a)
#else
#define ACMI_BT(var)
#define ACMI_BT_SET(var)
#define ACMI_BT_ADD(k, a, b)
#endif

template <typename WT, int EPI, int BIG>
__global__ __launch_bounds__(BigTile<BIG>::NW * 64, BIG ? 1 : 2) void lin_big_kernel(const BigArgs p) {
    using TC = BigTile<BIG>;
    constexpr int TS = TC::TS, FR = TC::FR, WI = TC::WI, SLOT = TC::SLOT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 4 ring slots x SLOT fragments x 1 KB
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> TC::WNS, wn = wave & ((1 << TC::WNS) - 1);
    const int tiles_n = (p.N + TC::T - 1) >> TS, tiles_m = (p.M + TC::T - 1) >> TS, nb = tiles_m * tiles_n;
    // Workgroup b runs on XCD b % 8 and every XCD has its own 4 MB L2.  XCD x owns tiles_n / 8 ADJACENT COLUMN tiles and
    // sweeps the row tiles: its weight panels stay in its L2 and every activation panel is read once per XCD.  A remainder
    // of 1, 2 or 4 column tiles is shared by 8 / remainder XCDs each, which take its row tiles in turn (big_grid below sizes
    // the launch: workgroups past an XCD's list leave at once).  (Round 3 gave each XCD a contiguous run of ROW-major tiles:
    // the tiles it works on at a time then touch all weight panels -- far beyond its L2 -- once per row tile: 1.4 GB from the
    // memory side per FFN1 GEMM of a 600-position prefill.)
    int bid = blockIdx.x, tm, tn;
    const int P = tiles_n >> 3, R = tiles_n & 7;
    if (R == 0) {
        const int x = bid & 7, l = bid >> 3;
        tm = l / P;
        tn = x * P + (l - tm * P);
    } else if (P > 0 && (R == 1 || R == 2 || R == 4)) {
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
    // the 4 fragments of a stage this wave moves: the lower half of the waves the activation's (row block wq * 4 + h), the
    // upper half the weight's; LDS fragment index within the slot = wave * 4 + h in both cases.  sb[h]: byte address of
    // fragment (block, K fragment 0), pre-decremented by the immediate its request carries
    constexpr int HW = TC::NW / 2;
    const bool mine_a = wave < HW;
    const int wq = wave & (HW - 1);
    unsigned long long sb[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const size_t blk = mine_a ? (size_t)min(tm * FR + wq * 4 + h, MT16 - 1) * p.a_rbs : (size_t)min(tn * FR + wq * 4 + h, NT16 - 1) * p.NKC;
        sb[h] = reinterpret_cast<unsigned long long>(mine_a ? p.a : p.w) + blk * 1024 - (size_t)h * 1024;
    }
    const u32x4* lds = reinterpret_cast<const u32x4*>(smem);
    f32x4 acc[WI][4];
#pragma unroll
    for (int i = 0; i < WI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nst = p.NKC;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
#ifdef ACMI_BIG_TRACE
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_epi = 0;
    ACMI_BT(t_start);
#endif
    auto dma = [&](int st) {   // this wave's 4 fragments of stage st (K fragment st) into ring slot st & 3
        const unsigned dst = lds0 + (unsigned)(((st & 3) * SLOT + wave * 4) * 1024);   // wave-uniform: M0
        const unsigned voff = (unsigned)lane * 16u + (unsigned)st * 1024u;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %3\n\t"
                     "global_load_lds_dwordx4 %1, %4 offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, %5 offset:2048\n\t"
                     "global_load_lds_dwordx4 %1, %6 offset:3072\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(dst), "s"(sb[0]), "s"(sb[1]), "s"(sb[2]), "s"(sb[3]) : "memory");
    };
    auto bar = [&]() {         // workgroup barrier WITHOUT the fence of __syncthreads (that fence is a vmcnt(0))
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    {
        constexpr int HI = WI / 2;
        u32x4 fb[2][4], flo[HI], fhi[HI];
        auto slot = [&](int s_) { return lds + (s_ & 3) * SLOT * 64 + lane; };
        auto rd_b = [&](int s_, u32x4 (&b)[4]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = slot(s_)[(FR + wn * 4 + j) * 64];
        };
        auto rd_lo = [&](int s_) {
#pragma unroll
            for (int i = 0; i < HI; ++i) flo[i] = slot(s_)[(wm * WI + i) * 64];
        };
        auto rd_hi = [&](int s_) {
#pragma unroll
            for (int i = 0; i < HI; ++i) fhi[i] = slot(s_)[(wm * WI + HI + i) * 64];
        };
        // MFMAs are pure register arithmetic to the compiler, which otherwise sinks them past the barriers and the requests
        // (a half's 16 MFMAs were found behind the NEXT barrier): an empty volatile asm that "updates" a row block's four
        // accumulators keeps them where they are written
        auto pin = [&](f32x4 (&c)[4]) { asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])); };
        auto wait_vm = [&](int young) {   // this wave's requests but the `young` most recent stages have landed
            if (young >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (young == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (young == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        // stage s_: b = its weight fragments (in registers, like flo); bn receives the next stage's; pre: there is a next
        // stage; req: stage s_ + 4 exists; young: own stages in flight behind s_ + 1 at the barrier
        auto stage = [&](int s_, const u32x4 (&b)[4], u32x4 (&bn)[4], bool pre, bool req, int young) {
            ACMI_BT(t0);
#pragma unroll
            for (int i = 0; i < HI; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) big_mma(b[j], flo[i], acc[i][j], WT());   // transposed: see the header
                pin(acc[i]);
                if (i == 0) { rd_hi(s_); __builtin_amdgcn_sched_barrier(0); }   // behind the first MFMAs: nothing waits for them yet
            }
            ACMI_BT(t1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // hi has arrived: this wave is done with the slot of stage s_
            wait_vm(young);                                       // stage s_ + 1 (this wave's part) has landed
            ACMI_BT(t2);
            bar();
            ACMI_BT(t3);
            if (pre) { rd_b(s_ + 1, bn); rd_lo(s_ + 1); }
            __builtin_amdgcn_sched_barrier(0);
            if (req) { dma(s_ + 4); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int i = 0; i < HI; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) big_mma(b[j], fhi[i], acc[HI + i][j], WT());
                pin(acc[HI + i]);
            }
            ACMI_BT(t4);
            ACMI_BT_ADD(1, t0, t1); ACMI_BT_ADD(3, t1, t2); ACMI_BT_ADD(4, t2, t3); ACMI_BT_ADD(2, t3, t4);
        };
        dma(0);
        if (nst > 1) dma(1);
        if (nst > 2) dma(2);
        if (nst > 3) dma(3);
        wait_vm(min(nst, 4) - 1);
        bar();
        rd_b(0, fb[0]);
        rd_lo(0);
        ACMI_BT(t_loop);
        ACMI_BT_ADD(0, t_start, t_loop);
        int s_ = 0;
        for (; s_ + 5 < nst; s_ += 2) {   // straight line: both stages of the iteration have a stage + 4 to request
            stage(s_, fb[0], fb[1], true, true, 2);
            stage(s_ + 1, fb[1], fb[0], true, true, 2);
        }
        ACMI_BT(t_drain);
        ACMI_BT_ADD(7, t_loop, t_drain);
        auto young_at = [&](int q) { return max(min(nst - 1, q + 3) - (q + 1), 0); };
        for (; s_ < nst; s_ += 2) {
            stage(s_, fb[0], fb[1], s_ + 1 < nst, s_ + 4 < nst, young_at(s_));
            if (s_ + 1 < nst) stage(s_ + 1, fb[1], fb[0], s_ + 2 < nst, s_ + 5 < nst, young_at(s_ + 1));
        }
        ACMI_BT_SET(t_epi);
        ACMI_BT_ADD(5, t_drain, t_epi);
    }

    // ---- epilogue: lane (kg, nl) of accumulator (i, j) holds row i * 16 + nl, features j * 16 + kg * 4 + (0..3)
    const int nl = lane & 15, kg = lane >> 4;
    const int row0 = (tm << TS) + wm * (WI * 16) + nl, col0 = (tn << TS) + wn * 64 + kg * 4;
    // every operand of the epilogue is requested before the first one is used
    int pos0 = 0;
    if (EPI == ACMI_BIG_QKV) pos0 = *p.pos;
    float biasv[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) biasv[j][r] = p.bias != nullptr ? p.bias[min(col0 + j * 16 + r, p.N - 1)] : 0.f;
    // f32 rows: 16-byte aligned groups of 4 features (N, ldo multiples of 4); the large tile is launched only then
    const bool vec = BIG ? true : p.vec != 0;
    // RESID (out += result): the old values of feature block j + 1 are requested before block j is added and stored -- one
    // block ahead, not all four at once: 4 x WI f32x4 next to the accumulators would not fit the register file
    auto load_old = [&](int j, f32x4 (&old)[WI]) {
        const int c = col0 + j * 16;
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const float* src_ = p.out + (size_t)min(row0 + i * 16, p.M - 1) * p.ldo;
            if (vec) {
                old[i] = *reinterpret_cast<const f32x4*>(src_ + min(c, p.N - 4));
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) old[i][r] = src_[min(c + r, p.N - 1)];
            }
        }
    };
    f32x4 oldv[2][WI];
    if (EPI == ACMI_BIG_RESID) load_old(0, oldv[0]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = col0 + j * 16;
        if (EPI == ACMI_BIG_RESID) {
            if (j < 3) load_old(j + 1, oldv[(j + 1) & 1]);
#pragma unroll
            for (int i = 0; i < WI; ++i) acc[i][j] += oldv[j & 1][i];
        }
        if (col >= p.N) continue;
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int row = row0 + i * 16;
            if (row >= p.M) continue;
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + biasv[j][r];
            if (EPI == ACMI_BIG_F32 || EPI == ACMI_BIG_RESID) {
                float* dst = p.out + (size_t)row * p.ldo + col;
                if (vec) {
                    *reinterpret_cast<f32x4*>(dst) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col + r < p.N) dst[r] = v[r];
                }
            } else if (EPI == ACMI_BIG_TILED) {   // N % 4 == 0: the 4 features are one half / the whole of a lane's 16 bytes
                if (p.act == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = sizeof(WT) == 2 ? big_gelu_bf16(v[r]) : big_gelu(v[r]);
                }
                WT* dst = reinterpret_cast<WT*>(p.out_t) + tiled_index<WT>(row, col, p.out_rbs);
                if constexpr (sizeof(WT) == 2) *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                else *reinterpret_cast<f32x4*>(dst) = v;
            } else {   // QKV (d % 16 == 0, hd % 4 == 0): q to scratch, K / V appended to the cache, V also time-minor
                const int part = col / p.d, f = col - part * p.d;
                if (part == 0) {
                    *reinterpret_cast<f32x4*>(p.q_out + (size_t)row * p.d + f) = v;
                } else {
                    const int h = f / p.hd, dd = f - h * p.hd;
                    const int brow = row / p.npos_pad, pidx = row - brow * p.npos_pad;
                    if (pidx >= p.npos) continue;   // pad rows of the position-minor layout
                    void* cache = part == 1 ? p.k_cache : p.v_cache;
                    const size_t ci = (((size_t)brow * p.H + h) * p.Tcap + pos0 + pidx) * p.hd + dd;
                    if (p.kv_bf16) *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(cache) + ci) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                    else *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(cache) + ci) = v;
                    if (part == 2 && p.vt != nullptr) {   // 16 lanes = 16 consecutive positions of one feature row
                        const size_t vi = (((size_t)brow * p.H + h) * p.hd + dd) * p.vt_tcap + pos0 + pidx;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (p.kv_bf16) reinterpret_cast<bf16_t*>(p.vt)[vi + (size_t)r * p.vt_tcap] = f32_to_bf16(v[r]);
                            else reinterpret_cast<float*>(p.vt)[vi + (size_t)r * p.vt_tcap] = v[r];
                        }
                    }
                }
            }
        }
    }
#ifdef ACMI_BIG_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ACMI_BT(t_end);
    ACMI_BT_ADD(6, t_epi, t_end);
    if (lane == 0 && blockIdx.x < 2048) {
#pragma unroll
        for (int k = 0; k < 8; ++k) g_big_trace[((size_t)blockIdx.x * 8 + wave) * 8 + k] = tr[k];
    }
#endif
}

template <typename WT, int E, int BIG>
static int launch_big_k(const BigArgs& a, int tiles, hipStream_t st) {
    constexpr size_t lds = 4 * BigTile<BIG>::SLOT * 1024;
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&lin_big_kernel<WT, E, BIG>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;   // once, thread safe
    if (!attr_ok) {
        acmi_set_error("acmi_linear_big: cannot set the dynamic LDS limit");
        return ACMI_ELAUNCH;
    }
    hipLaunchKernelGGL((lin_big_kernel<WT, E, BIG>), dim3(tiles), dim3(BigTile<BIG>::NW * 64), lds, st, a);
    return ACMI_OK;
}

// workgroups of a launch under lin_big_kernel's tile -> XCD mapping (ts = log2 of the tile edge)
static int big_grid(int M, int N, int ts) {
    const int t = 1 << ts, tiles_m = (M + t - 1) >> ts, tiles_n = (N + t - 1) >> ts, P = tiles_n >> 3, R = tiles_n & 7;
    if (P > 0 && (R == 1 || R == 2 || R == 4)) {
        const int share = 8 / R;
        return 8 * ((tiles_m + share - 1) / share) * (share * P + 1);
    }
    return tiles_m * tiles_n;
}

template <typename WT>
static int launch_big_t(const BigArgs& a, hipStream_t st) {
    // 256 x 256 tiles once they give most CUs a workgroup; ACMI_BIG_TILE = 0 / 1 forces the small / the large tile (A/B, tests)
    static const int force = getenv("ACMI_BIG_TILE") != nullptr ? atoi(getenv("ACMI_BIG_TILE")) : -1;
    const int t256 = ((a.M + 255) >> 8) * ((a.N + 255) >> 8);
    const bool vec_ok = a.vec || a.epi == ACMI_BIG_TILED || a.epi == ACMI_BIG_QKV;   // the large tile has no scalar f32 epilogue
    const int big = vec_ok && (force >= 0 ? (force != 0) : (t256 >= 128));
    const int tiles = big_grid(a.M, a.N, big ? 8 : 7);
#define ACMI_BIG_CASE(E)                                                                                            \
    case E: {                                                                                                       \
        const int rc = big ? launch_big_k<WT, E, 1>(a, tiles, st) : launch_big_k<WT, E, 0>(a, tiles, st);           \
        if (rc) return rc;                                                                                          \
        break;                                                                                                      \
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
    if (a.a_rbs <= 0) a.a_rbs = a.NKC;
    ACMI_REQUIRE(a.a_rbs >= a.NKC, "acmi_linear_big: a_rbs=%d < %d K tiles", a.a_rbs, a.NKC);
    if (a.epi == ACMI_BIG_F32 || a.epi == ACMI_BIG_RESID)
        a.vec = a.N % 4 == 0 && a.ldo % 4 == 0 && (reinterpret_cast<size_t>(a.out) & 15) == 0;
    ACMI_REQUIRE(a.epi != ACMI_BIG_TILED || a.N % 4 == 0, "acmi_linear_big: a tiled output needs N %% 4 == 0 (N=%d)", a.N);
    ACMI_REQUIRE(a.epi != ACMI_BIG_QKV || (a.d % 16 == 0 && a.hd % 4 == 0 && a.N == 3 * a.d && (reinterpret_cast<size_t>(a.q_out) & 15) == 0),
                 "acmi_linear_big: the QKV epilogue needs d %% 16 == 0, hd %% 4 == 0 and N = 3 d (d=%d hd=%d N=%d)", a.d, a.hd, a.N);
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
// What bounds it (round 4, rocprofv3 counters of the kernel alone at 16 rows x 24 heads x 600 positions,
// profiles/archive/r04_session25_prefill_attn.log): not the fragment loads -- halving them (two query blocks per wave) bought 9 % and
// prefetching the next key block nothing -- but the vector ALU: 360 VALU instructions per key block and query block (8
// libm expf, a hand-rolled bf16 rounding per probability, 64-bit address arithmetic per load, the 16 multiplies of the
// running output by alpha) against 8 MFMAs, the VALU 60 % busy.  Hence: scores kept in log2 units (scale * log2 e folded
// into one multiply, v_exp_f32 directly), v_cvt_pk_bf16_f32 for the probabilities, SGPR base + 32-bit lane offset
// addressing (the V^T offsets are loop invariant), and the output is rescaled only in the blocks where some query's
// running maximum moved: 175 -> 108 us (VALU instructions 54.7 M -> 26.5 M; what is left waits for the fragment loads
// half of the time at two waves per SIMD, and requesting the next key block one iteration early did not change it:
// r04_session26 / 27).
// QB: 16-query blocks per wave (consecutive positions): a K / V^T fragment is fetched once and multiplied QB times.
typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2_hw(float lo, float hi) {   // round to nearest even, like pack_bf16x2
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, hw_bf16x2));
}

template <typename KT, int HD, int QB>
__global__ __launch_bounds__(256) void attn_prefill_kernel(const PrefillAttnArgs p) {
    constexpr int E = 16 / (int)sizeof(KT);       // elements per lane of a fragment: 8 (bf16) / 4 (f32)
    constexpr int KTILE = 4 * E;                  // k columns per fragment: 32 / 16
    constexpr int TT = KTILE / 16;                // 16-key score tiles per probability fragment: 2 / 1
    constexpr int NKD = (HD + KTILE - 1) / KTILE; // fragments along the head dimension
    constexpr int ND = (HD + 15) / 16;            // 16-row tiles of O^T
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nl = lane & 15, kg = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = (blockIdx.x * 4 + wave) * 16 * QB;   // first query position (index inside this call) of the wave
    if (q0 >= p.npos) return;
    const int pos0 = *p.pos;
    const int d = p.H * HD;
    const char* __restrict__ kc = reinterpret_cast<const char*>(reinterpret_cast<const KT*>(p.k_cache) + ((size_t)b * p.H + h) * p.Tcap * HD);
    const char* __restrict__ vt = reinterpret_cast<const char*>(reinterpret_cast<const KT*>(p.vt) + ((size_t)b * p.H + h) * HD * p.vt_tcap);

    // queries as B fragments: lane (kg, n) holds Q[q0 + g * 16 + n][kc * KTILE + kg * E .. + E), in the cache's element type
    u32x4 qf[QB][NKD];
#pragma unroll
    for (int g = 0; g < QB; ++g) {
        const int qi = min(q0 + g * 16 + nl, p.npos - 1);
        const float* qrow = p.q + ((size_t)b * p.npos_pad + qi) * d + h * HD;
#pragma unroll
        for (int c = 0; c < NKD; ++c) {
            float t[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int f = c * KTILE + kg * E + e;
                t[e] = f < HD ? qrow[min(f, HD - 1)] : 0.f;
            }
            if (sizeof(KT) == 2) {
                qf[g][c] = u32x4{pack_bf16x2(t[0], t[1]), pack_bf16x2(t[2 % E], t[3 % E]), pack_bf16x2(t[4 % E], t[5 % E]),
                                 pack_bf16x2(t[6 % E], t[7 % E])};
            } else {
                qf[g][c] = u32x4{__float_as_uint(t[0]), __float_as_uint(t[1]), __float_as_uint(t[2 % E]), __float_as_uint(t[3 % E])};
            }
        }
    }
    // causal: keys <= the query's absolute position are visible; non-causal (cross-attention): keys [0, klen) of the row
    const int klen = p.causal ? 0 : (p.klen_rows != nullptr ? max(1, min(p.klen_rows[b], p.klen)) : p.klen);
    const int tq_last = p.causal ? pos0 + min(q0 + 16 * QB - 1, p.npos - 1) : klen - 1;
    int t_first = 0;
    if (p.causal && p.past_context > 0) t_first = max(0, pos0 + q0 - p.past_context) / KTILE * KTILE;
    const float sl2 = p.scale * 1.4426950408889634f;   // scores in log2 units: softmax by v_exp_f32
    float m_run[QB], l_run[QB];
    f32x4 o[QB][ND];
#pragma unroll
    for (int g = 0; g < QB; ++g) {
        m_run[g] = -INFINITY; l_run[g] = 0.f;
#pragma unroll
        for (int i = 0; i < ND; ++i) o[g][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // byte offsets of this lane inside a key block: K rows t0 + tt * 16 + nl (clamped to the cache: the scores of such keys
    // are masked), 16 bytes at head feature kg * E; V^T rows d = i * 16 + nl, keys t0 + kg * 4 .. (+ 16): loop invariant
    unsigned vlane[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) vlane[i] = (unsigned)((min(i * 16 + nl, HD - 1) * p.vt_tcap + kg * 4) * (int)sizeof(KT));
    const unsigned klane = (unsigned)(kg * 16);

    // K fragments (A operand of S^T) and V^T fragments (A operand of O^T) of a key block: all requested up front
    auto fetch = [&](int t0, u32x4 (&kf)[TT][NKD], u32x4 (&vf)[ND]) {
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const unsigned off = (unsigned)min(t0 + tt * 16 + nl, p.Tcap - 1) * (unsigned)(HD * sizeof(KT)) + klane;
#pragma unroll
            for (int c = 0; c < NKD; ++c)
                kf[tt][c] = c * KTILE + kg * E < HD ? *reinterpret_cast<const u32x4*>(kc + off + c * 64) : u32x4{0u, 0u, 0u, 0u};
        }
        const char* vb = vt + (size_t)t0 * sizeof(KT);   // wave-uniform
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            if (sizeof(KT) == 2) {
                const uint2 lo = *reinterpret_cast<const uint2*>(vb + vlane[i]);
                const uint2 hi = *reinterpret_cast<const uint2*>(vb + vlane[i] + 32);
                vf[i] = u32x4{lo.x, lo.y, hi.x, hi.y};
            } else {
                vf[i] = *reinterpret_cast<const u32x4*>(vb + vlane[i]);
            }
        }
    };
    for (int t0 = t_first; t0 <= tq_last; t0 += KTILE) {
        u32x4 kf[TT][NKD], vf[ND];
        fetch(t0, kf, vf);
#pragma unroll
        for (int g = 0; g < QB; ++g) {
            const int tq = p.causal ? pos0 + q0 + g * 16 + nl : klen - 1;
            // scores of this lane's query against keys t0 + tt * 16 + kg * 4 + r
            f32x4 s[TT];
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
                s[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < NKD; ++c) big_mma(kf[tt][c], qf[g][c], s[tt], KT());
            }
            float cmax = -INFINITY;
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = t0 + tt * 16 + kg * 4 + r;
                    const bool vis = t <= tq && (!p.causal || p.past_context <= 0 || t >= tq - p.past_context);
                    s[tt][r] = vis ? s[tt][r] * sl2 : -INFINITY;
                    cmax = fmaxf(cmax, s[tt][r]);
                }
            cmax = fmaxf(cmax, __shfl_xor(cmax, 16, 64));
            cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
            const float m_new = fmaxf(m_run[g], cmax);
            // a query with no visible key so far (its window starts in a later block) has m = -inf: subtract 0 instead
            // (every exponent is then 2^-inf = 0, and its running sums are still 0)
            const float m_use = m_new == -INFINITY ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run[g] - m_use);
            float psum = 0.f, pr[TT * 4];
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(s[tt][r] - m_use);
                    pr[tt * 4 + r] = e;
                    psum += e;
                }
            l_run[g] = l_run[g] * alpha + psum;
            m_run[g] = m_new;
            u32x4 pf;
            if (sizeof(KT) == 2) {
                pf = u32x4{pack_bf16x2_hw(pr[0], pr[1]), pack_bf16x2_hw(pr[2], pr[3]), pack_bf16x2_hw(pr[4 % (TT * 4)], pr[5 % (TT * 4)]),
                           pack_bf16x2_hw(pr[6 % (TT * 4)], pr[7 % (TT * 4)])};
            } else {
                pf = u32x4{__float_as_uint(pr[0]), __float_as_uint(pr[1]), __float_as_uint(pr[2]), __float_as_uint(pr[3])};
            }
            if (__ballot(alpha != 1.0f) != 0ull) {   // some query's maximum moved in this block (rare after the first ones)
#pragma unroll
                for (int i = 0; i < ND; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[g][i][r] *= alpha;
            }
#pragma unroll
            for (int i = 0; i < ND; ++i) big_mma(vf[i], pf, o[g][i], KT());
        }
    }
    // O^T: lane (kg, n = q) holds head features i * 16 + kg * 4 + r -> 4 consecutive columns of the tiled output row
#pragma unroll
    for (int g = 0; g < QB; ++g) {
        float l = l_run[g];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const int q = q0 + g * 16 + nl;
        if (q >= p.npos) continue;
        const int row = b * p.npos_pad + q;
        const float inv = 1.0f / l;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int f = i * 16 + kg * 4;
            if (f >= HD) continue;
            const int col = h * HD + f;
            if (p.out_bf16) {
                bf16_t* dst = reinterpret_cast<bf16_t*>(p.out) + tiled_index<bf16_t>(row, col, p.out_rbs);
                *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(o[g][i][0] * inv, o[g][i][1] * inv), pack_bf16x2(o[g][i][2] * inv, o[g][i][3] * inv));
            } else {
                float* dst = reinterpret_cast<float*>(p.out) + tiled_index<float>(row, col, p.out_rbs);
                *reinterpret_cast<float4*>(dst) = make_float4(o[g][i][0] * inv, o[g][i][1] * inv, o[g][i][2] * inv, o[g][i][3] * inv);
            }
        }
    }
}

int acmi_launch_prefill_attn(PrefillAttnArgs& a, int kvdtype, int hd, int Beff, hipStream_t st) {
    ACMI_REQUIRE(a.npos > 0 && a.npos_pad >= a.npos && a.npos_pad % 16 == 0, "acmi_attn_prefill: bad npos=%d npos_pad=%d", a.npos, a.npos_pad);
    ACMI_REQUIRE(a.vt_tcap % 32 == 0 && a.vt_tcap > 0, "acmi_attn_prefill: vt_tcap=%d must be a positive multiple of 32", a.vt_tcap);
    ACMI_REQUIRE(a.causal || (a.klen > 0 && a.klen <= a.Tcap && a.klen <= a.vt_tcap), "acmi_attn_prefill: klen=%d outside the caches", a.klen);
    ACMI_REQUIRE(hd % 4 == 0, "acmi_attn_prefill: head dim %d", hd);
    a.scale = 1.0f / sqrtf((float)hd);
    // ACMI_PFA_QB = 1: one query block per wave everywhere (same-box A/B)
    static const int qb = getenv("ACMI_PFA_QB") != nullptr ? atoi(getenv("ACMI_PFA_QB")) : 2;
    const int QBv = (qb == 2 && a.npos > 64 && hd <= 64) ? 2 : 1;   // hd 128: two query blocks do not fit the registers
    dim3 grid((a.npos + 64 * QBv - 1) / (64 * QBv), a.H, Beff), block(256);
#define ACMI_PFA_LAUNCH(KTv, HDv)                                                                                       \
    if (QBv == 2) hipLaunchKernelGGL((attn_prefill_kernel<KTv, HDv, 2>), grid, block, 0, st, a);                        \
    else hipLaunchKernelGGL((attn_prefill_kernel<KTv, HDv, 1>), grid, block, 0, st, a);
#define ACMI_PFA_CASE(HDv)                                                                                              \
    case HDv:                                                                                                           \
        if (kvdtype == ACMI_BF16) { ACMI_PFA_LAUNCH(bf16_t, HDv) } else { ACMI_PFA_LAUNCH(float, HDv) }                 \
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
#undef ACMI_PFA_LAUNCH
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
