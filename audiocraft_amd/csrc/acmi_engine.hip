// The tail of a MusicGen decode layer -- cross-attention out projection -> linear1 (+ norm2, GELU) -> linear2 -- as ONE
// persistent launch for gfx950 (CDNA4, wave64): acmi_ffn_engine (include/acmi.h).
// Reference semantics: audiocraft/modules/transformer.py:344-361 (cross-attention out_proj), :563-572 (the cross-attention
// and feed-forward blocks of StreamingTransformerLayer.forward), :54-67 (LayerNorm).
//
// Why: as three launches (acmi_gemm.hip) this sub-chain costs 4.9 + 6.3 + 7.3 us per layer (rocprofv3, MusicGen-medium, 16 rows)
// for 42.5 MB of weights, and the in-kernel timelines show HBM idle for ~3.3 us of every launch: the
// boundary, the first round trip, the epilogue.  A launch cannot request its weights before its predecessor has finished.
// Here the weights of ALL THREE matrices are requested from the first instruction on: every compute wave streams its slice
// (its K range of the workgroup's output features) through a private ring in LDS with LDS-DMA (`global_load_lds_dwordx4 nt`:
// no registers in flight), so the two inner dependency edges are crossed with the next GEMM's weights already on the CU.
//
// RESULT (round 5, MI355X, scripts/engine_lab.py, profiles/r05_engine_*): parity-green, bit-reproducible, and SLOWER than the
// launches -- 21.9 us per layer against 18.1 us at d = 1536 (1.21 x; d = 1024: 1.23 x, d = 2048: 1.28 x).  The prefetch credit is
// real (the FFN1 stage shrinks from 5.8 to 3.6 us, the FFN2 stage from 6.5 to 2.5 us) but an in-launch all-to-all edge costs
// ~4.2 us here (2.4 us waiting for the slowest of 192 producers, 1.8 us flag propagation + poll) plus ~1 us to pull the fresh
// activation, against 1.3 us for a kernel boundary; DESIGN.md section 5.8 has the timeline.  acmi_lm_step therefore keeps the
// three launches; this file stays as an exported, tested op and as the kernel behind that evidence.
//
// Work split (NWG = d / 8 workgroups, one per CU, all co-resident; NW compute waves + 1 control wave each):
//   op0  x2 = x1 + att W0^T          workgroup j owns features [8j, 8j + 8)          (half-tile order, K = d)
//   op1  h  = gelu(LN(x2) W1'^T)     workgroup j owns hidden features [32j, 32j + 32) = K tile j of h (tiled order, K = d)
//   op2  x3 = x2 + h W2^T            workgroup j owns features [8j, 8j + 8)          (half-tile order, K = 4d)
//   compute wave w: K tiles [w KF, (w + 1) KF) of d (op0, op1), [4 w KF, 4 (w + 1) KF) of the hidden width (op2);
//   its weight stream is 4.5 KF fragments of 1 KB: op0 KF / 2 units, op1 2 KF fragments, op2 2 KF units.
//   control wave: sums the waves' partial tiles in wave order (deterministic), applies the epilogues (the same arithmetic as
//   tl_epilogue of acmi_gemm.hip: folded LayerNorm with the row statistics from the fragments, exact GELU, residual), publishes
//   x2's / h's fragments with write-through (sc1) stores, drains them, sets the workgroup's flag; then polls the flags of all
//   producers (relaxed agent-scope loads, s_sleep in between, bounded by a timeout -> err word).
// The control wave issues no LDS-DMA: vmcnt retires in order, so a wave with weight requests in flight cannot drain its
// hand-off stores without also waiting for those.  Compute waves cross the edges at workgroup barriers (raw s_barrier: no
// vmcnt(0) fence, the ring keeps filling while they wait).
#include "acmi_lm_internal.h"

#include <stdlib.h>
#include <type_traits>

struct EngArgs {
    const u32x4* w0; const u32x4* w1; const u32x4* w2;
    const float* b0; const float* b1; const float* cs1; const float* b2;
    const u32x4* a0; float* x;
    u32x4* xt_mid; u32x4* xt_out;
    u32x4* hid;
    const float* shift;
    unsigned char* flags; unsigned char* flags_next; unsigned* err;
    unsigned long long* trace;
    int M, d, nwg; float eps, inv_d;
    int acq, chunk, g_epi, sleep;
};

#define ENG_NSTAMP 16
#define ENG_TIMEOUT_TICKS 5000000ull   // 50 ms of the 100 MHz s_memrealtime clock
// Flag layout of a set: ENG_NREP replicas x 2 edges x 1 KB; workgroup j's byte flag of edge e in replica r sits at
// (r * 2 + e) * 1024 + j.  A producer sets its flag in EVERY replica (one store instruction, one lane per replica); consumer j
// polls replica j % ENG_NREP only.  One array polled by all d / 8 CUs is a hot spot: every poll and every flag store of the
// chip queues at ONE memory channel (v2 of this kernel: the slowest workgroup's flag store landed 4.5 us after the median one).
#define ENG_NREP 8
#define ENG_FLAG_BYTES (ENG_NREP * 2 * 1024)

__device__ __forceinline__ float eng_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

__device__ __forceinline__ f32x4 eng_mma(const u32x4& a, const u32x4& b, const f32x4& acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}

// one 1 KB fragment global -> LDS without registers: lane l's 16 bytes at gbase + voff go to LDS byte lds_dst + 16 l.
// M0 (the LDS destination base) is compiler-reserved: saved and restored inside the statement.
__device__ __forceinline__ void eng_dma(const u32x4* gbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %3 nt\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(gbase) : "memory");
}

// workgroup barrier WITHOUT the vmcnt(0) fence of __syncthreads (LDS traffic of this wave is complete: lgkmcnt(0))
__device__ __forceinline__ void eng_bar() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// the compiler must have these registers' loads back here (it inserts the exact vmcnt itself), and nothing that reads LDS or
// memory moves above this point
template <int N>
__device__ __forceinline__ void eng_pin(u32x4 (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(v[i]) :: "memory");
}

// words of the workgroup's control block in LDS (written by the control wave, read by the compute waves)
enum { ENG_GRANT = 0, ENG_EDGE = 1 };
__device__ __forceinline__ int eng_lds_get(const unsigned* p) {
    return __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ void eng_lds_put(unsigned* p, int v) {
    __hip_atomic_store(p, (unsigned)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// NW compute waves, KF = K tiles of d per compute wave, RF = ring fragments (1 KB) per compute wave, C2 = chunks op2's
// activation is fetched in (register budget), SC1 = consumers read published fragments with agent-scope (sc1) loads
//
// The weight stream is METERED.  A CU's vector-memory path is a FIFO: a flag poll, an activation load or a hand-off store
// issued behind a burst of weight requests waits for that burst (v1 of this kernel requested a wave's whole ring at once and
// measured 4.2 us from "own flag stored" to "all flags seen" on every edge -- profiles/r05_engine_v1_burst_dma_*).  So the
// compute waves only request what the control wave has GRANTED (a cumulative fragment count per wave, in LDS): the weights of
// op0 at entry, then `chunk` fragments per wave right after every poll has been issued (the poll is ahead of them in the FIFO),
// and whatever an op still misses once its edge has resolved.
template <int NW, int KF, int RF, int C2, bool SC1>
__global__ __launch_bounds__((NW + 1) * 64) void ffn_engine_kernel(const EngArgs p) {
    constexpr int UF0 = KF / 2;                 // op0: weight units per wave (a unit = 8 features x 2 K tiles)
    constexpr int S1 = UF0, S2 = UF0 + 2 * KF, TOTAL = UF0 + 4 * KF;   // the wave's weight stream: [0, S1) op0, [S1, S2) op1, [S2, TOTAL) op2
    constexpr int NKC = NW * KF;                // K tiles of d
    static_assert(KF % 2 == 0 && RF >= 2 * KF && RF >= UF0 && (4 * KF) % C2 == 0 && ((4 * KF) / C2) % 2 == 0, "engine geometry");
    static_assert(512 % (NW * 64) == 0, "op1's epilogue: 512 outputs over the compute waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0: control wave, 1 .. NW: compute waves
    const int j = blockIdx.x;
    float* const red = reinterpret_cast<float*>(smem + (size_t)NW * RF * 1024);          // [NW][4 tiles][256]
    unsigned short* const stage = reinterpret_cast<unsigned short*>(smem + (size_t)NW * RF * 1024 + (size_t)NW * 4096);   // 1 KB
    unsigned* const ctl = reinterpret_cast<unsigned*>(smem + (size_t)NW * RF * 1024 + (size_t)NW * 4096 + 1024);           // 64 B
    unsigned long long ts[ENG_NSTAMP];
#pragma unroll
    for (int i = 0; i < ENG_NSTAMP; ++i) ts[i] = 0;
#define ENG_TS(i) do { if (p.trace != nullptr) ts[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
    ENG_TS(0);
    if (threadIdx.x == 0) { ctl[ENG_GRANT] = S1; ctl[ENG_EDGE] = 0; }
    eng_bar();                                       // B0: the control block is initialised

    if (wv != 0) {
        // ------------------------------------------------------------------------------------------ compute wave
        const int w = wv - 1;
        const u32x4* ring = reinterpret_cast<const u32x4*>(smem) + (size_t)w * RF * 64 + lane;
        const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem + (unsigned)(w * RF * 1024);
        const unsigned voff = (unsigned)lane * 16u;
        const u32x4* g0 = p.w0 + ((size_t)j * (NKC / 2) + (size_t)w * UF0) * 64;
        const u32x4* g1a = p.w1 + ((size_t)(2 * j) * NKC + (size_t)w * KF) * 64;
        const u32x4* g1b = g1a + (size_t)NKC * 64;
        const u32x4* g2 = p.w2 + ((size_t)j * (2 * NKC) + (size_t)w * 2 * KF) * 64;
        int issued = 0, slot = 0, freed = 0;         // wave-uniform: stream fragments requested / ring slot of the next one / consumed
        auto pump = [&](int limit) __attribute__((always_inline)) {                 // request stream fragments up to `limit`, as far as the ring has room
            const int lim = min(min(limit, freed + RF), TOTAL);
            while (issued < lim) {
                const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(ring_lds + (unsigned)(slot * 1024)));
                if (issued < S1) eng_dma(g0, voff + (unsigned)(issued * 1024), dst);
                else if (issued < S1 + KF) eng_dma(g1a, voff + (unsigned)((issued - S1) * 1024), dst);
                else if (issued < S2) eng_dma(g1b, voff + (unsigned)((issued - S1 - KF) * 1024), dst);
                else eng_dma(g2, voff + (unsigned)((issued - S2) * 1024), dst);
                ++issued;
                slot = slot + 1 == RF ? 0 : slot + 1;
            }
        };
        auto wait_edge = [&](int k) __attribute__((always_inline)) {                // until edge k has resolved: request what the control wave grants
            for (;;) {
                pump(eng_lds_get(ctl + ENG_GRANT));
                if (eng_lds_get(ctl + ENG_EDGE) >= k) break;
                __builtin_amdgcn_s_sleep(1);
            }
        };
        // op1's epilogue is shared by the compute waves: 512 / NW outputs each; this lane's: rows rw + (lane >> 5) (+ 2 per
        // further output), hidden feature 32 j + (lane & 31)
        constexpr int EO = 512 / (NW * 64);          // outputs per lane
        const float b1v = p.b1 != nullptr ? p.b1[32 * j + (lane & 31)] : 0.f;
        const float c1v = p.cs1[32 * j + (lane & 31)];
        // ---- op0: its weights, then the activation (written by the previous launch)
        pump(S1);
        u32x4 av[KF];
        {
            const u32x4* a0 = p.a0 + (size_t)(w * KF) * 64 + lane;
#pragma unroll
            for (int i = 0; i < KF; ++i) av[i] = a0[i * 64];
        }
        eng_pin(av);              // the activation is back, hence (in-order return) op0's weights have landed in LDS
        ENG_TS(1);
        {
            f32x4 ce = {0.f, 0.f, 0.f, 0.f}, co = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < UF0; ++u) {
                const u32x4 b = ring[(u % RF) * 64];
                ce = eng_mma(av[2 * u], b, ce);
                co = eng_mma(av[2 * u + 1], b, co);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(w * 4 + 0) * 256 + lane * 4 + r] = ce[r] + dpp_f32<0x128>(co[r]);
        }
        ENG_TS(2);
        eng_bar();                                   // B1: op0's partial tiles are in LDS
        freed = S1;
        wait_edge(1);                                // x2 is complete (every producer's flag seen by the control wave)
        ts[9] = (unsigned long long)issued;
        pump(S2);                                    // what op1 still misses
        ENG_TS(3);
        // ---- op1
        {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.xt_mid, 0, -1, 0x00020000);
#pragma unroll
            for (int i = 0; i < KF; ++i) {
                const unsigned off = (unsigned)(((w * KF + i) * 64 + lane) * 16);
                if (SC1) av[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
                else av[i] = p.xt_mid[(w * KF + i) * 64 + lane];
            }
        }
        eng_pin(av);              // every older request of this wave -- all of op1's weights -- has landed as well
        ENG_TS(4);
        {
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, cs = c0, cg = c0;
            const u32x4 ones = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
#pragma unroll
            for (int k = 0; k < KF; ++k) {
                const u32x4 b0 = ring[((S1 + k) % RF) * 64], b1 = ring[((S1 + KF + k) % RF) * 64];
                c0 = eng_mma(av[k], b0, c0);
                c1 = eng_mma(av[k], b1, c1);
                cs = eng_mma(av[k], ones, cs);        // row sums
                cg = eng_mma(av[k], av[k], cg);       // Gram matrix: its diagonal = the rows' sums of squares
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                red[(w * 4 + 0) * 256 + lane * 4 + r] = c0[r];
                red[(w * 4 + 1) * 256 + lane * 4 + r] = c1[r];
                red[(w * 4 + 2) * 256 + lane * 4 + r] = cs[r];
                red[(w * 4 + 3) * 256 + lane * 4 + r] = cg[r];
            }
        }
        ENG_TS(5);
        eng_bar();                                   // B3: op1's partial tiles are in LDS
        freed = S2;
        {   // epilogue of op1: folded LayerNorm (statistics from the fragments), bias, exact GELU -> bf16 in `stage`
            const float rk = p.inv_d;
#pragma unroll
            for (int i = 0; i < EO; ++i) {
                const int m = w * (2 * EO) + 2 * i + (lane >> 5), f = lane & 31, t = f >> 4, n = f & 15;
                const int idx = ((((m >> 2) * 16) + n) << 2) + (m & 3);
                const int idg = ((((m >> 2) * 16) + m) << 2) + (m & 3);
                float v = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) {
                    v += red[(ww * 4 + t) * 256 + idx];
                    s1 += red[(ww * 4 + 2) * 256 + idx];
                    s2 += red[(ww * 4 + 3) * 256 + idg];
                }
                const float mean_s = s1 * rk;
                const float rstd = __builtin_amdgcn_rsqf(fmaxf(s2 * rk - mean_s * mean_s, 0.f) + p.eps);
                v = rstd * (v - mean_s * c1v);
                v += b1v;
                v = eng_gelu(v);
                stage[m * 32 + f] = f32_to_bf16(v);
            }
        }
        eng_bar();                                   // B3b: h's fragment is staged
        wait_edge(2);                                // h is complete
        ts[10] = (unsigned long long)issued;
        pump(TOTAL);
        ENG_TS(6);
        // ---- op2
        {
            constexpr int AF = 4 * KF / C2;          // activation fragments per chunk
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.hid, 0, -1, 0x00020000);
            f32x4 ce = {0.f, 0.f, 0.f, 0.f}, co = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < C2; ++c) {
                u32x4 ah[AF];
#pragma unroll
                for (int i = 0; i < AF; ++i) {
                    const int kt = w * 4 * KF + c * AF + i;
                    if (SC1) ah[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)((kt * 64 + lane) * 16), 0, 16);
                    else ah[i] = p.hid[kt * 64 + lane];
                }
                eng_pin(ah);
                if (c == 0) ENG_TS(7);
#pragma unroll
                for (int u = 0; u < AF / 2; ++u) {
                    const u32x4 b = ring[((S2 + c * (AF / 2) + u) % RF) * 64];
                    ce = eng_mma(ah[2 * u], b, ce);
                    co = eng_mma(ah[2 * u + 1], b, co);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(w * 4 + 0) * 256 + lane * 4 + r] = ce[r] + dpp_f32<0x128>(co[r]);
        }
        ENG_TS(8);
        eng_bar();                                   // B5
        if (p.trace != nullptr && w == 0 && lane == 0) {
            unsigned long long* dst = p.trace + ((size_t)j * 2 + 1) * ENG_NSTAMP;
#pragma unroll
            for (int i = 0; i < ENG_NSTAMP; ++i) dst[i] = ts[i];
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------- control wave
    const int M = p.M, d = p.d, nwg = p.nwg;
    // half-tile epilogues (op0, op2): output e = lane + 64 i, i < 2: row e >> 3, feature 8 j + (e & 7)
    float res[2], b0v[2], b2v[2], shv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = lane + 64 * i, m = min(e >> 3, M - 1), f = 8 * j + (e & 7);
        res[i] = p.x[(size_t)m * d + f];
        b0v[i] = p.b0 != nullptr ? p.b0[f] : 0.f;
        b2v[i] = p.b2 != nullptr ? p.b2[f] : 0.f;
        shv[i] = p.shift != nullptr ? p.shift[m] : 0.f;
    }
    if (lane < 2 * ENG_NREP) __hip_atomic_store(p.flags_next + lane * 1024 + j, (unsigned char)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned char* const my_flags = p.flags + (j & (ENG_NREP - 1)) * 2048;   // the replica this workgroup polls
    const __amdgpu_buffer_rsrc_t rs_mid = __builtin_amdgcn_make_buffer_rsrc(p.xt_mid, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_hid = __builtin_amdgcn_make_buffer_rsrc(p.hid, 0, -1, 0x00020000);
    int grant = S1, npoll = 0;

    // all nwg byte flags == 1?  One dword (4 flags) per lane; after every poll's loads are on their way the compute waves
    // are granted `chunk` more fragments each (behind the poll in the CU's memory FIFO)
    auto poll = [&](const unsigned char* fl, unsigned code) __attribute__((always_inline)) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned* fw = reinterpret_cast<const unsigned*>(fl);
        for (;;) {
            unsigned v = 0x01010101u;
            if (lane * 4 < nwg) v = __hip_atomic_load(fw + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_sched_barrier(0);
            grant = min(grant + p.chunk, TOTAL);
            eng_lds_put(ctl + ENG_GRANT, grant);
            __builtin_amdgcn_sched_barrier(0);
            ++npoll;
            if (__all(v == 0x01010101u)) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > ENG_TIMEOUT_TICKS) {
                if (lane == 0) __hip_atomic_fetch_or(p.err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            for (int q = 0; q < p.sleep; ++q) __builtin_amdgcn_s_sleep(4);   // 4 x 64 clocks per unit
        }
        if (p.acq == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    };

    // x2 / x3 of this lane's two outputs; half-tile epilogue shared by op0 and op2
    float xv[2];
    auto half_epilogue = [&](const float (&resv)[2], const float (&bv)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = lane + 64 * i, m = e >> 3, c = e & 7;
            const int idx = (((m >> 2) * 16 + c) << 2) + (m & 3);
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += red[(w * 4 + 0) * 256 + idx];
            v += bv[i];
            v += resv[i];
            xv[i] = v;
            stage[m * 8 + c] = f32_to_bf16(v - shv[i]);
        }
    };

    eng_bar();                                       // B1
    ENG_TS(1);
    // ---- epilogue of op0: x2 (kept in registers: it is op2's residual), its fragments published
    half_epilogue(res, b0v);
    {
        const int m = lane & 15;
        const u32x4 q = *reinterpret_cast<const u32x4*>(stage + m * 8);
        // K tile j >> 2, lane group j & 3 of x2's fragments
        if (lane < 16 && m < M)
            __builtin_amdgcn_raw_buffer_store_b128(q, rs_mid, (unsigned)((((j >> 2) * 64) + (j & 3) * 16 + m) * 16), 0, 16 /* sc1 */);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane < ENG_NREP) __hip_atomic_store(p.flags + lane * 2048 + j, (unsigned char)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ENG_TS(2);
    poll(my_flags, 1u);
    grant = max(grant, S2);
    eng_lds_put(ctl + ENG_GRANT, grant);
    eng_lds_put(ctl + ENG_EDGE, 1);
    ENG_TS(3);
    ts[9] = (unsigned long long)npoll;
    eng_bar();                                       // B3: op1's partial tiles
    if (p.g_epi > 0) { grant = min(grant + p.g_epi, TOTAL); eng_lds_put(ctl + ENG_GRANT, grant); }
    eng_bar();                                       // B3b: K tile j of h is staged as bf16
    ENG_TS(4);
    {
        const int m = lane & 15, kg = lane >> 4;
        const u32x4 q4 = *reinterpret_cast<const u32x4*>(stage + m * 32 + kg * 8);
        if (m < M) __builtin_amdgcn_raw_buffer_store_b128(q4, rs_hid, (unsigned)((j * 64 + lane) * 16), 0, 16 /* sc1 */);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane < ENG_NREP) __hip_atomic_store(p.flags + lane * 2048 + 1024 + j, (unsigned char)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ENG_TS(5);
    npoll = 0;
    poll(my_flags + 1024, 2u);
    grant = TOTAL;
    eng_lds_put(ctl + ENG_GRANT, grant);
    eng_lds_put(ctl + ENG_EDGE, 2);
    ENG_TS(6);
    ts[10] = (unsigned long long)npoll;
    eng_bar();                                       // B5: op2's partial tiles
    ENG_TS(7);
    // ---- epilogue of op2: x3 = x2 + h W2^T + b2 -> x (f32) and the raw fragments for the next layer (plain stores: the
    // kernel boundary publishes them)
    {
        const float x2v[2] = {xv[0], xv[1]};
        half_epilogue(x2v, b2v);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = lane + 64 * i, m = e >> 3, f = 8 * j + (e & 7);
            if (m < M) p.x[(size_t)m * d + f] = xv[i];
        }
        const int m = lane & 15;
        const u32x4 q = *reinterpret_cast<const u32x4*>(stage + m * 8);
        if (lane < 16 && m < M) p.xt_out[((j >> 2) * 64) + (j & 3) * 16 + m] = q;
    }
    ENG_TS(8);
    if (p.trace != nullptr && lane == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        ts[15] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
        unsigned long long* dst = p.trace + ((size_t)j * 2) * ENG_NSTAMP;
#pragma unroll
        for (int i = 0; i < ENG_NSTAMP; ++i) dst[i] = ts[i];
    }
#undef ENG_TS
}

template <int NW, int KF, int RF, int C2, bool SC1>
static int eng_launch_k(const EngArgs& a, hipStream_t st) {
    constexpr size_t lds = (size_t)NW * RF * 1024 + (size_t)NW * 4096 + 1024 + 64;
    static_assert(lds <= 160 * 1024, "engine LDS budget");
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_engine_kernel<NW, KF, RF, C2, SC1>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;   // once, thread safe
    if (!attr_ok) {
        acmi_set_error("acmi_ffn_engine: cannot raise the dynamic LDS limit");
        return ACMI_ELAUNCH;
    }
    hipLaunchKernelGGL((ffn_engine_kernel<NW, KF, RF, C2, SC1>), dim3(a.nwg), dim3((NW + 1) * 64), lds, st, a);
    return acmi_check_launch("ffn_engine_kernel");
}

template <int NW, int KF, int RF, int C2>
static int eng_launch(const EngArgs& a, hipStream_t st) {
    return a.acq == 2 ? eng_launch_k<NW, KF, RF, C2, true>(a, st) : eng_launch_k<NW, KF, RF, C2, false>(a, st);
}

extern "C" int acmi_ffn_engine_supported(int M, int d, int ffn, int wdtype) {
    return wdtype == ACMI_BF16 && M >= 1 && M <= 16 && ffn == 4 * d && (d == 1024 || d == 1536 || d == 2048) ? 1 : 0;
}

extern "C" int acmi_ffn_engine(const acmi_ffn_engine_desc* c, void* stream) {
    ACMI_REQUIRE(c != nullptr, "acmi_ffn_engine: null descriptor");
    ACMI_REQUIRE(acmi_ffn_engine_supported(c->M, c->d, c->ffn, ACMI_BF16), "acmi_ffn_engine: unsupported geometry M=%d d=%d ffn=%d "
                 "(bf16, M <= 16, ffn == 4 d, d in {1024, 1536, 2048})", c->M, c->d, c->ffn);
    ACMI_REQUIRE(c->w0 && c->w1 && c->w2 && c->cs1 && c->a0 && c->x && c->xt_mid && c->xt_out && c->hidden && c->flags &&
                 c->flags_next && c->err && c->flags != c->flags_next, "acmi_ffn_engine: missing operand");
    ACMI_REQUIRE(c->acq_mode >= 0 && c->acq_mode <= 2, "acmi_ffn_engine: acq_mode %d", c->acq_mode);
    EngArgs a = {};
    a.w0 = (const u32x4*)c->w0; a.w1 = (const u32x4*)c->w1; a.w2 = (const u32x4*)c->w2;
    a.b0 = c->b0; a.b1 = c->b1; a.cs1 = c->cs1; a.b2 = c->b2;
    a.a0 = (const u32x4*)c->a0; a.x = c->x; a.xt_mid = (u32x4*)c->xt_mid; a.xt_out = (u32x4*)c->xt_out; a.hid = (u32x4*)c->hidden;
    a.shift = c->shift; a.flags = (unsigned char*)c->flags; a.flags_next = (unsigned char*)c->flags_next; a.err = c->err; a.trace = (unsigned long long*)c->trace;
    a.M = c->M; a.d = c->d; a.nwg = c->d / 8; a.eps = c->eps; a.inv_d = 1.0f / (float)c->d; a.acq = c->acq_mode;
    static const int env_chunk = getenv("ACMI_ENGINE_CHUNK") ? atoi(getenv("ACMI_ENGINE_CHUNK")) : 0;
    a.chunk = c->dma_chunk > 0 ? c->dma_chunk : (env_chunk > 0 ? env_chunk : 4);
    a.g_epi = c->dma_epi > 0 ? c->dma_epi : 0;
    a.sleep = c->poll_sleep >= 0 ? c->poll_sleep : 0;
    hipStream_t st = (hipStream_t)stream;
    static const int env_nw = getenv("ACMI_ENGINE_NW") ? atoi(getenv("ACMI_ENGINE_NW")) : 0;
    const int nw = c->waves > 0 ? c->waves : (env_nw > 0 ? env_nw : 4);
    const int kt = c->d / 32;
    if (nw == 4) {
        if (kt == 48) return eng_launch<4, 12, 34, 1>(a, st);
        if (kt == 32) return eng_launch<4, 8, 34, 1>(a, st);
        if (kt == 64) return eng_launch<4, 16, 34, 2>(a, st);
    } else if (nw == 8) {
        if (kt == 48) return eng_launch<8, 6, 15, 1>(a, st);
        if (kt == 32) return eng_launch<8, 4, 15, 1>(a, st);
    }
    acmi_set_error("acmi_ffn_engine: no instantiation for %d compute waves at d=%d", nw, c->d);
    return ACMI_EINVAL;
}
