// Lab (dev tool, not product): QKV GEMM -> single-query self-attention of one decode layer as ONE launch with a
// per-(row, head) hand-off, against the two launches acmi_lm_step uses today.
//   sep   : G launch (384 workgroups x 4 waves: 16 output features each of the [6144 x 1536] bf16 matrix, 12 weight + 12
//           activation fragments per wave, MFMA, LDS reduction, q -> f32 scratch, k / v -> bf16 cache row t) then A launch
//           (one workgroup of 4 waves per (row, head): acmi_attn.hip's online-softmax walk over 64-position chunks)
//   fused : ONE launch of 768 workgroups: [0, 384) run G and publish q | k | v as f32 words with write-through stores into
//           a hand-off row whose slots hold a sentinel (the value is its own flag); [384, 768) run A: P chunks of the
//           q-INDEPENDENT K / V stream are requested first, then the workgroup polls its 3 x 64 words, substitutes the new
//           position's k / v (rounded to the cache type: bit-identical arithmetic to `sep`), writes them into the cache and
//           re-arms the slots for the next replay.
// Geometry = MusicGen-medium, 8 prompts (16 CFG rows): d 1536, 24 heads x 64, context t (default 750), 48 layers with their own
// cold weights and caches in one hipGraph.
// Build: hipcc --offload-arch=gfx950 -O3 -o qkv_attn_lab qkv_attn_lab.hip     Run: ./qkv_attn_lab [t] [reps] [layers]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short bf16_t;
typedef bf16_t rawv __attribute__((ext_vector_type(8)));

constexpr int D = 1536, H = 24, HD = 64, ROWS = 16, TCAP = 1504, NF = 4 * D;   // q | k | v | r (x0 part of the cross query)
constexpr int NKC = D / 32;              // K fragments of a 16-feature tile (bf16: 32 columns each)
constexpr int GW = NF / 16;              // G workgroups
constexpr int AW = ROWS * H;             // A workgroups
constexpr unsigned SENT = 0x7fc0deadu;   // quiet NaN no arithmetic produces
constexpr int NST = 8;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float group_sum8(float v) {
    v += dpp_f32<0xB1>(v); v += dpp_f32<0x4E>(v); v += dpp_f32<0x141>(v);
    return v;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, -1, 0x00020000);
}

struct LayerArgs {
    const u32x4* w;      // [GW][NKC][64] weight fragments (18.9 MB)
    const u32x4* a;      // [NKC][64] activation fragments (16 rows x 1536 bf16)
    float* q;            // sep: [ROWS][D] f32
    float* r;            // [ROWS][D] f32 (the fourth block of features: not consumed here)
    unsigned* hand;      // fused: [ROWS][3 D] words, sentinel-armed
    bf16_t* kc; bf16_t* vc;   // [ROWS][H][TCAP][HD]
    float* out;          // [ROWS][D] attention output
    const int* tpos;     // device word: the position the new K / V rows are stored at (= context length before this step)
    unsigned* err;       // [0] poll timeouts
    unsigned long long* ts;   // [768][NST] stamps of workgroup-wave 0, or NULL
    int a_delay;         // fused: ticks of 10 ns the A role waits before its first request (the weight stream goes first)
};

// ------------------------------------------------------------------------------------------ G role
template <bool FUSED>
__device__ __forceinline__ void g_role(const LayerArgs& p, const int wg, float* red /* [4][256] */) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long t0 = 0;
    if (p.ts != nullptr && threadIdx.x == 0) t0 = wall_clock64();
    constexpr int C = NKC / 4;   // 12 fragments per wave
    u32x4 bv[C], av[C];
    const u32x4* wb = p.w + ((size_t)wg * NKC + wave * C) * 64 + lane;
#pragma unroll
    for (int i = 0; i < C; ++i) bv[i] = __builtin_nontemporal_load(wb + i * 64);
    __builtin_amdgcn_sched_barrier(0);
    const u32x4* ab = p.a + (wave * C) * 64 + lane;
#pragma unroll
    for (int i = 0; i < C; ++i) av[i] = ab[i * 64];
    const int tpos = *p.tpos;
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < C; ++i)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[i]), __builtin_bit_cast(bf16x8, bv[i]), acc, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wave * 256 + lane * 4 + e] = acc[e];
    __syncthreads();
    const int e = threadIdx.x, mm = (e >> 4) & 15, nn = e & 15;
    const int idx = (((mm >> 2) * 16 + nn) << 2) + (mm & 3);
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) v += red[w * 256 + idx];
    v *= 0.02f;   // keeps the scores of the random data moderate
    const int fb = __builtin_amdgcn_readfirstlane(wg * 16);
    const int part = (fb >= D) + (fb >= 2 * D) + (fb >= 3 * D);
    const int f = fb - part * D + nn;
    if (part == 3) {
        p.r[(size_t)mm * D + f] = v;
    } else if (FUSED) {
        __hip_atomic_store(p.hand + (size_t)mm * 3 * D + part * D + f, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (part == 0) {
        p.q[(size_t)mm * D + f] = v;
    } else {
        const int h = f >> 6, dd = f & 63;
        (part == 1 ? p.kc : p.vc)[(((size_t)mm * H + h) * TCAP + tpos) * HD + dd] = f2bf(v);
    }
    if (p.ts != nullptr && threadIdx.x == 0) {
        unsigned long long* d = p.ts + (size_t)wg * NST;
        d[0] = t0; d[1] = wall_clock64(); d[4] = 1 + __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
    }
}

// ------------------------------------------------------------------------------------------ prefetch role (no flag, no hand-off)
// Workgroup p of the prefetchers is ASSUMED to run on XCD p % 8 (workgroups are dealt round-robin over the XCDs) like the
// attention workgroup (row b, head h) of the NEXT launch runs on XCD (b H + h) % 8 = h % 8: it requests the first `npf`
// positions of K and V of the (p / 8)-th (row, head) pair whose head is congruent to its XCD, with plain loads (the lines
// allocate in this XCD's L2), and drops the data.
__device__ __forceinline__ void pf_role(const LayerArgs& p, const int pw, const int npf) {
    const int xcd = pw & 7, idx = pw >> 3;             // idx in [0, ROWS * H / 8)
    const int b = idx / (H / 8), h = xcd + 8 * (idx % (H / 8));
    const u32x4* kb = reinterpret_cast<const u32x4*>(p.kc + ((size_t)b * H + h) * TCAP * HD);
    const u32x4* vb = reinterpret_cast<const u32x4*>(p.vc + ((size_t)b * H + h) * TCAP * HD);
    const int tpos = __builtin_amdgcn_readfirstlane(*p.tpos);
    const int lim = min(npf, tpos) * (HD * 2 / 16);      // 16-byte units of the first positions
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (int i = threadIdx.x; i < lim; i += 4 * 256) {
        u32x4 a[4], c[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int ii = min(i + j * 256, lim - 1); a[j] = kb[ii]; c[j] = vb[ii]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc ^= a[j] ^ c[j];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u && tpos < 0) p.err[0] = 1;   // keeps the loads alive
    if (p.ts != nullptr && threadIdx.x == 0) p.ts[(size_t)(GW + AW) * NST + (size_t)pw * 2] = 1 + __builtin_amdgcn_s_getreg((31 << 11) | 20) + 16 * (b * H + h + 1);
}

// ------------------------------------------------------------------------------------------ A role
// P: chunks (64 positions per wave) requested before the query is waited for (register sets).  P = 1 is acmi_attn.hip's
// kernel; P = 2 keeps a second set in flight (processed alternately).
// one 1 KB wave-load global -> LDS without registers: lane l's 16 bytes at gbase + voff go to LDS byte lds_dst + 16 l
__device__ __forceinline__ void lds_dma(const void* gbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %3 nt\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(gbase) : "memory");
}
// KL: the K half of the SECOND round of chunks is staged in LDS (8 KB per wave) before the query is waited for
template <bool FUSED, int P, int NW = 4, bool KL = false>
__device__ __forceinline__ void a_role(const LayerArgs& p, const int wg, float* sm /* [NW][HD] + [NW] + [NW] */, unsigned char* kst = nullptr) {
    constexpr int NI = 8, PPI = 8, CH = NI * PPI, LPP = 8, DPL = 8;
    const int h = wg % H, b = wg / H;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane % LPP, pp = lane / LPP;
    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    if (p.ts != nullptr && threadIdx.x == 0) t0 = wall_clock64();
    if (FUSED && p.a_delay > 0) {
        const unsigned long long s0 = wall_clock64();
        while (wall_clock64() - s0 < (unsigned long long)p.a_delay) __builtin_amdgcn_s_sleep(2);
    }
    const bf16_t* kb = p.kc + ((size_t)b * H + h) * TCAP * HD + c * DPL;
    const bf16_t* vb = p.vc + ((size_t)b * H + h) * TCAP * HD + c * DPL;
    rawv kr[P][NI], vr[P][NI];
    int lim = TCAP;
    auto load_kv = [&](int s, int t0p) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int t = max(min(t0p + i * PPI + pp, lim - 1), 0);
            kr[s][i] = __builtin_nontemporal_load(reinterpret_cast<const rawv*>(kb + (size_t)t * HD));
            vr[s][i] = __builtin_nontemporal_load(reinterpret_cast<const rawv*>(vb + (size_t)t * HD));
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    const int tnew = __builtin_amdgcn_readfirstlane(__builtin_nontemporal_load(p.tpos));   // (first: vmcnt retires in order)
    // speculative: the chunks are requested against the cache capacity, masked once the length is known
#pragma unroll
    for (int s = 0; s < P; ++s) load_kv(s, (wave + NW * s) * CH);
    if (KL) {   // K of round 1 (this wave's second chunk) -> LDS, no registers in flight
        static_assert(!KL || P == 1, "LDS-staged K: one register set");
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(size_t)(__attribute__((address_space(3))) void*)kst + wave * 8192u));
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int t = min((wave + NW) * CH + i * PPI + pp, TCAP - 1);
            lds_dma(p.kc + ((size_t)b * H + h) * TCAP * HD, (unsigned)((t * HD + c * DPL) * 2), dst + i * 1024u);
        }
    }
    const int len = tnew + 1;   // the new position takes part
    lim = FUSED ? max(tnew, 1) : len;   // fused: position tnew is not in the cache yet (it comes through the hand-off)
    if (p.ts != nullptr && threadIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t1 = wall_clock64(); }
    float qv[DPL];
    rawv knew, vnew;
    if (FUSED) {
        // the 3 x 64 words of (row b, head h): lane (c, *) needs dims [8 c, 8 c + 8) of q, k, v
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.hand + (size_t)b * 3 * D + h * HD);
        u32x4 w[6];
        unsigned spins = 0;
        for (;;) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                w[2 * j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (j * D + c * DPL) * 4, 0, 16);
                w[2 * j + 1] = __builtin_amdgcn_raw_buffer_load_b128(rs, (j * D + c * DPL + 4) * 4, 0, 16);
            }
            bool bad = false;
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) bad |= w[j][e] == SENT;
            if (!__any(bad)) break;
            if (++spins > 200000u) { if (lane == 0) atomicAdd(p.err, 1u); break; }
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            qv[e] = __uint_as_float(w[0][e]); qv[4 + e] = __uint_as_float(w[1][e]);
            knew[e] = f2bf(__uint_as_float(w[2][e])); knew[4 + e] = f2bf(__uint_as_float(w[3][e]));
            vnew[e] = f2bf(__uint_as_float(w[4][e])); vnew[4 + e] = f2bf(__uint_as_float(w[5][e]));
        }
        if (wave == 0 && pp == 0) {   // the new rows enter the cache for the positions to come
            *reinterpret_cast<rawv*>(const_cast<bf16_t*>(kb) + (size_t)tnew * HD) = knew;
            *reinterpret_cast<rawv*>(const_cast<bf16_t*>(vb) + (size_t)tnew * HD) = vnew;
        }
    } else {
        const float* q = p.q + (size_t)b * D + h * HD + c * DPL;
#pragma unroll
        for (int e = 0; e < DPL; ++e) qv[e] = q[e];
    }
    if (p.ts != nullptr && threadIdx.x == 0) t2 = wall_clock64();
    const float scale = 0.125f;
    float m = -INFINITY, l = 0.f, o[DPL];
#pragma unroll
    for (int e = 0; e < DPL; ++e) o[e] = 0.f;
    auto process = [&](int s, int t0p) {
        if (t0p + CH > len - (FUSED ? 1 : 0)) {   // wave-uniform: the chunk reaches the end
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int t = t0p + i * PPI + pp;
                if (FUSED && t == tnew) { kr[s][i] = knew; vr[s][i] = vnew; }
                if (t >= len) {
#pragma unroll
                    for (int e = 0; e < DPL; ++e) vr[s][i][e] = (bf16_t)0;
                }
            }
        }
        float sc[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int t = t0p + i * PPI + pp;
            float part = 0.f;
#pragma unroll
            for (int e = 0; e < DPL; ++e) part = fmaf(qv[e], bf2f(kr[s][i][e]), part);
            part = group_sum8(part);
            sc[i] = (t < len) ? part * scale : -INFINITY;
        }
        float cmax = sc[0];
#pragma unroll
        for (int i = 1; i < NI; ++i) cmax = fmaxf(cmax, sc[i]);
        cmax = fmaxf(cmax, dpp_f32<0x128>(cmax));
        cmax = fmaxf(cmax, __shfl_xor(cmax, 16, 64));
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
        const float m_new = fmaxf(m, cmax);
        const float alpha = expf(m - m_new);
        l *= alpha;
#pragma unroll
        for (int e = 0; e < DPL; ++e) o[e] *= alpha;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int t = t0p + i * PPI + pp;
            const float pr = (t < len) ? expf(sc[i] - m_new) : 0.f;
            l += pr;
#pragma unroll
            for (int e = 0; e < DPL; ++e) o[e] = fmaf(pr, bf2f(vr[s][i][e]), o[e]);
        }
        m = m_new;
    };
    // wave w owns chunks w, w + 4, w + 8, ...; register set s holds chunk (w + 4 (s + P j))
    if (!KL) {
        for (int t0p = wave * CH; t0p < len; t0p += NW * P * CH) {
#pragma unroll
            for (int s = 0; s < P; ++s) {
                const int ts_ = t0p + NW * s * CH;
                if (ts_ < len) {
                    process(s, ts_);
                    if (ts_ + NW * P * CH < len) load_kv(s, ts_ + NW * P * CH);
                }
            }
        }
    } else {
        // round 0 from the registers; round 1: V requested now, K read back from LDS; later rounds as usual
        const int r0 = wave * CH, r1 = (wave + NW) * CH;
        if (r0 < len) process(0, r0);
        if (r1 < len) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (nothing is outstanding: the staged K landed before the poll's words)
            const unsigned char* src = kst + wave * 8192 + lane * 16;
#pragma unroll
            for (int i = 0; i < NI; ++i) kr[0][i] = *reinterpret_cast<const rawv*>(src + i * 1024);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int t = max(min(r1 + i * PPI + pp, lim - 1), 0);
                vr[0][i] = __builtin_nontemporal_load(reinterpret_cast<const rawv*>(vb + (size_t)t * HD));
            }
            __builtin_amdgcn_sched_barrier(0);
            process(0, r1);
            for (int t0p = r1 + NW * CH; t0p < len; t0p += NW * CH) { load_kv(0, t0p); process(0, t0p); }
        }
    }
#pragma unroll
    for (int off = LPP; off < 64; off <<= 1) {
        const bool dpp8 = off == 8;
        l += dpp8 ? dpp_f32<0x128>(l) : __shfl_xor(l, off, 64);
#pragma unroll
        for (int e = 0; e < DPL; ++e) o[e] += dpp8 ? dpp_f32<0x128>(o[e]) : __shfl_xor(o[e], off, 64);
    }
    float* sm_o = sm; float* sm_m = sm + NW * HD; float* sm_l = sm_m + NW;
    if (lane < LPP) {
#pragma unroll
        for (int e = 0; e < DPL; ++e) sm_o[wave * HD + c * DPL + e] = o[e];
    }
    if (lane == 0) { sm_m[wave] = m; sm_l[wave] = l; }
    __syncthreads();
    if (threadIdx.x < HD) {
        float M = sm_m[0];
        for (int w = 1; w < NW; ++w) M = fmaxf(M, sm_m[w]);
        float num = 0.f, den = 0.f;
        for (int w = 0; w < NW; ++w) {
            const float f = (sm_m[w] == -INFINITY) ? 0.f : expf(sm_m[w] - M);
            num += f * sm_o[w * HD + threadIdx.x];
            den += f * sm_l[w];
        }
        p.out[(size_t)b * D + h * HD + threadIdx.x] = den > 0.f ? num / den : 0.f;
    }
    if (FUSED && threadIdx.x < 3 * HD) {   // every wave has read its words (barrier above): re-arm for the next replay
        const int j = threadIdx.x / HD, dd = threadIdx.x % HD;
        __hip_atomic_store(p.hand + (size_t)b * 3 * D + j * D + h * HD + dd, SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (p.ts != nullptr && threadIdx.x == 0) {
        unsigned long long* d = p.ts + (size_t)(GW + wg) * NST;
        d[0] = t0; d[1] = t1; d[2] = t2; d[3] = wall_clock64(); d[4] = 1 + __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
}

#ifndef LAB_WPE
#define LAB_WPE 2
#endif
__global__ __launch_bounds__(256) void k_g(const LayerArgs p) {
    __shared__ float red[4 * 256];
    g_role<false>(p, blockIdx.x, red);
}
__global__ __launch_bounds__(256) void k_g_pf(const LayerArgs p, const int npf) {
    __shared__ float red[4 * 256];
    if ((int)blockIdx.x < GW) g_role<false>(p, blockIdx.x, red);
    else pf_role(p, (int)blockIdx.x - GW, npf);
}
template <int P, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW == 8 ? 4 : 2, NW == 8 ? 4 : 4))) void k_a(const LayerArgs p) {
    __shared__ float sm[NW * HD + 2 * NW];
    a_role<false, P, NW>(p, blockIdx.x, sm);
}
template <int P, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_fused(const LayerArgs p) {
    __shared__ float sm[4 * 256];
    if ((int)blockIdx.x < GW) g_role<true>(p, blockIdx.x, sm);
    else a_role<true, P>(p, (int)blockIdx.x - GW, sm);
}
template <int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_fused_kl(const LayerArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[4096 + 4 * 8192];
    if ((int)blockIdx.x < GW) g_role<true>(p, blockIdx.x, reinterpret_cast<float*>(lds));
    else a_role<true, 1, 4, true>(p, (int)blockIdx.x - GW, reinterpret_cast<float*>(lds), lds + 4096);
}
// A role first in the grid (for comparison only: safe while every workgroup is co-resident)
template <int P, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_fused_afirst(const LayerArgs p) {
    __shared__ float sm[4 * 256];
    if ((int)blockIdx.x >= AW) g_role<true>(p, (int)blockIdx.x - AW, sm);
    else a_role<true, P>(p, blockIdx.x, sm);
}

__global__ void k_fill_bf16(bf16_t* p, size_t n, unsigned seed, float amp) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)(i * 2654435761ull) ^ seed;
        x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15;
        p[i] = f2bf(amp * ((float)(x & 0xffffu) / 32768.0f - 1.0f));
    }
}
__global__ void k_fill_u32(unsigned* p, size_t n, unsigned v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void k_set(int* p, int v) { if (threadIdx.x == 0 && blockIdx.x == 0) *p = v; }

// ------------------------------------------------------------------------------------------ host
static hipStream_t s1;
static hipEvent_t ev_a, ev_b;

template <typename F>
static double time_graph(F&& body, int reps) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
    body();
    CK(hipStreamEndCapture(s1, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s1)); CK(hipGraphLaunch(ge, s1));
    CK(hipStreamSynchronize(s1));
    CK(hipEventRecord(ev_a, s1));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s1));
    CK(hipEventRecord(ev_b, s1));
    CK(hipStreamSynchronize(s1));
    float ms = 0; CK(hipEventElapsedTime(&ms, ev_a, ev_b));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return ms * 1e3 / reps;
}

static float bf2f_h(bf16_t v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
    const int t = argc > 1 ? atoi(argv[1]) : 750;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const int L = argc > 3 ? atoi(argv[3]) : 48;
    if (t < 0 || t >= TCAP) { printf("t out of range\n"); return 1; }
    CK(hipSetDevice(0));
    CK(hipStreamCreate(&s1));
    CK(hipEventCreate(&ev_a)); CK(hipEventCreate(&ev_b));
    const size_t wl = (size_t)GW * NKC * 64, al = (size_t)NKC * 64, cl = (size_t)ROWS * H * TCAP * HD;
    u32x4 *w, *a; bf16_t *kc, *vc; float *q, *r, *out_sep, *out_fused; unsigned *hand, *err; int* tpos; unsigned long long* ts;
    CK(hipMalloc(&w, wl * 16 * L)); CK(hipMalloc(&a, al * 16 * L));
    CK(hipMalloc(&kc, cl * 2 * L)); CK(hipMalloc(&vc, cl * 2 * L));
    CK(hipMalloc(&q, (size_t)ROWS * D * 4 * L)); CK(hipMalloc(&r, (size_t)ROWS * D * 4 * L));
    CK(hipMalloc(&out_sep, (size_t)ROWS * D * 4 * L)); CK(hipMalloc(&out_fused, (size_t)ROWS * D * 4 * L));
    CK(hipMalloc(&hand, (size_t)ROWS * 3 * D * 4 * L)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&tpos, 4));
    CK(hipMalloc(&ts, (size_t)(GW + AW) * NST * 8 + (size_t)AW * 2 * 8));
    hipLaunchKernelGGL(k_fill_bf16, dim3(2048), dim3(256), 0, s1, (bf16_t*)w, wl * 8 * L, 0x1234u, 1.0f);
    hipLaunchKernelGGL(k_fill_bf16, dim3(2048), dim3(256), 0, s1, (bf16_t*)a, al * 8 * L, 0x9876u, 1.0f);
    hipLaunchKernelGGL(k_fill_bf16, dim3(4096), dim3(256), 0, s1, kc, cl * L, 0x5555u, 1.0f);
    hipLaunchKernelGGL(k_fill_bf16, dim3(4096), dim3(256), 0, s1, vc, cl * L, 0x7777u, 1.0f);
    hipLaunchKernelGGL(k_fill_u32, dim3(256), dim3(256), 0, s1, hand, (size_t)ROWS * 3 * D * L, SENT);
    hipLaunchKernelGGL(k_set, dim3(1), dim3(64), 0, s1, tpos, t);
    CK(hipMemsetAsync(err, 0, 4, s1)); CK(hipMemsetAsync(ts, 0, (size_t)(GW + AW) * NST * 8 + (size_t)AW * 2 * 8, s1));
    CK(hipStreamSynchronize(s1));

    int g_delay = 0;
    auto layer = [&](int l, float* out, bool trace) {
        LayerArgs p;
        p.w = w + wl * l; p.a = a + al * l; p.q = q + (size_t)ROWS * D * l; p.r = r + (size_t)ROWS * D * l;
        p.hand = hand + (size_t)ROWS * 3 * D * l; p.kc = kc + cl * l; p.vc = vc + cl * l; p.out = out + (size_t)ROWS * D * l;
        p.tpos = tpos; p.err = err; p.ts = (trace && l == L / 2) ? ts : nullptr; p.a_delay = g_delay;
        return p;
    };
    const double mb = (wl * 16 + 2.0 * ROWS * H * (t + 1) * HD * 2) / 1e6;
    printf("context %d, %d layers, %.1f MB per layer (weights %.1f + K / V %.1f)\n", t, L, mb, wl * 16 / 1e6, mb - wl * 16 / 1e6);

    auto stamps = [&](const char* name) {
        std::vector<unsigned long long> h((size_t)(GW + AW) * NST);
        CK(hipMemcpy(h.data(), ts, h.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long origin = ~0ull;
        for (int i = 0; i < GW + AW; ++i) if (h[(size_t)i * NST]) origin = std::min(origin, h[(size_t)i * NST]);
        auto stat = [&](int i0, int i1, int k) {
            std::vector<double> v;
            for (int i = i0; i < i1; ++i) if (h[(size_t)i * NST + k]) v.push_back((h[(size_t)i * NST + k] - origin) * 0.01);
            std::sort(v.begin(), v.end());
            if (v.empty()) { printf(" -"); return; }
            printf(" %5.2f/%5.2f/%5.2f", v.front(), v[v.size() / 2], v.back());
        };
        printf("   %s stamps (us since the first wave; min/median/max over workgroups):\n      G start", name); stat(0, GW, 0);
        printf("  G end"); stat(0, GW, 1);
        printf("\n      A start"); stat(GW, GW + AW, 0); printf("  first chunks landed"); stat(GW, GW + AW, 1);
        printf("  q seen"); stat(GW, GW + AW, 2); printf("  A end"); stat(GW, GW + AW, 3); printf("\n");
        CK(hipMemset(ts, 0, h.size() * 8));
    };

    // ---- separate launches
    double us_g = time_graph([&]() { for (int l = 0; l < L; ++l) hipLaunchKernelGGL(k_g, dim3(GW), dim3(256), 0, s1, layer(l, out_sep, false)); }, reps);
    double us_a1 = time_graph([&]() { for (int l = 0; l < L; ++l) hipLaunchKernelGGL((k_a<1, 4>), dim3(AW), dim3(256), 0, s1, layer(l, out_sep, false)); }, reps);
    double us_a2 = time_graph([&]() { for (int l = 0; l < L; ++l) hipLaunchKernelGGL((k_a<2, 4>), dim3(AW), dim3(256), 0, s1, layer(l, out_sep, false)); }, reps);
    double us_a8 = time_graph([&]() { for (int l = 0; l < L; ++l) hipLaunchKernelGGL((k_a<1, 8>), dim3(AW), dim3(512), 0, s1, layer(l, out_sep, false)); }, reps);
    printf("G alone %.2f us   A alone: P=1 %.2f  P=2 %.2f  P=1 with 8 waves %.2f us per layer\n", us_g / L, us_a1 / L, us_a2 / L, us_a8 / L);
    double us_sep = time_graph([&]() {
        for (int l = 0; l < L; ++l) {
            hipLaunchKernelGGL(k_g, dim3(GW), dim3(256), 0, s1, layer(l, out_sep, true));
            hipLaunchKernelGGL((k_a<1, 4>), dim3(AW), dim3(256), 0, s1, layer(l, out_sep, true));
        }
    }, reps);
    printf("sep   (G, A<1>)          %6.2f us per layer  %.2f TB/s\n", us_sep / L, mb / (us_sep / L));
    stamps("sep");
    std::vector<float> ref((size_t)ROWS * D * L), got((size_t)ROWS * D * L);
    CK(hipMemcpy(ref.data(), out_sep, ref.size() * 4, hipMemcpyDeviceToHost));

    // ---- CPU check of the separate path: layer 0, a few (row, head) pairs, from the q / caches the device holds
    {
        std::vector<float> hq((size_t)ROWS * D);
        CK(hipMemcpy(hq.data(), q, hq.size() * 4, hipMemcpyDeviceToHost));
        std::vector<bf16_t> hk((size_t)TCAP * HD), hv((size_t)TCAP * HD);
        double worst = 0;
        for (int b : {0, 7, 15}) for (int h : {0, 11, 23}) {
            CK(hipMemcpy(hk.data(), kc + ((size_t)b * H + h) * TCAP * HD, hk.size() * 2, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hv.data(), vc + ((size_t)b * H + h) * TCAP * HD, hv.size() * 2, hipMemcpyDeviceToHost));
            std::vector<double> s(t + 1);
            double mx = -1e300;
            for (int i = 0; i <= t; ++i) {
                double d = 0;
                for (int e = 0; e < HD; ++e) d += (double)hq[(size_t)b * D + h * HD + e] * bf2f_h(hk[(size_t)i * HD + e]);
                s[i] = d * 0.125; mx = std::max(mx, s[i]);
            }
            double den = 0; std::vector<double> o(HD, 0.0);
            for (int i = 0; i <= t; ++i) { const double pr = exp(s[i] - mx); den += pr; for (int e = 0; e < HD; ++e) o[e] += pr * bf2f_h(hv[(size_t)i * HD + e]); }
            for (int e = 0; e < HD; ++e) worst = std::max(worst, fabs(o[e] / den - ref[(size_t)b * D + h * HD + e]));
        }
        printf("   CPU check (layer 0, 9 (row, head) pairs): max abs err %.3g\n", worst);
    }

    // ---- flag-free variant: prefetch workgroups inside the G launch warm the attention's first positions in the consumer XCD's L2
    for (int npf : {64, 128, 256, 512}) {
        CK(hipMemset(ts, 0, (size_t)(GW + AW) * NST * 8 + (size_t)AW * 2 * 8));
        double us_gp = time_graph([&]() { for (int l = 0; l < L; ++l) hipLaunchKernelGGL(k_g_pf, dim3(GW + AW), dim3(256), 0, s1, layer(l, out_sep, false), npf); }, reps);
        double us = time_graph([&]() {
            for (int l = 0; l < L; ++l) {
                hipLaunchKernelGGL(k_g_pf, dim3(GW + AW), dim3(256), 0, s1, layer(l, out_fused, true), npf);
                hipLaunchKernelGGL((k_a<1, 4>), dim3(AW), dim3(256), 0, s1, layer(l, out_fused, true));
            }
        }, reps);
        std::vector<unsigned long long> h((size_t)(GW + AW) * NST + (size_t)AW * 2);
        CK(hipMemcpy(h.data(), ts, h.size() * 8, hipMemcpyDeviceToHost));
        int match = 0, seen = 0;
        for (int pw = 0; pw < AW; ++pw) {
            const unsigned long long v = h[(size_t)(GW + AW) * NST + (size_t)pw * 2];
            if (!v) continue;
            const int xcc = (int)(v & 15) - 1, pair = (int)(v >> 4) - 1;
            const unsigned long long a = h[(size_t)(GW + pair) * NST + 4];
            if (a) { ++seen; match += ((int)a - 1) == xcc; }
        }
        printf("prefetch %3d positions in the G launch (%5.1f MB): G+pf alone %.2f us, pair %6.2f us per layer (%.2f us less than sep); "
               "prefetcher XCD == consumer XCD for %d of %d pairs\n", npf, 2.0 * AW * std::min(npf, t) * HD * 2 / 1e6, us_gp / L, us / L,
               (us_sep - us) / L, match, seen);
        stamps("G+prefetch, A");
        std::vector<float> g2((size_t)ROWS * D * L);
        CK(hipMemcpy(g2.data(), out_fused, g2.size() * 4, hipMemcpyDeviceToHost));
        printf("   outputs %s the separate launches'\n", memcmp(g2.data(), ref.data(), g2.size() * 4) == 0 ? "bit-identical to" : "DIFFER from");
        CK(hipMemset(out_fused, 0, g2.size() * 4));
    }
    auto check = [&](const char* name) {
        CK(hipMemcpy(got.data(), out_fused, got.size() * 4, hipMemcpyDeviceToHost));
        size_t nbits = 0; double worst = 0;
        for (size_t i = 0; i < got.size(); ++i) {
            if (memcmp(&got[i], &ref[i], 4) != 0) ++nbits;
            worst = std::max(worst, (double)fabsf(got[i] - ref[i]));
        }
        unsigned e; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
        printf("   %s vs sep: %zu of %zu outputs differ in bits, max abs diff %.3g, poll timeouts %u\n", name, nbits, got.size(), worst, e);
        CK(hipMemset(out_fused, 0, got.size() * 4)); CK(hipMemset(err, 0, 4));
    };
#define RUN_FUSED(KER, NAME)                                                                                         \
    {                                                                                                                \
        double us = time_graph([&]() {                                                                               \
            for (int l = 0; l < L; ++l) hipLaunchKernelGGL(KER, dim3(GW + AW), dim3(256), 0, s1, layer(l, out_fused, true)); \
        }, reps);                                                                                                    \
        printf("fused %-22s %6.2f us per layer  %.2f TB/s   (%.2f us less than sep)\n", NAME, us / L, mb / (us / L), (us_sep - us) / L); \
        stamps(NAME); check(NAME);                                                                                   \
    }
    RUN_FUSED((k_fused<1, 2>), "P=1 wpe=2")
    RUN_FUSED((k_fused<1, 3>), "P=1 wpe=3")
    RUN_FUSED((k_fused_kl<3>), "P=1+K(LDS) wpe=3")
    RUN_FUSED((k_fused_kl<2>), "P=1+K(LDS) wpe=2")
    for (int dl : {50, 100, 200, 300}) {
        g_delay = dl;
        char nm[64];
        snprintf(nm, sizeof nm, "P=1 wpe=3 delay %.1f", dl * 0.01);
        RUN_FUSED((k_fused<1, 3>), nm)
        snprintf(nm, sizeof nm, "P=1+K(LDS) w3 dl %.1f", dl * 0.01);
        RUN_FUSED((k_fused_kl<3>), nm)
    }
    g_delay = 0;
    printf("done\n");
    return 0;
}
