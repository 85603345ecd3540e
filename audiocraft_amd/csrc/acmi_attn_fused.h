// The self-attention role of the fused QKV -> self-attention launch (qkv_attn_kernel, acmi_gemm.hip).
// Same arithmetic, in the same order, as attn_decode_kernel<bf16_t, 64, false> (acmi_attn.hip) on the cache the separate QKV
// launch would have left: results are bit-identical.  What differs is WHEN things happen:
//   * the workgroup (row b, head h) runs INSIDE the launch that computes its q / k / v: it requests its first chunks of the
//     q-INDEPENDENT K / V stream at once (positions below the one this step appends), optionally stages the K half of its second
//     round in LDS by LDS-DMA (no registers in flight), and only then waits for its 3 x 64 words of the hand-off row;
//   * hand-off: the producers (tl_epilogue<.., EPI_QKVH>) store q | k | v of row b as f32 words with write-through (agent scope)
//     stores into slots that hold a sentinel -- the value is its own flag, no separate flag round trip; the consumer polls its
//     words with agent-scope loads (bounded: a time-out raises the state's error word, which later launches see at entry), rounds
//     k / v to the cache type exactly like the separate launch's epilogue, substitutes them for position `tnew` in the chunk that
//     holds it, writes them into the cache for the positions to come, and re-arms its slots after the workgroup's last read;
//   * deadlock freedom: producers occupy the lower block indexes (dispatched first, in order) and never wait on anything.
// Reference semantics: audiocraft/modules/transformer.py:266-298 (_complete_kv), :362-399, :412-414.
#pragma once
#include "acmi_lm_internal.h"

#define ACMI_HAND_SENTINEL 0x7fc0deadu   // a quiet NaN no arithmetic of the path produces

struct FusedAttnArgs {
    const int* len_dev;      // device word: positions already in the cache = the position this step appends
    unsigned* hand;          // [rows][3 d] words: q | k | v of the new position, sentinel-armed
    void* out;               // tiled activation (bf16 A-fragments) receiving the attention output
    int* err;                // [0]: poll time-outs (sticky: a non-zero word makes later launches skip their polls)
    int out_rbs, out_col0, len_bias, rows;
    float scale; int stage_k;   // stage_k: LDS-DMA the K half of the second round before the poll
    int d, pad0;
};

__device__ __forceinline__ void fa_lds_dma(const void* gbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;   // M0 (the LDS destination base) is compiler-reserved: saved and restored inside the statement
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %3 nt\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(gbase) : "memory");
}
template <int CTRL> __device__ __forceinline__ float fa_dpp(float v) { return dpp_f32<CTRL>(v); }
__device__ __forceinline__ float fa_group_sum8(float v) {   // sum over 8 consecutive lanes (acmi_attn.hip group_sum<8>)
    v += fa_dpp<0xB1>(v); v += fa_dpp<0x4E>(v); v += fa_dpp<0x141>(v);
    return v;
}

// aoff: byte offset of the FusedAttnArgs block in the kernarg segment.  smem: >= 1.1 KB (+ 32 KB when K is staged).
__device__ __forceinline__ void attn_fused_role(const void* hkc, const void* hvc, const int H, const int Tcap, const int wg,
                                                const int aoff, unsigned char* smem) {
    constexpr int HD = 64, DPL = 8, LPP = 8, PPI = 8, NI = 8, CH = NI * PPI, NW = 4;
    typedef bf16_t rawv __attribute__((ext_vector_type(DPL)));
    const int h = wg % H, b = wg / H;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane % LPP, pp = lane / LPP;
    const bf16_t* kbase = reinterpret_cast<const bf16_t*>(hkc) + ((size_t)b * H + h) * Tcap * HD;   // (uniform)
    const bf16_t* kb = kbase + c * DPL;
    const bf16_t* vb = reinterpret_cast<const bf16_t*>(hvc) + ((size_t)b * H + h) * Tcap * HD + c * DPL;
    // The argument block and the length word come FIRST (two dependent scalar round trips, ~1 us): unlike the stand-alone kernel
    // this role is not latency-bound at its start -- the launch is bandwidth-bound from its first request (delaying
    // these requests by 2 us changed nothing) -- and a speculative first chunk clamped to the cache capacity costs REAL bytes at
    // short contexts (up to 25 MB per launch next to the weight stream; PMC: 1.35 x the algorithmic bytes at a mean context of 67).
    FusedAttnArgs p;
    {
        const char __attribute__((address_space(4)))* ka = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
        __builtin_memcpy(&p, (const void __attribute__((address_space(4)))*)__builtin_assume_aligned(
                                 (const void __attribute__((address_space(4)))*)(ka + aoff), 8), sizeof(FusedAttnArgs));
    }
    const int tnew = __builtin_amdgcn_readfirstlane(*p.len_dev) + p.len_bias - 1;   // position appended by this step
    const int len = tnew + 1;
    const bool dead = __builtin_amdgcn_readfirstlane(*p.err) != 0;   // an earlier launch timed out: do not spin again
    rawv kr[NI], vr[NI];
    const int lim = max(tnew, 1);   // position tnew is not in the cache yet: it arrives through the hand-off
    const int r0 = wave * CH, r1 = (wave + NW) * CH;
    auto load_kv = [&](int t0) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int t = max(min(t0 + i * PPI + pp, lim - 1), 0);
            kr[i] = __builtin_nontemporal_load(reinterpret_cast<const rawv*>(kb + (size_t)t * HD));
            vr[i] = __builtin_nontemporal_load(reinterpret_cast<const rawv*>(vb + (size_t)t * HD));
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    load_kv(r0);   // (a wave past the context re-reads the last valid position: cache hits, no branch around the requests)
    unsigned char* kst = smem + 2048;   // [NW][8 KB] staged K of the second round
    const bool staged = p.stage_k != 0 && r1 < tnew;   // (wave-uniform) the second round exists in the cache
    if (staged) {
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(size_t)(__attribute__((address_space(3))) void*)kst + wave * 8192u));
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int t = min(r1 + i * PPI + pp, Tcap - 1);
            fa_lds_dma(kbase, (unsigned)((t * HD + c * DPL) * 2), dst + i * 1024u);
        }
    }
    // ---- the hand-off: q | k | v of (row b, head h); lane (c, *) needs dims [8 c, 8 c + 8) of each
    float qv[DPL];
    rawv knew, vnew;
    {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.hand + (size_t)b * 3 * p.d + h * HD, 0, -1, 0x00020000);
        u32x4 w[6];
        unsigned spins = 0;
        // ONE poller per workgroup: wave 0 polls and hands the 3 x 64 words over through LDS, the other waves wait at the barrier.
        // Same speed as every wave polling for itself (profiles/r06_fused_qkv_attn_one_poller_ab.txt) and the same HBM-side traffic by
        // PMC (34.3 MB per launch at a mean context of 67 either way): kept because it is a quarter of the poll requests in the
        // CU's memory queue.
        unsigned* hw = reinterpret_cast<unsigned*>(smem + 1280);   // [3][64] words
        if (wave == 0)
        for (;;) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                w[2 * j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (j * p.d + c * DPL) * 4, 0, 16);       // aux 16 = sc1: agent scope
                w[2 * j + 1] = __builtin_amdgcn_raw_buffer_load_b128(rs, (j * p.d + c * DPL + 4) * 4, 0, 16);
            }
            bool bad = false;
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) bad |= w[j][e] == ACMI_HAND_SENTINEL;
            if (!__any(bad) || dead) break;
            if (++spins > 20000u) { if (lane == 0) atomicAdd(p.err, 1); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (wave == 0 && pp == 0) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                *reinterpret_cast<u32x4*>(hw + j * 64 + c * DPL) = w[2 * j];
                *reinterpret_cast<u32x4*>(hw + j * 64 + c * DPL + 4) = w[2 * j + 1];
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            w[2 * j] = *reinterpret_cast<const u32x4*>(hw + j * 64 + c * DPL);
            w[2 * j + 1] = *reinterpret_cast<const u32x4*>(hw + j * 64 + c * DPL + 4);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            qv[e] = __uint_as_float(w[0][e]); qv[4 + e] = __uint_as_float(w[1][e]);
            knew[e] = f32_to_bf16(__uint_as_float(w[2][e])); knew[4 + e] = f32_to_bf16(__uint_as_float(w[3][e]));
            vnew[e] = f32_to_bf16(__uint_as_float(w[4][e])); vnew[4 + e] = f32_to_bf16(__uint_as_float(w[5][e]));
        }
        if (wave == 0 && pp == 0) {   // the new rows enter the cache for the positions to come
            *reinterpret_cast<rawv*>(const_cast<bf16_t*>(kb) + (size_t)tnew * HD) = knew;
            *reinterpret_cast<rawv*>(const_cast<bf16_t*>(vb) + (size_t)tnew * HD) = vnew;
        }
    }
    const float scale = p.scale;
    float m = -INFINITY, l = 0.f, o[DPL];
#pragma unroll
    for (int e = 0; e < DPL; ++e) o[e] = 0.f;
    auto process = [&](int t0) {   // kr / vr hold the chunk at t0 (acmi_attn.hip's loop body, + the substitution)
        if (t0 + CH > tnew) {      // (wave-uniform) the chunk reaches the new position / runs past the end
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int t = t0 + i * PPI + pp;
                if (t == tnew) { kr[i] = knew; vr[i] = vnew; }
                if (t >= len) {
#pragma unroll
                    for (int e = 0; e < DPL; ++e) vr[i][e] = (bf16_t)0;
                }
            }
        }
        float s[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int t = t0 + i * PPI + pp;
            float part = 0.f;
#pragma unroll
            for (int e = 0; e < DPL; ++e) part = fmaf(qv[e], bf16_to_f32(kr[i][e]), part);
            part = fa_group_sum8(part);
            s[i] = (t < len) ? part * scale : -INFINITY;
        }
        float cmax = s[0];
#pragma unroll
        for (int i = 1; i < NI; ++i) cmax = fmaxf(cmax, s[i]);
        cmax = fmaxf(cmax, fa_dpp<0x128>(cmax));
        cmax = fmaxf(cmax, __shfl_xor(cmax, 16, 64));
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
        const float m_new = fmaxf(m, cmax);
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
            for (int e = 0; e < DPL; ++e) o[e] = fmaf(pr, bf16_to_f32(vr[i][e]), o[e]);
        }
        m = m_new;
    };
    if (r0 < len) {
        process(r0);
        if (r1 < len) {
            if (staged) {
                // K of the second round comes back from LDS (it landed before the poll's words: vmcnt retires in order), V now
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned char* src = kst + wave * 8192 + lane * 16;
#pragma unroll
                for (int i = 0; i < NI; ++i) kr[i] = *reinterpret_cast<const rawv*>(src + i * 1024);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int t = max(min(r1 + i * PPI + pp, lim - 1), 0);
                    vr[i] = __builtin_nontemporal_load(reinterpret_cast<const rawv*>(vb + (size_t)t * HD));
                }
                __builtin_amdgcn_sched_barrier(0);
            } else {
                load_kv(r1);
            }
            process(r1);
            for (int t0 = r1 + NW * CH; t0 < len; t0 += NW * CH) { load_kv(t0); process(t0); }
        }
    }
#pragma unroll
    for (int off = LPP; off < 64; off <<= 1) {
        const bool dpp8 = off == 8;
        l += dpp8 ? fa_dpp<0x128>(l) : __shfl_xor(l, off, 64);
#pragma unroll
        for (int e = 0; e < DPL; ++e) o[e] += dpp8 ? fa_dpp<0x128>(o[e]) : __shfl_xor(o[e], off, 64);
    }
    float* sm_o = reinterpret_cast<float*>(smem);   // [4][HD]
    float* sm_m = sm_o + 4 * HD; float* sm_l = sm_m + 4;
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
        const float r = den > 0.f ? num / den : 0.f;
        const int f = h * HD + threadIdx.x;
        reinterpret_cast<bf16_t*>(p.out)[tiled_index<bf16_t>(b, p.out_col0 + f, p.out_rbs)] = f32_to_bf16(r);
    }
    if (threadIdx.x < 3 * HD / 4) {   // every wave has read its words (barrier above): re-arm the slots for the next launch (16-byte stores)
        const int j = threadIdx.x / (HD / 4), dd = (threadIdx.x % (HD / 4)) * 4;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.hand, 0, -1, 0x00020000);
        const u32x4 w4 = {ACMI_HAND_SENTINEL, ACMI_HAND_SENTINEL, ACMI_HAND_SENTINEL, ACMI_HAND_SENTINEL};
        __builtin_amdgcn_raw_buffer_store_b128(w4, rs, (int)(((size_t)b * 3 * p.d + j * p.d + h * HD + dd) * 4), 0, 16);   // aux 16 = sc1
    }
}
