// MusicGen LM decode step for gfx950 (CDNA4, wave64): embedding, sampler and acmi_lm_step, the per-position
// chain over the GEMMs of acmi_gemm.hip and the attention of acmi_attn.hip.
//
//   embed_kernel      sum of codebook embeddings (or prepended condition row) + sinusoidal position.
//   sample_kernel     CFG mix + softmax/top-k/top-p/multinomial (or argmax) + delay-pattern write-back.
//
// Reference semantics: audiocraft/models/lm.py:221-268,323-418,536-565;
// audiocraft/modules/transformer.py:70-89,550-574,693-713; audiocraft/utils/utils.py:88-141.
#include "acmi_lm_internal.h"
#include "acmi_attn_fused.h"

#include <math.h>
#include <stdlib.h>
#include <type_traits>

// =====================================================================================================
// embedding sum + sinusoidal position
// =====================================================================================================

__device__ __forceinline__ float block_sum(float v, float* sval);

struct EmbedArgs {
    const void* emb[16]; int w_bf16;
    const int64_t* gen_sequence; int B, Beff, K, S, card;
    const float* prepend; int P;
    const float* pos_table; float pos_scale;
    const int* pos;
    float* x; int d;
    float* stats;  // [M][1][2]: (mean, M2) of every produced row (one partial of d elements)
    void* xt_hi; void* xt_lo; int xt_nkc, xt_lo_nkc;  // folded LayerNorm: the raw row in fragment order (hi / lo), or NULL
    float* shift_out;  // [M] or NULL: single-term fragments bf16(x - mean), the row mean stored here (acmi_lm_state.xshift)
    const int* row_off; // per cache row: left padding of its stream (acmi_lm_state.row_off), or NULL
    int npos_pad, npos; // > 0: position-minor rows of the MFMA-tiled prefill (row = cache row * npos_pad + position; pad
                        // rows repeat the last position: finite, never stored to the caches); 0: row = position * Beff + cache row
    const float* input_add; int n_add;   // ADD variant only: acmi_lm_state.input_add / n_add (fuser 'sum' / 'input_interpolate')
};

// BF16: element type of the tables; KQ >= n_q: codebook tables read per row (compile time, so that every load of a phase
// is an unconditional, branch-free request: with `if (k < K)` / `bf16 ? .. : ..` around them hipcc put each load in its own
// basic block followed by its own s_waitcnt vmcnt(0) -- 24 serialised memory round trips, 17 us of an 18.7 us kernel)
// ADD: the fuser's 'sum' / 'input_interpolate' conditions (acmi_lm_state.input_add) -- a variant of its own so that the
// kernel every released model runs is unchanged.
template <bool BF16, int KQ, bool ADD = false>
__global__ __launch_bounds__(256) void embed_kernel(const EmbedArgs p) {
    __shared__ float sred[4], sred2[4];    // one slot set per reduction: a single barrier each
    const int m = blockIdx.x;              // row = position-of-the-call * Beff + CFG row
    int pidx, m0;
    if (p.npos_pad > 0) { m0 = m / p.npos_pad; pidx = min(m - m0 * p.npos_pad, p.npos - 1); }
    else { pidx = m / p.Beff; m0 = m - pidx * p.Beff; }
    const int g = *p.pos + pidx;
    const int gpos = p.row_off != nullptr ? max(g - p.row_off[m0], 0) : g;   // the row's own position (left-padded streams)
    const int b = m0 % p.B;
    float loc[8];  // d <= 2048
    float sum = 0.f;
    int cnt = 0;
    // Three dependent memory round trips in all: position -> tokens -> embedding rows.
    const int sidx = max(g - p.P, 0);      // (rows of the prepended condition never read their tokens)
    int64_t tok64[KQ];
#pragma unroll
    for (int k = 0; k < KQ; ++k) tok64[k] = p.gen_sequence[((size_t)b * p.K + min(k, p.K - 1)) * p.S + sidx];
    int toks[KQ];
#pragma unroll
    for (int k = 0; k < KQ; ++k) {
        int64_t tok = tok64[k];
        if (tok < 0 || tok > p.card) tok = p.card;  // never happens for a well-formed sequence
        toks[k] = (int)tok;
    }
    constexpr int CPT = 8;   // channels per thread, d <= 2048
    typedef typename std::conditional<BF16, bf16_t, float>::type ET;
    ET ev[CPT][KQ];
    float pv[CPT], pre[CPT];
    const bool is_prepend = g < p.P;       // block uniform
    const float* prow = p.prepend != nullptr ? p.prepend + ((size_t)m0 * p.P + min(g, max(p.P - 1, 0))) * p.d : p.pos_table;
    float av[CPT];
    const float* arow = p.pos_table;
    if constexpr (ADD) arow = p.input_add + ((size_t)m0 * p.n_add + min(sidx, p.n_add - 1)) * p.d;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int cch = min((int)threadIdx.x + i * 256, p.d - 1);
        pv[i] = p.pos_table[(size_t)gpos * p.d + cch];
        pre[i] = prow[cch];
        if constexpr (ADD) av[i] = arow[cch];
#pragma unroll
        for (int k = 0; k < KQ; ++k)
            ev[i][k] = reinterpret_cast<const ET*>(p.emb[min(k, p.K - 1)])[(size_t)toks[k] * p.d + cch];
    }
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int cch = (int)threadIdx.x + i * 256;
        if (cch >= p.d) break;
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < KQ; ++k)       // same order as the reference's sum over codebooks (lm.py:241)
            if (k < p.K) v += ld_f32<ET>(&ev[i][k]);
        if constexpr (ADD) v += av[i];     // `input += cond` of the fuser, before the positional embedding (conditioners.py:1733-1737)
        if (is_prepend) v = pre[i];
        v += p.pos_scale * pv[i];
        p.x[(size_t)m * p.d + cch] = v;
        loc[i] = v;
        cnt = i + 1;
        sum += v;
    }
    // two-pass (mean, M2) of the row for the first LayerNorm (same summation order as block_sum: wave butterflies, then the
    // waves in order; the second reduction uses its own slots, so neither needs a trailing barrier)
    auto bsum1 = [&](float x, float* slot) {
        x = wave_sum(x);
        if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = x;
        __syncthreads();
        float r = 0.f;
        for (int w = 0; w < 4; ++w) r += slot[w];
        return r;
    };
    const float mean = bsum1(sum, sred) / (float)p.d;
    const float shift = p.shift_out != nullptr ? mean : 0.f;   // single-term fragments are stored relative to the row mean
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i >= cnt) break;
        const int cch = (int)threadIdx.x + i * 256;
        q += (loc[i] - mean) * (loc[i] - mean);
        if (p.xt_hi != nullptr) {
            const float v = loc[i] - shift;
            if (BF16) {
                const size_t ti = tiled_index<bf16_t>(m, cch, p.xt_nkc);
                const bf16_t hi = f32_to_bf16(v);
                reinterpret_cast<bf16_t*>(p.xt_hi)[ti] = hi;
                if (p.xt_lo != nullptr)
                    reinterpret_cast<bf16_t*>(p.xt_lo)[tiled_index<bf16_t>(m, cch, p.xt_lo_nkc)] = f32_to_bf16(v - bf16_to_f32(hi));
            } else {
                reinterpret_cast<float*>(p.xt_hi)[tiled_index<float>(m, cch, p.xt_nkc)] = v;
            }
        }
    }
    q = bsum1(q, sred2);
    if (threadIdx.x == 0) {
        p.stats[(size_t)m * 2] = mean; p.stats[(size_t)m * 2 + 1] = q;
        if (p.shift_out != nullptr) p.shift_out[m] = mean;
    }
}

// create_sin_embedding (transformer.py:70-89): one block per position, computed once per run geometry
__global__ __launch_bounds__(256) void pos_table_kernel(const float* __restrict__ freq, float* __restrict__ table, int d) {
    const int t = blockIdx.x, half = d / 2;
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        const int i = c < half ? c : c - half;
        const float phase = (float)t / freq[i];
        table[(size_t)t * d + c] = c < half ? cosf(phase) : sinf(phase);
    }
}

extern "C" int acmi_pos_table(const float* freq, float* table, int T, int d, void* stream) {
    ACMI_REQUIRE(T > 0 && d > 0 && d % 2 == 0, "acmi_pos_table: bad shape T=%d d=%d", T, d);
    hipLaunchKernelGGL(pos_table_kernel, dim3(T), dim3(256), 0, (hipStream_t)stream, freq, table, d);
    return acmi_check_launch("pos_table_kernel");
}

// =====================================================================================================
// CFG + sampling + pattern write-back
// =====================================================================================================

#define ACMI_MAX_CARD 4096

struct SampleArgs {
    const float* logits;  // [Beff, K*card]
    int B, K, card, use_cfg;   // use_cfg: ACMI_CFG_NONE | ACMI_CFG_PAIR | ACMI_CFG_DOUBLE
    float cfg_coef, cfg_beta;
    int use_sampling; float temp; int top_k; float top_p;
    uint64_t seed; uint64_t step; int* pos;  // step counter = step + pos[0] when pos != NULL
    int advance;          // last block to finish does pos[0] += 1 (pos[1] is the ticket counter)
    int64_t* tokens_out;  // [B, K] or NULL
    float* mixed_out;     // [B, K, card] or NULL
    // write-back (NULL gen_sequence: skipped)
    int64_t* gen_sequence; const uint8_t* seq_mask; int S, P;
};

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// block-wide argmax with first-index tie break; all threads get the result
__device__ __forceinline__ int block_argmax(float v, int idx, float* sval, int* sidx) {
    // (value, index) butterflies: partners inside a row of 16 lanes through DPP (acmi_common.h), then xor 16 / 32
#define ACMI_ARGMAX_STEP(OV, OI) { const float ov = (OV); const int oi = (OI); if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; } }
    ACMI_ARGMAX_STEP(dpp_f32<0xB1>(v), dpp_i32<0xB1>(idx))
    ACMI_ARGMAX_STEP(dpp_f32<0x4E>(v), dpp_i32<0x4E>(idx))
    ACMI_ARGMAX_STEP(dpp_f32<0x141>(v), dpp_i32<0x141>(idx))
    ACMI_ARGMAX_STEP(dpp_f32<0x128>(v), dpp_i32<0x128>(idx))
    ACMI_ARGMAX_STEP(__shfl_xor(v, 16, 64), __shfl_xor(idx, 16, 64))
    ACMI_ARGMAX_STEP(__shfl_xor(v, 32, 64), __shfl_xor(idx, 32, 64))
#undef ACMI_ARGMAX_STEP
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) { sval[wave] = v; sidx[wave] = idx; }
    __syncthreads();
    float bv = sval[0]; int bi = sidx[0];
    for (int w = 1; w < nw; ++w)
        if (sval[w] > bv || (sval[w] == bv && sidx[w] < bi)) { bv = sval[w]; bi = sidx[w]; }
    __syncthreads();
    return bi;
}

__device__ __forceinline__ float block_max(float v, float* sval) {
    v = wave_max(v);
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) sval[wave] = v;
    __syncthreads();
    float r = sval[0];
    for (int w = 1; w < nw; ++w) r = fmaxf(r, sval[w]);
    __syncthreads();
    return r;
}
__device__ __forceinline__ float block_sum(float v, float* sval) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) sval[wave] = v;
    __syncthreads();
    float r = 0.f;
    for (int w = 0; w < nw; ++w) r += sval[w];
    __syncthreads();
    return r;
}

// Round 6: the same arithmetic (bit-identical probabilities, threshold and race as the round-2 kernel, so a (seed, logits) pair
// gives the same token), restructured around what the launch was waiting for (15.5 us per position for 32 workgroups):
//   * EPT elements per thread stay in REGISTERS (8 for card <= 2048: the loads of a row group are 8, not 16, requests);
//   * every block reduction has its own LDS slots: ONE barrier each instead of two;
//   * radix select: the FIRST pass looks at the top byte of a probability's bit pattern (sign + 7 exponent bits), which takes
//     a handful of distinct values -- 2048 LDS atomics on 3-4 addresses serialise; the lanes of a wave holding the same
//     digit now elect a leader that adds their count (wave-aggregated atomic), and the next pass's bins are cleared while
//     this pass is scanned (double-buffered histogram);
//   * the race runs over a COMPACTED candidate list (~top_k entries: one Philox per thread) instead of card / 256 rounds in
//     which every wave executes the Philox because some lane holds a candidate;
//   * the previous value of the written-back sequence slot is requested at kernel start, and the position counter's ticket
//     needs no fence (every block's read of pos[0] completed long before it takes its ticket; the kernel boundary publishes
//     the rest): __threadfence() is an L2 write-back + invalidate, microseconds on the tail of every position.
template <int EPT>
__global__ __launch_bounds__(256) void sample_kernel(const SampleArgs p) {
    __shared__ float vals[EPT * 256];          // top-p path: the probabilities; race: candidate probabilities
    __shared__ int cand[EPT * 256];            // race: candidate indexes
    __shared__ float red_a[4], red_b[4];
    __shared__ float sval[4];
    __shared__ int sidx[4];
    __shared__ unsigned hist[2][256];
    __shared__ unsigned sel[2];
    __shared__ unsigned wtot[4];
    const int k = blockIdx.x, b = blockIdx.y;
    const int card = p.card;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* cond = p.logits + ((size_t)b * p.K + k) * card;
    // row groups: PAIR [cond; uncond], DOUBLE [cond (text + wav); wav; uncond]; a group the mode does not have aliases
    // `cond`, so that the loads below are unconditional (a branch around a load costs a memory round trip of its own:
    // the loads of one phase must be requested back to back)
    const float* second = p.use_cfg != ACMI_CFG_NONE ? p.logits + ((size_t)(p.B + b) * p.K + k) * card : cond;
    const float* third = p.use_cfg == ACMI_CFG_DOUBLE ? p.logits + ((size_t)(2 * p.B + b) * p.K + k) * card : cond;
    float lc[EPT], l2[EPT], l3[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = min((int)threadIdx.x + e * 256, card - 1);
        lc[e] = cond[i]; l2[e] = second[i]; l3[e] = third[i];
    }
    const int gpos = p.pos ? p.pos[0] : 0;
    // the sequence slot this position fills (lm.py:540-562) and what it holds now (a prompt token stays): requested here
    const int offset = (gpos - p.P) + 1;
    const bool wb = p.gen_sequence != nullptr && offset >= 0 && offset < p.S;
    int64_t* const dst = wb ? p.gen_sequence + ((size_t)b * p.K + k) * p.S + offset : nullptr;
    int64_t prev = -1; uint8_t mk = 0;
    if (threadIdx.x == 0 && wb) { prev = *dst; mk = p.seq_mask[(size_t)k * p.S + offset]; }
    float v[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = (int)threadIdx.x + e * 256;
        float x = lc[e];
        if (p.use_cfg == ACMI_CFG_PAIR) {  // uncond + (cond - uncond) * coef, rounded after every op like the reference (no fma)
            const float u = l2[e];
            x = __fadd_rn(u, __fmul_rn(__fsub_rn(x, u), p.cfg_coef));
        } else if (p.use_cfg == ACMI_CFG_DOUBLE) {  // uncond + coef * (wav + beta * (cond - wav) - uncond), lm.py:372-376
            const float w = l2[e], u = l3[e];
            const float inner = __fsub_rn(__fadd_rn(w, __fmul_rn(p.cfg_beta, __fsub_rn(x, w))), u);
            x = __fadd_rn(u, __fmul_rn(p.cfg_coef, inner));
        }
        v[e] = x;
        if (p.mixed_out && i < card) p.mixed_out[((size_t)b * p.K + k) * card + i] = x;
    }
    // one barrier per reduction: the partials of consecutive reductions live in different slots
    auto bsum = [&](float x, float* slot) {
        x = wave_sum(x);
        if (lane == 0) slot[wave] = x;
        __syncthreads();
        float r = 0.f;
        for (int w = 0; w < 4; ++w) r += slot[w];
        return r;
    };
    int token;
    if (!(p.use_sampling && p.temp > 0.f)) {
        float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = (int)threadIdx.x + e * 256;
            if (i < card && v[e] > bv) { bv = v[e]; bi = i; }
        }
        token = block_argmax(bv, bi, sval, sidx);
    } else {
        // softmax(logits / temp)
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = (int)threadIdx.x + e * 256;
            v[e] = v[e] / p.temp;
            if (i < card) mx = fmaxf(mx, v[e]);
        }
        mx = wave_max(mx);
        if (lane == 0) red_a[wave] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red_a[0], red_a[1]), fmaxf(red_a[2], red_a[3]));
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = (int)threadIdx.x + e * 256;
            const float ex = expf(v[e] - mx);
            v[e] = ex;
            if (i < card) sum += ex;
        }
        sum = bsum(sum, red_b);
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = (int)threadIdx.x + e * 256;
            v[e] = i < card ? v[e] / sum : 0.f;    // (elements past the vocabulary: probability 0, never a candidate)
        }
        float thr = 0.f;
        if (p.top_p > 0.f) {
            // utils.sample_top_p (utils.py:125-141): in descending order, keep i unless the mass sorted strictly before it
            // exceeds top_p.  The kept set is {p_i >= x*} with x* the smallest value whose strictly-greater mass
            // f(x) = sum_{p_j > x} p_j is <= top_p; f is a non-increasing step function, so x* is found by bisection on
            // the float bit pattern (non-negative floats order like their bits): 31 block reductions instead of the
            // card^2 comparisons of a direct evaluation.  Only the group of values equal to x* can be kept partially
            // (ties are ordered by index, like a stable sort).
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int i = (int)threadIdx.x + e * 256;
                if (i < card) vals[i] = v[e];
            }
            __syncthreads();
            auto mass_above = [&](float x) {
                float m = 0.f;
                for (int i = threadIdx.x; i < card; i += blockDim.x) m += vals[i] > x ? vals[i] : 0.f;
                return block_sum(m, sval);
            };
            unsigned lo = 0u, hi = 0x3f800000u;   // f(1.0) = 0 <= top_p always
            while (lo < hi) {                       // block-uniform: every thread sees the same reductions
                const unsigned mid = lo + ((hi - lo) >> 1);
                if (mass_above(__uint_as_float(mid)) <= p.top_p) hi = mid; else lo = mid + 1;
            }
            const float xs = __uint_as_float(lo);
            const float fstar = mass_above(xs);
            float ties = 0.f;
            for (int i = threadIdx.x; i < card; i += blockDim.x) ties += vals[i] == xs ? 1.f : 0.f;
            ties = block_sum(ties, sval);
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int i = (int)threadIdx.x + e * 256;
                if (i >= card) continue;
                const float pi = v[e];
                bool keep = pi >= xs;
                if (keep && pi == xs && ties > 1.5f) {   // rank of this element inside its tie group (rare path)
                    int r = 0;
                    for (int j = 0; j < i; ++j) r += vals[j] == xs ? 1 : 0;
                    keep = fstar + (float)r * xs <= p.top_p;
                }
                if (!keep) v[e] = 0.f;
            }
            __syncthreads();   // every thread has read what it needs of the un-filtered probabilities (vals is reused below)
        } else if (p.top_k > 0 && p.top_k < card) {
            // k-th largest probability by 4-pass radix select on the (non-negative) float bit patterns
            unsigned prefix = 0u, mask = 0u, remaining = (unsigned)p.top_k;
            hist[0][threadIdx.x] = 0u;
            __syncthreads();
#pragma unroll
            for (int pass = 3; pass >= 0; --pass) {
                unsigned* const hc = hist[pass & 1 ? 0 : 1];
                unsigned* const hn = hist[pass & 1 ? 1 : 0];
#pragma unroll
                for (int e = 0; e < EPT; ++e) {
                    const int i = (int)threadIdx.x + e * 256;
                    const unsigned bits = __float_as_uint(v[e]);
                    const bool act = i < card && (bits & mask) == prefix;
                    const unsigned digit = (bits >> (8 * pass)) & 255u;
                    if (pass == 3) {
                        // wave-aggregated: one atomic per distinct digit of the wave (a few exponent values), not one per lane
                        unsigned long long todo = __ballot(act);
                        while (todo) {
                            const int leader = __ffsll((long long)todo) - 1;
                            const unsigned dl = (unsigned)__shfl((int)digit, leader, 64);
                            const unsigned long long same = __ballot(act && digit == dl);
                            if (lane == leader) atomicAdd(&hc[dl], (unsigned)__popcll(same));
                            todo &= ~same;
                        }
                    } else if (act) {
                        atomicAdd(&hc[digit], 1u);
                    }
                }
                __syncthreads();
                {   // suffix sums S[t] = sum_{j >= t} hist[j] with 256 threads; digit d is selected when
                    // S[d] >= remaining > S[d+1]
                    const unsigned hcnt = hc[threadIdx.x];
                    hn[threadIdx.x] = 0u;     // the next pass's bins
                    unsigned sfx = hcnt;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) {
                        const unsigned o = __shfl_down(sfx, off, 64);
                        if (lane + off < 64) sfx += o;
                    }
                    if (lane == 0) wtot[wave] = sfx;
                    __syncthreads();
                    for (int w2 = wave + 1; w2 < 4; ++w2) sfx += wtot[w2];
                    if (sfx >= remaining && sfx - hcnt < remaining) { sel[0] = threadIdx.x; sel[1] = remaining - (sfx - hcnt); }
                }
                __syncthreads();
                prefix |= sel[0] << (8 * pass);
                mask |= 255u << (8 * pass);
                remaining = sel[1];
            }
            thr = __uint_as_float(prefix);
            __syncthreads();   // (sel / wtot are rewritten by nobody below; vals / cand are free)
        }
        // multinomial(num_samples=1) as an exponential race: argmax_i p_i / q_i, q_i ~ Exp(1), over the compacted support
        int mine = 0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) mine += (v[e] >= thr && v[e] > 0.f) ? 1 : 0;
        int incl = mine;   // inclusive prefix over the block: wave scan, then the waves' totals
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) sidx[wave] = incl;
        __syncthreads();
        int base = incl - mine, total = 0;
        for (int w = 0; w < 4; ++w) { if (w < wave) base += sidx[w]; total += sidx[w]; }
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            if (v[e] >= thr && v[e] > 0.f) { cand[base] = (int)threadIdx.x + e * 256; vals[base] = v[e]; ++base; }
        }
        __syncthreads();
        const uint64_t stepc = p.step + (uint64_t)gpos;
        float bv = -1.f; int bi = 0x7fffffff;
        for (int j = threadIdx.x; j < total; j += 256) {
            const int i = cand[j];
            const float pi = vals[j];
            uint32_t c4[4] = {(uint32_t)i, (uint32_t)(b * p.K + k), (uint32_t)stepc, (uint32_t)(stepc >> 32)};
            philox4x32_10(c4, (uint32_t)p.seed, (uint32_t)(p.seed >> 32));
            const float u = ((float)(c4[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
            const float r = pi / (-logf(u));
            if (r > bv || (r == bv && i < bi)) { bv = r; bi = i; }
        }
        token = block_argmax(bv, bi, sval, sidx);
    }
    if (threadIdx.x == 0) {
        if (p.tokens_out) p.tokens_out[(size_t)b * p.K + k] = token;
        if (wb && prev == -1) *dst = mk ? (int64_t)token : (int64_t)p.card;
        if (p.advance) {  // every block has read pos[0] (it computed with it) before it takes its ticket
            const int ticket = atomicAdd(&p.pos[1], 1);
            if (ticket == (int)(gridDim.x * gridDim.y) - 1) {
                p.pos[1] = 0;
                p.pos[0] = gpos + 1;
            }
        }
    }
}

static int launch_sample(const SampleArgs& a, hipStream_t st) {
    ACMI_REQUIRE(a.card > 0 && a.card <= ACMI_MAX_CARD, "acmi_sample: card=%d unsupported (max %d)", a.card, ACMI_MAX_CARD);
    ACMI_REQUIRE(a.B > 0 && a.K > 0, "acmi_sample: bad shape");
    if (a.card <= 2048) hipLaunchKernelGGL(sample_kernel<8>, dim3(a.K, a.B), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(sample_kernel<ACMI_MAX_CARD / 256>, dim3(a.K, a.B), dim3(256), 0, st, a);
    return acmi_check_launch("sample_kernel");
}

extern "C" int acmi_sample(const float* logits, int64_t* tokens_out, float* mixed_out, int B, int K, int card,
                           int cfg_mode, float cfg_coef, float cfg_coef_beta, int use_sampling, float temp, int top_k,
                           float top_p, uint64_t seed, uint64_t step, void* stream) {
    ACMI_REQUIRE(cfg_mode >= ACMI_CFG_NONE && cfg_mode <= ACMI_CFG_DOUBLE, "acmi_sample: bad cfg_mode %d", cfg_mode);
    SampleArgs a = {};
    a.logits = logits; a.B = B; a.K = K; a.card = card; a.use_cfg = cfg_mode; a.cfg_coef = cfg_coef; a.cfg_beta = cfg_coef_beta;
    a.use_sampling = use_sampling; a.temp = temp; a.top_k = top_k; a.top_p = top_p; a.seed = seed; a.step = step;
    a.tokens_out = tokens_out; a.mixed_out = mixed_out;
    return launch_sample(a, (hipStream_t)stream);
}

__global__ void advance_kernel(int* pos, int n) { if (threadIdx.x == 0 && blockIdx.x == 0) pos[0] += n; }

// =====================================================================================================
// one decode position
// =====================================================================================================

// LayerNorm in front of a GEMM, two interchangeable forms (ACMI_LN_MODE = fold | tile):
//   tile   a separate standardisation kernel writes the tiled `xn` (one extra launch per LayerNorm);
//   fold   the producers of x also emit it raw in fragment order (hi / lo), the consumer runs the plain tiled
//          GEMM on it and applies rstd * (acc - mean * colsum) in its epilogue from the (mean, M2) partials the
//          producer wrote next to x (no launch, no staging).  Default.
// (A third form -- the consumer staging the row-major x through LDS and standardising it there -- measured
// no faster than `tile` and was removed.)
enum { LN_TILE = 0, LN_FOLD = 2 };
enum { LN_X = 100 };  // step_lin: the activation is LayerNorm(x)
static int ln_mode_of(const acmi_lm_model* m, const acmi_lm_state* s) {
    static const int want = [] { const char* e = getenv("ACMI_LN_MODE"); return (e && e[0] == 't') ? (int)LN_TILE : (int)LN_FOLD; }();
    if (want == LN_FOLD) {
        const bool have = m->cs_head != nullptr && m->layers[0].cs_qkv != nullptr && m->layers[0].cs_ff1 != nullptr &&
                          (m->wdtype != ACMI_BF16 || s->xlo != nullptr || s->xshift != nullptr) && m->dim / 16 <= 128 &&
                          m->dim % 16 == 0;
        return have ? LN_FOLD : LN_TILE;
    }
    return LN_TILE;
}
// bf16 weights: the raw x fragments are single-term with a per-row shift when the state carries `xshift` (default), the
// hi / lo pair of 0.1.2 otherwise or with ACMI_LN_LO=1 (A/B switch); ACMI_LN_LO=0 without xshift = hi only, unshifted.
static int fold_lo_env() {
    static const int v = [] { const char* e = getenv("ACMI_LN_LO"); return e == nullptr ? -1 : (e[0] == '0' ? 0 : 1); }();
    return v;
}

// ---- the step's GEMM chain ---------------------------------------------------------------------------
// Rotary positions (include/acmi.h, acmi_lm_model.rope_freq; rope.py:75-114) as a launch of their own after the QKV GEMM:
// one thread per complex pair (features 2i, 2i+1 of a head) rotates q in place and the k row the GEMM has just appended
// to the cache.  Inside the GEMM's epilogue the same code cost EVERY variant of the kernel ~20 spilled SGPRs (six more
// kernel arguments, the sincosf / powf bodies) and the non-rotary decode step 3.4 % -- no released model uses rotary
// positions, so they pay for themselves here (one more launch per layer).  With a bf16 cache k is rounded twice (by the
// GEMM's store and by this one); exact in f32 parity mode.
struct RopeArgs {
    float* q; void* kc; int kv_bf16; int H, hd, Tcap, d, rpp; const int* pos;
    const float* freq; const float* decay; float scale, base; int first, shift;
    int npos_pad, npos;   // > 0: position-minor rows (see EmbedArgs); pad rows are skipped
    const int* row_off;   // left padding of each cache row's stream (acmi_lm_state.row_off) or NULL: the rotary position is the row's OWN
};

__global__ __launch_bounds__(1024) void rope_qk_kernel(const RopeArgs p) {
    const int gm = blockIdx.x, pi = threadIdx.x, half = p.hd >> 1;
    const int h = pi / half, i = pi - h * half;
    int pidx, brow;                                             // several positions per call (prefill)
    if (p.npos_pad > 0) { brow = gm / p.npos_pad; pidx = gm - brow * p.npos_pad; if (pidx >= p.npos) return; }
    else { pidx = gm / p.rpp; brow = gm - pidx * p.rpp; }
    const int tpos = *p.pos + pidx;
    // rotary position (acmi_lm_state.rope_first / rope_shift); of a left-padded stream: counted from the row's own first position
    // (the reference keeps one streaming state per pass, lm.py:378-390) -- the cache slot stays the stream position tpos
    const int own = tpos - (p.row_off != nullptr ? p.row_off[brow] : 0);
    const int rp = own >= p.first ? own - p.shift : own;
    float sn, cs;
    sincosf((float)rp * p.freq[i], &sn, &cs);
    float dq = 1.0f, dk = 1.0f;
    if (p.decay != nullptr) {
        dq = powf(p.decay[i], (float)rp / p.base);
        dk = 1.0f / dq;                                         // keys: inverted decay (rope.py:111-114)
    }
    const float one_m = 1.0f - p.scale;
    {   // q
        float* qp = p.q + (size_t)gm * p.d + h * p.hd + 2 * i;
        const float re = cs * dq * p.scale + one_m, im = sn * dq * p.scale;
        const float a = qp[0], b = qp[1];
        qp[0] = a * re - b * im;
        qp[1] = a * im + b * re;
    }
    const size_t ci = (((size_t)brow * p.H + h) * p.Tcap + tpos) * p.hd + 2 * i;
    const float re = cs * dk * p.scale + one_m, im = sn * dk * p.scale;
    if (p.kv_bf16) {
        bf16_t* kp = reinterpret_cast<bf16_t*>(p.kc) + ci;
        const float a = bf16_to_f32(kp[0]), b = bf16_to_f32(kp[1]);
        kp[0] = f32_to_bf16(a * re - b * im);
        kp[1] = f32_to_bf16(a * im + b * re);
    } else {
        float* kp = reinterpret_cast<float*>(p.kc) + ci;
        const float a = kp[0], b = kp[1];
        kp[0] = a * re - b * im;
        kp[1] = a * im + b * re;
    }
}

// qk_layer_norm (include/acmi.h, acmi_lm_layer.q_ln_g ..; transformer.py:216-222, 358-360, 388-392) as a launch of its own after
// the QKV GEMM and before the rotary positions -- like those, an option no released model switches on, kept out of the GEMM
// epilogues.  One workgroup per (row, operand): blockIdx.y = 0 normalises the q row in place (row-major f32), blockIdx.y = 1
// the k row the GEMM has just appended to the cache, whose d features lie in H pieces of hd (one per head).  Two-pass
// statistics over the full model dimension, biased variance, like nn.LayerNorm.
struct QkLnArgs {
    float* q; void* kc; int kv_bf16; int H, hd, Tcap, d, rpp; const int* pos;
    const float* qg; const float* qb; const float* kg; const float* kb; float eps;
    int npos_pad, npos;   // > 0: position-minor rows (see EmbedArgs); pad rows are skipped
};

__global__ __launch_bounds__(256) void qk_ln_kernel(const QkLnArgs p) {
    __shared__ float sred[4];
    const int gm = blockIdx.x;
    const bool is_k = blockIdx.y == 1;
    int pidx, brow;
    if (p.npos_pad > 0) { brow = gm / p.npos_pad; pidx = gm - brow * p.npos_pad; if (pidx >= p.npos) return; }
    else { pidx = gm / p.rpp; brow = gm - pidx * p.rpp; }
    const int tpos = is_k ? *p.pos + pidx : 0;
    const float* g = is_k ? p.kg : p.qg;
    const float* b = is_k ? p.kb : p.qb;
    float loc[8];   // d <= 2048
    size_t at[8];
    float sum = 0.f;
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = (int)threadIdx.x + i * 256;
        if (c >= p.d) break;
        if (is_k) {
            const int h = c / p.hd, j = c - h * p.hd;
            at[i] = (((size_t)brow * p.H + h) * p.Tcap + tpos) * p.hd + j;
            loc[i] = p.kv_bf16 ? bf16_to_f32(reinterpret_cast<const bf16_t*>(p.kc)[at[i]]) : reinterpret_cast<const float*>(p.kc)[at[i]];
        } else {
            at[i] = (size_t)gm * p.d + c;
            loc[i] = p.q[at[i]];
        }
        sum += loc[i];
        cnt = i + 1;
    }
    const float mean = block_sum(sum, sred) / (float)p.d;
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i >= cnt) break;
        m2 += (loc[i] - mean) * (loc[i] - mean);
    }
    const float rstd = 1.0f / sqrtf(block_sum(m2, sred) / (float)p.d + p.eps);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i >= cnt) break;
        const int c = (int)threadIdx.x + i * 256;
        float v = (loc[i] - mean) * rstd;
        if (g != nullptr) v *= g[c];
        if (b != nullptr) v += b[c];
        if (!is_k) p.q[at[i]] = v;
        else if (p.kv_bf16) reinterpret_cast<bf16_t*>(p.kc)[at[i]] = f32_to_bf16(v);
        else reinterpret_cast<float*>(p.kc)[at[i]] = v;
    }
}

// q (and, with kc, the freshly appended k rows) of M rows of this call
static int launch_qk_ln(const QkLnArgs& a, int M, bool with_k, hipStream_t st) {
    ACMI_REQUIRE(a.d > 0 && a.d <= 2048 && M > 0, "qk_layer_norm: d=%d (<= 2048) M=%d", a.d, M);
    hipLaunchKernelGGL(qk_ln_kernel, dim3(M, with_k ? 2 : 1), dim3(256), 0, st, a);
    return acmi_check_launch("qk_ln_kernel");
}

extern "C" int acmi_layer_norm_rows(const float* x, const float* gamma, const float* beta, float* y, int M, int d, float eps,
                                    void* stream) {
    ACMI_REQUIRE(x && y && M > 0 && d > 0 && d <= 2048, "acmi_layer_norm_rows: bad arguments (M=%d d=%d, d <= 2048)", M, d);
    hipStream_t st = (hipStream_t)stream;
    if (y != x) {
        hipError_t e = hipMemcpyAsync(y, x, (size_t)M * d * sizeof(float), hipMemcpyDeviceToDevice, st);
        ACMI_REQUIRE(e == hipSuccess, "acmi_layer_norm_rows: copy failed: %s", hipGetErrorString(e));
    }
    QkLnArgs a = {};
    a.q = y; a.d = d; a.rpp = 1; a.qg = gamma; a.qb = beta; a.eps = eps;
    return launch_qk_ln(a, M, false, st);
}

// The residual stream x lives in four forms (include/acmi.h, acmi_lm_state): f32 row-major `x`, raw fragments
// xh / xl (hi / lo) and the statistics partials.  StepCtx tracks which fragment buffers currently hold x.
struct StepCtx {
    const acmi_lm_model* m; const acmi_lm_state* s; hipStream_t st;
    int lnm, kt, nkc_d, rbs;   // LayerNorm mode, K tile, K tiles of d, K tiles per row block of the xh buffers
    void* xh; void* xl;        // fragments of the current x
    int np, cnt;               // statistics partials of the current x: np partials of cnt elements
    int rows;                  // rows of this call (Beff x positions)
    // single-term mode (acmi_lm_state.xshift): xsh = the shift the CURRENT x fragments were stored with, nsh = the shift
    // this layer's producers store with (the means the layer's QKV launch wrote); both NULL in hi / lo mode
    const float* xsh; float* nsh;
    bool use_lo;               // bf16 hi / lo pair
    // round 4: in single-term + shift mode with <= 32 rows the LayerNorm-consuming GEMMs take their row statistics from the
    // fragments themselves (acmi_linear_desc: colsum without a_stats), so the producers of x write partials only for the
    // one consumer that is not a GEMM: the cross-attention kernel's query hook (the paired out-projection's x1)
    bool gram;
    const float* stats_in;     // the partials the next LayerNorm-consuming GEMM reads (s->stats; s->q behind the folded cross block)
};

// out = act(LayerNorm(x) W'^T + bias): folded into the GEMM, or standardisation kernel + plain GEMM
static int gemm_ln_x(StepCtx& c, LinArgs& p, const void* w, const float* bias, const float* colsum, int N) {
    const acmi_lm_model* m = c.m; const acmi_lm_state* s = c.s;
    p.a_tiled = 1; p.w = w; p.bias = bias; p.M = c.rows; p.N = N; p.K = m->dim;
    if (c.lnm == LN_FOLD) {
        p.a = c.xh; p.a_rbs = c.rbs; p.a_lo = c.use_lo ? c.xl : nullptr;
        p.a_stats = c.gram ? nullptr : c.stats_in; p.a_np = c.np; p.a_cnt = c.cnt; p.eps = m->eps; p.colsum = colsum;
        p.a_shift = c.xsh;
    } else {
        // post-norm layers consume x itself (eps < 0: fragment order, no standardisation); pre-norm: the standardised rows
        int rc = acmi_launch_ln_tile(s->x, c.xh, m->wdtype, c.rows, m->dim, m->post_norm ? -1.0f : m->eps, nullptr, 0, c.st);
        if (rc) return rc;
        p.a = c.xh;
    }
    return acmi_launch_lin(p, m->wdtype, c.st);
}

// post-norm layers (acmi_lm_model.post_norm; transformer.py:567-573): x <- LayerNorm(x) in place after a block's residual add
static int post_ln(StepCtx& c, const float* g, const float* b) {
    ACMI_REQUIRE(g != nullptr && b != nullptr, "acmi_lm_step: post_norm needs n1_g .. n2_b on every layer");
    return acmi_layer_norm_rows(c.s->x, g, b, c.s->x, c.rows, c.m->dim, c.m->eps, (void*)c.st);
}

// describes x <- x + a W^T (in place on the f32 copy), also emitted as fragments into (xh, xl) + statistics
static void gemm_produce_x_args(StepCtx& c, LinArgs& p, const void* a, int a_rbs, const void* w, int K, void* xh, void* xl,
                                const float* bias = nullptr, bool stats_needed = false) {
    const acmi_lm_model* m = c.m; const acmi_lm_state* s = c.s;
    p.bias = bias;
    p.a = a; p.a_tiled = 1; p.a_rbs = a_rbs; p.w = w; p.residual = s->x; p.out = s->x; p.out_mode = ACMI_OUT_F32;
    p.M = c.rows; p.N = m->dim; p.K = K;
    if (c.lnm == LN_FOLD) {
        p.stats_out = (c.gram && !stats_needed) ? nullptr : s->stats;
        p.xt_hi = xh; p.xt_lo = c.use_lo ? xl : nullptr; p.xt_nkc = c.rbs; p.xt_lo_nkc = c.nkc_d;
        p.xt_shift = c.nsh;
    }
}
static int gemm_produce_x(StepCtx& c, const void* a, const void* w, int K, bool w_half = false, const float* bias = nullptr) {
    LinArgs p = {};
    gemm_produce_x_args(c, p, a, 0, w, K, c.xh, c.xl, bias);
    p.w_half = w_half;
    int rc = acmi_launch_lin(p, c.m->wdtype, c.st);
    c.cnt = w_half ? 8 : 16; c.np = c.m->dim / c.cnt;   // 8-feature workgroups leave 8-element partials
    c.xsh = c.nsh;                                       // the fragments of the new x carry this layer's shift
    return rc;
}

// ---- the prompt / prefix positions through ONE forward: MFMA-tiled GEMMs + causal prefill attention (acmi_prefill.hip)
// Same arithmetic as the reference's first streaming call (lm.py:540-543): per layer LayerNorm (its own launch: rows
// standardised into the weight's element type; the affine part lives in the folded matrices) -> QKV GEMM whose epilogue
// appends K / V to the caches -> causal attention over [0, position] -> out projection accumulating onto x -> cross
// attention (the decode kernel on every row: the source is short) -> FFN.  Rows are position-minor (include/acmi.h).
// CONTRACT: the stream is empty (pos[0] == 0, a device word this host code cannot read without a sync): pf_vt / pf_tcap
// hold positions [0, npos_pad) only.  LMModel._prefill(start) routes a non-empty stream to the chunked path.
static int lm_prefill_big(const acmi_lm_model* m, const acmi_lm_state* s, hipStream_t st) {
    const int npos = s->n_pos, npp = (npos + 15) / 16 * 16, M = s->Beff * npp;
    const int d = m->dim, H = m->num_heads, hd = d / H, F = m->ffn_dim;
    const int wbf = m->wdtype == ACMI_BF16, kvbf = m->kvdtype == ACMI_BF16, kt = wbf ? 32 : 16;
    const int nkc_d = d / kt;
    ACMI_REQUIRE(d % (2 * kt) == 0 && F % (2 * kt) == 0, "acmi_lm_step: the tiled prefill needs d and ffn to be multiples of %d", 2 * kt);
    ACMI_REQUIRE(s->pf_vt != nullptr && s->pf_tcap % 32 == 0 && s->pf_tcap >= npos, "acmi_lm_step: pf_vt / pf_tcap=%d for %d positions",
                 s->pf_tcap, npos);
    ACMI_REQUIRE(m->layers[0].w_qkv != nullptr && m->layers[0].b_qkv != nullptr && m->layers[0].b_ff1 != nullptr,
                 "acmi_lm_step: the tiled prefill needs the folded-LayerNorm matrices");
    int rc;
    EmbedArgs e = {};
    for (int k = 0; k < m->n_q; ++k) e.emb[k] = m->emb[k];
    e.w_bf16 = wbf; e.gen_sequence = s->gen_sequence; e.B = s->B; e.Beff = s->Beff; e.K = m->n_q; e.S = s->S; e.card = m->card;
    e.prepend = s->prepend; e.P = s->prepend ? s->n_prepend : 0; e.pos_table = m->pos_table;
    e.pos_scale = m->positional_scale; e.pos = s->pos; e.x = s->x; e.d = d; e.stats = s->stats; e.row_off = s->row_off;
    e.npos_pad = npp; e.npos = npos;
    e.input_add = s->input_add; e.n_add = s->n_add;
#define ACMI_EMBED_CASE(KQv)                                                                        \
    if (m->n_q <= KQv) {                                                                           \
        if (e.input_add != nullptr) {                                                              \
            if (wbf) hipLaunchKernelGGL((embed_kernel<true, KQv, true>), dim3(M), dim3(256), 0, st, e);   \
            else hipLaunchKernelGGL((embed_kernel<false, KQv, true>), dim3(M), dim3(256), 0, st, e);      \
        } else if (wbf) hipLaunchKernelGGL((embed_kernel<true, KQv>), dim3(M), dim3(256), 0, st, e);      \
        else hipLaunchKernelGGL((embed_kernel<false, KQv>), dim3(M), dim3(256), 0, st, e);         \
    } else
    ACMI_EMBED_CASE(4) ACMI_EMBED_CASE(8) ACMI_EMBED_CASE(16) {}
#undef ACMI_EMBED_CASE
    if ((rc = acmi_check_launch("embed_kernel"))) return rc;

    auto big = [&](const void* a, const void* w, const float* bias, int N, int K, int epi) {
        BigArgs b = {};
        b.a = a; b.w = w; b.bias = bias; b.M = M; b.N = N; b.K = K; b.epi = epi;
        return b;
    };
    // post-norm layers (acmi_lm_model.post_norm): the GEMMs read x itself (raw fragments), each block's LayerNorm follows its
    // residual add as a launch of its own, and the cross-attention's query is projected from the LAYER INPUT
    const bool post = m->post_norm != 0;
    const float tile_eps = post ? -1.0f : m->eps;
    auto post_ln = [&](const float* g, const float* bta) {
        ACMI_REQUIRE(g != nullptr && bta != nullptr, "acmi_lm_step: post_norm needs n1_g .. n2_b on every layer");
        return acmi_layer_norm_rows(s->x, g, bta, s->x, M, d, m->eps, (void*)st);
    };
    for (int li = 0; li < m->num_layers; ++li) {
        const acmi_lm_layer& L = m->layers[li];
        if ((rc = acmi_launch_ln_tile(s->x, s->pf_xn, m->wdtype, M, d, tile_eps, nullptr, 0, st))) return rc;
        {
            BigArgs b = big(s->pf_xn, L.w_qkv, L.b_qkv, 3 * d, d, ACMI_BIG_QKV);
            b.q_out = s->q; b.k_cache = L.k_cache; b.v_cache = L.v_cache; b.vt = s->pf_vt; b.kv_bf16 = kvbf;
            b.H = H; b.hd = hd; b.Tcap = s->Tmax; b.d = d; b.npos = npos; b.npos_pad = npp; b.vt_tcap = s->pf_tcap; b.pos = s->pos;
            if ((rc = acmi_launch_big(b, m->wdtype, st))) return rc;
        }
        if (L.q_ln_g != nullptr || L.k_ln_g != nullptr) {   // qk_layer_norm on the q rows and the K rows just stored (before the rotary positions)
            ACMI_REQUIRE(L.q_ln_g != nullptr && L.k_ln_g != nullptr, "acmi_lm_step: q_ln_g and k_ln_g come together");
            QkLnArgs qa = {};
            qa.q = s->q; qa.kc = L.k_cache; qa.kv_bf16 = kvbf; qa.H = H; qa.hd = hd; qa.Tcap = s->Tmax; qa.d = d; qa.rpp = s->Beff;
            qa.pos = s->pos; qa.qg = L.q_ln_g; qa.qb = L.q_ln_b; qa.kg = L.k_ln_g; qa.kb = L.k_ln_b; qa.eps = m->eps;
            qa.npos_pad = npp; qa.npos = npos;
            if ((rc = launch_qk_ln(qa, M, true, st))) return rc;
        }
        if (m->rope_freq != nullptr) {
            RopeArgs ra = {};
            ra.q = s->q; ra.kc = L.k_cache; ra.kv_bf16 = kvbf; ra.H = H; ra.hd = hd; ra.Tcap = s->Tmax; ra.d = d; ra.rpp = s->Beff;
            ra.pos = s->pos; ra.freq = m->rope_freq; ra.decay = m->rope_decay; ra.scale = m->rope_scale; ra.base = m->rope_base;
            ra.first = s->rope_first > 0 ? s->rope_first : 0x7fffffff; ra.shift = s->rope_shift;
            ra.npos_pad = npp; ra.npos = npos;
            hipLaunchKernelGGL(rope_qk_kernel, dim3(M), dim3(d / 2), 0, st, ra);
            if ((rc = acmi_check_launch("rope_qk_kernel"))) return rc;
        }
        if (li == m->num_layers - 1) break;   // the last layer only owes its K / V: no position of this call is sampled from
        {
            PrefillAttnArgs pa = {};
            pa.q = s->q; pa.k_cache = L.k_cache; pa.vt = s->pf_vt; pa.out = s->att; pa.out_bf16 = wbf; pa.out_rbs = nkc_d;
            pa.H = H; pa.Tcap = s->Tmax; pa.vt_tcap = s->pf_tcap; pa.npos = npos; pa.npos_pad = npp; pa.pos = s->pos;
            pa.past_context = m->past_context; pa.causal = 1;
            if ((rc = acmi_launch_prefill_attn(pa, m->kvdtype, hd, s->Beff, st))) return rc;
        }
        {
            BigArgs b = big(s->att, L.w_out, L.b_out, d, d, ACMI_BIG_RESID);
            b.out = s->x; b.ldo = d;
            if ((rc = acmi_launch_big(b, m->wdtype, st))) return rc;
        }
        if (post && (rc = post_ln(L.n1_g, L.n1_b))) return rc;
        if (m->cross_attention) {
            ACMI_REQUIRE(s->Lc > 0 && L.ck_cache && L.cv_cache && L.w_cq && L.b_cq, "acmi_lm_step: cross-attention operands missing");
            // post-norm: pf_xn still holds the layer input's fragments, which is what the reference projects the query from
            if (!post && (rc = acmi_launch_ln_tile(s->x, s->pf_xn, m->wdtype, M, d, m->eps, nullptr, 0, st))) return rc;
            BigArgs b = big(s->pf_xn, L.w_cq, L.b_cq, d, d, ACMI_BIG_F32);
            b.out = s->q; b.ldo = d;
            if ((rc = acmi_launch_big(b, m->wdtype, st))) return rc;
            if (L.cq_ln_g != nullptr) {   // qk_layer_norm_cross on the queries (the keys were normalised when the cache was filled)
                QkLnArgs qa = {};
                qa.q = s->q; qa.d = d; qa.rpp = s->Beff; qa.qg = L.cq_ln_g; qa.qb = L.cq_ln_b; qa.eps = m->eps;
                if ((rc = launch_qk_ln(qa, M, false, st))) return rc;
            }
            if (L.cvt_cache != nullptr && s->cvt_tcap >= s->Lc) {
                // the same MFMA attention kernel, non-causal over the Lc source positions (V time-minor: built once per generate)
                PrefillAttnArgs pa = {};
                pa.q = s->q; pa.k_cache = L.ck_cache; pa.vt = L.cvt_cache; pa.out = s->att; pa.out_bf16 = wbf; pa.out_rbs = nkc_d;
                pa.H = H; pa.Tcap = s->Lc; pa.vt_tcap = s->cvt_tcap; pa.npos = npos; pa.npos_pad = npp; pa.pos = s->pos;
                pa.causal = 0; pa.klen = s->Lc; pa.klen_rows = s->cross_len_rows;
                if ((rc = acmi_launch_prefill_attn(pa, m->kvdtype, hd, s->Beff, st))) return rc;
            } else {
                acmi_attn_desc ca = {};
                ca.q = s->q; ca.k_cache = L.ck_cache; ca.v_cache = L.cv_cache; ca.kvdtype = m->kvdtype; ca.out = s->att;
                ca.out_mode = ACMI_OUT_TILED; ca.out_dtype = m->wdtype; ca.out_rbs = nkc_d; ca.Beff = M; ca.H = H; ca.hd = hd;
                ca.Tcap = s->Lc; ca.len = s->Lc; ca.cache_rows = s->Beff; ca.len_rows = s->cross_len_rows; ca.pos_minor_rows = npp;
                if ((rc = acmi_attn_decode_ex(&ca, (void*)st))) return rc;
            }
            BigArgs c = big(s->att, L.w_cout, L.b_cout, d, d, ACMI_BIG_RESID);
            c.out = s->x; c.ldo = d;
            if ((rc = acmi_launch_big(c, m->wdtype, st))) return rc;
            if (post && (rc = post_ln(L.nc_g, L.nc_b))) return rc;
        }
        if ((rc = acmi_launch_ln_tile(s->x, s->pf_xn, m->wdtype, M, d, tile_eps, nullptr, 0, st))) return rc;
        {
            BigArgs b = big(s->pf_xn, L.w_ff1, L.b_ff1, F, d, ACMI_BIG_TILED);
            b.out_t = s->hidden; b.out_rbs = F / kt; b.act = 1;
            if ((rc = acmi_launch_big(b, m->wdtype, st))) return rc;
            BigArgs c = big(s->hidden, L.w_ff2, L.b_ff2, d, F, ACMI_BIG_RESID);
            c.out = s->x; c.ldo = d;
            if ((rc = acmi_launch_big(c, m->wdtype, st))) return rc;
        }
        if (post && (rc = post_ln(L.n2_g, L.n2_b))) return rc;
    }
    hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(64), 0, st, s->pos, npos);
    return acmi_check_launch("advance_kernel");
}

extern "C" int acmi_lm_step(const acmi_lm_model* m, const acmi_lm_state* s, int mode, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    ACMI_REQUIRE(m && s, "acmi_lm_step: null argument");
    if (mode == ACMI_STEP_PREFILL && s->pf_xn != nullptr && s->n_pos > 1) {
        ACMI_REQUIRE(s->row_off == nullptr, "acmi_lm_step: the one-forward prefill does not take left-padded streams (row_off)");
        ACMI_REQUIRE(m->dim % m->num_heads == 0 && m->n_q <= 16 && s->use_cfg >= ACMI_CFG_NONE && s->use_cfg <= ACMI_CFG_DOUBLE &&
                     s->Beff == s->B * (s->use_cfg + 1), "acmi_lm_step: bad geometry");
        return lm_prefill_big(m, s, st);
    }
    // rows of this call: Beff per position; a PREFILL call may run several consecutive positions at once (row
    // p * Beff + b = position pos[0] + p of CFG row b): K / V of all of them are in the cache before any attends
    const int npos = (mode == ACMI_STEP_PREFILL && s->n_pos > 1) ? s->n_pos : 1;
    const int d = m->dim, H = m->num_heads, hd = d / H, M = s->Beff * npos, F = m->ffn_dim;
    ACMI_REQUIRE(d % H == 0 && d % 16 == 0 && d <= 2048 && F % 4 == 0, "acmi_lm_step: bad dims d=%d H=%d", d, H);
    ACMI_REQUIRE(m->n_q <= 16, "acmi_lm_step: n_q=%d > 16", m->n_q);
    ACMI_REQUIRE(s->use_cfg >= ACMI_CFG_NONE && s->use_cfg <= ACMI_CFG_DOUBLE, "acmi_lm_step: bad use_cfg %d", s->use_cfg);
    ACMI_REQUIRE(s->Beff == s->B * (s->use_cfg + 1), "acmi_lm_step: Beff=%d is not B=%d x %d row groups", s->Beff, s->B,
                 s->use_cfg + 1);
    ACMI_REQUIRE(mode == ACMI_STEP_PREFILL || s->n_pos <= 1, "acmi_lm_step: n_pos=%d only with ACMI_STEP_PREFILL", s->n_pos);
    ACMI_REQUIRE(s->row_off == nullptr || m->rope_freq == nullptr || (m->past_context <= 0 && s->rope_shift == 0),
                 "acmi_lm_step: row_off (left-padded streams) with rotary positions needs an unbounded context (past_context <= 0)");
    const int wbf = m->wdtype == ACMI_BF16, kvbf = m->kvdtype == ACMI_BF16;
    int rc;

    const bool post = m->post_norm != 0;
    ACMI_REQUIRE(!post || (m->cs_head == nullptr && m->layers[0].cs_qkv == nullptr && m->layers[0].w_qkvx == nullptr),
                 "acmi_lm_step: post_norm takes the plain matrices (cs_* / w_qkvx NULL)");
    StepCtx c = {};
    c.m = m; c.s = s; c.st = st; c.lnm = post ? (int)LN_TILE : ln_mode_of(m, s);
    c.kt = wbf ? 32 : 16; c.nkc_d = (d + c.kt - 1) / c.kt;
    c.rbs = s->x_rbs > 0 ? s->x_rbs : c.nkc_d;
    ACMI_REQUIRE(c.rbs >= c.nkc_d, "acmi_lm_step: x_rbs=%d < %d K tiles of d", c.rbs, c.nkc_d);
    c.xh = s->xn; c.xl = s->xlo; c.np = 1; c.cnt = d; c.rows = M; c.stats_in = s->stats;
    // bf16 fragments of x: ACMI_LN_LO=1 -> hi / lo pair; else with `xshift` (and not ACMI_LN_SHIFT=0) -> single term, per-row
    // shift (default); else ACMI_LN_LO=0 / ACMI_LN_SHIFT=0 -> single term, unshifted (A/B only); else the hi / lo pair
    static const bool shift_off = getenv("ACMI_LN_SHIFT") != nullptr && getenv("ACMI_LN_SHIFT")[0] == '0';
    const bool foldbf = c.lnm == LN_FOLD && wbf;
    const bool shifted = foldbf && s->xshift != nullptr && fold_lo_env() != 1 && !shift_off;
    c.use_lo = foldbf && !shifted && s->xlo != nullptr && (fold_lo_env() == 1 || (fold_lo_env() == -1 && !shift_off));
    static const bool gram_off = getenv("ACMI_LN_GRAM") != nullptr && getenv("ACMI_LN_GRAM")[0] == '0';   // A/B: partials
    c.gram = shifted && M <= 32 && !gram_off;
    float* const shbuf[2] = {s->xshift, s->xshift != nullptr ? s->xshift + M : nullptr};
    c.xsh = nullptr; c.nsh = nullptr;
    // Cross-attention query without a launch of its own (include/acmi.h, acmi_linear_pair): needs the folded
    // LayerNorm, the [W_cq' | W_cq' W_out] matrices, xh buffers wide enough for [x | att] and a second pair.
    static const bool pair_enabled = !(getenv("ACMI_CROSS_FUSED") != nullptr && getenv("ACMI_CROSS_FUSED")[0] == '0');
    const bool pair = pair_enabled && m->cross_attention && c.lnm == LN_FOLD && m->layers[0].w_qkvx != nullptr &&
                      m->layers[0].cq_ln_g == nullptr &&   // qk_layer_norm_cross: the query is not linear in x1 any more
                      m->layers[0].w_mq != nullptr && s->xn2 != nullptr && (!c.use_lo || s->xlo2 != nullptr) && s->r != nullptr &&
                      c.rbs >= 2 * c.nkc_d;
    void* const xh2[2] = {s->xn, s->xn2};
    void* const xl2[2] = {s->xlo, s->xlo2};
    int cur = 0;
    // Score-folded cross-attention (include/acmi.h, acmi_lm_layer.w_qkvs): the raw scores of the conditioned rows ride in the
    // QKV and the paired launch as xs_n = R * H * Lc more output features, ONE launch (acmi_crossfold.hip) replaces the
    // cross-attention launch and the cross-out GEMM.  The CALLER opts in per generate (xs_rows > 0 + the tables); measured
    // slower than the separate launches at every batch size on MI355X, so LMModel leaves xs_rows at 0 unless
    // ACMI_CROSS_FOLD=1.
    const int xs_hl = H * s->Lc, xs_n = (s->xs_rows * xs_hl + 15) / 16 * 16;   // (whole 16-feature tiles: zero rows behind R H Lc)
    const bool xs = pair && npos == 1 && s->xs_rows > 0 && s->xs_rows <= s->Beff && m->layers[0].w_qkvs != nullptr &&
                    !c.use_lo && s->cross_len_rows == nullptr && xs_hl <= 1024 && d / 16 <= 128;

    // QKV -> self-attention as one launch (acmi_lm_state.qkv_hand; acmi_attn_fused.h): the caller opts in per state, the geometry
    // decides per step; ACMI_QKV_STAGE_K=0 keeps the K half of the second round out of LDS (A/B)
    static const bool stage_k = !(getenv("ACMI_QKV_STAGE_K") != nullptr && getenv("ACMI_QKV_STAGE_K")[0] == '0');
    // rows: one 16-row block by default; the kernel also takes two (<= 32 rows: bit-identical, but measured neutral -- melody 16 x 30 s
    // RTF 102.3 vs 101.9, medium 16 x 30 s 96.6 vs 96.8: 167 VGPRs, a third of the attention workgroups wait for a slot) --
    // ACMI_QKV_ATTN_ROWS=32 lets them in
    static const int fuse_rows = getenv("ACMI_QKV_ATTN_ROWS") != nullptr ? atoi(getenv("ACMI_QKV_ATTN_ROWS")) : 16;
    const bool fuse_qkv = mode == ACMI_STEP_DECODE && npos == 1 && s->qkv_hand != nullptr && s->hand_err != nullptr && wbf && kvbf &&
                          hd == 64 && c.lnm == LN_FOLD && c.gram && M <= fuse_rows && !post && !xs && m->rope_freq == nullptr &&
                          m->past_context <= 0 && s->row_off == nullptr && c.nkc_d % 4 == 0 && c.nkc_d / 4 <= 16 && s->Tmax <= 0xffff &&
                          d % c.kt == 0;

    EmbedArgs e = {};
    for (int k = 0; k < m->n_q; ++k) e.emb[k] = m->emb[k];
    e.w_bf16 = wbf; e.gen_sequence = s->gen_sequence; e.B = s->B; e.Beff = s->Beff; e.K = m->n_q; e.S = s->S; e.card = m->card;
    e.prepend = s->prepend; e.P = s->prepend ? s->n_prepend : 0; e.pos_table = m->pos_table;
    e.pos_scale = m->positional_scale; e.pos = s->pos; e.x = s->x; e.d = d; e.stats = s->stats; e.row_off = s->row_off;
    if (c.lnm == LN_FOLD) { e.xt_hi = c.xh; e.xt_lo = c.use_lo ? c.xl : nullptr; e.xt_nkc = c.rbs; e.xt_lo_nkc = c.nkc_d; }
    e.input_add = s->input_add; e.n_add = s->n_add;
    if (shifted) { e.shift_out = shbuf[0]; c.xsh = shbuf[0]; }
#define ACMI_EMBED_CASE(KQv)                                                                        \
    if (m->n_q <= KQv) {                                                                           \
        if (e.input_add != nullptr) {                                                              \
            if (wbf) hipLaunchKernelGGL((embed_kernel<true, KQv, true>), dim3(M), dim3(256), 0, st, e);   \
            else hipLaunchKernelGGL((embed_kernel<false, KQv, true>), dim3(M), dim3(256), 0, st, e);      \
        } else if (wbf) hipLaunchKernelGGL((embed_kernel<true, KQv>), dim3(M), dim3(256), 0, st, e);      \
        else hipLaunchKernelGGL((embed_kernel<false, KQv>), dim3(M), dim3(256), 0, st, e);         \
    } else
    ACMI_EMBED_CASE(4) ACMI_EMBED_CASE(8) ACMI_EMBED_CASE(16) {}
#undef ACMI_EMBED_CASE
    if ((rc = acmi_check_launch("embed_kernel"))) return rc;

    for (int li = 0; li < m->num_layers; ++li) {
        const acmi_lm_layer& L = m->layers[li];
        const float* const sh_l = c.xsh;   // shift of the layer's input fragments (the cross-attention query's r is built on them)
        if (shifted) c.nsh = shbuf[(li + 1) & 1];
        // norm1 -> QKV ; K, V appended in place at position g, q to scratch
        if (fuse_qkv && L.q_ln_g == nullptr && L.k_ln_g == nullptr) {
            // ... and the self-attention over positions [0, g] in the SAME launch: q | k | v go through the hand-off row
            LinArgs a = {};
            a.mean_out = c.nsh;
            a.qkv = 1; a.q_out = reinterpret_cast<float*>(s->qkv_hand); a.k_cache = L.k_cache; a.v_cache = L.v_cache; a.kv_bf16 = 1;
            a.H = H; a.hd = hd; a.Tcap = s->Tmax; a.d = d; a.pos = s->pos; a.rpp = s->Beff;
            a.a_tiled = 1; a.M = c.rows; a.K = d; a.a = c.xh; a.a_rbs = c.rbs; a.a_np = c.np; a.a_cnt = c.cnt; a.eps = m->eps;
            a.a_shift = c.xsh;
            if (pair) { a.r_out = s->r; a.w = L.w_qkvx; a.bias = L.b_qkvx; a.colsum = L.cs_qkvx; a.N = 4 * d; }
            else { a.w = L.w_qkv; a.bias = L.b_qkv; a.colsum = L.cs_qkv; a.N = 3 * d; }
            FusedAttnArgs f = {};
            f.len_dev = s->pos; f.hand = reinterpret_cast<unsigned*>(s->qkv_hand); f.err = s->hand_err; f.len_bias = 1;
            f.scale = 1.0f / sqrtf((float)hd); f.stage_k = stage_k ? 1 : 0;
            if (pair) { f.out = c.xh; f.out_rbs = c.rbs; f.out_col0 = c.nkc_d * c.kt; }
            else { f.out = s->att; f.out_rbs = c.nkc_d; f.out_col0 = 0; }
            if ((rc = acmi_launch_qkv_attn(a, f, L.k_cache, L.v_cache, H, s->Tmax, st))) return rc;
            goto attn_done;
        }
        {
            LinArgs a = {};
            a.mean_out = c.nsh;   // single-term mode: the row means of x0 = the shift of this layer's producers
            a.qkv = 1; a.q_out = s->q; a.k_cache = L.k_cache; a.v_cache = L.v_cache; a.kv_bf16 = kvbf;
            a.H = H; a.hd = hd; a.Tcap = s->Tmax; a.d = d; a.pos = s->pos; a.rpp = s->Beff;
            if (xs) {     // + the x0 part of the folded cross-attention scores: R H Lc raw features -> r viewed as [M, R H Lc]
                ACMI_REQUIRE(L.w_qkvs && L.b_qkvs && L.cs_qkvs && L.w_g2 && L.xs_u && L.xs_cs && L.xs_bs,
                             "acmi_lm_step: xs_rows > 0 needs the folded cross-attention tables on every layer");
                a.r_out = s->r; a.r_ld = xs_n;
                if ((rc = gemm_ln_x(c, a, L.w_qkvs, L.b_qkvs, L.cs_qkvs, 3 * d + xs_n))) return rc;
            } else if (pair) {   // + the x0 part of the cross-attention query as a fourth, raw block of features -> r
                a.r_out = s->r;
                if ((rc = gemm_ln_x(c, a, L.w_qkvx, L.b_qkvx, L.cs_qkvx, 4 * d))) return rc;
            } else if ((rc = gemm_ln_x(c, a, L.w_qkv, L.b_qkv, L.cs_qkv, 3 * d))) return rc;
            if (L.q_ln_g != nullptr || L.k_ln_g != nullptr) {   // qk_layer_norm on the new q / k rows (before the rotary positions)
                ACMI_REQUIRE(L.q_ln_g != nullptr && L.k_ln_g != nullptr, "acmi_lm_step: q_ln_g and k_ln_g come together");
                QkLnArgs qa = {};
                qa.q = s->q; qa.kc = L.k_cache; qa.kv_bf16 = kvbf; qa.H = H; qa.hd = hd; qa.Tcap = s->Tmax; qa.d = d; qa.rpp = s->Beff;
                qa.pos = s->pos; qa.qg = L.q_ln_g; qa.qb = L.q_ln_b; qa.kg = L.k_ln_g; qa.kb = L.k_ln_b; qa.eps = m->eps;
                if ((rc = launch_qk_ln(qa, M, true, st))) return rc;
            }
            if (m->rope_freq != nullptr) {   // rotary positions on the new q / k rows (rotary models only)
                RopeArgs ra = {};
                ra.q = s->q; ra.kc = L.k_cache; ra.kv_bf16 = kvbf; ra.H = H; ra.hd = hd; ra.Tcap = s->Tmax; ra.d = d; ra.rpp = s->Beff;
                ra.pos = s->pos; ra.freq = m->rope_freq; ra.decay = m->rope_decay; ra.scale = m->rope_scale; ra.base = m->rope_base;
                ra.first = s->rope_first > 0 ? s->rope_first : 0x7fffffff; ra.shift = s->rope_shift; ra.row_off = s->row_off;
                hipLaunchKernelGGL(rope_qk_kernel, dim3(M), dim3(d / 2), 0, st, ra);
                if ((rc = acmi_check_launch("rope_qk_kernel"))) return rc;
            }
        }
        // self attention over positions [0, g]; output in A-fragment order for the out projection: into `att`,
        // or next to x ([x | att], columns d_pad ..) when the out projection is paired with the cross query
        {
        acmi_attn_desc sa = {};
        sa.q = s->q; sa.k_cache = L.k_cache; sa.v_cache = L.v_cache; sa.kvdtype = m->kvdtype;
        sa.out_mode = ACMI_OUT_TILED; sa.out_dtype = m->wdtype; sa.Beff = M; sa.H = H; sa.hd = hd; sa.Tcap = s->Tmax;
        sa.len_dev = s->pos; sa.len_bias = 1; sa.cache_rows = s->Beff; sa.past_context = m->past_context;
        sa.start_rows = s->row_off;
        if (pair) { sa.out = c.xh; sa.out_rbs = c.rbs; sa.out_col0 = c.nkc_d * c.kt; }
        else sa.out = s->att;
        if ((rc = acmi_attn_decode_ex(&sa, stream))) return rc;
        }
    attn_done:
        if (!m->cross_attention) {
            if ((rc = gemm_produce_x(c, s->att, L.w_out, d, false, L.b_out))) return rc;
            if (post && (rc = post_ln(c, L.n1_g, L.n1_b))) return rc;
        } else {
            ACMI_REQUIRE(s->Lc > 0 && L.ck_cache && L.cv_cache, "acmi_lm_step: cross-attention caches missing");
            acmi_attn_desc ca = {};
            ca.k_cache = L.ck_cache; ca.v_cache = L.cv_cache; ca.kvdtype = m->kvdtype; ca.out = s->att;
            ca.out_mode = ACMI_OUT_TILED; ca.out_dtype = m->wdtype; ca.Beff = M; ca.H = H; ca.hd = hd; ca.Tcap = s->Lc;
            ca.len = s->Lc; ca.cache_rows = s->Beff; ca.len_rows = s->cross_len_rows;
            ca.active_rows = (pair && s->cross_active_rows > 0 && s->cross_active_rows < s->Beff) ? s->cross_active_rows : 0;
            if (pair) {
                // ONE launch: x1 = x0 + att W_out^T (fragments of x1 into the other buffer pair) and
                // r += att (W_cq' W_out)^T, which completes r = x1 W_cq'^T (its x0 part came out of the QKV launch)
                LinArgs p0 = {}, p1 = {};
                const void* att_half = reinterpret_cast<const unsigned char*>(c.xh) + (size_t)c.nkc_d * 1024;  // K tile nkc_d
                gemm_produce_x_args(c, p0, att_half, c.rbs, L.w_out, d, xh2[cur ^ 1], xl2[cur ^ 1], L.b_out, true);
                p1.a = att_half; p1.a_tiled = 1; p1.a_rbs = c.rbs; p1.bias = xs ? L.b_gs : L.b_mq;
                p1.w = xs ? L.w_g2 : L.w_mq; p1.residual = s->r; p1.out = s->r; p1.out_mode = ACMI_OUT_F32; p1.M = M;
                p1.N = xs ? xs_n : d; p1.K = d;
                if ((rc = acmi_launch_pair(p0, p1, m->wdtype, st))) return rc;
                cur ^= 1; c.xh = xh2[cur]; c.xl = xl2[cur]; c.np = d / 16; c.cnt = 16; c.xsh = c.nsh;
                if (xs) {
                    // x2 = x1 + softmax(folded LayerNorm of the raw scores) U: the x-producer of this block (what
                    // gemm_produce_x would set up for the cross-out GEMM: f32 in place, fragments with this layer's shift,
                    // partials only when the consumers do not take their statistics from the fragments)
                    CrossFoldArgs f = {};
                    f.s_raw = s->r; f.s_ld = xs_n; f.stats = s->stats; f.np = c.np; f.cnt = c.cnt; f.shift = sh_l;
                    f.cs = L.xs_cs; f.bs = L.xs_bs; f.u = L.xs_u; f.x = s->x; f.bias = L.b_cout;
                    // (it reads x1's partials while it writes x2's: those go to s->q, free between the self-attention and the next
                    // layer's QKV launch, and FFN1 -- their only consumer -- is pointed there)
                    f.xt = c.xh; f.xt_nkc = c.rbs; f.xt_shift = c.nsh; f.stats_out = c.gram ? nullptr : s->q;
                    c.stats_in = s->q;
                    f.R = s->xs_rows; f.HL = xs_hl; f.Lc = s->Lc; f.d = d; f.FB = 0; f.eps = m->eps;
                    if ((rc = acmi_launch_cross_fold(f, m->wdtype, M, st))) return rc;
                    c.cnt = 16; c.np = d / 16; c.xsh = c.nsh;
                    goto ffn;
                }
                // the cross-attention kernel applies norm_cross to r from the statistics of x1; r = (x1 - shift of x0's
                // fragments) W_cq'^T, since its x0 part was accumulated on them
                ca.q = s->r; ca.q_stats = s->stats; ca.q_stats_np = c.np; ca.q_stats_cnt = c.cnt; ca.eps = m->eps;
                ca.q_colsum = L.cs_cq; ca.q_bias = L.b_cq; ca.q_shift = sh_l;
            } else {
                LinArgs a = {};
                a.out = s->q; a.out_mode = ACMI_OUT_F32;
                if (post) {
                    // the reference's post-norm layer takes the cross-attention's query from the LAYER INPUT (`src`,
                    // transformer.py:569-572): its fragments are still in xh (the QKV launch's), s->q is free (self-attention done)
                    a.a = c.xh; a.a_tiled = 1; a.w = L.w_cq; a.bias = L.b_cq; a.M = c.rows; a.N = d; a.K = d;
                    if ((rc = acmi_launch_lin(a, m->wdtype, st))) return rc;
                    if ((rc = gemm_produce_x(c, s->att, L.w_out, d, false, L.b_out))) return rc;
                    if ((rc = post_ln(c, L.n1_g, L.n1_b))) return rc;
                } else {
                    if ((rc = gemm_produce_x(c, s->att, L.w_out, d, false, L.b_out))) return rc;
                    if ((rc = gemm_ln_x(c, a, L.w_cq, L.b_cq, L.cs_cq, d))) return rc;
                }
                if (L.cq_ln_g != nullptr) {   // qk_layer_norm_cross on the queries (the keys were normalised when the cache was filled)
                    QkLnArgs qa = {};
                    qa.q = s->q; qa.d = d; qa.rpp = s->Beff; qa.qg = L.cq_ln_g; qa.qb = L.cq_ln_b; qa.eps = m->eps;
                    if ((rc = launch_qk_ln(qa, M, false, st))) return rc;
                }
                ca.q = s->q;
            }
            if ((rc = acmi_attn_decode_ex(&ca, stream))) return rc;
            if ((rc = gemm_produce_x(c, s->att, L.w_cout, d, false, L.b_cout))) return rc;
            if (post && (rc = post_ln(c, L.nc_g, L.nc_b))) return rc;
        }
    ffn:
        {   // norm2 -> linear1 + GELU -> hidden (A-fragment order) ; linear2 -> x
            LinArgs a = {};
            a.out = s->hidden; a.out_mode = ACMI_OUT_TILED; a.act = 1;
            if ((rc = gemm_ln_x(c, a, L.w_ff1, L.b_ff1, L.cs_ff1, F))) return rc;
            c.stats_in = s->stats;
            // FFN2: N = d is narrow and K = 4d long: 8-feature workgroups put it on twice the CUs (small calls only:
            // their consumers hold d / 8 statistics partials per row in registers)
            static const bool half_ok = !(getenv("ACMI_FFN2_HALF") != nullptr && getenv("ACMI_FFN2_HALF")[0] == '0');
            const bool half = half_ok && L.w_ff2h != nullptr && c.lnm == LN_FOLD && M <= 32 && d % 8 == 0 && d / 8 <= 256 &&
                              F % (2 * c.kt) == 0;
            if ((rc = gemm_produce_x(c, s->hidden, half ? L.w_ff2h : L.w_ff2, F, half, L.b_ff2))) return rc;
            if (post && (rc = post_ln(c, L.n2_g, L.n2_b))) return rc;
        }
    }
    if (mode == ACMI_STEP_DECODE) {
        LinArgs hl = {};
        hl.out = s->logits; hl.out_mode = ACMI_OUT_F32;
        if ((rc = gemm_ln_x(c, hl, m->w_head, m->b_head, m->cs_head, m->n_q * m->card))) return rc;
        SampleArgs a = {};
        a.logits = s->logits; a.B = s->B; a.K = m->n_q; a.card = m->card; a.use_cfg = s->use_cfg;
        a.cfg_coef = s->cfg_coef; a.cfg_beta = s->cfg_coef_beta; a.use_sampling = s->use_sampling; a.temp = s->temp; a.top_k = s->top_k;
        a.top_p = s->top_p; a.seed = s->seed; a.step = 0; a.pos = s->pos; a.advance = 1; a.mixed_out = s->step_logits;
        a.gen_sequence = s->gen_sequence; a.seq_mask = s->seq_mask; a.S = s->S; a.P = s->prepend ? s->n_prepend : 0;
        return launch_sample(a, st);
    }
    hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(64), 0, st, s->pos, npos);
    return acmi_check_launch("advance_kernel");
}
