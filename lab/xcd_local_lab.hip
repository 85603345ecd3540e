// Lab (dev tool, not product; written at the end of round 2): is a hand-off between two workgroups of the
// SAME XCD cheaper than the ~2 us (+ ~2 us to pull the payload) a memory-side hand-off costs?
//
// Why it matters: with 16 rows a GEMM workgroup pulls as many activation as weight bytes through its CU (DESIGN.md section
// 4, FFN2 on 8-feature workgroups).  A K split over TWO workgroups halves both -- if their partial sums can meet through
// the XCD's own L2 (coherent inside an XCD) instead of through the memory side.  The dispatcher places workgroup i on XCD
// i % 8 in an otherwise idle chip, so (w, w + 8) should be neighbours; the lab first RECORDS the XCC id of every
// workgroup (s_getreg XCC_ID) instead of trusting that.
//
//   mode 0  memory-side: write-through stores (sc0 sc1) + agent-scope loads          -- what section 5.1 measured
//   mode 1  L2-side: plain stores, s_waitcnt vmcnt(0), flag by an agent-scope atomic exchange; the consumer polls with an
//           atomic (executes at the L2, never in the stale L1) and reads the payload with PLAIN loads from addresses it
//           has not touched before in this launch (a fresh slot per round: no stale L1 line can exist)
//   pairing 0: partner = w ^ 8 (same XCD if the mapping is round-robin)   pairing 1: partner = w ^ 1 (different XCDs)
//
// Every payload word is checked (mode 1 with pairing 1 is EXPECTED to show stale reads: the L2s of two XCDs are not
// coherent -- the count is the evidence); every spin is bounded.  One-way latency = launch time / (2 * rounds).
// Build: hipcc --offload-arch=gfx950 -O3 -o xcd_local_lab xcd_local_lab.hip      Run: ./xcd_local_lab [workgroups=192]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

#define PAYLOAD 256          // words per hand-off (1 KB: the partial sums of one 16 x 16 tile)
#define SPIN_LIMIT 2000000

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

__global__ void probe_kernel(unsigned* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

__device__ __forceinline__ unsigned expect_val(unsigned round, unsigned from, unsigned i) { return round * 2654435761u + from * 40503u + i + 1u; }

// slots: [G][rounds][PAYLOAD] words written by workgroup g in round r;  flags: [G][rounds] (0 = not yet)
template <int MODE>
__global__ __launch_bounds__(256) void pingpong_kernel(unsigned* slots, unsigned* flags, int rounds, int pair_xor, unsigned* err) {
    const unsigned w = blockIdx.x, partner = w ^ (unsigned)pair_xor;
    const bool starts = w < partner;   // the lower id of a pair sends first
    for (int r = 0; r < rounds; ++r) {
        for (int half = 0; half < 2; ++half) {
            const bool sending = (half == 0) == starts;
            if (sending) {
                unsigned* dst = slots + ((size_t)w * rounds + r) * PAYLOAD + threadIdx.x;
                const unsigned v = expect_val(r, w, threadIdx.x);
                if (MODE == 0) asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(dst), "v"(v) : "memory");
                else *dst = v;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (threadIdx.x == 0) {
                    if (MODE == 0) asm volatile("global_store_dword %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" :: "v"(flags + (size_t)w * rounds + r), "v"(1u) : "memory");
                    else __hip_atomic_exchange(flags + (size_t)w * rounds + r, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                __shared__ int ok;
                if (threadIdx.x == 0) {
                    unsigned* f = flags + (size_t)partner * rounds + r;
                    int spins = 0;
                    unsigned seen = 0;
                    while (spins++ < SPIN_LIMIT) {
                        if (MODE == 0) asm volatile("global_load_dword %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(seen) : "v"(f) : "memory");
                        else seen = __hip_atomic_fetch_add(f, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (seen) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                    ok = seen != 0;
                    if (!seen) atomicAdd(err + 1, 1u);
                }
                __syncthreads();
                if (!ok) return;   // bounded: a lost flag ends this workgroup (its partner times out too)
                const unsigned* src = slots + ((size_t)partner * rounds + r) * PAYLOAD + threadIdx.x;
                unsigned v;
                if (MODE == 0) asm volatile("global_load_dword %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(src) : "memory");
                else v = *src;   // first touch of this address by this CU in this launch
                if (v != expect_val(r, partner, threadIdx.x)) atomicAdd(err, 1u);
                __syncthreads();
            }
        }
    }
}

template <int MODE>
static void run(const char* name, int G, int rounds, int pair_xor, unsigned* slots, unsigned* flags, unsigned* err) {
    CK(hipMemset(flags, 0, (size_t)G * rounds * 4));
    CK(hipMemset(slots, 0, (size_t)G * rounds * PAYLOAD * 4));
    CK(hipMemset(err, 0, 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(pingpong_kernel<MODE>, dim3(G), dim3(256), 0, 0, slots, flags, rounds, pair_xor, err);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned h[2];
    CK(hipMemcpy(h, err, 8, hipMemcpyDeviceToHost));
    printf("%-34s partner = w ^ %d: %7.3f us per one-way hand-off (1 KB)   mismatches %u   timeouts %u\n", name, pair_xor,
           ms * 1e3 / (2.0 * rounds), h[0], h[1]);
}

int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 192, rounds = 200;
    if (G % 16 != 0 || G > 256) { printf("workgroups must be a multiple of 16, <= 256 (one per CU: all resident)\n"); return 1; }
    unsigned *xcc, *slots, *flags, *err;
    CK(hipMalloc(&xcc, G * 4)); CK(hipMalloc(&slots, (size_t)G * rounds * PAYLOAD * 4));
    CK(hipMalloc(&flags, (size_t)G * rounds * 4)); CK(hipMalloc(&err, 8));
    hipLaunchKernelGGL(probe_kernel, dim3(G), dim3(64), 0, 0, xcc);
    std::vector<unsigned> h(G);
    CK(hipMemcpy(h.data(), xcc, G * 4, hipMemcpyDeviceToHost));
    int same8 = 0, same1 = 0;
    for (int w = 0; w < G; ++w) { same8 += h[w] == h[w ^ 8]; same1 += h[w] == h[w ^ 1]; }
    printf("XCC id of workgroups 0..15:");
    for (int w = 0; w < 16; ++w) printf(" %u", h[w]);
    printf("\npairs on the same XCD: (w, w^8) %d / %d   (w, w^1) %d / %d   [the probe launch, not the timed ones]\n", same8, G, same1, G);
    for (int px : {8, 1}) {
        run<0>("memory-side (sc0 sc1 stores / loads)", G, rounds, px, slots, flags, err);
        run<1>("L2-side (plain stores, atomic flag)", G, rounds, px, slots, flags, err);
    }
    return 0;
}
