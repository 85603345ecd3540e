// Lab (dev tool, not product): a "touch" kernel for the concurrent MALL prefetcher experiment (scripts/mall_prefetch_lab.py).
// Reads `bytes` bytes with `wgs` workgroups of 256 threads, 16 bytes per lane and request, `unroll` requests in flight per
// lane; the data is dropped.  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o lab/libprefetch_lab.so lab/prefetch_lab.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void touch_kernel(const u32x4* __restrict__ p, size_t n16, unsigned* sink) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned acc = 0;
    for (; i + 7 * stride < n16; i += 8 * stride) {
        u32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[i + k * stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc ^= v[k][0] ^ v[k][3];
    }
    for (; i < n16; i += stride) acc ^= p[i][0];
    if (acc == 0x12345u) *sink = acc;   // never true in practice: keeps the loads alive
}

extern "C" int lab_touch(const void* p, size_t bytes, int wgs, unsigned* sink, void* stream) {
    hipLaunchKernelGGL(touch_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, (const u32x4*)p, bytes / 16, sink);
    return (int)hipGetLastError();
}

// ---- a PERSISTENT, time-paced prefetcher: one launch beside the whole chain (one cross-stream fork per graph replay).
// Workgroup w reads its 1 / W slice of matrix k once the chip-wide clock (s_memrealtime, 100 MHz) has passed
// start + (k - lead) * period: k - lead launches of the chain are then expected to have finished.  Loose pacing is enough: the
// data only has to reach the 256 MB memory-side cache before its launch and survive there until it.
struct PacedEntry { const void* p; unsigned long long bytes; };

__global__ __launch_bounds__(256) void paced_touch_kernel(const PacedEntry* __restrict__ tab, int n, int period_ticks, int lead,
                                                          int frac_pct, unsigned* sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned acc = 0;
    for (int k = 0; k < n; ++k) {
        const long long due = (long long)(k - lead) * period_ticks;
        while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < due) __builtin_amdgcn_s_sleep(8);
        // too late for this matrix (its launch has probably started): drop it instead of falling further behind
        if ((long long)(__builtin_amdgcn_s_memrealtime() - t0) > due + (long long)lead * period_ticks) continue;
        const u32x4* p = (const u32x4*)tab[k].p;
        const size_t n16 = (size_t)(tab[k].bytes / 16 * (unsigned long long)frac_pct / 100), per = (n16 + gridDim.x - 1) / gridDim.x;
        const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n16 ? lo + per : n16;
        size_t i = lo + threadIdx.x;
        for (; i + 15 * 256 < hi; i += 16 * 256) {
            u32x4 v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = p[i + q * 256];
#pragma unroll
            for (int q = 0; q < 16; ++q) acc ^= v[q][0] ^ v[q][3];
        }
        for (; i < hi; i += 256) acc ^= p[i][0];
    }
    if (acc == 0x12345u) *sink = acc;
}

extern "C" int lab_paced_touch(const void* table, int n, int period_ticks, int lead, int wgs, int frac_pct, unsigned* sink, void* stream) {
    hipLaunchKernelGGL(paced_touch_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, (const PacedEntry*)table, n, period_ticks,
                       lead, frac_pct, sink);
    return (int)hipGetLastError();
}
