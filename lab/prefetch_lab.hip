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
