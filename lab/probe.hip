#include <hip/hip_runtime.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, -1, 0x00020000); }
__device__ __forceinline__ i32x4 make_rsrc_old(const void* p) {
    unsigned long long a = (unsigned long long)p;
    i32x4 r; r[0] = (int)(unsigned)a; r[1] = (int)(unsigned)(a >> 32); r[2] = -1; r[3] = 0x00020000; return r;
}
__global__ void k(const u32x4* in, u32x4* out, unsigned* flag, float* f) {
    auto rs = make_rsrc(in);
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, threadIdx.x * 16, 0, 16);
    u32x4 v2 = __builtin_amdgcn_raw_buffer_load_b128(rs, threadIdx.x * 16 + 4096, 0, 2);
    unsigned x = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(f + threadIdx.x, (float)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    auto ro = make_rsrc(out);
    __builtin_amdgcn_raw_buffer_store_b128(v + v2, ro, threadIdx.x * 16, 0, 16);
    __hip_atomic_fetch_add(flag + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long t = wall_clock64();
    out[1000][0] = (unsigned)t;
}
