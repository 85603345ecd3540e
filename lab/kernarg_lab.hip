// Lab (dev tool, not product): what does a kernel pay for its ARGUMENTS at the head of every wave?
// A decode position is ~340 dependent launches, each with its own 320-byte kernarg block that is read once per replay,
// ~3 GB of weight / KV traffic after its previous use: cold in the scalar cache and in L2.  hipcc loads such a block in
// several dependent s_load rounds (SGPR pressure), all of them in front of the first HBM request of the wave.
//   empty   : no arguments                                   -> the boundary itself
//   one     : 320-byte struct, three fields, ONE s_load round
//   three   : the same fields in THREE dependent rounds (what lin_tiled_kernel's prologue does)
//   preload : the fields arrive in SGPRs (kernarg preload, -mllvm -amdgpu-kernarg-preload-count=14): no s_load at all
// Each graph = one streaming node that evicts L2 / MALL (512 MB copy) + CHAIN dependent nodes of 256 x 512 threads; every
// kernel stamps s_memrealtime at its end, the period between consecutive stamps is the cost per launch.
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=14 -o kernarg_lab kernarg_lab.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Big { int pad0[4]; const unsigned* a; int pad1[30]; const unsigned* b; int pad2[30]; const unsigned* c; int pad3[8]; };
static_assert(sizeof(Big) >= 300, "size");

__device__ __forceinline__ void finish(unsigned long long* ts, int idx, unsigned v) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ts[idx] = __builtin_amdgcn_s_memrealtime() + v;   // the words are zero (the compiler does not know)
    }
}

__global__ __launch_bounds__(512) void k_empty(unsigned long long* ts, int idx) { finish(ts, idx, 0); }

__global__ __launch_bounds__(512) void k_one(const Big p, unsigned long long* ts, int idx) {
    const unsigned v = *p.a + *p.b + *p.c;   // pointers from one s_load round, then three scalar data loads (warm: 3 words)
    finish(ts, idx, v);
}

// three dependent rounds: the offset of each round is made opaque and dependent on the previous round's value
__global__ __launch_bounds__(512) void k_three(const Big p, unsigned long long* ts, int idx) {
    typedef const unsigned* const __attribute__((address_space(4)))* slot_t;
    const char __attribute__((address_space(4)))* ka = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    const unsigned* pa = *(slot_t)(ka + offsetof(Big, a));
    unsigned z = (unsigned)(size_t)pa;
    asm volatile("s_and_b32 %0, %0, 0" : "+s"(z));
    const unsigned* pb = *(slot_t)(ka + (offsetof(Big, b) + z));
    unsigned z2 = (unsigned)(size_t)pb;
    asm volatile("s_and_b32 %0, %0, 0" : "+s"(z2));
    const unsigned* pc = *(slot_t)(ka + (offsetof(Big, c) + z2));
    const unsigned v = *pa + *pb + *pc;
    finish(ts, idx, v);
}

__global__ __launch_bounds__(512) void k_preload(const unsigned* a, const unsigned* b, const unsigned* c, unsigned long long* ts,
                                                 int idx) {
    const unsigned v = *a + *b + *c;
    finish(ts, idx, v);
}

__global__ void k_stream(const float4* src, float4* dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main(int argc, char** argv) {
    const int CHAIN = 240, REPS = 12;
    unsigned long long* ts; unsigned* words; float4 *src, *dst;
    const size_t nstream = (512u << 20) / 16;
    CK(hipMalloc(&ts, (CHAIN + 1) * 8)); CK(hipMalloc(&words, 4096)); CK(hipMemset(words, 0, 4096));
    CK(hipMalloc(&src, nstream * 16)); CK(hipMalloc(&dst, nstream * 16)); CK(hipMemset(src, 1, nstream * 16));
    int rate_khz = 0; CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    printf("wall clock rate %d kHz\n", rate_khz);
    hipStream_t st; CK(hipStreamCreate(&st));
    const char* names[4] = {"empty", "one", "three", "preload"};
    for (int cold = 0; cold < 2; ++cold)
    for (int mode = 0; mode < 4; ++mode) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        if (cold) hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, st, src, dst, nstream);
        hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, st, ts, 0);
        for (int i = 0; i < CHAIN; ++i) {
            Big p = {}; p.a = words + (i % 64) * 4; p.b = words + 256 + (i % 64) * 4; p.c = words + 512 + (i % 64) * 4;
            if (mode == 0) hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, st, ts, i + 1);
            if (mode == 1) hipLaunchKernelGGL(k_one, dim3(256), dim3(512), 0, st, p, ts, i + 1);
            if (mode == 2) hipLaunchKernelGGL(k_three, dim3(256), dim3(512), 0, st, p, ts, i + 1);
            if (mode == 3) hipLaunchKernelGGL(k_preload, dim3(256), dim3(512), 0, st, p.a, p.b, p.c, ts, i + 1);
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        std::vector<double> per;
        std::vector<unsigned long long> h(CHAIN + 1);
        for (int r = 0; r < REPS; ++r) {
            CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
            CK(hipMemcpy(h.data(), ts, (CHAIN + 1) * 8, hipMemcpyDeviceToHost));
            if (r >= 2) per.push_back((double)(h[CHAIN] - h[0]) / CHAIN / (rate_khz * 1e-3));
        }
        std::sort(per.begin(), per.end());
        printf("%s %-8s : %.3f us per launch (median of %d replays; min %.3f max %.3f)\n", cold ? "cold" : "warm", names[mode],
               per[per.size() / 2], (int)per.size(), per.front(), per.back());
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
