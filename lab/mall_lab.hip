// Lab (dev tool, not product):
//  (1) is a re-read of a buffer that an earlier kernel streamed (Infinity Cache warm) faster than a cold HBM read,
//      by size and load policy (plain / non-temporal)?  -> decides whether a side-branch prefetcher can pay;
//  (2) what does the shared activation read of a skinny-GEMM workgroup cost, by size and access order?
// Build: hipcc --offload-arch=gfx950 -O3 -o mall_lab mall_lab.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// streaming read: grid-stride over 1-KB wave fragments, 8 loads in flight per lane
template <bool NT>
__global__ __launch_bounds__(256) void k_read(const u32x4* __restrict__ p, size_t nfrag, unsigned* sink) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
    unsigned acc = 0;
    for (size_t f = wave * 8; f < nfrag; f += nwaves * 8) {
        u32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const size_t ff = std::min(f + i, nfrag - 1);
            v[i] = NT ? __builtin_nontemporal_load(p + ff * 64 + lane) : p[ff * 64 + lane];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc ^= v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
    }
    if (acc == 0x9e3779b9u) sink[threadIdx.x] = acc;
}

// skinny-GEMM-like stage: G workgroups x 8 waves, NF weight fragments per wave (nt, up front) + AF activation
// fragments per wave from ONE shared buffer (every workgroup reads the same AF*8 KB), optionally rotated per workgroup
template <int NF, int AF, int ROT>
__global__ __launch_bounds__(512) void k_stage(const u32x4* __restrict__ w, const u32x4* __restrict__ act, unsigned* sink) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wg = blockIdx.x;
    u32x4 wv[NF], av[AF > 0 ? AF : 1];
    const u32x4* wb = w + ((size_t)(wg * 8 + wave) * NF) * 64 + lane;
#pragma unroll
    for (int i = 0; i < NF; ++i) wv[i] = __builtin_nontemporal_load(wb + i * 64);
    constexpr int TOT = AF * 8;
#pragma unroll
    for (int i = 0; i < AF; ++i) {
        int f = wave * AF + i;
        if (ROT == 1) f = (f + wg * 5) % TOT;            // every workgroup starts somewhere else
        if (ROT == 2) f = (wave + i * 8 + wg * 3) % TOT;  // waves interleaved + rotated
        av[i] = act[(size_t)f * 64 + lane];
    }
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < NF; ++i) acc ^= wv[i][0] ^ wv[i][1] ^ wv[i][2] ^ wv[i][3];
#pragma unroll
    for (int i = 0; i < AF; ++i) acc ^= av[i][0] ^ av[i][1] ^ av[i][2] ^ av[i][3];
    sink[(size_t)wg * 512 + threadIdx.x] = acc;
}
__global__ void k_touch(unsigned* act, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) act[i] = act[i] * 3 + 1; }

static hipStream_t s1;
static hipEvent_t ev_a, ev_b;
template <typename F>
static double time_graph(F&& body, int reps) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
    body();
    CK(hipStreamEndCapture(s1, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s1)); CK(hipStreamSynchronize(s1));
    CK(hipEventRecord(ev_a, s1));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s1));
    CK(hipEventRecord(ev_b, s1));
    CK(hipStreamSynchronize(s1));
    float ms = 0; CK(hipEventElapsedTime(&ms, ev_a, ev_b));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return ms * 1e3 / reps;
}

template <int NF, int AF, int ROT>
static void run_stage(const u32x4* w, size_t wbytes, unsigned* act, unsigned* sink, int G, int reps, const char* name) {
    const size_t stage_bytes = (size_t)G * 8 * NF * 1024;
    int S = (int)std::min<size_t>(160, std::max<size_t>(48, (640u << 20) / stage_bytes));
    if (stage_bytes * S > wbytes) S = (int)(wbytes / stage_bytes);
    double us = time_graph([&]() {
        for (int s = 0; s < S; ++s) {
            hipLaunchKernelGGL((k_stage<NF, AF, ROT>), dim3(G), dim3(512), 0, s1,
                               reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(w) + (size_t)s * stage_bytes),
                               reinterpret_cast<const u32x4*>(act), sink);
        }
    }, reps);
    printf("G=%3d W=%5.1f MB act=%3d KB/WG %-10s %6.2f us/launch\n", G, stage_bytes / 1e6, AF * 8, name, us / S);
}

int main(int argc, char** argv) {
    int reps = argc > 1 ? atoi(argv[1]) : 10;
    CK(hipSetDevice(0));
    CK(hipStreamCreate(&s1));
    CK(hipEventCreate(&ev_a)); CK(hipEventCreate(&ev_b));
    const size_t pool = (size_t)2048 << 20;
    char* buf; CK(hipMalloc(&buf, pool)); CK(hipMemset(buf, 0x5a, pool));
    unsigned* sink; CK(hipMalloc(&sink, 4 << 20));
    unsigned* act; CK(hipMalloc(&act, 1 << 20)); CK(hipMemset(act, 1, 1 << 20));

    printf("=== (1) cold vs warm re-read (1024 WGs x 256 threads; 'flush' = 1 GB of other data read in between)\n");
    const size_t sizes[] = {(size_t)19 << 20, (size_t)74 << 20, (size_t)148 << 20, (size_t)220 << 20, (size_t)400 << 20};
    const u32x4* X = reinterpret_cast<const u32x4*>(buf);
    const u32x4* F = reinterpret_cast<const u32x4*>(buf + ((size_t)1024 << 20));
    const size_t flush_frags = ((size_t)1000 << 20) / 1024;
    auto rd = [&](bool nt, const u32x4* p, size_t frags, int grid) {
        if (nt) hipLaunchKernelGGL(k_read<true>, dim3(grid), dim3(256), 0, s1, p, frags, sink);
        else hipLaunchKernelGGL(k_read<false>, dim3(grid), dim3(256), 0, s1, p, frags, sink);
    };
    for (size_t sz : sizes) {
        const size_t fr = sz / 1024;
        const double t_flush = time_graph([&]() { rd(false, F, flush_frags, 2048); }, reps);
        for (int first = 0; first < 2; ++first)       // policy of the FIRST (prefetching) read
            for (int second = 0; second < 2; ++second) {
                const double t1 = time_graph([&]() { rd(false, F, flush_frags, 2048); rd(first, X, fr, 1024); }, reps) - t_flush;
                const double t2 = time_graph([&]() { rd(false, F, flush_frags, 2048); rd(first, X, fr, 1024); rd(second, X, fr, 1024); }, reps) - t_flush - t1;
                printf("%4zu MB: first read (%s) %7.2f us = %5.2f TB/s | re-read (%s) %7.2f us = %5.2f TB/s\n", sz >> 20,
                       first ? "nt" : "plain", t1, sz / t1 / 1e6, second ? "nt" : "plain", t2, sz / t2 / 1e6);
            }
    }
    // a small prefetch grid (what a side branch could afford): 64 WGs
    for (size_t sz : {(size_t)74 << 20}) {
        const size_t fr = sz / 1024;
        const double t_flush = time_graph([&]() { rd(false, F, flush_frags, 2048); }, reps);
        const double t1 = time_graph([&]() { rd(false, F, flush_frags, 2048); rd(false, X, fr, 64); }, reps) - t_flush;
        printf("%4zu MB cold read by only 64 WGs: %7.2f us = %5.2f TB/s\n", sz >> 20, t1, sz / t1 / 1e6);
    }

    printf("=== (2) shared activation read of a stage (weights cold, nt; activation rewritten before every chain)\n");
    const u32x4* W = reinterpret_cast<const u32x4*>(buf);
#define RUN(NF, AF, ROT, G, name) run_stage<NF, AF, ROT>(W, pool, act, sink, G, reps, name)
    RUN(6, 0, 0, 96, "none");   RUN(6, 6, 0, 96, "48KB");    RUN(6, 12, 0, 96, "96KB");  RUN(6, 12, 1, 96, "96KB rot");  RUN(6, 12, 2, 96, "96KB rot2");
    RUN(12, 0, 0, 192, "none"); RUN(12, 6, 0, 192, "48KB");  RUN(12, 12, 0, 192, "96KB"); RUN(12, 12, 1, 192, "96KB rot"); RUN(12, 12, 2, 192, "96KB rot2");
    RUN(12, 0, 0, 144, "none"); RUN(12, 6, 0, 144, "48KB");  RUN(12, 12, 0, 144, "96KB"); RUN(12, 12, 1, 144, "96KB rot");
    RUN(24, 0, 0, 96, "none");  RUN(24, 12, 0, 96, "96KB");  RUN(24, 24, 0, 96, "192KB"); RUN(24, 24, 1, 96, "192KB rot"); RUN(24, 24, 2, 96, "192KB rot2");
    RUN(12, 12, 0, 192, "96KB");  RUN(6, 6, 0, 256, "48KB"); RUN(6, 12, 0, 256, "96KB");
    printf("done\n");
    return 0;
}
