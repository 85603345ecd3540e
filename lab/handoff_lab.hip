// Lab (dev tool, not product): what does a dependency edge of the decode chain cost on MI355X, by form?
//   plain   : one launch per stage, stream order (what acmi_lm_step does today)
//   merged2 : two dependent stages in ONE launch; the second stage's workgroups request their weights, then
//             wait on an arrival counter of the first stage's workgroups (write-through stores + sc1 loads)
//   merged4 : four stages per launch
//   twostr  : one launch per stage, alternating between two graph branches, every edge a counter wait
// A stage = G workgroups x 8 waves; every wave requests NF 1-KB weight fragments (non-temporal) up front,
// then (after the edge) 12 activation fragments (96 KB per workgroup, written by the previous stage), checks
// them, and writes 2 KB per workgroup for the next stage.
// Build: hipcc --offload-arch=gfx950 -O3 -o handoff_lab handoff_lab.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, -1, 0x00020000);
}

struct StageArgs {
    const u32x4* w;          // this stage's weights: [G][8 waves][NF][64 lanes] u32x4
    const unsigned* act_in;  // >= 96 KB, written by stage `verify_stage`
    unsigned* act_out;       // [G * 512]
    unsigned* sink;          // [G * 512]
    unsigned* cnt_in;        // 8 words, 32 words apart (monotonic arrival counters of the producing stage)
    unsigned* cnt_out;
    int n_in;                // producer workgroups to wait for (0: ordered by the stream)
    int verify_stage;        // stage id that wrote act_in in this round, or -1
    int stage_id;
    const unsigned* round;   // device word, incremented once per replay
    unsigned* err;           // [0] mismatches, [1] timeouts
    unsigned long long* ts;  // [G][4] or NULL
};

struct MultiArgs { StageArgs st[4]; int off[5]; int n; };

__device__ __forceinline__ unsigned expect_val(unsigned round, int stage, unsigned i) {
    return round * 2654435761u + (unsigned)stage * 40503u + i;
}

template <int NF, bool FLAGS, bool ACT>
__device__ __forceinline__ void stage_body(const StageArgs& a, const int wg) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long t0 = 0, t1 = 0;
    if (threadIdx.x == 0) t0 = wall_clock64();
    // ---- weights: everything requested up front
    u32x4 wv[NF > 0 ? NF : 1];
    const u32x4* wb = a.w + ((size_t)(wg * 8 + wave) * NF) * 64 + lane;
#pragma unroll
    for (int i = 0; i < NF; ++i) wv[i] = __builtin_nontemporal_load(wb + i * 64);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned round = *a.round;
    // ---- the edge
    if (FLAGS && a.n_in > 0) {
        if (wave == 0) {
            const unsigned want = (unsigned)a.n_in * round;
            unsigned spins = 0;
            for (;;) {
                unsigned c = 0;
                if (lane < 8) c = __hip_atomic_load(a.cnt_in + lane * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int o = 4; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
                c = __builtin_amdgcn_readfirstlane(c);
                if (c >= want) break;
                if (++spins > 50000u) { if (lane == 0) atomicAdd(a.err + 1, 1u); break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) t1 = wall_clock64();
    // ---- activation written by the previous stage: 12 fragments per wave
    unsigned acc = 0, bad = 0;
    if (ACT) {
        u32x4 av[12];
        if (FLAGS) {
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.act_in);
#pragma unroll
            for (int i = 0; i < 12; ++i) av[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((wave * 12 + i) * 64 + lane) * 16, 0, 16);
        } else {
            const u32x4* ab = reinterpret_cast<const u32x4*>(a.act_in) + (wave * 12) * 64 + lane;
#pragma unroll
            for (int i = 0; i < 12; ++i) av[i] = ab[i * 64];
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const unsigned base = (unsigned)((wave * 12 + i) * 64 + lane) * 4u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc ^= av[i][e];
                if (a.verify_stage >= 0 && av[i][e] != expect_val(round, a.verify_stage, base + e)) ++bad;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NF; ++i) acc ^= wv[i][0] ^ wv[i][1] ^ wv[i][2] ^ wv[i][3];
    if (bad) atomicAdd(a.err, bad);
    // ---- outputs: 4 B per thread (an epilogue's natural store width)
    const unsigned oi = (unsigned)wg * 512u + threadIdx.x;
    a.sink[oi] = acc;
    if (FLAGS) {
        __hip_atomic_store(a.act_out + oi, expect_val(round, a.stage_id, oi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(a.cnt_out + (wg & 7) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        a.act_out[oi] = expect_val(round, a.stage_id, oi);
    }
    if (threadIdx.x == 0 && a.ts != nullptr) {
        a.ts[(size_t)wg * 4 + 0] = t0; a.ts[(size_t)wg * 4 + 1] = t1; a.ts[(size_t)wg * 4 + 2] = wall_clock64();
    }
}

template <int NF, bool FLAGS, bool ACT>
__global__ __launch_bounds__(512) void k_multi(const MultiArgs m) {
    const int b = blockIdx.x;
    int s = 0;
    while (s + 1 < m.n && b >= m.off[s + 1]) ++s;
    stage_body<NF, FLAGS, ACT>(m.st[s], b - m.off[s]);
}

__global__ void k_round(unsigned* round) { if (threadIdx.x == 0 && blockIdx.x == 0) *round += 1; }
__global__ void k_empty(unsigned* p) { if (p == (unsigned*)16) *p = 0; }
__global__ void k_spin(unsigned long long ticks, unsigned* p) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (p == (unsigned*)16) *p = 0;
}

// ------------------------------------------------------------------------------------------ host
static hipStream_t s1, s2;
static hipEvent_t ev_a, ev_b;

template <typename F>
static double time_graph(F&& body, int reps, bool two_streams) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
    if (two_streams) { hipEvent_t f; CK(hipEventCreate(&f)); CK(hipEventRecord(f, s1)); CK(hipStreamWaitEvent(s2, f, 0)); }
    body();
    if (two_streams) { hipEvent_t j; CK(hipEventCreate(&j)); CK(hipEventRecord(j, s2)); CK(hipStreamWaitEvent(s1, j, 0)); }
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

struct Bufs {
    u32x4* w; size_t wbytes;
    unsigned* act[2]; unsigned* sink; unsigned* cnt; unsigned* round; unsigned* err; unsigned long long* ts;
};

template <int NF, bool ACT>
static void run_config(Bufs& B, int G, int reps) {
    const size_t stage_bytes = (size_t)G * 8 * NF * 1024;
    int S = (int)std::min<size_t>(192, std::max<size_t>(48, (640u << 20) / std::max<size_t>(stage_bytes, 1)));
    S = (S / 4) * 4;
    if (stage_bytes * S > B.wbytes) S = (int)(B.wbytes / stage_bytes) / 4 * 4;
    printf("--- G=%d NF=%d (%.1f MB weights per stage, %d stages, act %s)\n", G, NF, stage_bytes / 1e6, S, ACT ? "96 KB/WG" : "none");
    auto stage = [&](int s, int n_in) {
        StageArgs a = {};
        a.w = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(B.w) + (size_t)s * stage_bytes);
        a.act_in = B.act[(s + 1) & 1]; a.act_out = B.act[s & 1]; a.sink = B.sink;
        a.cnt_in = B.cnt + (size_t)((s + S - 1) % S) * 256; a.cnt_out = B.cnt + (size_t)s * 256;
        a.n_in = n_in; a.verify_stage = s > 0 ? s - 1 : -1; a.stage_id = s; a.round = B.round; a.err = B.err;
        a.ts = B.ts + (size_t)s * 512 * 4;
        return a;
    };
    auto reset = [&]() {
        CK(hipMemset(B.cnt, 0, 256 * 4 * 256)); CK(hipMemset(B.round, 0, 4)); CK(hipMemset(B.err, 0, 8));
        CK(hipMemset(B.ts, 0, sizeof(unsigned long long) * 256 * 512 * 4));
    };
    auto report = [&](const char* name, double us, int launches) {
        unsigned err[2]; CK(hipMemcpy(err, B.err, 8, hipMemcpyDeviceToHost));
        printf("%-8s %8.2f us/stage  (%d launches/replay)  %.2f TB/s  mismatches=%u timeouts=%u\n", name, us / S, launches,
               stage_bytes / (us / S) / 1e6, err[0], err[1]);
    };
    // plain
    reset();
    double us = time_graph([&]() {
        hipLaunchKernelGGL(k_round, dim3(1), dim3(64), 0, s1, B.round);
        for (int s = 0; s < S; ++s) {
            MultiArgs m = {}; m.n = 1; m.st[0] = stage(s, 0); m.off[0] = 0; m.off[1] = G;
            hipLaunchKernelGGL((k_multi<NF, false, ACT>), dim3(G), dim3(512), 0, s1, m);
        }
    }, reps, false);
    report("plain", us, S);
    if (!ACT) return;
    // plain launches but with the flag-style stores / loads (cost of the write-through forms alone)
    reset();
    us = time_graph([&]() {
        hipLaunchKernelGGL(k_round, dim3(1), dim3(64), 0, s1, B.round);
        for (int s = 0; s < S; ++s) {
            MultiArgs m = {}; m.n = 1; m.st[0] = stage(s, 0); m.off[0] = 0; m.off[1] = G;
            hipLaunchKernelGGL((k_multi<NF, true, ACT>), dim3(G), dim3(512), 0, s1, m);
        }
    }, reps, false);
    report("plain+sc1", us, S);
    for (int per = 2; per <= 4; per *= 2) {
        if (G * per > 1024) continue;
        reset();
        us = time_graph([&]() {
            hipLaunchKernelGGL(k_round, dim3(1), dim3(64), 0, s1, B.round);
            for (int s = 0; s < S; s += per) {
                MultiArgs m = {}; m.n = per;
                for (int j = 0; j < per; ++j) { m.st[j] = stage(s + j, j == 0 ? 0 : G); m.off[j] = j * G; }
                m.off[per] = per * G;
                hipLaunchKernelGGL((k_multi<NF, true, ACT>), dim3(G * per), dim3(512), 0, s1, m);
            }
        }, reps, false);
        report(per == 2 ? "merged2" : "merged4", us, S / per);
        // timestamp breakdown of the last replay, first launch pair
        std::vector<unsigned long long> ts((size_t)per * 512 * 4);
        CK(hipMemcpy(ts.data(), B.ts, ts.size() * 8, hipMemcpyDeviceToHost));
        for (int j = 0; j < per; ++j) {
            unsigned long long s_min = ~0ull, s_max = 0, f_min = ~0ull, f_max = 0, e_min = ~0ull, e_max = 0;
            for (int wg = 0; wg < G; ++wg) {
                const unsigned long long* t = &ts[((size_t)j * 512 + wg) * 4];
                s_min = std::min(s_min, t[0]); s_max = std::max(s_max, t[0]);
                f_min = std::min(f_min, t[1]); f_max = std::max(f_max, t[1]);
                e_min = std::min(e_min, t[2]); e_max = std::max(e_max, t[2]);
            }
            static unsigned long long origin; if (j == 0) origin = s_min;
            printf("   stage %d of launch: start %.2f..%.2f  edge-passed %.2f..%.2f  end %.2f..%.2f us\n", j,
                   (s_min - origin) * 0.01, (s_max - origin) * 0.01, (f_min - origin) * 0.01, (f_max - origin) * 0.01,
                   (e_min - origin) * 0.01, (e_max - origin) * 0.01);
        }
    }
    // two graph branches, every edge a counter wait
    reset();
    us = time_graph([&]() {
        hipLaunchKernelGGL(k_round, dim3(1), dim3(64), 0, s1, B.round);
        hipEvent_t f; CK(hipEventCreate(&f)); CK(hipEventRecord(f, s1)); CK(hipStreamWaitEvent(s2, f, 0));
        for (int s = 0; s < S; ++s) {
            MultiArgs m = {}; m.n = 1; m.st[0] = stage(s, s == 0 ? 0 : G); m.off[0] = 0; m.off[1] = G;
            hipLaunchKernelGGL((k_multi<NF, true, ACT>), dim3(G), dim3(512), 0, (s & 1) ? s2 : s1, m);
        }
    }, reps, true);
    report("twostr", us, S);
}

int main(int argc, char** argv) {
    int reps = argc > 1 ? atoi(argv[1]) : 20;
    CK(hipSetDevice(0));
    CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    CK(hipEventCreate(&ev_a)); CK(hipEventCreate(&ev_b));
    Bufs B;
    B.wbytes = (size_t)1500 << 20;
    CK(hipMalloc(&B.w, B.wbytes)); CK(hipMemset(B.w, 0x5a, B.wbytes));
    for (int i = 0; i < 2; ++i) { CK(hipMalloc(&B.act[i], 4 << 20)); CK(hipMemset(B.act[i], 0, 4 << 20)); }
    CK(hipMalloc(&B.sink, 4 << 20));
    CK(hipMalloc(&B.cnt, 256 * 4 * 256)); CK(hipMalloc(&B.round, 4)); CK(hipMalloc(&B.err, 8));
    CK(hipMalloc(&B.ts, sizeof(unsigned long long) * 256 * 512 * 4));

    // ---- A: empty launches
    printf("=== empty kernel chains (us per launch, hipGraph, one stream)\n");
    const int geo[][2] = {{1, 64}, {16, 256}, {96, 512}, {144, 512}, {192, 512}, {256, 512}, {256, 256}, {512, 256}, {384, 256}, {1024, 256}};
    for (auto& ge : geo) {
        double us = time_graph([&]() { for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(ge[0]), dim3(ge[1]), 0, s1, B.err); }, reps, false);
        printf("grid %4d x %3d threads: %.2f us\n", ge[0], ge[1], us / 200);
    }
    {   // dynamic LDS makes a difference?
        double us = time_graph([&]() { for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 40 * 1024, s1, B.err); }, reps, false);
        printf("grid  256 x 512 threads, 40 KB dynamic LDS: %.2f us\n", us / 200);
    }
    // ---- E: do two graph branches run concurrently?
    {
        auto spin_chain = [&](bool two) {
            return time_graph([&]() {
                for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, (two && (i & 1)) ? s2 : s1, 1000ull, B.err);
            }, reps, two);
        };
        printf("=== 40 x 10-us spin kernels: one stream %.1f us, two graph branches %.1f us\n", spin_chain(false), spin_chain(true));
    }
    printf("=== stages\n");
    run_config<6, false>(B, 96, reps);
    run_config<6, true>(B, 96, reps);
    run_config<12, false>(B, 192, reps);
    run_config<12, true>(B, 192, reps);
    run_config<24, false>(B, 96, reps);
    run_config<24, true>(B, 96, reps);
    run_config<12, true>(B, 128, reps);
    run_config<6, true>(B, 256, reps);
    printf("done\n");
    return 0;
}
