// Lab (dev tool, not product): does gfx950 dispatch a kernel launched with hipExtAnyOrderLaunch WITHOUT the AQL barrier
// bit, i.e. while its predecessor on the same stream is still running (what a programmatic dependent launch would need)?
//   period : N kernels of 256 x 512 threads back to back on one stream, each stamping s_memrealtime when it ends:
//            period between stamps with flags = 0 (in order: one kernel boundary each) and flags = hipExtAnyOrderLaunch
//   overlap: kernel A spins SPIN us, kernel B (launched right behind it) stamps its entry: entry(B) - entry(A) << SPIN
//            means B was dispatched beside A
// Build: hipcc --offload-arch=gfx950 -O3 -o anyorder_lab anyorder_lab.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(512) void k_stamp(unsigned long long* ts, int idx) {
    if (blockIdx.x == 0 && threadIdx.x == 0) ts[idx] = __builtin_amdgcn_s_memrealtime();
}

__global__ __launch_bounds__(512) void k_spin(unsigned long long* ts, int idx, int ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) ts[idx] = t0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(2);
    if (blockIdx.x == 0 && threadIdx.x == 0) ts[idx + 1] = __builtin_amdgcn_s_memrealtime();
}

static double period(hipStream_t st, unsigned long long* d_ts, int n, int flags, int grid) {
    std::vector<unsigned long long> h(n);
    CK(hipMemsetAsync(d_ts, 0, n * sizeof(unsigned long long), st));
    for (int i = 0; i < n; ++i) {
        int idx = i;
        void* args[] = {&d_ts, &idx};
        CK(hipExtLaunchKernel((const void*)k_stamp, dim3(grid), dim3(512), args, 0, st, nullptr, nullptr, flags));
    }
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h.data(), d_ts, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    // second half only (the host is ahead by then); stamps need not be monotonic in any-order mode
    const unsigned long long lo = *std::min_element(h.begin() + n / 2, h.end()), hi = *std::max_element(h.begin() + n / 2, h.end());
    return (hi - lo) / 100.0 / (n - n / 2 - 1);
}

int main() {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    unsigned long long* d_ts;
    const int N = 4000;
    CK(hipMalloc(&d_ts, N * sizeof(unsigned long long)));
    for (int grid : {1, 256, 1024}) {
        for (int rep = 0; rep < 2; ++rep) {
            const double p0 = period(st, d_ts, N, 0, grid);
            const double p1 = period(st, d_ts, N, hipExtAnyOrderLaunch, grid);
            printf("grid %4d x 512: in-order %.3f us per launch, any-order %.3f us per launch\n", grid, p0, p1);
        }
    }
    for (int spin_us : {10, 50}) {
        for (int flags : {0, (int)hipExtAnyOrderLaunch}) {
            unsigned long long h[4];
            CK(hipMemsetAsync(d_ts, 0, 4 * sizeof(unsigned long long), st));
            // warm both kernels, then the measured pair
            for (int pass = 0; pass < 2; ++pass) {
                int i0 = 0, i2 = 2, ticks = spin_us * 100;
                void* a0[] = {&d_ts, &i0, &ticks};
                void* a1[] = {&d_ts, &i2, &ticks};
                CK(hipExtLaunchKernel((const void*)k_spin, dim3(64), dim3(512), a0, 0, st, nullptr, nullptr, flags));
                CK(hipExtLaunchKernel((const void*)k_spin, dim3(64), dim3(512), a1, 0, st, nullptr, nullptr, flags));
                CK(hipStreamSynchronize(st));
            }
            CK(hipMemcpy(h, d_ts, sizeof(h), hipMemcpyDeviceToHost));
            printf("spin %2d us, flags %d: A [0, %.2f]  B [%.2f, %.2f] us\n", spin_us, flags, (h[1] - h[0]) / 100.0,
                   ((long long)h[2] - (long long)h[0]) / 100.0, ((long long)h[3] - (long long)h[0]) / 100.0);
        }
    }
    return 0;
}
