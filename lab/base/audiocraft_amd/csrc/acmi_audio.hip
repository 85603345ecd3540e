// Audio-rate utilities either side of the generation path, gfx950: fractional resampling (the windowed-sinc polyphase FIR
// of julius.resample_frac, which audiocraft/data/audio_utils.py:54-59 calls for prompts and melodies that arrive at
// another sample rate).  One-off per generate: a direct form, one output sample per thread.
#include "acmi_common.h"

struct ResampleArgs {
    const float* x; float* y; const float* kernel;   // x [rows, T], y [rows, Tout], kernel [new_sr, K]
    int T, Tout, old_sr, new_sr, width, K;
};

// grid (ceil(frames / 256), new_sr, rows): the block shares one polyphase branch i (kernel row read as a broadcast),
// thread = output frame n; y[n * new_sr + i] = sum_j kernel[i][j] * xpad[n * old_sr + j], xpad = x replicate-padded by
// `width` in front and `width + old_sr` behind (julius ResampleFrac.forward).
__global__ __launch_bounds__(256) void resample_kernel(const ResampleArgs p) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y, row = blockIdx.z;
    const long long o = (long long)n * p.new_sr + i;
    if (o >= p.Tout) return;
    const float* xr = p.x + (size_t)row * p.T;
    const float* kr = p.kernel + (size_t)i * p.K;
    const int base = n * p.old_sr - p.width;
    float acc = 0.f;
    for (int j = 0; j < p.K; ++j) {
        const int src = min(max(base + j, 0), p.T - 1);
        acc = fmaf(kr[j], xr[src], acc);
    }
    p.y[(size_t)row * p.Tout + o] = acc;
}

extern "C" int acmi_resample_frac(const float* x, float* y, const float* kernel, int rows, int T, int Tout, int old_sr,
                                  int new_sr, int width, void* stream) {
    ACMI_REQUIRE(rows > 0 && T > 0 && Tout > 0 && old_sr > 0 && new_sr > 0 && width > 0, "acmi_resample_frac: bad shape");
    ACMI_REQUIRE(new_sr <= 65535 && rows <= 65535, "acmi_resample_frac: reduce the rates by their gcd first (new_sr=%d)", new_sr);
    ACMI_REQUIRE((long long)Tout <= ((long long)T * new_sr + old_sr - 1) / old_sr, "acmi_resample_frac: Tout=%d too long", Tout);
    ResampleArgs a = {x, y, kernel, T, Tout, old_sr, new_sr, width, 2 * width + old_sr};
    const int frames = (Tout + new_sr - 1) / new_sr;
    hipLaunchKernelGGL(resample_kernel, dim3((frames + 255) / 256, new_sr, rows), dim3(256), 0, (hipStream_t)stream, a);
    return acmi_check_launch("resample_kernel");
}
