// MultiBandDiffusion decoder option for gfx950: the operations of the reference's diffusion
// U-Net and reverse process that are not convolutions (those run on acmi_conv1d) --
//   acmi_group_norm     nn.GroupNorm (+ the ReLU that follows it everywhere in unet.py:32-104)
//   acmi_channel_add    z += embedding(step)[b, c]                       (unet.py:176-181)
//   acmi_add_cropped    z[..., :Ts] + skip  (decoder input, unet.py:209-212), also the residual-free plain add
//   acmi_interp_add     z += F.interpolate(condition_emb, T) (nearest; unet.py:191-193)
//   acmi_ddpm_step      one step of NoiseSchedule.generate / generate_subsampled (diffusion_schedule.py:205-272)
//   acmi_fir_bank       julius.LowPassFilters as a direct FIR bank with replicate padding (SplitBands of the
//                       MultiBandProcessor and of MultiBandDiffusion.re_eq), acmi_band_stats the per-band sums re_eq needs
// All f32, activation-streaming (HBM bound) except the FIR bank (LDS / VALU).
#include "acmi_common.h"

#include <math.h>

// ---------------------------------------------------------------------------------------------------------------------
// GroupNorm: statistics over (C / G channels x T) per (batch item, group), two launches so that long signals (240 000
// samples x 12 channels at the first level) spread over the chip: (1) per-chunk (mean, M2) partials -- two-pass inside the
// chunk, exact -- (2) every block combines the partials of its (b, g) in a fixed order (Chan) and normalises its chunk.
// ---------------------------------------------------------------------------------------------------------------------
#define GN_CHUNK 4096   // elements per block: 256 threads x 16

struct GnArgs {
    const float* x; const float* gamma; const float* beta; float* y; float* part;   // part [B G][nchunks][2]
    int C, T, G, nchunks; float eps; int relu;
};

__device__ __forceinline__ float gn_block_sum(float v, float* s4) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void gn_partial_kernel(const GnArgs p) {
    __shared__ float s4[4];
    const int bg = blockIdx.y, chunk = blockIdx.x;
    const size_t n = (size_t)(p.C / p.G) * p.T, base = (size_t)bg * n;   // a group's channels are contiguous in [B, C, T]
    const size_t lo = (size_t)chunk * GN_CHUNK;
    const int cnt = (int)min((size_t)GN_CHUNK, n - lo);
    float v[16];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = (int)threadIdx.x + i * 256;
        v[i] = k < cnt ? p.x[base + lo + k] : 0.f;
        sum += v[i];
    }
    const float mean = gn_block_sum(sum, s4) / (float)cnt;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = (int)threadIdx.x + i * 256;
        if (k < cnt) q += (v[i] - mean) * (v[i] - mean);
    }
    q = gn_block_sum(q, s4);
    if (threadIdx.x == 0) {
        p.part[((size_t)bg * p.nchunks + chunk) * 2] = mean;
        p.part[((size_t)bg * p.nchunks + chunk) * 2 + 1] = q;
    }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const GnArgs p) {
    __shared__ float s4[4];
    __shared__ float stat[2];
    const int bg = blockIdx.y, chunk = blockIdx.x, g = bg % p.G;
    const int cpg = p.C / p.G;
    const size_t n = (size_t)cpg * p.T, base = (size_t)bg * n, lo = (size_t)chunk * GN_CHUNK;
    // combine the partials: counts are GN_CHUNK except for the last chunk
    float ws = 0.f;
    for (int c = threadIdx.x; c < p.nchunks; c += 256) {
        const float cntc = (float)min((size_t)GN_CHUNK, n - (size_t)c * GN_CHUNK);
        ws += p.part[((size_t)bg * p.nchunks + c) * 2] * cntc;
    }
    const float mean = gn_block_sum(ws, s4) / (float)n;
    float m2 = 0.f;
    for (int c = threadIdx.x; c < p.nchunks; c += 256) {
        const float cntc = (float)min((size_t)GN_CHUNK, n - (size_t)c * GN_CHUNK);
        const float dlt = p.part[((size_t)bg * p.nchunks + c) * 2] - mean;
        m2 += p.part[((size_t)bg * p.nchunks + c) * 2 + 1] + cntc * dlt * dlt;
    }
    m2 = gn_block_sum(m2, s4);
    if (threadIdx.x == 0) { stat[0] = mean; stat[1] = 1.0f / sqrtf(m2 / (float)n + p.eps); }
    __syncthreads();
    const float mu = stat[0], rstd = stat[1];
    const int cnt = (int)min((size_t)GN_CHUNK, n - lo);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = (int)threadIdx.x + i * 256;
        if (k >= cnt) break;
        const size_t e = lo + k;
        const int ch = g * cpg + (int)(e / p.T);
        float v = (p.x[base + e] - mu) * rstd * p.gamma[ch] + p.beta[ch];
        if (p.relu) v = fmaxf(v, 0.f);
        p.y[base + e] = v;
    }
}

extern "C" size_t acmi_group_norm_work_floats(int B, int C, int T, int groups) {
    if (B <= 0 || C <= 0 || T <= 0 || groups <= 0) return 0;
    const size_t n = (size_t)(C / groups) * T;
    return (size_t)B * groups * ((n + GN_CHUNK - 1) / GN_CHUNK) * 2;
}

int acmi_launch_gn_partial(const float* x, float* part, int B, int C, int T, int groups, int* nchunks, int* chunk, hipStream_t st) {
    ACMI_REQUIRE(B > 0 && C > 0 && T > 0 && groups > 0 && C % groups == 0, "acmi_group_norm: bad shape B=%d C=%d T=%d groups=%d", B, C, T, groups);
    ACMI_REQUIRE(B * groups <= 65535, "acmi_group_norm: B x groups = %d exceeds the grid", B * groups);
    GnArgs a = {x, nullptr, nullptr, nullptr, part, C, T, groups, 0, 0.f, 0};
    const size_t n = (size_t)(C / groups) * T;
    a.nchunks = (int)((n + GN_CHUNK - 1) / GN_CHUNK);
    *nchunks = a.nchunks; *chunk = GN_CHUNK;
    hipLaunchKernelGGL(gn_partial_kernel, dim3(a.nchunks, B * groups), dim3(256), 0, st, a);
    return acmi_check_launch("gn_partial_kernel");
}

extern "C" int acmi_group_norm(const float* x, const float* gamma, const float* beta, float* y, float* work, int B, int C, int T,
                               int groups, float eps, int relu, void* stream) {
    ACMI_REQUIRE(B > 0 && C > 0 && T > 0 && groups > 0 && C % groups == 0, "acmi_group_norm: bad shape B=%d C=%d T=%d groups=%d", B, C, T, groups);
    GnArgs a = {x, gamma, beta, y, work, C, T, groups, 0, eps, relu};
    const size_t n = (size_t)(C / groups) * T;
    a.nchunks = (int)((n + GN_CHUNK - 1) / GN_CHUNK);
    ACMI_REQUIRE(B * groups <= 65535, "acmi_group_norm: B x groups = %d exceeds the grid", B * groups);
    dim3 grid(a.nchunks, B * groups), block(256);
    hipLaunchKernelGGL(gn_partial_kernel, grid, block, 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(gn_apply_kernel, grid, block, 0, (hipStream_t)stream, a);
    return acmi_check_launch("gn_apply_kernel");
}

// ---------------------------------------------------------------------------------------------------------------------
// element-wise pieces of the U-Net
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void channel_add_kernel(float* z, const float* table, const int64_t* steps, int C, int T, size_t total) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t bc = i / T;
        const int b = (int)(bc / C), c = (int)(bc - (size_t)b * C);
        z[i] += table[(size_t)steps[b] * C + c];
    }
}

extern "C" int acmi_channel_add(float* z, const float* table, const int64_t* steps, int B, int C, int T, void* stream) {
    ACMI_REQUIRE(B > 0 && C > 0 && T > 0, "acmi_channel_add: bad shape");
    const size_t total = (size_t)B * C * T;
    const int blocks = (int)min((total + 255) / 256, (size_t)8192);
    hipLaunchKernelGGL(channel_add_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, z, table, steps, C, T, total);
    return acmi_check_launch("channel_add_kernel");
}

// out[b, c, t] = a[b, c, t] (row pitch Ta >= T) + s[b, c, t] (row pitch T), t < T
__global__ __launch_bounds__(256) void add_cropped_kernel(const float* a, const float* s, float* out, int Ta, int T, size_t total) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t row = i / T;
        const int t = (int)(i - row * T);
        out[i] = a[row * Ta + t] + s[i];
    }
}

extern "C" int acmi_add_cropped(const float* a, int Ta, const float* s, float* out, int rows, int T, void* stream) {
    ACMI_REQUIRE(rows > 0 && T > 0 && Ta >= T, "acmi_add_cropped: bad shape rows=%d T=%d Ta=%d", rows, T, Ta);
    const size_t total = (size_t)rows * T;
    const int blocks = (int)min((total + 255) / 256, (size_t)8192);
    hipLaunchKernelGGL(add_cropped_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, s, out, Ta, T, total);
    return acmi_check_launch("add_cropped_kernel");
}

// z[b, c, t] += ce[b, c, min(floor(t * (Tc / T)), Tc - 1)]   (F.interpolate(mode='nearest') to length T)
__global__ __launch_bounds__(256) void interp_add_kernel(float* z, const float* ce, int T, int Tc, float scale, size_t total) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t row = i / T;
        const int t = (int)(i - row * T);
        const int src = min((int)floorf((float)t * scale), Tc - 1);
        z[i] += ce[row * Tc + src];
    }
}

extern "C" int acmi_interp_add(float* z, const float* ce, int rows, int T, int Tc, void* stream) {
    ACMI_REQUIRE(rows > 0 && T > 0 && Tc > 0, "acmi_interp_add: bad shape");
    const size_t total = (size_t)rows * T;
    const int blocks = (int)min((total + 255) / 256, (size_t)8192);
    hipLaunchKernelGGL(interp_add_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, z, ce, T, Tc, (float)Tc / (float)T, total);
    return acmi_check_launch("interp_add_kernel");
}

// previous = clamp((current - c_est * (estimate * est_scale)) / sqrt_alpha + sigma * noise, -clip, clip) * out_scale
// with the reference's operation order (diffusion_schedule.py:251-268); noise may be NULL (sigma = 0), clip <= 0: none
__global__ __launch_bounds__(256) void ddpm_step_kernel(const float* cur, const float* est, const float* noise, float* out, size_t n,
                                                        float c_est, float sqrt_alpha, float sigma, float clip, float est_scale,
                                                        float out_scale) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v = (cur[i] - c_est * (est[i] * est_scale)) / sqrt_alpha;
        if (noise != nullptr) v += sigma * noise[i];
        if (clip > 0.f) v = fminf(fmaxf(v, -clip), clip);
        out[i] = v * out_scale;
    }
}

extern "C" int acmi_ddpm_step(const float* current, const float* estimate, const float* noise, float* out, size_t n, float c_est,
                              float sqrt_alpha, float sigma, float clip, float est_scale, float out_scale, void* stream) {
    ACMI_REQUIRE(n > 0 && sqrt_alpha > 0.f, "acmi_ddpm_step: bad arguments");
    const int blocks = (int)min((n + 255) / 256, (size_t)8192);
    hipLaunchKernelGGL(ddpm_step_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, current, estimate, noise, out, n, c_est,
                       sqrt_alpha, sigma, clip, est_scale, out_scale);
    return acmi_check_launch("ddpm_step_kernel");
}

// ---------------------------------------------------------------------------------------------------------------------
// FIR bank with replicate padding: y[f][row][t] = sum_k filt[f][k] * x[row][clamp(t + k - half)]
// 256 outputs per block, the span in LDS, one LDS read per tap shared by up to 8 filters per pass (weights wave uniform).
// ---------------------------------------------------------------------------------------------------------------------
#define FIR_NF 8
struct FirArgs { const float* x; const float* filt; float* y; int rows, T, nf, half, f0; };

__global__ __launch_bounds__(256) void fir_bank_kernel(const FirArgs p) {
    extern __shared__ float span[];   // [256 + 2 half]
    const int row = blockIdx.y, t0 = blockIdx.x * 256, K = 2 * p.half + 1;
    const float* xr = p.x + (size_t)row * p.T;
    for (int i = threadIdx.x; i < 256 + 2 * p.half; i += 256) span[i] = xr[min(max(t0 + i - p.half, 0), p.T - 1)];
    __syncthreads();
    float acc[FIR_NF];
#pragma unroll
    for (int f = 0; f < FIR_NF; ++f) acc[f] = 0.f;
    const int nf = min(FIR_NF, p.nf - p.f0);
    for (int k = 0; k < K; ++k) {
        const float xv = span[threadIdx.x + k];
#pragma unroll
        for (int f = 0; f < FIR_NF; ++f) {
            const float w = p.filt[(size_t)(p.f0 + min(f, nf - 1)) * K + k];   // wave uniform
            acc[f] = fmaf(w, xv, acc[f]);
        }
    }
    const int t = t0 + threadIdx.x;
    if (t < p.T)
        for (int f = 0; f < nf; ++f) p.y[((size_t)(p.f0 + f) * p.rows + row) * p.T + t] = acc[f];
}

extern "C" int acmi_fir_bank(const float* x, const float* filters, float* y, int rows, int T, int n_filters, int half, void* stream) {
    ACMI_REQUIRE(rows > 0 && T > 0 && n_filters > 0 && half >= 0, "acmi_fir_bank: bad shape");
    const size_t lds = (size_t)(256 + 2 * half) * sizeof(float);
    ACMI_REQUIRE(lds <= 64 * 1024 && rows <= 65535, "acmi_fir_bank: half=%d too long (LDS) or rows=%d > 65535", half, rows);
    for (int f0 = 0; f0 < n_filters; f0 += FIR_NF) {
        FirArgs a = {x, filters, y, rows, T, n_filters, half, f0};
        hipLaunchKernelGGL(fir_bank_kernel, dim3((T + 255) / 256, rows), dim3(256), lds, (hipStream_t)stream, a);
    }
    return acmi_check_launch("fir_bank_kernel");
}

// per-band sums of julius.SplitBands' output without materialising the bands: band 0 = low 0, band i = low i - low i-1,
// band n-1 = x - low n-2; stats[band][2] += (sum, sum of squares) over all rows and samples, accumulated in double by one
// block per band and chunk through a fixed-order second pass on the host side (chunk partials [n_bands][chunks][2] f64).
struct BandStatArgs { const float* x; const float* lows; double* part; int n_bands; size_t n; int chunks; };

__global__ __launch_bounds__(256) void band_stats_kernel(const BandStatArgs p) {
    __shared__ double sd[2][4];
    const int band = blockIdx.y, chunk = blockIdx.x;
    const size_t per = (p.n + p.chunks - 1) / p.chunks, lo = (size_t)chunk * per, hi = min(p.n, lo + per);
    const float* hi_src = band == p.n_bands - 1 ? p.x : p.lows + (size_t)band * p.n;
    const float* lo_src = band == 0 ? nullptr : p.lows + (size_t)(band - 1) * p.n;
    double s = 0.0, q = 0.0;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
        const float v = hi_src[i] - (lo_src != nullptr ? lo_src[i] : 0.f);
        s += (double)v; q += (double)v * (double)v;
    }
    for (int off = 32; off >= 1; off >>= 1) { s += __shfl_down(s, off, 64); q += __shfl_down(q, off, 64); }
    if ((threadIdx.x & 63) == 0) { sd[0][threadIdx.x >> 6] = s; sd[1][threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        p.part[((size_t)band * p.chunks + chunk) * 2] = (sd[0][0] + sd[0][1]) + (sd[0][2] + sd[0][3]);
        p.part[((size_t)band * p.chunks + chunk) * 2 + 1] = (sd[1][0] + sd[1][1]) + (sd[1][2] + sd[1][3]);
    }
}

extern "C" int acmi_band_stats(const float* x, const float* lows, double* partials, int n_bands, size_t n, int chunks, void* stream) {
    ACMI_REQUIRE(n_bands >= 2 && n > 0 && chunks > 0 && chunks <= 65535, "acmi_band_stats: bad arguments");
    BandStatArgs a = {x, lows, partials, n_bands, n, chunks};
    hipLaunchKernelGGL(band_stats_kernel, dim3(chunks, n_bands), dim3(256), 0, (hipStream_t)stream, a);
    return acmi_check_launch("band_stats_kernel");
}

// out = gain_last * x + sum_i (gain[i] - gain[i + 1]) * low[i] + offset     (any per-band gain applied to SplitBands' output
// and summed again: MultiBandProcessor.return_sample / project_sample, MultiBandDiffusion.re_eq)
__global__ __launch_bounds__(256) void band_mix_kernel(const float* x, const float* lows, const float* gains, float* out, int n_bands,
                                                       size_t n, float offset) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v = gains[n_bands - 1] * x[i] + offset;
        for (int b = 0; b < n_bands - 1; ++b) v += (gains[b] - gains[b + 1]) * lows[(size_t)b * n + i];
        out[i] = v;
    }
}

extern "C" int acmi_band_mix(const float* x, const float* lows, const float* gains, float* out, int n_bands, size_t n, float offset,
                             void* stream) {
    ACMI_REQUIRE(n_bands >= 1 && n > 0, "acmi_band_mix: bad arguments");
    const int blocks = (int)min((n + 255) / 256, (size_t)8192);
    hipLaunchKernelGGL(band_mix_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, lows, gains, out, n_bands, n, offset);
    return acmi_check_launch("band_mix_kernel");
}
