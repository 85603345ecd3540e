// Melody front-end for gfx950 (CDNA4, wave64): ChromaExtractor.forward of the reference
// (audiocraft/modules/chroma.py:46-66) = torchaudio Spectrogram(power 2, centre reflect padding, periodic Hann, "window"
// normalisation) -> librosa chroma filterbank -> inf-norm over the chroma axis -> optional argmax one-hot.
//
// One workgroup per (frame, batch row).  The frame (n_fft <= 16384 real samples) is windowed on load, packed as
// n_fft / 2 complex numbers, transformed by a radix-2 FFT that lives entirely in LDS (64 KB at n_fft = 16384), unpacked
// to the real spectrum on the fly and contracted with the [n_chroma, n_fft/2 + 1] filterbank; the spectrogram never
// exists in memory (the reference materialises [B, 8193, 235] floats).  Twiddles e^{-2 pi i k / n_fft} come from a table
// the host computes in double precision: sincosf on the device would put ~1e-7 of phase error into every stage.
#include "acmi_common.h"

#define ACMI_CHROMA_MAXC 16

struct ChromaArgs {
    const float* wav; int T, stride;     // row b: wav + b * stride, T valid samples
    int log2n;                            // n_fft = 1 << log2n
    int pad_left, Tv;                     // T < n_fft: the row is zero padded to Tv = n_fft, pad_left zeros in front
    const float2* tw;                     // [n_fft / 2]: (cos, -sin)(2 pi k / n_fft)
    const float* fb; int C;               // [C][n_fft / 2 + 1]
    float inv_wsum2;                      // 1 / sum_n w[n]^2
    int n_frames; float* out; float* raw; int argmax;
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

__global__ __launch_bounds__(1024) void chroma_kernel(const ChromaArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2* z = reinterpret_cast<float2*>(smem);               // [M]
    const int N = 1 << p.log2n, M = N >> 1, lm = p.log2n - 1;
    float* red = reinterpret_cast<float*>(z + M);              // [16 waves][ACMI_CHROMA_MAXC]
    const int frame = blockIdx.x, b = blockIdx.y;
    const float* row = p.wav + (size_t)b * p.stride;
    const int hop = N >> 2, start = frame * hop - M;          // centre = True: frame t is centred on sample t * hop

    auto sample = [&](int n) -> float {                        // windowed sample n of this frame
        int j = start + n;
        if (j < 0) j = -j;                                     // reflect padding of n_fft / 2 (Tv > n_fft / 2 always)
        if (j >= p.Tv) j = 2 * (p.Tv - 1) - j;
        const int jj = j - p.pad_left;
        const float v = (jj >= 0 && jj < p.T) ? row[jj] : 0.f;
        // periodic Hann: 0.5 - 0.5 cos(2 pi n / N), the cosine read from the twiddle table
        const float c = n < M ? p.tw[n].x : -p.tw[n - M].x;
        return v * (0.5f - 0.5f * c);
    };
    // ---- load: z[k] = x[2k] + i x[2k+1], stored bit reversed for the decimation-in-time passes below
    for (int k = threadIdx.x; k < M; k += blockDim.x) {
        const int r = (int)(__brev((unsigned)k) >> (32 - lm));
        z[r] = make_float2(sample(2 * k), sample(2 * k + 1));
    }
    __syncthreads();
    // ---- radix-2 FFT of size M in LDS
    for (int s = 0; s < lm; ++s) {
        const int half = 1 << s;
        for (int i = threadIdx.x; i < (M >> 1); i += blockDim.x) {
            const int j = i & (half - 1), a = ((i >> s) << (s + 1)) + j, bb = a + half;
            const float2 w = p.tw[j << (lm - s)];              // e^{-2 pi i j / (2 half)} = tw[j * M / half]
            const float2 t = cmul(z[bb], w), u = z[a];
            z[a] = make_float2(u.x + t.x, u.y + t.y);
            z[bb] = make_float2(u.x - t.x, u.y - t.y);
        }
        __syncthreads();
    }
    // ---- real spectrum X[k], k = 0 .. M, power, filterbank
    float acc[ACMI_CHROMA_MAXC];
#pragma unroll
    for (int c = 0; c < ACMI_CHROMA_MAXC; ++c) acc[c] = 0.f;
    const int F = M + 1;
    for (int k = threadIdx.x; k <= M; k += blockDim.x) {
        const float2 zk = z[k & (M - 1)], zm = z[(M - k) & (M - 1)];
        const float2 e = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));      // (Z[k] + conj Z[M-k]) / 2
        const float2 o = make_float2(0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x));     // -i (Z[k] - conj Z[M-k]) / 2
        const float2 w = k < M ? p.tw[k] : make_float2(-1.f, 0.f);
        const float2 t = cmul(o, w);
        const float xr = e.x + t.x, xi = e.y + t.y;
        const float pw = (xr * xr + xi * xi) * p.inv_wsum2;
#pragma unroll
        for (int c = 0; c < ACMI_CHROMA_MAXC; ++c)
            if (c < p.C) acc[c] = fmaf(p.fb[(size_t)c * F + k], pw, acc[c]);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int c = 0; c < ACMI_CHROMA_MAXC; ++c) {
        const float v = wave_sum(acc[c]);
        if (lane == 0) red[wave * ACMI_CHROMA_MAXC + c] = v;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = threadIdx.x;
        float v = 0.f;
        if (c < p.C)
            for (int w = 0; w < nw; ++w) v += red[w * ACMI_CHROMA_MAXC + c];
        // F.normalize(p = inf, dim = chroma, eps = 1e-6), then argmax (first index on ties) -> one-hot
        const float mx = fmaxf(wave_max(c < p.C ? fabsf(v) : 0.f), 1e-6f);
        const float nv = v / mx;
        float bv = c < p.C ? nv : -INFINITY;
        int bi = c;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (c < p.C) {
            const size_t oidx = ((size_t)b * p.n_frames + frame) * p.C + c;
            p.out[oidx] = p.argmax ? (c == bi ? 1.f : 0.f) : nv;
            if (p.raw != nullptr) p.raw[oidx] = v;
        }
    }
}

extern "C" int acmi_chroma_frames(int T, int radix2_exp) {
    if (radix2_exp < 6 || radix2_exp > 14 || T < 1) return -1;
    const int N = 1 << radix2_exp;
    return 1 + (T < N ? N : T) / (N >> 2);
}

extern "C" int acmi_chroma(const float* wav, int B, int T, int wav_stride, int radix2_exp, const float* twiddle,
                           const float* fbanks, int n_chroma, int argmax, float* out, float* raw_out, void* stream) {
    ACMI_REQUIRE(radix2_exp >= 6 && radix2_exp <= 14, "acmi_chroma: n_fft = 2^%d unsupported (2^6 .. 2^14)", radix2_exp);
    ACMI_REQUIRE(B > 0 && T > 0 && wav_stride >= T, "acmi_chroma: bad shape B=%d T=%d stride=%d", B, T, wav_stride);
    ACMI_REQUIRE(n_chroma >= 1 && n_chroma <= ACMI_CHROMA_MAXC, "acmi_chroma: n_chroma=%d unsupported (<= %d)", n_chroma,
                 ACMI_CHROMA_MAXC);
    const int N = 1 << radix2_exp, M = N >> 1;
    ChromaArgs a = {};
    a.wav = wav; a.T = T; a.stride = wav_stride; a.log2n = radix2_exp;
    a.Tv = T < N ? N : T;
    a.pad_left = T < N ? (N - T) / 2 : 0;   // chroma.py:50-54: pad // 2 in front, the odd sample behind
    a.tw = reinterpret_cast<const float2*>(twiddle); a.fb = fbanks; a.C = n_chroma;
    // sum of the squared periodic Hann window: sum (0.5 - 0.5 cos)^2 = N * 3 / 8 exactly for N >= 4
    a.inv_wsum2 = 1.0f / (0.375f * (float)N);
    a.n_frames = 1 + a.Tv / (N >> 2); a.out = out; a.raw = raw_out; a.argmax = argmax;
    const size_t lds = (size_t)M * sizeof(float2) + 16 * ACMI_CHROMA_MAXC * sizeof(float);
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&chroma_kernel),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;   // once, thread safe
    if (!attr_ok) {
        acmi_set_error("acmi_chroma: cannot raise the dynamic LDS limit");
        return ACMI_ELAUNCH;
    }
    const int threads = M >= 2048 ? 1024 : (M >= 512 ? 256 : 64);
    hipLaunchKernelGGL(chroma_kernel, dim3(a.n_frames, B), dim3(threads), lds, (hipStream_t)stream, a);
    return acmi_check_launch("chroma_kernel");
}
