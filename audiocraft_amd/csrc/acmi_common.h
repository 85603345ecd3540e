// Shared device helpers + error plumbing for libacmi (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "acmi.h"

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------- host-side error plumbing
void acmi_set_error(const char* fmt, ...);
int acmi_check_launch(const char* what);
// GroupNorm statistics pass alone (acmi_diffusion.hip): (mean, M2) partials of chunks of *chunk elements per (batch item, group),
// part [B groups][*nchunks][2] (acmi_group_norm_work_floats floats); consumed by acmi_conv1d_gn's pack pass
int acmi_launch_gn_partial(const float* x, float* part, int B, int C, int T, int groups, int* nchunks, int* chunk, hipStream_t st);

#define ACMI_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            acmi_set_error(__VA_ARGS__);   \
            return ACMI_EINVAL;            \
        }                                  \
    } while (0)

// ---------------------------------------------------------------- bf16 <-> f32
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

__device__ __forceinline__ bf16_t f32_to_bf16(float f) {  // round to nearest even
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

// element load / store by storage type
template <typename T> __device__ __forceinline__ float ld_f32(const T* p);
template <> __device__ __forceinline__ float ld_f32<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld_f32<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }
template <typename T> __device__ __forceinline__ void st_f32(T* p, float v);
template <> __device__ __forceinline__ void st_f32<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st_f32<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }

// 8 consecutive elements -> 8 floats (16 B for bf16, 32 B for f32); p must be 16-B aligned
__device__ __forceinline__ void ld8(const bf16_t* p, float (&o)[8]) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
    o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
    o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ void ld8(const float* p, float (&o)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}

// N (= 4 or 8) consecutive elements -> floats
template <int N, typename T> __device__ __forceinline__ void ldn(const T* p, float (&o)[N]) {
    if constexpr (N == 8) {
        ld8(p, o);
    } else {
#pragma unroll
        for (int e = 0; e < N; ++e) o[e] = ld_f32<T>(p + e);
    }
}

// ---------------------------------------------------------------- wave / block reductions
// Cross-lane exchange through the DPP path of the VALU wherever the partner lies in the same row of 16 lanes: __shfl_xor
// compiles to ds_bpermute_b32, a round trip through the LDS crossbar (~60 cycles of dependent latency per step).
//   0xB1 = quad_perm [1,0,3,2] (xor 1)   0x4E = quad_perm [2,3,0,1] (xor 2)   0x128 = row_ror:8 (xor 8)
//   0x141 = row_half_mirror (i <-> 7 - i): once the quad steps are done every lane of a quad holds the same bits, so
//   the mirror partner is bit-for-bit the xor-4 partner.  The four steps below therefore equal the xor butterfly exactly.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true); }

__device__ __forceinline__ float row16_sum(float v) {   // sum over the 16 consecutive lanes of a row, in every lane
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    v += dpp_f32<0x128>(v);
    return v;
}
__device__ __forceinline__ float row8_sum(float v) {    // sum over the 8 consecutive lanes of a half row, in every lane
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f32<0xB1>(v));
    v = fmaxf(v, dpp_f32<0x4E>(v));
    v = fmaxf(v, dpp_f32<0x141>(v));
    v = fmaxf(v, dpp_f32<0x128>(v));
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}
