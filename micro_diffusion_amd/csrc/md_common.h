// Common device helpers for the MicroDiT gfx950 (CDNA4) kernels.
// Wave = 64 lanes everywhere; nothing in this tree is written for 32-wide warps.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define MD_WAVE 64

// Error convention of the C ABI: 0 = ok, otherwise a hipError_t (or -1 for a bad argument).
#define MD_LAUNCH_CHECK()                                  \
    do {                                                   \
        hipError_t e__ = hipGetLastError();                \
        if (e__ != hipSuccess) return (int)e__;            \
    } while (0)

#define MD_BAD_ARG (-1)

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }

union U128 {
    uint4 u;
    bf16x8 h;
    bf16 e[8];
    bf16x4 h4[2];
    float f[4];
};

union U64 {
    uint2 u;
    bf16x4 h;
    bf16 e[4];
    s16x4 s;
};

__device__ __forceinline__ bf16x8 ld_bf16x8(const bf16* p) {
    U128 t;
    t.u = *reinterpret_cast<const uint4*>(p);
    return t.h;
}
__device__ __forceinline__ void st_bf16x8(bf16* p, bf16x8 v) {
    U128 t;
    t.h = v;
    *reinterpret_cast<uint4*>(p) = t.u;
}
__device__ __forceinline__ bf16x4 ld_bf16x4(const bf16* p) {
    U64 t;
    t.u = *reinterpret_cast<const uint2*>(p);
    return t.h;
}
__device__ __forceinline__ void st_bf16x4(bf16* p, bf16x4 v) {
    U64 t;
    t.h = v;
    *reinterpret_cast<uint2*>(p) = t.u;
}

// 64-lane butterfly reductions (every lane ends with the result).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// tanh-approx GELU (torch.nn.GELU(approximate='tanh')) and its derivative.
__device__ __forceinline__ float gelu_tanh_f(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    float u = k0 * (x + k1 * x * x * x);
    return 0.5f * x * (1.f + tanhf(u));
}
__device__ __forceinline__ float dgelu_tanh_f(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    float x2 = x * x;
    float u = k0 * (x + k1 * x * x2);
    float t = tanhf(u);
    float du = k0 * (1.f + 3.f * k1 * x2);
    return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * du;
}
// exact (erf) GELU used by the expert-choice MoE (reference dit.py:124).
__device__ __forceinline__ float gelu_erf_f(float x) {
    return 0.5f * x * (1.f + erff(x * 0.7071067811865476f));
}
__device__ __forceinline__ float dgelu_erf_f(float x) {
    float cdf = 0.5f * (1.f + erff(x * 0.7071067811865476f));
    float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ float dsilu_f(float x) {
    float s = 1.f / (1.f + __expf(-x));
    return s * (1.f + x * (1.f - s));
}
