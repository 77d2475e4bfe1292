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

// 64-lane reductions (every lane ends with the result).  DPP row shifts + row broadcasts (6 VALU instructions with a DPP
// modifier, then v_readlane of lane 63): the __shfl_xor butterfly compiles to six DEPENDENT ds_bpermute_b32 -- six LDS
// round trips, ~600 cycles per reduction -- and the LayerNorm kernels do two or three reductions per 2 KB row.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<0x111, 0xf>(v);   // row_shr:1   (lanes without a source keep old = 0)
    v += dpp_f<0x112, 0xf>(v);   // row_shr:2
    v += dpp_f<0x114, 0xf>(v);   // row_shr:4
    v += dpp_f<0x118, 0xf>(v);   // row_shr:8   -> lane 15 of every row holds the row sum
    v += dpp_f<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v += dpp_f<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Activations on the hardware transcendentals (v_exp_f32, v_rcp_f32; ~1 ulp) instead of the libm routines: the fused
// GEMM epilogues evaluate 64-128 of them per lane, and erff / tanhf (40-50 instructions each, range-split) made the
// MoE fc1 epilogue as long as its k-loop.  Absolute error of these forms is <= 2e-7, three orders below bf16 rounding.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// tanh(u) = 1 - 2 / (1 + e^{2u});  e^{2u} -> inf gives 1, -> 0 gives -1.
__device__ __forceinline__ float fast_tanh(float u) { return 1.f - 2.f * fast_rcp(1.f + __expf(2.f * u)); }

// tanh-approx GELU (torch.nn.GELU(approximate='tanh')) and its derivative.
__device__ __forceinline__ float gelu_tanh_f(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    float u = k0 * (x + k1 * x * x * x);
    return 0.5f * x * (1.f + fast_tanh(u));
}
__device__ __forceinline__ float dgelu_tanh_f(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    float x2 = x * x;
    float u = k0 * (x + k1 * x * x2);
    float t = fast_tanh(u);
    float du = k0 * (1.f + 3.f * k1 * x2);
    return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * du;
}
// exact (erf) GELU used by the expert-choice MoE (reference dit.py:124).  Normal CDF through Abramowitz-Stegun 7.1.26
// (|erf error| <= 1.5e-7):  Q(|x|) = 0.5 * poly(t) * exp(-x^2 / 2), t = 1 / (1 + p |x| / sqrt 2);  Phi(x) = x < 0 ? Q : 1 - Q
// (no cancellation in the negative tail).  `pdf_e` returns exp(-x^2 / 2) for the derivative.
__device__ __forceinline__ float normal_cdf_f(float x, float& pdf_e) {
    const float z = fabsf(x) * 0.7071067811865476f;
    const float t = fast_rcp(1.f + 0.3275911f * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    pdf_e = __expf(-z * z);
    const float q = 0.5f * poly * pdf_e;
    return x < 0.f ? q : 1.f - q;
}
__device__ __forceinline__ float gelu_erf_f(float x) {
    float e;
    return x * normal_cdf_f(x, e);
}
__device__ __forceinline__ float dgelu_erf_f(float x) {
    float e;
    const float cdf = normal_cdf_f(x, e);
    return cdf + x * 0.3989422804014327f * e;
}
// The same two functions on PAIRS, for the fused GEMM epilogues (128 evaluations per lane and output tile; with the scalar
// form above the MoE fc1 epilogue needed more VALU cycles than its k-loop needs MFMA cycles).  One transcendental instead
// of two and packed fp32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32):
//   Q(u) = Phi(-u) = exp(-u^2 / 2) * R(u),  R(u) = erfcx(u / sqrt 2) / 2 ~ degree-8 polynomial on [0, 6] (weighted minimax fit,
//   scripts/fit_normal_tail.py: |Phi error| <= 4e-6, |GELU error| <= 1.6e-6 -- three orders below bf16 rounding);
//   beyond |x| = 6 the argument is clamped (Q < 1e-9 there).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void normal_tail2(f32x2 x, f32x2& q, f32x2& e) {
    const f32x2 a = {fminf(fabsf(x.x), 6.f), fminf(fabsf(x.y), 6.f)};
    f32x2 p = 2.5696488483e-05f * a + -4.5856760922e-04f;
    p = p * a + 3.5954925116e-03f;
    p = p * a + -1.6710778400e-02f;
    p = p * a + 5.3111584618e-02f;
    p = p * a + -1.2762509357e-01f;
    p = p * a + 2.4839277447e-01f;
    p = p * a + -3.9874859017e-01f;
    p = p * a + 4.9999610713e-01f;
    const f32x2 t = (a * a) * -0.7213475204444817f;          // -u^2 / 2 * log2(e)
    e = f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    q = p * e;
}
__device__ __forceinline__ f32x2 gelu_erf_2(f32x2 x) {
    f32x2 q, e;
    normal_tail2(x, q, e);
    const f32x2 r = 1.f - q;
    return x * f32x2{x.x < 0.f ? q.x : r.x, x.y < 0.f ? q.y : r.y};
}
__device__ __forceinline__ f32x2 dgelu_erf_2(f32x2 x) {
    f32x2 q, e;
    normal_tail2(x, q, e);
    const f32x2 r = 1.f - q;
    return f32x2{x.x < 0.f ? q.x : r.x, x.y < 0.f ? q.y : r.y} + (x * 0.3989422804014327f) * e;
}
// Both at once (the MoE fc1 forward epilogue with md_gemm_args.dact_cached: the backward multiplies by the stored derivative).
__device__ __forceinline__ void gelu_dgelu_erf_2(f32x2 x, f32x2& g, f32x2& d) {
    f32x2 q, e;
    normal_tail2(x, q, e);
    const f32x2 r = 1.f - q;
    const f32x2 cdf = {x.x < 0.f ? q.x : r.x, x.y < 0.f ? q.y : r.y};
    g = x * cdf;
    d = cdf + (x * 0.3989422804014327f) * e;
}
__device__ __forceinline__ float silu_f(float x) { return x * fast_rcp(1.f + __expf(-x)); }
__device__ __forceinline__ float dsilu_f(float x) {
    float s = fast_rcp(1.f + __expf(-x));
    return s * (1.f + x * (1.f - s));
}
