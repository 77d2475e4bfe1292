// Shared by the two GEMM translation units (gemm.hip: 2-stage tile kernels; gemm_pp.hip: persistent ping-pong kernel).
#pragma once
#include "md_common.h"
#include "../../include/microdit_hip.h"

constexpr int BKT = 64;            // k-tile depth of every GEMM kernel

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) void glb_void_t;

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == MD_ACT_GELU_TANH) return gelu_tanh_f(v);
    if (act == MD_ACT_GELU_ERF) return gelu_erf_f(v);
    if (act == MD_ACT_SILU) return silu_f(v);
    return v;
}
__device__ __forceinline__ float apply_dact(float v, int act) {
    if (act == MD_ACT_GELU_TANH) return dgelu_tanh_f(v);
    if (act == MD_ACT_GELU_ERF) return dgelu_erf_f(v);
    if (act == MD_ACT_SILU) return dsilu_f(v);
    return 1.f;
}

// gemm_pp.hip
bool md_gemm_pp_eligible(const md_gemm_args* a);
bool md_gemm_pp_shape_ok(const md_gemm_args* a, int epi);
int md_gemm_pp_launch(const md_gemm_args* a, hipStream_t stream);

// gemm_w4.hip
bool md_gemm_w4_eligible(const md_gemm_args* a);
int md_gemm_w4_launch(const md_gemm_args* a, hipStream_t stream);
