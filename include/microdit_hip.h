/* microdit_hip.h — C ABI of libmicrodit_hip.so: the MI355X (gfx950) kernels of the MicroDiT training path.
 *
 * The reference (SonyResearch/micro_diffusion) has no FFI layer: every kernel it runs is reached implicitly
 * through PyTorch ATen / cuBLAS / SDPA from micro_diffusion/models/{dit,utils,model}.py.  This header is the
 * boundary a maintainer binds instead (ctypes stub in INTEGRATION.md).  Each entry cites the reference code
 * whose implicit kernels it replaces.
 *
 * Conventions
 *  - every function returns int: 0 = ok, -1 = bad argument, otherwise the hipError_t of the failed launch;
 *    nothing throws across the ABI.
 *  - all memory is owned by the caller (PyTorch): arguments are raw device pointers, explicit sizes / strides
 *    in ELEMENTS as int64_t, and an explicit hipStream_t.  No allocation, no synchronisation inside.
 *  - re-entrant, no global or thread-local device state: safe to call from the autograd worker thread.
 *  - "bf16" pointers are device uint16 storage of bfloat16; "f32" are float.
 */
#ifndef MICRODIT_HIP_H
#define MICRODIT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP_PLATFORM_AMD__
typedef struct ihipStream_t* hipStream_t;
#endif

#define MD_ABI_VERSION 1
int md_abi_version(void);

/* ------------------------------------------------------------------------------------------------ GEMM */
enum md_act { MD_ACT_NONE = 0, MD_ACT_GELU_TANH = 1, MD_ACT_GELU_ERF = 2, MD_ACT_SILU = 3 };

enum md_epilogue {
    MD_EPI_STORE_BF16 = 0, /* C = bf16(act(alpha*acc + bias)); optional C2 = bf16(alpha*acc + bias)            */
    MD_EPI_RESIDUAL = 1,   /* C = bf16(res + gate[row / rows_per_sample, col] * (alpha*acc + bias)); opt. C2   */
    MD_EPI_STORE_F32 = 2,  /* C(f32)  = alpha*acc + bias                                                       */
    MD_EPI_ACCUM_F32 = 3,  /* C(f32) += alpha*acc + bias   (single writer per element)                         */
    MD_EPI_ATOMIC_F32 = 4, /* atomicAdd(C(f32), alpha*acc) (split-K weight gradients)                          */
    MD_EPI_DACT = 5        /* C = bf16(alpha*acc * act'(aux))  (dgrad through an activation)                   */
};

/* C[m,n] (+)= alpha * sum_k A(m,k) * B(n,k).  a_kcontig: A(m,k) = A[m*lda + k], else A[k*lda + m]; same for B
 * with n.  Replaces nn.Linear / einsum GEMMs: dit.py:84-89 (SwiGLU), dit.py:131-142 (MoE experts, batch = 8
 * experts), dit.py:222-225 (adaLN), utils.py:58-61, 109-111, 172-173, 225-233, and their autograd backward. */
typedef struct md_gemm_args {
    const void* A;      /* bf16 */
    const void* B;      /* bf16 */
    void* C;            /* bf16 or f32 depending on mode */
    void* C2;           /* optional bf16 second output (STORE_BF16 / RESIDUAL) */
    const void* bias;   /* optional f32 [N] */
    const void* res;    /* RESIDUAL: bf16 [M, ldr] */
    const void* gate;   /* RESIDUAL: optional bf16 [M / rows_per_sample, ldg] */
    const void* aux;    /* DACT: bf16 [M, ldaux] pre-activation */
    int64_t M, N, K;
    int64_t lda, ldb, ldc, ldc2, ldr, ldg, ldaux;
    int64_t sA, sB, sC, sC2, sBias, sAux; /* batch strides (elements) */
    int64_t rows_per_sample;
    int32_t batch;
    int32_t ksplit;
    int32_t a_kcontig, b_kcontig;
    int32_t mode; /* md_epilogue */
    int32_t act;  /* md_act */
    float alpha;
} md_gemm_args;

int md_gemm_bf16(const md_gemm_args* args, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MICRODIT_HIP_H */
