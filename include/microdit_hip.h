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
 *  - re-entrant, no global or thread-local state on host or device (no environment reads, no device globals):
 *    safe to call from the autograd worker thread.
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

#define MD_ABI_VERSION 6
int md_abi_version(void);

/* ------------------------------------------------------------------------------------------------ GEMM */
enum md_act { MD_ACT_NONE = 0, MD_ACT_GELU_TANH = 1, MD_ACT_GELU_ERF = 2, MD_ACT_SILU = 3 };

enum md_epilogue {
    MD_EPI_STORE_BF16 = 0, /* C = bf16(act(alpha*acc + bias)); optional C2 = bf16(alpha*acc + bias)            */
    MD_EPI_RESIDUAL = 1,   /* C = bf16(res + gate[row / rows_per_sample, col] * (alpha*acc + bias)); opt. C2   */
    MD_EPI_STORE_F32 = 2,  /* C(f32)  = alpha*acc + bias                                                       */
    MD_EPI_ACCUM_F32 = 3,  /* C(f32) += alpha*acc + bias   (single writer per element)                         */
    MD_EPI_ATOMIC_F32 = 4, /* atomicAdd(C(f32), alpha*acc) (split-K weight gradients)                          */
    MD_EPI_DACT = 5,       /* C = bf16(alpha*acc * act'(aux))  (dgrad through an activation)                   */
    MD_EPI_SWIGLU_BWD = 6  /* ABI 6: the SwiGLU backward (dit.py:88-89) fused into the w3 data gradient da = dy W3: aux = h12
                              [M, 2N] (ldaux), C = dh12 [M, 2N] (ldc); with g = bf16(acc): C[:, n] = g * h2 * silu'(h1),
                              C[:, N + n] = g * silu(h1).  Built into the 4-wave kernel only (A K-contiguous, M % 256 == N % 256
                              == 0, no cu_limit): any other problem returns MD_NOT_ELIGIBLE and launches nothing -- the caller
                              then runs the plain data gradient + md_swiglu_bwd.                                  */
};

/* C[m,n] (+)= alpha * sum_k A(m,k) * B(n,k).  a_kcontig: A(m,k) = A[m*lda + k], else A[k*lda + m]; same for B
 * with n.  Replaces nn.Linear / einsum GEMMs: dit.py:84-89 (SwiGLU), dit.py:131-142 (MoE experts, batch = 8
 * experts), dit.py:222-225 (adaLN), utils.py:58-61, 109-111, 172-173, 225-233, and their autograd backward. */
/* One problem of a grouped launch (md_gemm_args.problems): C_p[M, N] = A_p B_p^T with the launch's K, ksplit and operand layouts.
 * The fp32 result of split s goes to C + s * sSplit + c_off (dense rows of N): the slices of a group mirror the layout of the
 * gradient tensors they are reduced into, so ONE md_splitk_reduce_flat call per contiguous run finishes the whole group. */
typedef struct md_gemm_problem {
    const void* A;
    const void* B;
    int64_t lda, ldb;
    int64_t M, N;
    int64_t c_off;   /* elements, multiple of 4 */
} md_gemm_problem;
#define MD_GEMM_MAX_PROBLEMS 8

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
    int64_t sSplit;                       /* STORE_F32 with ksplit > 1: split s writes its partial at C + s * sSplit */
    int64_t rows_per_sample;
    int32_t batch;
    int32_t ksplit;
    int32_t a_kcontig, b_kcontig;
    int32_t mode; /* md_epilogue */
    int32_t act;  /* md_act */
    float alpha;
    int32_t variant;        /* md_gemm_variant; 0 = the library picks per shape (tests force each kernel) */
    int32_t raster_group_n; /* column-tiles per L2 raster group; 0 = the library picks */
    void* timeline;         /* profiling aid, normally NULL: every workgroup of the 2-stage kernels writes 8 int64
                               {t_entry, t_prologue_done, t_loop_done, t_stores_drained (shader clock), wall clock
                               (100 MHz), XCC_ID << 32 | HW_ID, 0, 0} at timeline[linear_workgroup_id * 8] */
    int32_t* chosen_variant; /* optional HOST pointer: receives the md_gemm_variant that was actually launched */
    /* Operand lists (PP256 fp32-slice kernels only; NULL = the strided form above).  DEVICE arrays of
     * batch * ksplit * list_segments device pointers.  Item (batch b, split s) contracts, in order, the list_segments operand
     * pairs A_list[i] / B_list[i], i = (b * ksplit + s) * list_segments + j, each over K / (ksplit * list_segments) elements
     * starting at its element 0, accumulating all of them in registers before its one fp32 slice is written:
     *   list_segments = 1, ksplit = G : G separate operand pairs whose products md_splitk_reduce sums (the adaLN
     *                                   condition-vector gradient: sum over the layers of dmod_l W_l in ONE launch);
     *   list_segments = G, ksplit = 1 : ONE contraction over the concatenation of G operand pairs (the caption-token gradient
     *                                   sum_l dkv_l Wkv_l of all 28 blocks: no slices, no reduction pass). */
    const void* const* A_list;
    const void* const* B_list;
    int32_t list_segments;   /* 0 or 1 = one pair per item */
    /* Grouped launch (PP256, both operands K-strided, MD_EPI_STORE_F32 slices: the weight gradients of one DiT block that
     * contract over the same tokens -- qkv, proj, q_linear, ... -- as ONE launch instead of one launch + one reduction each,
     * every one of them too small to fill the chip without a deep split).  HOST array of n_problems <= MD_GEMM_MAX_PROBLEMS
     * entries; A / B / M / N / lda / ldb / sC of the struct are ignored, K / ksplit / sSplit are shared. */
    const md_gemm_problem* problems;
    int32_t n_problems;
    /* PP256 only: upper bound on the workgroups (= CUs) the persistent kernel occupies; 0 = all 256.  The data-parallel step
     * sets 256 - (RCCL channels) while a collective is in flight: an RCCL kernel holds one CU per channel for its whole duration
     * and a PP256 workgroup fills a CU (128 KiB LDS, every VGPR), so with 256 workgroups the ones that find their CU taken start
     * only when another workgroup has finished its whole tile list -- the launch takes up to twice as long; with the smaller grid
     * the same tiles are dealt to the CUs that are free (ceil(tiles / (256 - k)) rounds). */
    int32_t cu_limit;
    /* PP256 only, bf16-output epilogues (STORE_BF16 +- GELU, RESIDUAL, DACT): "whole rounds + split-K tail".  When the work items
     * of a launch do not make whole rounds of its G workgroups (256 tiles on the 248 CUs RCCL leaves: one round + 8 tiles, i.e.
     * TWO rounds), the r left-over tiles are cut along K into s units each (r * s <= G); workgroup u runs unit u after its own
     * whole tiles, in the same k-tile stream, and writes its 256 x 256 fp32 partial to tail_ws + u * 256 KiB; a second, small
     * launch sums the s partials of every left-over tile and applies the launch's epilogue.  The caller provides the workspace
     * (tail_ws_bytes >= 256 KiB * 256 covers every case; NULL = the form is never used).  tail_mode: 0 = the library decides
     * (only when its cost model predicts a gain), 1 = never, 2 = whenever it is structurally possible (tests, A/B runs). */
    void* tail_ws;
    int64_t tail_ws_bytes;
    int32_t tail_mode;
    int32_t* tail_used; /* optional HOST pointer: receives units per left-over tile (s) when the tail form was launched, else 0 */
    /* Activation derivative cached by the forward (MD_ACT_GELU_ERF only; the expert-choice MoE of dit.py:124,131-142, whose
     * pre-activation is needed by NOTHING but gelu' in the backward).  MD_EPI_STORE_BF16 with C2: C2 receives
     * bf16(gelu'(bf16(alpha*acc + bias))) instead of the pre-activation (the forward epilogue has Phi and the density in registers
     * anyway: +2 operations per pair).  MD_EPI_DACT: aux already holds the derivative, C = bf16(bf16(alpha*acc) * aux) -- the
     * backward epilogue loses its transcendental and its polynomial.  0 = the plain forms above. */
    int32_t dact_cached;
} md_gemm_args;

/* Kernels behind md_gemm_bf16.  AUTO applies the measured per-shape rules; a kernel that cannot
 * run the requested problem (PP256 needs K / ksplit to be a multiple of 128, N a multiple of 8, no atomics) is
 * rejected with MD_NOT_ELIGIBLE (-2; nothing was launched) rather than silently replaced; a malformed problem is -1. */
#define MD_NOT_ELIGIBLE (-2)
enum md_gemm_variant {
    MD_GEMM_AUTO = 0,
    MD_GEMM_REG128 = 1,   /* 128 x 128 tile, register-staged global -> LDS, 3 workgroups / CU                        */
    MD_GEMM_DMA128 = 2,   /* 128 x 128 tile, LDS-DMA double buffer                                                   */
    MD_GEMM_PACED256 = 3, /* 256 x 256 tile, LDS-DMA double buffer, DMA issue paced over the k-steps                 */
    MD_GEMM_PP256 = 4,    /* 256 x 256 tile, persistent, two wave groups half a phase apart, 8-slot half-tile ring   */
    MD_GEMM_W4 = 5        /* 256 x 256 tile, persistent, 4 waves x 128 x 128 (accumulators in the AGPR half of the file),
                             register-staged operands; K-contiguous x K-contiguous (nn.Linear forward) only          */
};

int md_gemm_bf16(const md_gemm_args* args, hipStream_t stream);
/* out[b] (+)= sum over the ksplit dense fp32 [M, N] slices a split-K md_gemm_bf16 left in ws (deterministic).
 * accumulate: 0 = store fp32, 1 = add to the fp32 out, 2 (ABI 6) = round the sum to bf16 and STORE it: out is then a bf16 matrix
 * (ldo / sOut in bf16 elements) -- the gradient-exchange buffer of a step that has one microbatch (configs/res_256_pretrain.yaml:24,111:
 * a rank of the 8-GPU run), which needs no fp32 accumulator pass.  Same for the flat form. */
int md_splitk_reduce(const float* ws, float* out, int64_t M, int64_t N, int64_t ldo, int64_t sOut, int32_t ksplit,
                     int32_t batch, int32_t accumulate, hipStream_t stream);

/* out[i] (+)= sum_s ws[s * slice_stride + i], i < n: the flat form for grouped launches (n % 4 == 0). */
int md_splitk_reduce_flat(const float* ws, float* out, int64_t n, int64_t slice_stride, int32_t ksplit, int32_t accumulate,
                          hipStream_t stream);

/* ------------------------------------------------------------------------------------------- LayerNorm */
/* y = LN(act(x + pos[row % pos_rows])) * w;  out = y * (1 + scale[row / rows_per_sample]) + shift[...].
 * w (f32 [C]), pos (f32 [pos_rows, C]), act, scale/shift (bf16 [samples, ldmod]) are all optional.
 * Replaces nn.LayerNorm(bias=False) under low-precision LayerNorm (utils.py:71-78, train.py:81-84), modulate
 * (utils.py:28-30), the `+ pos_embed` of dit.py:479 and the act->norm order of Mlp (utils.py:63-68). */
typedef struct md_ln_args {
    const void* x;     /* bf16 [rows, ldx] */
    const void* w;     /* f32 [C] or NULL */
    const void* shift; /* bf16 or NULL */
    const void* scale; /* bf16 or NULL */
    const void* pos;   /* f32 or NULL */
    void* out;         /* bf16 [rows, ldo] (forward only) */
    void* mean;        /* f32 [rows]: written by fwd (optional), read by bwd */
    void* rstd;        /* f32 [rows] */
    int64_t rows, C, ldx, ldo, ldmod, rows_per_sample, pos_rows;
    float eps;
    int32_t act; /* md_act applied to x before the norm */
} md_ln_args;

typedef struct md_ln_bwd_args {
    const void* dz;  /* bf16 [rows, lddz]: grad of the (modulated) output */
    void* dx;        /* bf16 [rows, lddx] or NULL */
    void* dscale;    /* f32 [samples, ldg], ZERO on entry: receives the per-sample sums dS = sum_t dz * xhat; if
                        dscale_is_output it is turned into the modulation-scale gradient w * dS.  Required when dw is set
                        (plain LayerNorms pass a zeroed scratch [samples, C]). */
    void* dshift;    /* f32 [samples, ldg] (+=) or NULL */
    void* dw;        /* f32 [C] (+= sum_b (1 + scale_b) * dS_b) or NULL */
    int64_t lddz, lddx, ldg;
    int64_t rows_per_block; /* rows of one sample handled by one workgroup (column-sum granularity) */
    int32_t accumulate;     /* dx += instead of dx = */
    int32_t dscale_is_output;
} md_ln_bwd_args;

int md_ln_fwd(const md_ln_args* a, hipStream_t stream);
int md_ln_bwd(const md_ln_args* a, const md_ln_bwd_args* b, hipStream_t stream);

/* Non-parametric LayerNorm over nseg column segments of every row of buf, in place: segment s covers columns
 * [col0 + s * seg_stride, ... + width); rstd_out is [nseg][rows].  nseg = 2, seg_stride = hidden normalises the q and the k
 * half of a packed qkv row in one launch.
 * (ln_q / ln_k over ALL heads concatenated: utils.py:113-114,122-125,175-176,183-186.) */
int md_qkln_fwd(void* buf, int64_t rows, int64_t ld, int64_t col0, int64_t width, int32_t nseg, int64_t seg_stride,
                float* rstd_out, float eps, hipStream_t stream);
/* d: grad buffer (in place, dy -> dx); y: the normalised forward output; same segment addressing in both. */
int md_qkln_bwd(void* d, int64_t ldd, int64_t dcol0, const void* y, int64_t ldy, int64_t ycol0, int64_t rows,
                int64_t width, int32_t nseg, int64_t dseg_stride, int64_t yseg_stride, const float* rstd, hipStream_t stream);

/* Head-major forms (ABI 6).  The forward reads the row-major segments like md_qkln_fwd and writes the normalised values
 * HEAD-MAJOR into a separate buffer: element (row = b * S + s, segment g, column c = h * hd + e) goes to
 * out[g * out_seg_stride + ((b * H + h) * S + s) * hd + e], H = width / hd -- one contiguous [S, hd] block per (sample, head), which is
 * what the attention kernels read (md_attn_args.hsq / hsk); buf is left untouched.  The backward reads dL/dy (written head-major by
 * md_attn_bwd) and y in that layout and writes dL/dx row-major into d (the operand of the qkv / q / kv weight- and data-gradient
 * GEMMs).  rows % S == 0.  The [B, N, 3, H, hd] reshape + permute of utils.py:177-182 (116-121) is the layout freedom used. */
int md_qkln_fwd_hm(const void* buf, int64_t rows, int64_t ld, int64_t col0, int64_t width, int32_t nseg, int64_t seg_stride,
                   void* out, int64_t out_seg_stride, int64_t S, int32_t hd, float* rstd_out, float eps, hipStream_t stream);
int md_qkln_bwd_hm(const void* dy, int64_t dy_seg_stride, const void* y, int64_t y_seg_stride, void* d, int64_t ldd, int64_t dcol0,
                   int64_t dseg_stride, int64_t rows, int64_t width, int32_t nseg, int64_t S, int32_t hd, const float* rstd,
                   hipStream_t stream);

/* ------------------------------------------------------------------------------------------- attention */
/* softmax(scale * Q K^T) V per (batch, head), non-causal, no mask.  Row r of head h of batch b of X lives at
 * X + b*sX + r*ldX + h*hsX (bf16; hsX = hd unless set), so packed qkv / kv projection buffers are addressed in place.
 * lse / delta: f32 [B, H, Sq].  Replaces F.scaled_dot_product_attention (utils.py:127-132,188-193) + backward. */
typedef struct md_attn_args {
    const void *q, *k, *v;
    void* o;          /* fwd: output; bwd: the forward output (input) */
    void* lse;        /* fwd: written; bwd: read */
    const void* d_o;  /* bwd: grad of o */
    void *dq, *dk, *dv;
    void* delta;      /* bwd workspace f32 [B, H, Sq] */
    int64_t B, H, Sq, Skv;
    int64_t ldq, ldk, ldv, ldo, sq, sk, sv, so;
    int64_t lddq, lddk, lddv, lddo, sdq, sdk, sdv, sdo;
    float scale;
    int32_t hd;        /* 32 or 64 */
    int32_t bwd_split; /* backward kernel: 0 = the library picks (one fused launch per (batch, head) for Sq, Skv <= 256; the streaming
                          pair (5) for longer sequences);
                          forced (tests, A/B runs): 2 = fused with Q, dO, K, V in LDS together, 3 = fused in two phases on half the
                          LDS, 4 = the same with dK / dV as two passes for every size (3 does that for the 256-row buckets only)
                          -- all three for Sq, Skv <= 256 only; 5 = the streaming pair (two launches, 128-row chunks, register
                          prefetch, 8-wave workgroups; any size).  A forced variant that does not cover the problem returns -1 and
                          launches nothing.  (1, the round-4 dQ + dK/dV pair, and the block-pair form of 3 / 4 for long sequences
                          were removed in ABI 6.) */
    /* ABI 6: element offset of head h inside its row, per tensor (0 = hd: heads packed side by side in the row).  A head-major
     * tensor [B, H, S, hd] (what md_qkln_fwd_hm writes) is ldX = hd, sX = H * S * hd, hsX = S * hd. */
    int64_t hsq, hsk, hsv, hso, hsdq, hsdk, hsdv, hsdo;
} md_attn_args;

int md_attn_fwd(const md_attn_args* a, hipStream_t stream);
int md_attn_bwd(const md_attn_args* a, hipStream_t stream);

/* ------------------------------------------------------------------------------------------- elementwise */
/* a = silu(h12[:, :f]) * h12[:, f:]  (FeedForward, dit.py:88-89) and its backward (dh12 from da). */
int md_swiglu_fwd(const void* h12, int64_t ldh, void* a, int64_t lda, int64_t M, int64_t f, hipStream_t stream);
int md_swiglu_bwd(const void* da, int64_t ldda, const void* h12, int64_t ldh, void* dh12, int64_t lddh, int64_t M, int64_t f,
                  hipStream_t stream);
/* adaLN-Zero gate backward (dit.py:236,238): dbr = gate[b] * dx; dgate[b, :] += sum_t dx * br.  All [rows, C] dense. */
int md_gate_bwd(const void* dx, const void* br, const void* gate, int64_t ldgate, void* dbr, float* dgate, int64_t lddg,
                int64_t rows, int64_t C, int64_t rows_per_sample, int64_t rows_per_block, hipStream_t stream);
int md_act_fwd(const void* x, void* y, int64_t n, int32_t act, hipStream_t stream);               /* bf16 -> bf16 */
int md_act_bwd(const float* dy, const void* x, void* dx, int64_t n, int32_t act, hipStream_t stream); /* dx = dy*act'(x) */
int md_colsum(const void* x, int32_t x_is_f32, int64_t ld, float* out, int64_t rows, int64_t C, hipStream_t stream); /* out += */
int md_cast_f32_bf16(const float* x, void* y, int64_t n, const float* scale_ptr, hipStream_t stream);
/* y = bf16(x), x = 0: stages one bucket of the fp32 gradient accumulators for the data-parallel exchange and clears it in the
 * same pass (the reduce-scatter form of FSDP's gradient reduction, configs/res_256_pretrain.yaml:117-118 SHARD_GRAD_OP). */
int md_cast_f32_bf16_clear(float* x, void* y, int64_t n, hipStream_t stream);
/* y(bf16)[r, :] = x[r, :] * rowscale[r / rows_per_sample]; x_dtype 0 = f16, 1 = f32 (caption cast + drop, model.py:132-139) */
int md_cast_rows_bf16(const void* x, int32_t x_dtype, void* y, int64_t rows, int64_t C, const float* rowscale,
                      int64_t rows_per_sample, hipStream_t stream);
int md_mean_tokens(const void* y, void* out, int64_t B, int64_t L, int64_t C, hipStream_t stream);     /* dit.py:484 */
int md_mean_tokens_bwd(const void* dpool, float* dy, int64_t B, int64_t L, int64_t C, hipStream_t stream); /* dy += */
int md_add_bf16(const void* a, const void* b, void* y, int64_t n, hipStream_t stream);
/* zero-fill (fp32 gradient accumulators, scatter targets): hipMemsetAsync on the caller's stream */
int md_fill_zero(void* p, int64_t bytes, hipStream_t stream);

/* ------------------------------------------------------------------------------------------- masking / MoE routing */
/* utils.py:382-403 given the uniform noise: stable ascending rank.  keep_rows[b*len_keep + r] = b*T + token with rank r
 * (absolute row for md_gather_rows), ids_restore[b, t] = rank, mask[b, t] = rank >= len_keep. */
int md_get_mask(const float* noise, int64_t B, int64_t T, int64_t len_keep, int32_t* keep_rows, int32_t* ids_restore,
                float* mask, hipStream_t stream);
int md_gather_rows(const void* src, int64_t ld_src, const int32_t* idx, void* dst, int64_t ld_dst, int64_t n, int64_t C,
                   hipStream_t stream);  /* dst[i] = src[idx[i]]   (mask_out_token utils.py:406-414; MoE dispatch) */
int md_scatter_rows(const void* src, int64_t ld_src, const int32_t* idx, void* dst, int64_t ld_dst, int64_t n, int64_t C,
                    hipStream_t stream); /* dst[idx[i]] = src[i]   (their backward; dst pre-zeroed) */
/* Expert-choice routing (dit.py:131-133): probs = softmax(logits) (f32, rows padded to ld); expert e of sample n takes
 * its k best tokens: rowidx/gval [E, B*k], slot [B*S, E] (position in the expert's list, or -1). */
int md_moe_route(const float* logits, float* probs, int64_t ld, int64_t B, int64_t S, int32_t E, int32_t k, int32_t* rowidx,
                 float* gval, int32_t* slot, hipStream_t stream);
/* dit.py:141-142 + the gated residual of dit.py:238: br = sum_e g*h2, out = res + gate*br. */
int md_moe_combine(const void* h2, const float* gval, const int32_t* slot, const void* res, const void* gate, int64_t ldgate,
                   void* br, void* out, int64_t B, int64_t S, int32_t E, int32_t k, int64_t C, hipStream_t stream);
int md_moe_combine_bwd(const void* dbr, const void* h2, const int32_t* rowidx, const float* gval, void* dh2, float* dgval,
                       int64_t R, int64_t C, hipStream_t stream);
int md_moe_dispatch_bwd(const void* dxin, const int32_t* slot, void* dx, const float* probs, int64_t ldp, const float* dgval,
                        void* dlogits, int64_t ldo, int64_t B, int64_t S, int32_t E, int32_t k, int64_t C, hipStream_t stream);

/* ------------------------------------------------------------------------------------------- EDM front / back end */
int md_edm_prepare(const float* x0, const float* eps, const float* rnd, float* xn, float* sigma, float* cin, float* cnoise,
                   int64_t B, int64_t per_sample, float p_mean, float p_std, float sigma_data, hipStream_t stream);
/* The same for the fp16 latents the precomputed-latent datasets hold (datasets/latents_loader.py:55-66): also writes the
 * fp32 copy of the clean latents the loss reads, so no torch cast runs on the step path. */
int md_edm_prepare_f16(const void* x0_f16, const float* eps, const float* rnd, float* xn, float* x0_f32, float* sigma, float* cin,
                       float* cnoise, int64_t B, int64_t per_sample, float p_mean, float p_std, float sigma_data,
                       hipStream_t stream);
int md_patchify(const float* x, const float* scale, void* out, int64_t B, int32_t C, int32_t H, int32_t W, int32_t p,
                hipStream_t stream);
int md_timestep_embed(const float* t, void* out, int64_t B, int32_t dim, hipStream_t stream);
int md_unpatchify(const void* tok, const int32_t* ids_restore, int64_t Tk, const float* mask_token, float* img, int64_t B,
                  int32_t C, int32_t H, int32_t W, int32_t p, hipStream_t stream);
/* model.py:177,199-210 on the kept tokens; dtok (optional) = d(batch-mean loss)/d(network output), f32 [B*Tk, C*p*p]. */
int md_edm_loss(const void* tok, const int32_t* keep_rows, const float* xn, const float* x0, const float* sigma,
                float* loss_per_sample, float* loss_mean, float* dtok, int64_t B, int64_t Tk, int32_t C, int32_t H, int32_t W,
                int32_t p, float sigma_data, hipStream_t stream);
/* The form one microbatch of a training step uses (replaces `(loss * n_micro / n_rank).backward()` of Composer's microbatch
 * loop around model.py:104-142 and the scalar torch ops that come with it): dtok is written as bf16 (what the hand-written
 * backward starts from) and pre-multiplied by grad_scale (the microbatch weight); loss_accum (optional) += accum_weight *
 * batch-mean loss, so the rank-mean loss of the step accumulates on the device. */
int md_edm_loss_train(const void* tok, const int32_t* keep_rows, const float* xn, const float* x0, const float* sigma,
                      float* loss_per_sample, float* loss_mean, void* dtok_bf16, float grad_scale, float* loss_accum,
                      float accum_weight, int64_t B, int64_t Tk, int32_t C, int32_t H, int32_t W, int32_t p, float sigma_data,
                      hipStream_t stream);

/* Sampler (model.py:231-297): the arithmetic around each network evaluation of the Heun loop, fused; fp64 state, fp32
 * preconditioning (model.py:144-179) and classifier-free-guidance combine (dit.py:542-550: F = [cond; uncond] halves). */
int md_edm_sampler_input(const double* x, float* out, int64_t n, float sigma, float sigma_data, int32_t duplicate, hipStream_t stream);
int md_edm_heun_update(const double* x_hat, const double* x_in, const float* F, double* d_cur, double* x_next, int64_t n, float cfg,
                       int32_t has_uncond, double t_in, double t_hat, double t_next, float sigma_data, int32_t second,
                       hipStream_t stream);

/* ------------------------------------------------------------------------------------------- optimiser */
/* Sum of squares of a gradient buffer (fp32, or bf16 when g_is_bf16), deterministic: workgroup b of a fixed grid writes
 * partials[b] (MD_SUMSQ_PARTIALS floats per call); md_sumsq_finish adds `count` partials (several calls' worth, e.g. one per
 * data-parallel bucket) in a fixed order into out[0].  n must be a multiple of 8. */
#define MD_SUMSQ_PARTIALS 1024
int md_sumsq(const void* g, int32_t g_is_bf16, int64_t n, float* partials, hipStream_t stream);
int md_sumsq_finish(const float* partials, int64_t count, float* out, hipStream_t stream);
/* Exact integer checksum of n 16-bit words (n % 8 == 0, x 16-byte aligned): out2[0] += sum of the words, out2[1] += sum of
 * word_i * h(i) with h an odd 32-bit hash of the index; out2 (two uint64 on the device) must be zero on entry.  Order-independent
 * and exact, so bit-identical on identical data: the replica-consistency check of the data-parallel step compares it across ranks
 * (a single bf16 ulp, a sign flip or a permutation changes it; no reference counterpart: FSDP keeps one sharded copy). */
int md_checksum_u16(const void* x, int64_t n, uint64_t* out2, hipStream_t stream);
/* clip_grad_norm_ (train.py:85-86) + torch.optim.AdamW (train.py:39-43) + bf16 shadow emit (+ EMA of the weights,
 * configs/res_512_*.yaml:4-9), one pass. */
typedef struct md_adamw_args {
    void *p, *g, *m, *v; /* f32 [n]; g is zeroed when zero_grad */
    void* shadow;        /* bf16 [n] or NULL */
    const void* sumsq;   /* f32 [1] device: sum of squared (unscaled) grads, or NULL for no clipping */
    const void* g_bf16;  /* optional bf16 [n]: take the gradient from here (the data-parallel exchange buffer) instead of g */
    void* ema;           /* optional f32 [n] */
    int64_t n;
    float lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2, max_norm, grad_scale;
    float ema_smoothing;
    int32_t zero_grad;
    int32_t ema_mode;    /* 0 = none, 1 = ema <- updated weights (ema_start), 2 = ema <- s * ema + (1 - s) * weights */
} md_adamw_args;
int md_adamw_step(const md_adamw_args* a, hipStream_t stream);
/* The sharded form of the same pass (FSDP SHARD_GRAD_OP, configs/res_256_pretrain.yaml:117-118: every rank updates its slice of
 * the optimiser state): ONE launch over n_ranges <= MD_ADAMW_MAX_RANGES chunks of the flat buffers.  Range j covers flat
 * elements [flat_off[j], flat_off[j] + count[j]) of p / m / v / ema (a->p ... are the flat bases); a->g_bf16 (the
 * reduce-scattered gradient) and a->shadow (the bf16 weights to all-gather) are PACKED: range j starts at sum(count[0..j)).
 * a->n is ignored; zero_grad must be 0 (the staging cast cleared the accumulators).  flat_off / count are HOST arrays. */
#define MD_ADAMW_MAX_RANGES 64
int md_adamw_step_ranges(const md_adamw_args* a, const int64_t* flat_off, const int64_t* count, int32_t n_ranges, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MICRODIT_HIP_H */
