// bf16 MFMA GEMM family for the MicroDiT linear layers (forward, dgrad, wgrad, grouped MoE experts).
//
//   C[m, n] (+)= alpha * sum_k A(m, k) * B(n, k)          (fp32 accumulate on v_mfma_f32_32x32x16_bf16)
//
// Either operand may be stored "K-contiguous" (element (r, k) at base[r * ld + k]; torch Linear weights
// [out, in] and activations [tokens, channels] feeding a forward GEMM) or "K-strided" (element (r, k) at
// base[k * ld + r]; both operands of a weight-gradient GEMM, the [in, out] expert weights of the MoE, and
// the weight operand of a dgrad).  K-strided tiles are staged in LDS exactly as they lie in HBM (coalesced
// 16-byte rows) and the MFMA fragments are formed with ds_read_b64_tr_b16, CDNA4's transposing LDS read, so
// no operand is ever transposed through HBM.
//
// Tile: 128 x 128 x 64 per 256-thread workgroup (4 waves, 64 x 64 per wave = 2 x 2 MFMA 32x32x16 tiles),
// register-staged global->LDS prefetch (loads of tile t+1 are in flight while tile t is multiplied),
// XCD-aware tile order (neighbouring tiles that share an A row-panel run on the same XCD / L2).
// The epilogue is staged through LDS so that every global access is a 16-byte row-contiguous chunk, and
// fuses: bias, GELU(tanh|erf), gated residual (adaLN-Zero), activation-derivative multiply (dgrad through
// GELU), fp32 store / accumulate / split-K atomic accumulate (wgrad).
//
// Replaces the implicit cuBLAS calls behind every nn.Linear / einsum of the reference
// (micro_diffusion/models/dit.py:84-89,131-142,224; utils.py:58-61,109-111,172-173,225-233).
#include "md_common.h"
#include "../../include/microdit_hip.h"

namespace {

constexpr int BM = 128, BN = 128, BKT = 64;
constexpr int PITCH_KC = (BKT + 8) * 2;   // bytes per row of a K-contiguous tile  [128][64+8] bf16
constexpr int PITCH_KS = (BM + 32) * 2;   // bytes per row of a K-strided tile     [64][128+32] bf16
constexpr int TILE_BYTES = (BM * PITCH_KC > BKT * PITCH_KS) ? BM * PITCH_KC : BKT * PITCH_KS;  // 20480
constexpr int SLAB_PITCH = 68;            // floats per row of the epilogue slab [32][64+4]
constexpr int SLAB_FLOATS = 32 * SLAB_PITCH;
static_assert(4 * SLAB_FLOATS * 4 <= 2 * TILE_BYTES, "epilogue slab must fit in the staging LDS");

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ bf16x4 lds_tr_read(const unsigned char* p) {
    // 16 lanes read one [4 k][16 col] block; lane i of the group receives column i (4 consecutive k).
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    U64 t;
    t.s = v;
    return t.h;
}

template <int KC>
__device__ __forceinline__ void load_tile(uint4 (&r)[4], const bf16* __restrict__ base, int64_t ld, int64_t r0,
                                          int64_t rmax, int64_t k0, int64_t kend, int tid) {
    if (KC) {
        const int c = tid & 7;
        const int64_t gk = k0 + c * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t gr = r0 + (tid >> 3) + 32 * i;
            if (gr < rmax && gk < kend)
                r[i] = *reinterpret_cast<const uint4*>(base + gr * ld + gk);
            else
                r[i] = make_uint4(0, 0, 0, 0);
        }
    } else {
        const int c = tid & 15;
        const int64_t gr = r0 + c * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t gk = k0 + (tid >> 4) + 16 * i;
            if (gr < rmax && gk < kend)
                r[i] = *reinterpret_cast<const uint4*>(base + gk * ld + gr);
            else
                r[i] = make_uint4(0, 0, 0, 0);
        }
    }
}

template <int KC>
__device__ __forceinline__ void store_tile(const uint4 (&r)[4], unsigned char* s, int tid) {
    if (KC) {
        const int c = tid & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (tid >> 3) + 32 * i;
            *reinterpret_cast<uint4*>(s + row * PITCH_KC + c * 16) = r[i];
        }
    } else {
        const int c = tid & 15;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kk = (tid >> 4) + 16 * i;
            *reinterpret_cast<uint4*>(s + kk * PITCH_KS + c * 16) = r[i];
        }
    }
}

// Fragment of 32 rows x 16 k for v_mfma_f32_32x32x16_bf16: lane l holds row (l & 31), k = (l >> 5) * 8 .. +7.
template <int KC>
__device__ __forceinline__ bf16x8 load_frag(const unsigned char* s, int row0, int ks, int lane) {
    if (KC) {
        const int row = row0 + (lane & 31);
        const int kb = ks * 16 + (lane >> 5) * 8;
        U128 t;
        t.u = *reinterpret_cast<const uint4*>(s + row * PITCH_KC + kb * 2);
        return t.h;
    } else {
        const int li = lane & 15;
        const int col = row0 + ((lane >> 4) & 1) * 16 + (li & 3) * 4;
        const int kk = ks * 16 + (lane >> 5) * 8 + (li >> 2);
        const unsigned char* p = s + kk * PITCH_KS + col * 2;
        bf16x4 lo = lds_tr_read(p);
        bf16x4 hi = lds_tr_read(p + 4 * PITCH_KS);
        bf16x8 f;
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
        f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        return f;
    }
}

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

template <int AKC, int BKC>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(md_gemm_args p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILE_BYTES];
    unsigned char* sA = smem;
    unsigned char* sB = smem + TILE_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- XCD-aware tile order (block b runs on XCD b % 8; give each XCD a contiguous tile range) ----
    const int ntn = (int)((p.N + BN - 1) / BN);
    const int nwg = gridDim.x;
    int logical;
    {
        const int bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int64_t m0 = (int64_t)(logical / ntn) * BM;
    const int64_t n0 = (int64_t)(logical % ntn) * BN;

    const int batch = blockIdx.y / p.ksplit;
    const int split = blockIdx.y % p.ksplit;

    const bf16* A = reinterpret_cast<const bf16*>(p.A) + (int64_t)batch * p.sA;
    const bf16* B = reinterpret_cast<const bf16*>(p.B) + (int64_t)batch * p.sB;

    // split-K range, in whole 64-wide k-tiles
    const int64_t ntk = (p.K + BKT - 1) / BKT;
    const int64_t tps = (ntk + p.ksplit - 1) / p.ksplit;
    const int64_t kbeg = (int64_t)split * tps * BKT;
    int64_t kend = kbeg + tps * BKT;
    if (kend > p.K) kend = p.K;
    const int nt = kbeg < kend ? (int)((kend - kbeg + BKT - 1) / BKT) : 0;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    uint4 ra[4], rb[4];
    if (nt > 0) {
        load_tile<AKC>(ra, A, p.lda, m0, p.M, kbeg, kend, tid);
        load_tile<BKC>(rb, B, p.ldb, n0, p.N, kbeg, kend, tid);
    }
    for (int t = 0; t < nt; ++t) {
        __syncthreads();  // everyone finished reading the previous tile
        store_tile<AKC>(ra, sA, tid);
        store_tile<BKC>(rb, sB, tid);
        __syncthreads();
        if (t + 1 < nt) {
            const int64_t k0 = kbeg + (int64_t)(t + 1) * BKT;
            load_tile<AKC>(ra, A, p.lda, m0, p.M, k0, kend, tid);
            load_tile<BKC>(rb, B, p.ldb, n0, p.N, k0, kend, tid);
        }
#pragma unroll
        for (int ks = 0; ks < BKT / 16; ++ks) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = load_frag<AKC>(sA, wm * 64 + i * 32, ks, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = load_frag<BKC>(sB, wn * 64 + j * 32, ks, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }

    // ------------------------------------------------------------------ epilogue
    __syncthreads();
    float* slab = reinterpret_cast<float*>(smem) + wave * SLAB_FLOATS;
    const int mode = p.mode;
    const float alpha = p.alpha;
    const int erow = lane >> 3;        // 0..7
    const int ecol = (lane & 7) * 8;   // 0..56
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        // C/D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                slab[row * SLAB_PITCH + ni * 32 + (lane & 31)] = acc[mi][ni][r];
            }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int lr = it * 8 + erow;
            const int64_t gr = m0 + wm * 64 + mi * 32 + lr;
            const int64_t gc = n0 + wn * 64 + ecol;
            if (gr >= p.M || gc >= p.N) continue;
            float v[8];
            {
                const float4 v0 = *reinterpret_cast<const float4*>(slab + lr * SLAB_PITCH + ecol);
                const float4 v1 = *reinterpret_cast<const float4*>(slab + lr * SLAB_PITCH + ecol + 4);
                v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w;
                v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= alpha;
            if (p.bias) {
                const float* bp = reinterpret_cast<const float*>(p.bias) + (int64_t)batch * p.sBias + gc;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += bp[e];
            }
            if (mode == MD_EPI_STORE_BF16 || mode == MD_EPI_RESIDUAL) {
                if (p.C2) {  // raw (pre-activation / pre-gate) copy for the backward pass
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
                    st_bf16x8(reinterpret_cast<bf16*>(p.C2) + (int64_t)batch * p.sC2 + gr * p.ldc2 + gc, o);
                }
                if (mode == MD_EPI_STORE_BF16) {
                    if (p.act != MD_ACT_NONE) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = apply_act(v[e], p.act);
                    }
                } else {
                    const bf16x8 rs = ld_bf16x8(reinterpret_cast<const bf16*>(p.res) + gr * p.ldr + gc);
                    if (p.gate) {
                        const int64_t smp = gr / p.rows_per_sample;
                        const bf16x8 g = ld_bf16x8(reinterpret_cast<const bf16*>(p.gate) + smp * p.ldg + gc);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = bf2f(rs[e]) + bf2f(g[e]) * bf2f(f2bf(v[e]));
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = bf2f(rs[e]) + bf2f(f2bf(v[e]));
                    }
                }
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
                st_bf16x8(reinterpret_cast<bf16*>(p.C) + (int64_t)batch * p.sC + gr * p.ldc + gc, o);
            } else if (mode == MD_EPI_DACT) {
                const bf16x8 ax =
                    ld_bf16x8(reinterpret_cast<const bf16*>(p.aux) + (int64_t)batch * p.sAux + gr * p.ldaux + gc);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e] * apply_dact(bf2f(ax[e]), p.act));
                st_bf16x8(reinterpret_cast<bf16*>(p.C) + (int64_t)batch * p.sC + gr * p.ldc + gc, o);
            } else {
                float* cp = reinterpret_cast<float*>(p.C) + (int64_t)batch * p.sC + gr * p.ldc + gc;
                if (mode == MD_EPI_STORE_F32) {
                    *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(cp + 4) = make_float4(v[4], v[5], v[6], v[7]);
                } else if (mode == MD_EPI_ACCUM_F32) {
                    float4 c0 = *reinterpret_cast<const float4*>(cp);
                    float4 c1 = *reinterpret_cast<const float4*>(cp + 4);
                    c0.x += v[0]; c0.y += v[1]; c0.z += v[2]; c0.w += v[3];
                    c1.x += v[4]; c1.y += v[5]; c1.z += v[6]; c1.w += v[7];
                    *reinterpret_cast<float4*>(cp) = c0;
                    *reinterpret_cast<float4*>(cp + 4) = c1;
                } else {  // MD_EPI_ATOMIC_F32 (split-K weight gradients)
#pragma unroll
                    for (int e = 0; e < 8; ++e) unsafeAtomicAdd(cp + e, v[e]);
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int md_gemm_bf16(const md_gemm_args* a, hipStream_t stream) {
    if (!a || !a->A || !a->B || !a->C) return MD_BAD_ARG;
    if (a->M <= 0 || a->N <= 0 || a->K <= 0 || a->batch <= 0 || a->ksplit <= 0) return MD_BAD_ARG;
    // All global accesses are 16-byte chunks of 8 bf16 along the contiguous dimension: leading dimensions must be
    // multiples of 8 (4 for fp32 outputs).  A contiguous extent that is not a multiple of 8 (the MoE gate: N = K =
    // num_experts) is allowed as long as the caller pads the rows to the next multiple of 8 with finite values
    // (zeros): whole chunks are read / written whenever their first element is in range.
    if (a->lda % 8 || a->ldb % 8) return MD_BAD_ARG;
    if ((a->mode == MD_EPI_STORE_BF16 || a->mode == MD_EPI_RESIDUAL || a->mode == MD_EPI_DACT) ? (a->ldc % 8) : (a->ldc % 4))
        return MD_BAD_ARG;
    if (a->C2 && a->ldc2 % 8) return MD_BAD_ARG;
    if (a->ksplit > 1 && a->mode != MD_EPI_ATOMIC_F32) return MD_BAD_ARG;
    if (a->mode == MD_EPI_RESIDUAL && (!a->res || (a->gate && a->rows_per_sample <= 0))) return MD_BAD_ARG;
    if (a->mode == MD_EPI_DACT && !a->aux) return MD_BAD_ARG;
    const int64_t tiles = ((a->M + BM - 1) / BM) * ((a->N + BN - 1) / BN);
    dim3 grid((unsigned)tiles, (unsigned)(a->batch * a->ksplit), 1);
    dim3 block(256, 1, 1);
    if (a->a_kcontig && a->b_kcontig)
        hipLaunchKernelGGL((gemm_bf16_kernel<1, 1>), grid, block, 0, stream, *a);
    else if (a->a_kcontig && !a->b_kcontig)
        hipLaunchKernelGGL((gemm_bf16_kernel<1, 0>), grid, block, 0, stream, *a);
    else if (!a->a_kcontig && a->b_kcontig)
        hipLaunchKernelGGL((gemm_bf16_kernel<0, 1>), grid, block, 0, stream, *a);
    else
        hipLaunchKernelGGL((gemm_bf16_kernel<0, 0>), grid, block, 0, stream, *a);
    MD_LAUNCH_CHECK();
    return 0;
}
