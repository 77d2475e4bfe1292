// Discrete-selection kernels of the MicroDiT path (integer / index work, bit-exact by construction):
//   * random patch masking: stable rank of the uniform noise -> ids_keep / ids_restore / mask (utils.py:382-403)
//   * token gather / scatter (mask_out_token utils.py:406-414 and its backward)
//   * expert-choice MoE routing (dit.py:126-143): fp32 softmax over experts, per-(sample, expert) top-k over
//     tokens, dispatch gather, gate-weighted combine (a scatter-ADD in the reference, done here as a
//     deterministic per-token gather through an inverse slot map), and the backward of all of it.
// The reference burns FLOPs on one-hot dispatch matmuls (dit.py:134-136,142); here dispatch/combine are copies.
#include "md_common.h"
#include "../../include/microdit_hip.h"

namespace {

// ---------------------------------------------------------------------------------------------- masking
// One workgroup per sample.  rank_i = #{j : noise_j < noise_i  or  (noise_j == noise_i and j < i)}  is exactly
// the position of i in a STABLE ascending sort, so ids_shuffle[rank_i] = i and ids_restore[i] = rank_i.
__global__ __launch_bounds__(256) void get_mask_kernel(const float* noise, int64_t T, int64_t len_keep, int32_t* keep_rows,
                                                       int32_t* ids_restore, float* mask) {
    extern __shared__ float s_noise[];
    const int64_t b = blockIdx.x;
    for (int64_t i = threadIdx.x; i < T; i += 256) s_noise[i] = noise[b * T + i];
    __syncthreads();
    for (int64_t i = threadIdx.x; i < T; i += 256) {
        const float v = s_noise[i];
        int rank = 0;
        for (int64_t j = 0; j < T; ++j) {
            const float u = s_noise[j];
            rank += (u < v) || (u == v && j < i);
        }
        ids_restore[b * T + i] = rank;
        mask[b * T + i] = rank >= len_keep ? 1.f : 0.f;
        if (rank < len_keep) keep_rows[b * len_keep + rank] = (int32_t)(b * T + i);  // absolute source row
    }
}

// dst[i, :] = src[idx[i], :]
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16* src, int64_t lds_, const int32_t* idx, bf16* dst,
                                                          int64_t ldd, int64_t n, int64_t C) {
    const int64_t cpr = C / 8, total = n * cpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cpr, c = (i % cpr) * 8;
        *reinterpret_cast<uint4*>(dst + r * ldd + c) = *reinterpret_cast<const uint4*>(src + (int64_t)idx[r] * lds_ + c);
    }
}

// dst[idx[i], :] = src[i, :]   (indices unique; dst pre-zeroed by the caller)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const bf16* src, int64_t lds_, const int32_t* idx, bf16* dst,
                                                           int64_t ldd, int64_t n, int64_t C) {
    const int64_t cpr = C / 8, total = n * cpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cpr, c = (i % cpr) * 8;
        *reinterpret_cast<uint4*>(dst + (int64_t)idx[r] * ldd + c) = *reinterpret_cast<const uint4*>(src + r * lds_ + c);
    }
}

// ---------------------------------------------------------------------------------------------- MoE routing
// probs[m, e] = softmax_e(logits[m, e]) in fp32 (dit.py:132); E <= 16, rows padded to ld.
__global__ __launch_bounds__(256) void moe_softmax_kernel(const float* logits, float* probs, int64_t M, int E, int64_t ld) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    float v[16];
    float mx = -1e30f;
    for (int e = 0; e < E; ++e) {
        v[e] = bf2f(f2bf(logits[m * ld + e]));  // the gate Linear runs under bf16 autocast: logits are bf16
        mx = fmaxf(mx, v[e]);
    }
    float s = 0.f;
    for (int e = 0; e < E; ++e) {
        v[e] = expf(v[e] - mx);
        s += v[e];
    }
    const float inv = 1.f / s;
    for (int e = 0; e < E; ++e) probs[m * ld + e] = v[e] * inv;
    for (int e = E; e < ld; ++e) probs[m * ld + e] = 0.f;
}

// grid = (E, B): expert e of sample n picks its k highest-probability tokens (dit.py:133).  Descending order,
// ties -> lower token index first.  Outputs: rowidx[e, n*k + j] = absolute token row (n*S + tau);
// gval[e, n*k + j] = its probability; slot[n*S + tau, e] = j or -1.
__global__ __launch_bounds__(256) void moe_topk_kernel(const float* probs, int64_t ld, int64_t S, int E, int k,
                                                       int64_t Bk /* = B*k */, int32_t* rowidx, float* gval, int32_t* slot) {
    extern __shared__ float s_p[];
    const int e = blockIdx.x;
    const int64_t n = blockIdx.y;
    for (int64_t t = threadIdx.x; t < S; t += 256) s_p[t] = probs[(n * S + t) * ld + e];
    __syncthreads();
    for (int64_t t = threadIdx.x; t < S; t += 256) {
        const float v = s_p[t];
        int rank = 0;
        for (int64_t j = 0; j < S; ++j) {
            const float u = s_p[j];
            rank += (u > v) || (u == v && j < t);
        }
        if (rank < k) {
            rowidx[(int64_t)e * Bk + n * k + rank] = (int32_t)(n * S + t);
            gval[(int64_t)e * Bk + n * k + rank] = v;
            slot[(n * S + t) * E + e] = rank;
        } else {
            slot[(n * S + t) * E + e] = -1;
        }
    }
}

// Combine + gated residual.  One wave per token row:
//   br[row, :]  = sum_e [slot >= 0] bf16(g * h2[e, n*k + slot, :])      (fp32 accumulate -> bf16; dit.py:141-142)
//   out[row, :] = res[row, :] + gate[n, :] * br[row, :]                  (dit.py:238)
// The expert loop is written in two sweeps -- all slot indices of the row, then every selected expert row's load, then the sums
// in expert order -- because the obvious form (slot -> branch -> gate value -> row load, per expert) is a chain of E dependent
// global loads per row and chunk: the kernel ran at 2.8 TB/s with 8 of its 9 loads waiting for the previous one.
template <int EMAX>
__global__ __launch_bounds__(256) void moe_combine_kernel(const bf16* h2, const float* gval, const int32_t* slot,
                                                          const bf16* res, const bf16* gate, int64_t ldgate, bf16* br,
                                                          bf16* out, int64_t M, int64_t S, int E, int k, int64_t Bk,
                                                          int64_t C) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t row = wave; row < M; row += nwaves) {
        const int64_t n = row / S;
        int sl[EMAX];
        float g[EMAX];
#pragma unroll
        for (int e = 0; e < EMAX; ++e) sl[e] = e < E ? slot[row * E + e] : -1;
#pragma unroll
        for (int e = 0; e < EMAX; ++e) g[e] = sl[e] >= 0 ? gval[(int64_t)e * Bk + n * k + sl[e]] : 0.f;
        for (int c = lane * 8; c < C; c += 512) {
            bf16x8 h[EMAX];
#pragma unroll
            for (int e = 0; e < EMAX; ++e)
                if (sl[e] >= 0) h[e] = ld_bf16x8(h2 + ((int64_t)e * Bk + n * k + sl[e]) * C + c);
            const bf16x8 rs = ld_bf16x8(res + row * C + c);
            const bf16x8 gt = ld_bf16x8(gate + n * ldgate + c);
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < EMAX; ++e)
                if (sl[e] >= 0) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] += bf2f(f2bf(g[e] * bf2f(h[e][i])));
                }
            bf16x8 ob, ox;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                ob[i] = f2bf(acc[i]);
                ox[i] = f2bf(bf2f(rs[i]) + bf2f(gt[i]) * bf2f(ob[i]));
            }
            st_bf16x8(br + row * C + c, ob);
            st_bf16x8(out + row * C + c, ox);
        }
    }
}

// Backward of the combine w.r.t. the expert outputs and gate values.  One wave per routed row r = (e, n*k + j):
//   dh2[r, :] = g[r] * dbr[token(r), :] ;  dg[r] = <dbr[token(r), :], h2[r, :]>
__global__ __launch_bounds__(256) void moe_combine_bwd_kernel(const bf16* dbr, const bf16* h2, const int32_t* rowidx,
                                                              const float* gval, bf16* dh2, float* dgval, int64_t R,
                                                              int64_t C) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave; r < R; r += nwaves) {
        const int64_t tok = rowidx[r];
        const float g = gval[r];
        float dot = 0.f;
        for (int c = lane * 8; c < C; c += 512) {
            const bf16x8 d = ld_bf16x8(dbr + tok * C + c), h = ld_bf16x8(h2 + r * C + c);
            bf16x8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float dv = bf2f(d[i]);
                dot += dv * bf2f(h[i]);
                o[i] = f2bf(g * dv);
            }
            st_bf16x8(dh2 + r * C + c, o);
        }
        dot = wave_sum(dot);
        if (lane == 0) dgval[r] = dot;
    }
}

// Backward of the dispatch gather: dx[row, :] = sum_e [slot >= 0] dxin[e, n*k + slot, :]   (wave per token row)
template <int EMAX>
__global__ __launch_bounds__(256) void moe_scatter_sum_kernel(const bf16* dxin, const int32_t* slot, bf16* dx, int64_t M,
                                                              int64_t S, int E, int k, int64_t Bk, int64_t C) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t row = wave; row < M; row += nwaves) {
        const int64_t n = row / S;
        int sl[EMAX];              // all slot indices first, then all row loads (see moe_combine_kernel)
#pragma unroll
        for (int e = 0; e < EMAX; ++e) sl[e] = e < E ? slot[row * E + e] : -1;
        for (int c = lane * 8; c < C; c += 512) {
            bf16x8 h[EMAX];
#pragma unroll
            for (int e = 0; e < EMAX; ++e)
                if (sl[e] >= 0) h[e] = ld_bf16x8(dxin + ((int64_t)e * Bk + n * k + sl[e]) * C + c);
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < EMAX; ++e)
                if (sl[e] >= 0) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] += bf2f(h[e][i]);
                }
            bf16x8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = f2bf(acc[i]);
            st_bf16x8(dx + row * C + c, o);
        }
    }
}

// dlogits[m, e] = p_e * (dp_e - sum_e' dp_e' p_e'),  dp_e = slot >= 0 ? dgval[e, n*k + slot] : 0.
// Written as bf16 rows padded with zeros to ldo (>= 8) so they can feed the gate dgrad / wgrad GEMMs.
__global__ __launch_bounds__(256) void moe_softmax_bwd_kernel(const float* probs, int64_t ldp, const float* dgval,
                                                              const int32_t* slot, bf16* dlogits, int64_t ldo, int64_t M,
                                                              int64_t S, int E, int k, int64_t Bk) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const int64_t n = m / S;
    float dp[16], p[16];
    float dot = 0.f;
    for (int e = 0; e < E; ++e) {
        const int sl = slot[m * E + e];
        p[e] = probs[m * ldp + e];
        dp[e] = sl >= 0 ? dgval[(int64_t)e * Bk + n * k + sl] : 0.f;
        dot += dp[e] * p[e];
    }
    for (int e = 0; e < ldo; ++e) dlogits[m * ldo + e] = f2bf(e < E ? p[e] * (dp[e] - dot) : 0.f);
}

inline int rgrid(int64_t work) {
    int64_t g = (work + 255) / 256;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return (int)g;
}
inline int wgrid(int64_t rows) {
    int64_t g = (rows + 3) / 4;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int md_get_mask(const float* noise, int64_t B, int64_t T, int64_t len_keep, int32_t* keep_rows,
                           int32_t* ids_restore, float* mask, hipStream_t st) {
    if (!noise || !keep_rows || !ids_restore || !mask || B <= 0 || T <= 0 || len_keep <= 0 || len_keep > T || T > 16384)
        return MD_BAD_ARG;
    hipLaunchKernelGGL(get_mask_kernel, dim3((unsigned)B), dim3(256), (size_t)T * 4, st, noise, T, len_keep, keep_rows,
                       ids_restore, mask);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_gather_rows(const void* src, int64_t ld_src, const int32_t* idx, void* dst, int64_t ld_dst, int64_t n,
                              int64_t C, hipStream_t st) {
    if (!src || !idx || !dst || n <= 0 || C <= 0 || C % 8 || ld_src % 8 || ld_dst % 8) return MD_BAD_ARG;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(rgrid(n * C / 8)), dim3(256), 0, st, (const bf16*)src, ld_src, idx,
                       (bf16*)dst, ld_dst, n, C);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_scatter_rows(const void* src, int64_t ld_src, const int32_t* idx, void* dst, int64_t ld_dst, int64_t n,
                               int64_t C, hipStream_t st) {
    if (!src || !idx || !dst || n <= 0 || C <= 0 || C % 8 || ld_src % 8 || ld_dst % 8) return MD_BAD_ARG;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(rgrid(n * C / 8)), dim3(256), 0, st, (const bf16*)src, ld_src, idx,
                       (bf16*)dst, ld_dst, n, C);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_moe_route(const float* logits, float* probs, int64_t ld, int64_t B, int64_t S, int32_t E, int32_t k,
                            int32_t* rowidx, float* gval, int32_t* slot, hipStream_t st) {
    if (!logits || !probs || !rowidx || !gval || !slot || B <= 0 || S <= 0 || E <= 0 || E > 16 || ld < E || k <= 0 ||
        k > S || S > 16384)
        return MD_BAD_ARG;
    const int64_t M = B * S;
    hipLaunchKernelGGL(moe_softmax_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, logits, probs, M, E, ld);
    hipLaunchKernelGGL(moe_topk_kernel, dim3((unsigned)E, (unsigned)B), dim3(256), (size_t)S * 4, st, probs, ld, S, E, k,
                       B * k, rowidx, gval, slot);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_moe_combine(const void* h2, const float* gval, const int32_t* slot, const void* res, const void* gate,
                              int64_t ldgate, void* br, void* out, int64_t B, int64_t S, int32_t E, int32_t k, int64_t C,
                              hipStream_t st) {
    if (!h2 || !gval || !slot || !res || !gate || !br || !out || B <= 0 || S <= 0 || E <= 0 || k <= 0 || C <= 0 || C % 8 ||
        ldgate % 8)
        return MD_BAD_ARG;
    if (E > 16) return MD_BAD_ARG;
    if (E <= 8)
        hipLaunchKernelGGL(moe_combine_kernel<8>, dim3(wgrid(B * S)), dim3(256), 0, st, (const bf16*)h2, gval, slot,
                           (const bf16*)res, (const bf16*)gate, ldgate, (bf16*)br, (bf16*)out, B * S, S, E, k, B * k, C);
    else
        hipLaunchKernelGGL(moe_combine_kernel<16>, dim3(wgrid(B * S)), dim3(256), 0, st, (const bf16*)h2, gval, slot,
                           (const bf16*)res, (const bf16*)gate, ldgate, (bf16*)br, (bf16*)out, B * S, S, E, k, B * k, C);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_moe_combine_bwd(const void* dbr, const void* h2, const int32_t* rowidx, const float* gval, void* dh2,
                                  float* dgval, int64_t R, int64_t C, hipStream_t st) {
    if (!dbr || !h2 || !rowidx || !gval || !dh2 || !dgval || R <= 0 || C <= 0 || C % 8) return MD_BAD_ARG;
    hipLaunchKernelGGL(moe_combine_bwd_kernel, dim3(wgrid(R)), dim3(256), 0, st, (const bf16*)dbr, (const bf16*)h2, rowidx,
                       gval, (bf16*)dh2, dgval, R, C);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_moe_dispatch_bwd(const void* dxin, const int32_t* slot, void* dx, const float* probs, int64_t ldp,
                                   const float* dgval, void* dlogits, int64_t ldo, int64_t B, int64_t S, int32_t E,
                                   int32_t k, int64_t C, hipStream_t st) {
    if (!dxin || !slot || !dx || !probs || !dgval || !dlogits || B <= 0 || S <= 0 || E <= 0 || E > 16 || k <= 0 || C <= 0 ||
        C % 8 || ldo < E || ldo > 16)
        return MD_BAD_ARG;
    const int64_t M = B * S;
    if (E <= 8)
        hipLaunchKernelGGL(moe_scatter_sum_kernel<8>, dim3(wgrid(M)), dim3(256), 0, st, (const bf16*)dxin, slot, (bf16*)dx, M, S,
                           E, k, B * k, C);
    else
        hipLaunchKernelGGL(moe_scatter_sum_kernel<16>, dim3(wgrid(M)), dim3(256), 0, st, (const bf16*)dxin, slot, (bf16*)dx, M, S,
                           E, k, B * k, C);
    hipLaunchKernelGGL(moe_softmax_bwd_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, probs, ldp, dgval, slot,
                       (bf16*)dlogits, ldo, M, S, E, k, B * k);
    MD_LAUNCH_CHECK();
    return 0;
}
