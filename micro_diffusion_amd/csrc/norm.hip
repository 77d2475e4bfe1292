// LayerNorm family of the MicroDiT path: one 64-lane wave per token row, 16-byte bf16 loads, fp32 statistics by
// wave shuffles (no LDS, no barriers in the forward), adaLN modulate fused into the same pass.
//
//  md_ln_fwd   y = LN(act(x + pos)) * w ; z = y * (1 + scale[b]) + shift[b]        (w, pos, act, modulate optional)
//  md_ln_bwd   dx (+)= d/dx ; dscale[b] += w * sum_t dz * xhat ; dshift[b] += sum_t dz ; dw += (1 + scale[b]) * sum dz * xhat
//  md_qkln_*   non-parametric LN over the whole q (or k) hidden width, in place (all heads concatenated)
//
// Reference: create_norm / nn.LayerNorm(bias=False) (utils.py:71-78) under Composer's low-precision LayerNorm
// (train.py:81-84: bf16 in/out, fp32 statistics), modulate (utils.py:28-30), the Mlp act->norm order
// (utils.py:63-68) and the q/k LayerNorms of utils.py:122-125,183-186.
#include "md_common.h"
#include "../../include/microdit_hip.h"

namespace {

constexpr int MAXCH = 4;  // chunks of 8 channels per lane: C <= 64 * 8 * 4 = 2048 (kernels are
                          // instantiated for 1, 2 and 4 chunks so narrow rows keep few registers)

__device__ __forceinline__ float act_fwd(float v, int act) {
    if (act == MD_ACT_GELU_TANH) return gelu_tanh_f(v);
    if (act == MD_ACT_GELU_ERF) return gelu_erf_f(v);
    if (act == MD_ACT_SILU) return silu_f(v);
    return v;
}
__device__ __forceinline__ float act_bwd(float v, int act) {
    if (act == MD_ACT_GELU_TANH) return dgelu_tanh_f(v);
    if (act == MD_ACT_GELU_ERF) return dgelu_erf_f(v);
    if (act == MD_ACT_SILU) return dsilu_f(v);
    return 1.f;
}

// Loads row `row` of the LN input into registers: v = act(x + pos).  raw keeps the pre-activation value.
template <int NCH, bool KEEP_RAW>
__device__ __forceinline__ void load_row(const md_ln_args& p, int64_t row, int lane, float (&v)[NCH][8],
                                         float (&raw)[NCH][8]) {
    const bf16* xr = reinterpret_cast<const bf16*>(p.x) + row * p.ldx;
    const float* pr = p.pos ? reinterpret_cast<const float*>(p.pos) + (row % p.pos_rows) * p.C : nullptr;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = lane * 8 + j * 512;
        if (c < p.C) {
            const bf16x8 h = ld_bf16x8(xr + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = bf2f(h[e]);
                if (pr) f += pr[c + e];
                if (KEEP_RAW) raw[j][e] = f;
                v[j][e] = p.act ? act_fwd(f, p.act) : f;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[j][e] = 0.f;
                if (KEEP_RAW) raw[j][e] = 0.f;
            }
        }
    }
}

template <int NCH>
__global__ __launch_bounds__(256) void ln_fwd_kernel(md_ln_args p) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const float invC = 1.f / (float)p.C;
    for (int64_t row = wave; row < p.rows; row += nwaves) {
        float v[NCH][8], raw[NCH][8];
        load_row<NCH, false>(p, row, lane, v, raw);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[j][e];
        const float mean = wave_sum(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = lane * 8 + j * 512;
            if (c < p.C) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = v[j][e] - mean;
                    q += d * d;
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) * invC + p.eps);
        if (lane == 0) {
            if (p.mean) reinterpret_cast<float*>(p.mean)[row] = mean;
            if (p.rstd) reinterpret_cast<float*>(p.rstd)[row] = rstd;
        }
        const int64_t smp = p.rows_per_sample > 0 ? row / p.rows_per_sample : 0;
        const float* w = reinterpret_cast<const float*>(p.w);
        const bf16* sc = p.scale ? reinterpret_cast<const bf16*>(p.scale) + smp * p.ldmod : nullptr;
        const bf16* sh = p.shift ? reinterpret_cast<const bf16*>(p.shift) + smp * p.ldmod : nullptr;
        bf16* orow = reinterpret_cast<bf16*>(p.out) + row * p.ldo;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = lane * 8 + j * 512;
            if (c < p.C) {
                bf16x8 o;
                bf16x8 scv, shv;
                if (sc) scv = ld_bf16x8(sc + c);
                if (sh) shv = ld_bf16x8(sh + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float y = (v[j][e] - mean) * rstd;
                    if (w) y *= w[c + e];
                    if (sc) y = bf2f(f2bf(y)) * (1.f + bf2f(scv[e]));
                    if (sh) y += bf2f(shv[e]);
                    o[e] = f2bf(y);
                }
                st_bf16x8(orow + c, o);
            }
        }
    }
}

// grid = (row chunks per sample, samples); every block stays inside one sample so the per-sample column sums
// (dscale / dshift) are reduced in registers + LDS and published with one atomicAdd per column per block.
// ACT = false is the hot instantiation (no pre-norm activation): it keeps ~100 VGPRs -> 4-5 waves / SIMD.
template <int NCH, bool ACT>
__global__ __launch_bounds__(256) void ln_bwd_kernel(md_ln_args p, md_ln_bwd_args b) {
    __shared__ float red[4][64 * 8 * NCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t rps = p.rows_per_sample > 0 ? p.rows_per_sample : p.rows;
    const int64_t smp = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * b.rows_per_block;
    int64_t r1 = r0 + b.rows_per_block;
    if (r1 > rps) r1 = rps;
    const float invC = 1.f / (float)p.C;
    const float* w = reinterpret_cast<const float*>(p.w);
    const bf16* sc = p.scale ? reinterpret_cast<const bf16*>(p.scale) + smp * p.ldmod : nullptr;
    const bool want_cols = b.dscale || b.dshift || b.dw;

    float accS[NCH][8], accD[NCH][8], wm[NCH][8];   // wm = w * (1 + scale): d(out)/d(xhat) per column
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = lane * 8 + j * 512;
        bf16x8 scv;
        if (sc && c < p.C) scv = ld_bf16x8(sc + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            accS[j][e] = 0.f;
            accD[j][e] = 0.f;
            const float wv = (w && c < p.C) ? w[c + e] : 1.f;
            const float mv = (sc && c < p.C) ? 1.f + bf2f(scv[e]) : 1.f;
            wm[j][e] = wv * mv;
        }
    }
    for (int64_t lr = r0 + wave; lr < r1; lr += 4) {
        const int64_t row = smp * rps + lr;
        float v[NCH][8], raw[ACT ? NCH : 1][8];
        if (ACT) {
            float (&rw)[NCH][8] = reinterpret_cast<float (&)[NCH][8]>(raw);
            load_row<NCH, true>(p, row, lane, v, rw);
        } else {
            float dummy[NCH][8];
            load_row<NCH, false>(p, row, lane, v, dummy);
        }
        const float mean = reinterpret_cast<const float*>(p.mean)[row];
        const float rstd = reinterpret_cast<const float*>(p.rstd)[row];
        const bf16* dzr = reinterpret_cast<const bf16*>(b.dz) + row * b.lddz;
        float g[NCH][8];  // dL/dxhat
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = lane * 8 + j * 512;
            if (c < p.C) {
                const bf16x8 dzv = ld_bf16x8(dzr + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float dz = bf2f(dzv[e]);
                    const float xh = (v[j][e] - mean) * rstd;
                    v[j][e] = xh;
                    accS[j][e] += dz * xh;
                    accD[j][e] += dz;
                    const float gg = dz * wm[j][e];
                    g[j][e] = gg;
                    s1 += gg;
                    s2 += gg * xh;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) g[j][e] = 0.f;
            }
        }
        s1 = wave_sum(s1) * invC;
        s2 = wave_sum(s2) * invC;
        if (b.dx) {
            bf16* dxr = reinterpret_cast<bf16*>(b.dx) + row * b.lddx;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int c = lane * 8 + j * 512;
                if (c < p.C) {
                    bf16x8 o;
                    bf16x8 prev;
                    if (b.accumulate) prev = ld_bf16x8(dxr + c);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float d = rstd * (g[j][e] - s1 - v[j][e] * s2);
                        if (ACT) d *= act_bwd(raw[j][e], p.act);
                        if (b.accumulate) d += bf2f(prev[e]);
                        o[e] = f2bf(d);
                    }
                    st_bf16x8(dxr + c, o);
                }
            }
        }
    }
    if (!want_cols) return;
    // ---- cross-wave reduction of the column sums, then one atomic per column
    float* dscale = b.dscale ? reinterpret_cast<float*>(b.dscale) + smp * b.ldg : nullptr;
    float* dshift = b.dshift ? reinterpret_cast<float*>(b.dshift) + smp * b.ldg : nullptr;
    float* dw = reinterpret_cast<float*>(b.dw);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1 && !dshift) break;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NCH; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[wave][lane * 8 + j * 512 + e] = pass == 0 ? accS[j][e] : accD[j][e];
        __syncthreads();
        for (int c = threadIdx.x; c < p.C; c += 256) {
            const float t = red[0][c] + red[1][c] + red[2][c] + red[3][c];
            if (pass == 0) {
                const float wc = w ? w[c] : 1.f;
                if (dscale) unsafeAtomicAdd(dscale + c, wc * t);
                if (dw) {
                    const float m = sc ? 1.f + bf2f(sc[c]) : 1.f;
                    unsafeAtomicAdd(dw + c, m * t);
                }
            } else {
                unsafeAtomicAdd(dshift + c, t);
            }
        }
    }
}

template <int NCH>
__global__ __launch_bounds__(256) void qkln_fwd_kernel(bf16* buf, int64_t rows, int64_t ld, int64_t col0, int C,
                                                       float* rstd_out, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const float invC = 1.f / (float)C;
    for (int64_t row = wave; row < rows; row += nwaves) {
        bf16* r = buf + row * ld + col0;
        float v[NCH][8];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = lane * 8 + j * 512;
            if (c < C) {
                const bf16x8 h = ld_bf16x8(r + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[j][e] = bf2f(h[e]);
                    s += v[j][e];
                }
            }
        }
        const float mean = wave_sum(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = lane * 8 + j * 512;
            if (c < C) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = v[j][e] - mean;
                    q += d * d;
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) * invC + eps);
        if (lane == 0) rstd_out[row] = rstd;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = lane * 8 + j * 512;
            if (c < C) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = f2bf((v[j][e] - mean) * rstd);
                st_bf16x8(r + c, o);
            }
        }
    }
}

// d (in place) holds dL/dy on entry, dL/dx on exit; y = the normalised values written by the forward.
template <int NCH>
__global__ __launch_bounds__(256) void qkln_bwd_kernel(bf16* d, int64_t ldd, int64_t dcol0, const bf16* y, int64_t ldy,
                                                       int64_t ycol0, int64_t rows, int C, const float* rstd_in) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const float invC = 1.f / (float)C;
    for (int64_t row = wave; row < rows; row += nwaves) {
        bf16* dr = d + row * ldd + dcol0;
        const bf16* yr = y + row * ldy + ycol0;
        float g[NCH][8], xh[NCH][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = lane * 8 + j * 512;
            if (c < C) {
                const bf16x8 gv = ld_bf16x8(dr + c);
                const bf16x8 yv = ld_bf16x8(yr + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    g[j][e] = bf2f(gv[e]);
                    xh[j][e] = bf2f(yv[e]);
                    s1 += g[j][e];
                    s2 += g[j][e] * xh[j][e];
                }
            }
        }
        s1 = wave_sum(s1) * invC;
        s2 = wave_sum(s2) * invC;
        const float rstd = rstd_in[row];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = lane * 8 + j * 512;
            if (c < C) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = f2bf(rstd * (g[j][e] - s1 - xh[j][e] * s2));
                st_bf16x8(dr + c, o);
            }
        }
    }
}

inline int ln_grid(int64_t rows) {
    int64_t g = (rows + 3) / 4;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

inline bool ln_args_ok(const md_ln_args* a) {
    return a && a->x && a->rows > 0 && a->C > 0 && a->C % 8 == 0 && a->C <= 64 * 8 * MAXCH && a->ldx % 8 == 0 &&
           (!a->scale || (a->rows_per_sample > 0 && a->ldmod % 8 == 0)) &&
           (!a->shift || (a->rows_per_sample > 0 && a->ldmod % 8 == 0)) && (!a->pos || a->pos_rows > 0);
}

}  // namespace

extern "C" int md_ln_fwd(const md_ln_args* a, hipStream_t stream) {
    if (!ln_args_ok(a) || !a->out || a->ldo % 8) return MD_BAD_ARG;
    if (a->C <= 512)
        hipLaunchKernelGGL(ln_fwd_kernel<1>, dim3(ln_grid(a->rows)), dim3(256), 0, stream, *a);
    else if (a->C <= 1024)
        hipLaunchKernelGGL(ln_fwd_kernel<2>, dim3(ln_grid(a->rows)), dim3(256), 0, stream, *a);
    else
        hipLaunchKernelGGL(ln_fwd_kernel<4>, dim3(ln_grid(a->rows)), dim3(256), 0, stream, *a);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_ln_bwd(const md_ln_args* a, const md_ln_bwd_args* b, hipStream_t stream) {
    if (!ln_args_ok(a) || !b || !b->dz || !a->mean || !a->rstd || b->lddz % 8) return MD_BAD_ARG;
    if (b->dx && b->lddx % 8) return MD_BAD_ARG;
    if (b->rows_per_block <= 0) return MD_BAD_ARG;
    const int64_t rps = a->rows_per_sample > 0 ? a->rows_per_sample : a->rows;
    if (a->rows % rps) return MD_BAD_ARG;
    dim3 grid((unsigned)((rps + b->rows_per_block - 1) / b->rows_per_block), (unsigned)(a->rows / rps), 1);
#define LNB(N, A) hipLaunchKernelGGL((ln_bwd_kernel<N, A>), grid, dim3(256), 0, stream, *a, *b)
    if (a->act) {
        if (a->C <= 512) LNB(1, true); else if (a->C <= 1024) LNB(2, true); else LNB(4, true);
    } else {
        if (a->C <= 512) LNB(1, false); else if (a->C <= 1024) LNB(2, false); else LNB(4, false);
    }
#undef LNB
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_qkln_fwd(void* buf, int64_t rows, int64_t ld, int64_t col0, int64_t width, float* rstd_out, float eps,
                           hipStream_t stream) {
    if (!buf || !rstd_out || rows <= 0 || width <= 0 || width % 8 || width > 64 * 8 * MAXCH || ld % 8 || col0 % 8)
        return MD_BAD_ARG;
#define QKF(N) hipLaunchKernelGGL(qkln_fwd_kernel<N>, dim3(ln_grid(rows)), dim3(256), 0, stream, (bf16*)buf, rows, ld, \
                                  col0, (int)width, rstd_out, eps)
    if (width <= 512) QKF(1); else if (width <= 1024) QKF(2); else QKF(4);
#undef QKF
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_qkln_bwd(void* d, int64_t ldd, int64_t dcol0, const void* y, int64_t ldy, int64_t ycol0, int64_t rows,
                           int64_t width, const float* rstd, hipStream_t stream) {
    if (!d || !y || !rstd || rows <= 0 || width <= 0 || width % 8 || width > 64 * 8 * MAXCH || ldd % 8 || ldy % 8 ||
        dcol0 % 8 || ycol0 % 8)
        return MD_BAD_ARG;
#define QKB(N) hipLaunchKernelGGL(qkln_bwd_kernel<N>, dim3(ln_grid(rows)), dim3(256), 0, stream, (bf16*)d, ldd, dcol0, \
                                  (const bf16*)y, ldy, ycol0, rows, (int)width, rstd)
    if (width <= 512) QKB(1); else if (width <= 1024) QKB(2); else QKB(4);
#undef QKB
    MD_LAUNCH_CHECK();
    return 0;
}
