// LayerNorm family of the MicroDiT path: one 64-lane wave per token row, 16-byte bf16 loads, fp32 statistics by
// wave shuffles (no LDS, no barriers in the forward), adaLN modulate fused into the same pass.
//
//  md_ln_fwd   y = LN(act(x + pos)) * w ; z = y * (1 + scale[b]) + shift[b]        (w, pos, act, modulate optional)
//  md_ln_bwd   dx (+)= d/dx ; dS[b] += sum_t dz * xhat ; dshift[b] += sum_t dz ; then (finish) dscale = w * dS, dw += sum_b (1 + scale[b]) * dS
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
// GENERIC = false is the hot path (no pos table, no pre-norm activation): a pure 16-byte load + convert.
template <int NCH, bool GENERIC, bool KEEP_RAW>
__device__ __forceinline__ void load_row(const md_ln_args& p, int64_t row, int lane, float (&v)[NCH][8],
                                         float (&raw)[KEEP_RAW ? NCH : 1][8]) {
    const bf16* xr = reinterpret_cast<const bf16*>(p.x) + row * p.ldx;
    const float* pr = (GENERIC && p.pos) ? reinterpret_cast<const float*>(p.pos) + (row % p.pos_rows) * p.C : nullptr;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = lane * 8 + j * 512;
        if (c < p.C) {
            const bf16x8 h = ld_bf16x8(xr + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = bf2f(h[e]);
                if (GENERIC) {
                    if (pr) f += pr[c + e];
                    if (KEEP_RAW) raw[j][e] = f;
                    if (p.act) f = act_fwd(f, p.act);
                }
                v[j][e] = f;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[j][e] = 0.f;
                if (GENERIC && KEEP_RAW) raw[j][e] = 0.f;
            }
        }
    }
}

// Hot path (no pos table, no pre-norm activation): the 16-byte loads of a row, kept packed so that the NEXT row's loads can
// be issued before the current row's reductions (one wave has two rows of loads in flight instead of one).
template <int NCH>
__device__ __forceinline__ void issue_row(const bf16* xr, int C, int lane, bf16x8 (&h)[NCH]) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = lane * 8 + j * 512;
        if (c < C) h[j] = ld_bf16x8(xr + c);
        else
#pragma unroll
            for (int e = 0; e < 8; ++e) h[j][e] = (bf16)0.f;
    }
}
template <int NCH>
__device__ __forceinline__ void unpack_row(const bf16x8 (&h)[NCH], float (&v)[NCH][8]) {
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[j][e] = bf2f(h[j][e]);
}

template <int NCH>
__device__ __forceinline__ void load_cols_f32(const float* w, int C, int lane, float (&o)[NCH][8], float fill) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = lane * 8 + j * 512;
        if (w && c < C) {
            const float4 a = *reinterpret_cast<const float4*>(w + c), b = *reinterpret_cast<const float4*>(w + c + 4);
            o[j][0] = a.x; o[j][1] = a.y; o[j][2] = a.z; o[j][3] = a.w;
            o[j][4] = b.x; o[j][5] = b.y; o[j][6] = b.z; o[j][7] = b.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[j][e] = fill;
        }
    }
}

// Each wave owns a run of consecutive rows (so the per-sample scale / shift vectors are re-read only when the
// sample changes); the LN weight lives in registers for the whole run.
template <int NCH, bool GENERIC>
__global__ __launch_bounds__(256) void ln_fwd_kernel(md_ln_args p, int64_t rows_per_wave) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t r0 = wave * rows_per_wave;
    int64_t r1 = r0 + rows_per_wave;
    if (r1 > p.rows) r1 = p.rows;
    const float invC = 1.f / (float)p.C;
    float wk[NCH][8];
    load_cols_f32<NCH>(reinterpret_cast<const float*>(p.w), (int)p.C, lane, wk, 1.f);
    const bool mod = p.scale != nullptr;
    bf16x8 nxt[NCH];
    if (!GENERIC && r0 < r1) issue_row<NCH>(reinterpret_cast<const bf16*>(p.x) + r0 * p.ldx, (int)p.C, lane, nxt);
    for (int64_t row = r0; row < r1; ++row) {
        float v[NCH][8], raw[1][8];
        if (GENERIC) {
            load_row<NCH, GENERIC, false>(p, row, lane, v, raw);
        } else {
            unpack_row<NCH>(nxt, v);
            if (row + 1 < r1) issue_row<NCH>(reinterpret_cast<const bf16*>(p.x) + (row + 1) * p.ldx, (int)p.C, lane, nxt);
        }
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
        bf16* orow = reinterpret_cast<bf16*>(p.out) + row * p.ldo;
        const int64_t smp = mod ? row / p.rows_per_sample : 0;
        const bf16* sc = mod ? reinterpret_cast<const bf16*>(p.scale) + smp * p.ldmod : nullptr;
        const bf16* sh = (mod && p.shift) ? reinterpret_cast<const bf16*>(p.shift) + smp * p.ldmod : nullptr;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = lane * 8 + j * 512;
            if (c < p.C) {
                bf16x8 o, scv, shv;
                if (mod) scv = ld_bf16x8(sc + c);     // L1/L2-resident per-sample vectors
                if (sh) shv = ld_bf16x8(sh + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float y = (v[j][e] - mean) * rstd * wk[j][e];
                    if (mod) y = bf2f(f2bf(y)) * (1.f + bf2f(scv[e])) + (sh ? bf2f(shv[e]) : 0.f);
                    o[e] = f2bf(y);
                }
                st_bf16x8(orow + c, o);
            }
        }
    }
}

// grid = (row chunks per sample, samples); every workgroup stays inside one sample: its per-column sums
//   dS[b, c] += sum_t dz * xhat        dshift[b, c] += sum_t dz
// are reduced in registers + LDS and published with one atomicAdd per column (few workgroups per sample -> little
// contention).  dscale = w * dS and dw = sum_b (1 + scale_b) * dS are finished by ln_bwd_finish_kernel: summing dw with
// atomics straight from here serialised ~2000 adds per address and tripled the kernel time.
template <int NCH, bool GENERIC>
__global__ __launch_bounds__(256) void ln_bwd_kernel(md_ln_args p, md_ln_bwd_args b) {
    __shared__ float red[4][64 * 8 * NCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t rps = p.rows_per_sample > 0 ? p.rows_per_sample : p.rows;
    const int64_t smp = blockIdx.y;
    const int64_t b0 = (int64_t)blockIdx.x * b.rows_per_block;
    int64_t b1 = b0 + b.rows_per_block;
    if (b1 > rps) b1 = rps;
    const int64_t per_wave = (b1 - b0 + 3) / 4;
    const int64_t r0 = b0 + wave * per_wave;
    int64_t r1 = r0 + per_wave;
    if (r1 > b1) r1 = b1;
    const float invC = 1.f / (float)p.C;
    const bf16* sc = p.scale ? reinterpret_cast<const bf16*>(p.scale) + smp * p.ldmod : nullptr;
    const bool want_cols = b.dscale || b.dshift;

    float accS[NCH][8], accD[NCH][8], wm[NCH][8];   // wm = w * (1 + scale): d(out)/d(xhat) per column
    load_cols_f32<NCH>(reinterpret_cast<const float*>(p.w), (int)p.C, lane, wm, 1.f);
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = lane * 8 + j * 512;
        bf16x8 scv;
        if (sc && c < p.C) scv = ld_bf16x8(sc + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            accS[j][e] = 0.f;
            accD[j][e] = 0.f;
            if (sc && c < p.C) wm[j][e] *= 1.f + bf2f(scv[e]);
        }
    }
    bf16x8 nx[NCH], nz[NCH];
    float nmean = 0.f, nrstd = 0.f;
    if (r0 < r1) {
        const int64_t row = smp * rps + r0;
        if (!GENERIC) issue_row<NCH>(reinterpret_cast<const bf16*>(p.x) + row * p.ldx, (int)p.C, lane, nx);
        issue_row<NCH>(reinterpret_cast<const bf16*>(b.dz) + row * b.lddz, (int)p.C, lane, nz);
        nmean = reinterpret_cast<const float*>(p.mean)[row];
        nrstd = reinterpret_cast<const float*>(p.rstd)[row];
    }
    for (int64_t lr = r0; lr < r1; ++lr) {
        const int64_t row = smp * rps + lr;
        float v[NCH][8], raw[GENERIC ? NCH : 1][8];
        if (GENERIC) load_row<NCH, GENERIC, GENERIC>(p, row, lane, v, raw);
        else unpack_row<NCH>(nx, v);
        const float mean = nmean, rstd = nrstd;
        // dL/dxhat = dz * wm is formed twice (here for the two row sums, below for dx) from the PACKED dz kept in dzk: half the
        // registers of an fp32 copy, which is what takes the NCH = 2 kernel from 134 to under 128 VGPRs (4 waves / SIMD)
        bf16x8 dzk[NCH];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = lane * 8 + j * 512;
            dzk[j] = nz[j];
            if (c < p.C) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float dz = bf2f(dzk[j][e]);
                    const float xh = (v[j][e] - mean) * rstd;
                    v[j][e] = xh;
                    accS[j][e] += dz * xh;
                    accD[j][e] += dz;
                    const float gg = dz * wm[j][e];
                    s1 += gg;
                    s2 += gg * xh;
                }
            }
        }
        if (lr + 1 < r1) {    // the next row's loads fly during this row's reductions and dx pass (nx / nz are consumed: no extra registers)
            if (!GENERIC) issue_row<NCH>(reinterpret_cast<const bf16*>(p.x) + (row + 1) * p.ldx, (int)p.C, lane, nx);
            issue_row<NCH>(reinterpret_cast<const bf16*>(b.dz) + (row + 1) * b.lddz, (int)p.C, lane, nz);
            nmean = reinterpret_cast<const float*>(p.mean)[row + 1];
            nrstd = reinterpret_cast<const float*>(p.rstd)[row + 1];
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
                        float d = rstd * (bf2f(dzk[j][e]) * wm[j][e] - s1 - v[j][e] * s2);
                        if (GENERIC && p.act) d *= act_bwd(raw[j][e], p.act);
                        if (b.accumulate) d += bf2f(prev[e]);
                        o[e] = f2bf(d);
                    }
                    st_bf16x8(dxr + c, o);
                }
            }
        }
    }
    if (!want_cols) return;
    float* dS = b.dscale ? reinterpret_cast<float*>(b.dscale) + smp * b.ldg : nullptr;
    float* dshift = b.dshift ? reinterpret_cast<float*>(b.dshift) + smp * b.ldg : nullptr;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        float* dst = pass == 0 ? dS : dshift;
        if (!dst) continue;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NCH; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[wave][lane * 8 + j * 512 + e] = pass == 0 ? accS[j][e] : accD[j][e];
        __syncthreads();
        for (int c = threadIdx.x; c < p.C; c += 256) unsafeAtomicAdd(dst + c, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
    }
}

// dw[c] += sum_b (1 + scale[b, c]) * dS[b, c];  optionally dS[b, c] *= w[c]  (dS becomes dscale).
// grid = (column blocks of 256, sample chunks of 16): independent loads, one atomic per column per chunk.
__global__ __launch_bounds__(256) void ln_bwd_finish_kernel(float* dS, int64_t ldg, const float* w, const bf16* scale, int64_t ldmod,
                                                            float* dw, int64_t B, int64_t C, int to_dscale) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int64_t b0 = (int64_t)blockIdx.y * 16;
    const float wc = w ? w[c] : 1.f;
    float sv[16], mv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t b = b0 + i;
        sv[i] = b < B ? dS[b * ldg + c] : 0.f;
        mv[i] = (scale && b < B) ? 1.f + bf2f(scale[b * ldmod + c]) : 1.f;
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        acc += mv[i] * sv[i];
        if (to_dscale && b0 + i < B) dS[(b0 + i) * ldg + c] = sv[i] * wc;
    }
    if (dw) unsafeAtomicAdd(dw + c, acc);
}

template <int NCH>
__global__ __launch_bounds__(256) void qkln_fwd_kernel(bf16* buf, int64_t rows, int64_t ld, int64_t col0, int C, int nseg,
                                                       int64_t seg_stride, float* rstd_out, float eps) {
    // work item vr = row * nseg + seg: segment seg (q, k, ...) of a row starts seg_stride elements after the previous one, so the
    // q and k halves of a packed qkv row are normalised by ONE launch (they were two, each too short to fill the chip at 16 k rows)
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int64_t items = rows * nseg;
    const float invC = 1.f / (float)C;
    auto at = [&](int64_t vr) { return buf + (vr / nseg) * ld + col0 + (vr % nseg) * seg_stride; };
    bf16x8 nxt[NCH];
    if (wave < items) issue_row<NCH>(at(wave), C, lane, nxt);
    for (int64_t vr = wave; vr < items; vr += nwaves) {
        bf16* r = at(vr);
        float v[NCH][8];
        unpack_row<NCH>(nxt, v);
        if (vr + nwaves < items) issue_row<NCH>(at(vr + nwaves), C, lane, nxt);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[j][e];          // columns >= C were loaded as zeros
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
        if (lane == 0) rstd_out[(vr % nseg) * rows + vr / nseg] = rstd;
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
                                                       int64_t ycol0, int64_t rows, int C, int nseg, int64_t dseg_stride,
                                                       int64_t yseg_stride, const float* rstd_in) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int64_t items = rows * nseg;                      // item vr = row * nseg + seg (see qkln_fwd_kernel)
    const float invC = 1.f / (float)C;
    auto dat = [&](int64_t vr) { return d + (vr / nseg) * ldd + dcol0 + (vr % nseg) * dseg_stride; };
    auto yat = [&](int64_t vr) { return y + (vr / nseg) * ldy + ycol0 + (vr % nseg) * yseg_stride; };
    bf16x8 ng[NCH], ny[NCH];
    float nrstd = 0.f;
    if (wave < items) {
        issue_row<NCH>(dat(wave), C, lane, ng);
        issue_row<NCH>(yat(wave), C, lane, ny);
        nrstd = rstd_in[(wave % nseg) * rows + wave / nseg];
    }
    for (int64_t vr = wave; vr < items; vr += nwaves) {
        bf16* dr = dat(vr);
        float g[NCH][8], xh[NCH][8];
        unpack_row<NCH>(ng, g);
        unpack_row<NCH>(ny, xh);
        const float rstd = nrstd;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s1 += g[j][e];
                s2 += g[j][e] * xh[j][e];
            }
        if (vr + nwaves < items) {
            const int64_t nx = vr + nwaves;
            issue_row<NCH>(dat(nx), C, lane, ng);
            issue_row<NCH>(yat(nx), C, lane, ny);
            nrstd = rstd_in[(nx % nseg) * rows + nx / nseg];
        }
        s1 = wave_sum(s1) * invC;
        s2 = wave_sum(s2) * invC;
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

// ---------------------------------------------------------------------------------------------------------------------
// Head-major forms (round 6): the normalised q / k go to a separate [seg][B, H, S, hd] buffer -- one contiguous [S, hd] block per
// (sample, head), the tile an attention workgroup stages -- and dL/dy comes back in that layout.  A lane's 16-byte piece (8 columns)
// lies inside one head, 8 lanes cover a head's 128 bytes (hd = 64): every store / load instruction moves whole 128-byte lines,
// as the row-major form does.  Same arithmetic, same statistics buffer as the in-place kernels above.
// ---------------------------------------------------------------------------------------------------------------------
struct HmMap {
    int64_t S, H;
    int hd_shift;      // hd = 1 << hd_shift (32 or 64)
    __device__ __forceinline__ int64_t row_base(int64_t row) const {      // offset of (b, head 0, s, 0)
        const int64_t b = row / S, s = row - b * S;
        return (b * H * S + s) << hd_shift;
    }
    __device__ __forceinline__ int64_t col_off(int c) const {             // + offset of column c = h * hd + e
        return ((int64_t)(c >> hd_shift) * S << hd_shift) + (c & ((1 << hd_shift) - 1));
    }
};
template <int NCH>
__device__ __forceinline__ void issue_row_hm(const bf16* base, const HmMap& m, int C, int lane, bf16x8 (&h)[NCH]) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = lane * 8 + j * 512;
        if (c < C) h[j] = ld_bf16x8(base + m.col_off(c));
        else
#pragma unroll
            for (int e = 0; e < 8; ++e) h[j][e] = (bf16)0.f;
    }
}

template <int NCH>
__global__ __launch_bounds__(256) void qkln_fwd_hm_kernel(const bf16* buf, int64_t rows, int64_t ld, int64_t col0, int C, int nseg,
                                                          int64_t seg_stride, bf16* out, int64_t out_seg_stride, HmMap m,
                                                          float* rstd_out, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int64_t items = rows * nseg;                      // item vr = row * nseg + seg (see qkln_fwd_kernel)
    const float invC = 1.f / (float)C;
    auto at = [&](int64_t vr) { return buf + (vr / nseg) * ld + col0 + (vr % nseg) * seg_stride; };
    bf16x8 nxt[NCH];
    if (wave < items) issue_row<NCH>(at(wave), C, lane, nxt);
    for (int64_t vr = wave; vr < items; vr += nwaves) {
        float v[NCH][8];
        unpack_row<NCH>(nxt, v);
        if (vr + nwaves < items) issue_row<NCH>(at(vr + nwaves), C, lane, nxt);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[j][e];          // columns >= C were loaded as zeros
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
        const int64_t row = vr / nseg, seg = vr % nseg;
        if (lane == 0) rstd_out[seg * rows + row] = rstd;
        bf16* orow = out + seg * out_seg_stride + m.row_base(row);
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = lane * 8 + j * 512;
            if (c < C) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = f2bf((v[j][e] - mean) * rstd);
                st_bf16x8(orow + m.col_off(c), o);
            }
        }
    }
}

template <int NCH>
__global__ __launch_bounds__(256) void qkln_bwd_hm_kernel(const bf16* dy, int64_t dy_seg_stride, const bf16* y, int64_t y_seg_stride,
                                                          bf16* d, int64_t ldd, int64_t dcol0, int64_t dseg_stride, int64_t rows, int C,
                                                          int nseg, HmMap m, const float* rstd_in) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int64_t items = rows * nseg;
    const float invC = 1.f / (float)C;
    bf16x8 ng[NCH], ny[NCH];
    float nrstd = 0.f;
    auto issue = [&](int64_t vr) {
        const int64_t row = vr / nseg, seg = vr % nseg, rb = m.row_base(row);
        issue_row_hm<NCH>(dy + seg * dy_seg_stride + rb, m, C, lane, ng);
        issue_row_hm<NCH>(y + seg * y_seg_stride + rb, m, C, lane, ny);
        nrstd = rstd_in[seg * rows + row];
    };
    if (wave < items) issue(wave);
    for (int64_t vr = wave; vr < items; vr += nwaves) {
        bf16* dr = d + (vr / nseg) * ldd + dcol0 + (vr % nseg) * dseg_stride;
        float g[NCH][8], xh[NCH][8];
        unpack_row<NCH>(ng, g);
        unpack_row<NCH>(ny, xh);
        const float rstd = nrstd;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s1 += g[j][e];
                s2 += g[j][e] * xh[j][e];
            }
        if (vr + nwaves < items) issue(vr + nwaves);
        s1 = wave_sum(s1) * invC;
        s2 = wave_sum(s2) * invC;
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
    // consecutive rows per wave: enough waves to fill the chip (>= 8192), at most 8 rows each
    int64_t rpw = (a->rows + 16383) / 16384;
    if (rpw < 1) rpw = 1;
    if (rpw > 8) rpw = 8;
    const int64_t waves = (a->rows + rpw - 1) / rpw;
    dim3 grid((unsigned)((waves + 3) / 4));
    const bool generic = a->act != MD_ACT_NONE || a->pos != nullptr;
#define LNF(N, G) hipLaunchKernelGGL((ln_fwd_kernel<N, G>), grid, dim3(256), 0, stream, *a, rpw)
    if (generic) {
        if (a->C <= 512) LNF(1, true); else if (a->C <= 1024) LNF(2, true); else LNF(4, true);
    } else {
        if (a->C <= 512) LNF(1, false); else if (a->C <= 1024) LNF(2, false); else LNF(4, false);
    }
#undef LNF
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_ln_bwd(const md_ln_args* a, const md_ln_bwd_args* b, hipStream_t stream) {
    if (!ln_args_ok(a) || !b || !b->dz || !a->mean || !a->rstd || b->lddz % 8) return MD_BAD_ARG;
    if (b->dx && b->lddx % 8) return MD_BAD_ARG;
    if (b->rows_per_block <= 0) return MD_BAD_ARG;
    if (b->dw && !b->dscale) return MD_BAD_ARG;   // dw is finished from the per-sample sums: a dS buffer is required
    const int64_t rps = a->rows_per_sample > 0 ? a->rows_per_sample : a->rows;
    if (a->rows % rps) return MD_BAD_ARG;
    const int64_t nsmp = a->rows / rps;
    dim3 grid((unsigned)((rps + b->rows_per_block - 1) / b->rows_per_block), (unsigned)nsmp, 1);
    const bool generic = a->act != MD_ACT_NONE || a->pos != nullptr;
#define LNB(N, G) hipLaunchKernelGGL((ln_bwd_kernel<N, G>), grid, dim3(256), 0, stream, *a, *b)
    if (generic) {
        if (a->C <= 512) LNB(1, true); else if (a->C <= 1024) LNB(2, true); else LNB(4, true);
    } else {
        if (a->C <= 512) LNB(1, false); else if (a->C <= 1024) LNB(2, false); else LNB(4, false);
    }
#undef LNB
    if (b->dscale && (b->dw || b->dscale_is_output))
        hipLaunchKernelGGL(ln_bwd_finish_kernel, dim3((unsigned)((a->C + 255) / 256), (unsigned)((nsmp + 15) / 16)), dim3(256), 0, stream,
                           reinterpret_cast<float*>(b->dscale), b->ldg, reinterpret_cast<const float*>(a->w),
                           reinterpret_cast<const bf16*>(a->scale), a->ldmod, reinterpret_cast<float*>(b->dw), nsmp, a->C,
                           b->dscale_is_output);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_qkln_fwd(void* buf, int64_t rows, int64_t ld, int64_t col0, int64_t width, int32_t nseg, int64_t seg_stride,
                           float* rstd_out, float eps, hipStream_t stream) {
    if (!buf || !rstd_out || rows <= 0 || width <= 0 || width % 8 || width > 64 * 8 * MAXCH || ld % 8 || col0 % 8 || nseg < 1 ||
        nseg > 4 || seg_stride % 8 || (nseg > 1 && seg_stride < width))
        return MD_BAD_ARG;
#define QKF(N) hipLaunchKernelGGL(qkln_fwd_kernel<N>, dim3(ln_grid(rows * nseg)), dim3(256), 0, stream, (bf16*)buf, rows, ld, \
                                  col0, (int)width, (int)nseg, seg_stride, rstd_out, eps)
    if (width <= 512) QKF(1); else if (width <= 1024) QKF(2); else QKF(4);
#undef QKF
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_qkln_bwd(void* d, int64_t ldd, int64_t dcol0, const void* y, int64_t ldy, int64_t ycol0, int64_t rows,
                           int64_t width, int32_t nseg, int64_t dseg_stride, int64_t yseg_stride, const float* rstd,
                           hipStream_t stream) {
    if (!d || !y || !rstd || rows <= 0 || width <= 0 || width % 8 || width > 64 * 8 * MAXCH || ldd % 8 || ldy % 8 ||
        dcol0 % 8 || ycol0 % 8 || nseg < 1 || nseg > 4 || dseg_stride % 8 || yseg_stride % 8)
        return MD_BAD_ARG;
#define QKB(N) hipLaunchKernelGGL(qkln_bwd_kernel<N>, dim3(ln_grid(rows * nseg)), dim3(256), 0, stream, (bf16*)d, ldd, dcol0, \
                                  (const bf16*)y, ldy, ycol0, rows, (int)width, (int)nseg, dseg_stride, yseg_stride, rstd)
    if (width <= 512) QKB(1); else if (width <= 1024) QKB(2); else QKB(4);
#undef QKB
    MD_LAUNCH_CHECK();
    return 0;
}

namespace {
inline bool hm_ok(int64_t rows, int64_t width, int64_t S, int32_t hd) {
    return S > 0 && rows % S == 0 && (hd == 32 || hd == 64) && width % hd == 0;
}
}  // namespace

extern "C" int md_qkln_fwd_hm(const void* buf, int64_t rows, int64_t ld, int64_t col0, int64_t width, int32_t nseg, int64_t seg_stride,
                              void* out, int64_t out_seg_stride, int64_t S, int32_t hd, float* rstd_out, float eps, hipStream_t stream) {
    if (!buf || !out || !rstd_out || rows <= 0 || width <= 0 || width % 8 || width > 64 * 8 * MAXCH || ld % 8 || col0 % 8 || nseg < 1 ||
        nseg > 4 || seg_stride % 8 || (nseg > 1 && (seg_stride < width || out_seg_stride < rows * width)) || out_seg_stride % 8 ||
        !hm_ok(rows, width, S, hd))
        return MD_BAD_ARG;
    const HmMap m{S, width / hd, hd == 64 ? 6 : 5};
#define QKF(N) hipLaunchKernelGGL(qkln_fwd_hm_kernel<N>, dim3(ln_grid(rows * nseg)), dim3(256), 0, stream, (const bf16*)buf, rows, ld, \
                                  col0, (int)width, (int)nseg, seg_stride, (bf16*)out, out_seg_stride, m, rstd_out, eps)
    if (width <= 512) QKF(1); else if (width <= 1024) QKF(2); else QKF(4);
#undef QKF
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_qkln_bwd_hm(const void* dy, int64_t dy_seg_stride, const void* y, int64_t y_seg_stride, void* d, int64_t ldd,
                              int64_t dcol0, int64_t dseg_stride, int64_t rows, int64_t width, int32_t nseg, int64_t S, int32_t hd,
                              const float* rstd, hipStream_t stream) {
    if (!dy || !y || !d || !rstd || rows <= 0 || width <= 0 || width % 8 || width > 64 * 8 * MAXCH || ldd % 8 || dcol0 % 8 ||
        nseg < 1 || nseg > 4 || dseg_stride % 8 || dy_seg_stride % 8 || y_seg_stride % 8 || !hm_ok(rows, width, S, hd))
        return MD_BAD_ARG;
    const HmMap m{S, width / hd, hd == 64 ? 6 : 5};
#define QKB(N) hipLaunchKernelGGL(qkln_bwd_hm_kernel<N>, dim3(ln_grid(rows * nseg)), dim3(256), 0, stream, (const bf16*)dy, dy_seg_stride, \
                                  (const bf16*)y, y_seg_stride, (bf16*)d, ldd, dcol0, dseg_stride, rows, (int)width, (int)nseg, m, rstd)
    if (width <= 512) QKB(1); else if (width <= 1024) QKB(2); else QKB(4);
#undef QKB
    MD_LAUNCH_CHECK();
    return 0;
}
