// HBM-bound elementwise / column-reduction kernels of the MicroDiT path (16-byte accesses, grid-stride loops).
//   SwiGLU (dit.py:88-89), adaLN-Zero gate backward (dit.py:236,238), GELU on the condition vector (dit.py:223),
//   bias-gradient column sums, token mean pooling (dit.py:484), dtype casts (model.py:132-139).
#include "md_common.h"
#include "../../include/microdit_hip.h"

namespace {

__device__ __forceinline__ float act_f(float v, int act) {
    if (act == MD_ACT_GELU_TANH) return gelu_tanh_f(v);
    if (act == MD_ACT_GELU_ERF) return gelu_erf_f(v);
    if (act == MD_ACT_SILU) return silu_f(v);
    return v;
}
__device__ __forceinline__ float dact_f(float v, int act) {
    if (act == MD_ACT_GELU_TANH) return dgelu_tanh_f(v);
    if (act == MD_ACT_GELU_ERF) return dgelu_erf_f(v);
    if (act == MD_ACT_SILU) return dsilu_f(v);
    return 1.f;
}

inline int ew_grid(int64_t work_items) {
    int64_t g = (work_items + 255) / 256;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return (int)g;
}

// SwiGLU, a[m, c] = silu(h12[m, c]) * h12[m, f + c], and its backward.  grid = (column blocks of 64 lanes x 8 elements, row
// groups): a lane keeps its column chunk, the four waves of a workgroup take interleaved rows, and a thread walks RPT rows per
// iteration with every load issued before the first use.  (The first version was a flat grid-stride loop over 16-byte chunks:
// ~60 instructions of 64-bit i / cpr, i % cpr per chunk and ONE chunk per thread in flight; 3.3 TB/s in the step.)
__device__ __forceinline__ void swiglu_fwd8(const bf16x8& h1, const bf16x8& h2, bf16x8& o) {
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(f2bf(silu_f(bf2f(h1[e])))) * bf2f(h2[e]));
}
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16* __restrict__ h12, int64_t ldh, bf16* __restrict__ a, int64_t lda,
                                                         int64_t M, int64_t f, int64_t rows_per_block) {
    constexpr int RPT = 4;
    const int64_t c = ((int64_t)blockIdx.x * 64 + (threadIdx.x & 63)) * 8;
    if (c >= f) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block + (threadIdx.x >> 6);
    int64_t r1 = (int64_t)blockIdx.y * rows_per_block + rows_per_block;
    if (r1 > M) r1 = M;
    const bf16* p1 = h12 + r0 * ldh + c;
    bf16* po = a + r0 * lda + c;
    const int64_t sh = 4 * ldh, so = 4 * lda;                       // the next row of this wave
    int64_t m = r0;
    for (; m + 4 * (RPT - 1) < r1; m += 4 * RPT, p1 += RPT * sh, po += RPT * so) {
        bf16x8 h1[RPT], h2[RPT];
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            h1[i] = ld_bf16x8(p1 + i * sh);
            h2[i] = ld_bf16x8(p1 + i * sh + f);
        }
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            bf16x8 o;
            swiglu_fwd8(h1[i], h2[i], o);
            st_bf16x8(po + i * so, o);
        }
    }
    for (; m < r1; m += 4, p1 += sh, po += so) {
        bf16x8 o;
        swiglu_fwd8(ld_bf16x8(p1), ld_bf16x8(p1 + f), o);
        st_bf16x8(po, o);
    }
}

__device__ __forceinline__ void swiglu_bwd8(const bf16x8& h1, const bf16x8& h2, const bf16x8& d, bf16x8& o1, bf16x8& o2) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x1 = bf2f(h1[e]), x2 = bf2f(h2[e]), g = bf2f(d[e]);
        const float sg = fast_rcp(1.f + __expf(-x1));            // one exp + one rcp for silu AND its derivative
        o1[e] = f2bf(g * x2 * (sg * (1.f + x1 * (1.f - sg))));
        o2[e] = f2bf(g * (x1 * sg));
    }
}

__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16* __restrict__ da, int64_t ldda, const bf16* __restrict__ h12,
                                                         int64_t ldh, bf16* __restrict__ dh12, int64_t lddh, int64_t M, int64_t f,
                                                         int64_t rows_per_block) {
    constexpr int RPT = 2;                                           // three streams in: 2 rows x 3 x 16 bytes per thread in flight
    const int64_t c = ((int64_t)blockIdx.x * 64 + (threadIdx.x & 63)) * 8;
    if (c >= f) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block + (threadIdx.x >> 6);
    int64_t r1 = (int64_t)blockIdx.y * rows_per_block + rows_per_block;
    if (r1 > M) r1 = M;
    const bf16* ph = h12 + r0 * ldh + c;
    const bf16* pd = da + r0 * ldda + c;
    bf16* po = dh12 + r0 * lddh + c;
    const int64_t sh = 4 * ldh, sd = 4 * ldda, so = 4 * lddh;
    int64_t m = r0;
    for (; m + 4 * (RPT - 1) < r1; m += 4 * RPT, ph += RPT * sh, pd += RPT * sd, po += RPT * so) {
        bf16x8 h1[RPT], h2[RPT], d[RPT];
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            h1[i] = ld_bf16x8(ph + i * sh);
            h2[i] = ld_bf16x8(ph + i * sh + f);
            d[i] = ld_bf16x8(pd + i * sd);
        }
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            bf16x8 o1, o2;
            swiglu_bwd8(h1[i], h2[i], d[i], o1, o2);
            st_bf16x8(po + i * so, o1);
            st_bf16x8(po + i * so + f, o2);
        }
    }
    for (; m < r1; m += 4, ph += sh, pd += sd, po += so) {
        bf16x8 o1, o2;
        swiglu_bwd8(ld_bf16x8(ph), ld_bf16x8(ph + f), ld_bf16x8(pd), o1, o2);
        st_bf16x8(po, o1);
        st_bf16x8(po + f, o2);
    }
}

// rows per workgroup of the two kernels above (a multiple of 4: one row per wave and step): >= ~16 workgroups per CU, 16..256 rows
inline int64_t swiglu_rows_per_block(int64_t M, int64_t f) {
    const int64_t colblocks = (f / 8 + 63) / 64;
    int64_t rpb = 256;
    while (rpb > 16 && colblocks * ((M + rpb - 1) / rpb) < 4096) rpb /= 2;
    return rpb;
}

// dbr = gate[b] * dx ; dgate[b, c] += sum_t dx * br.   grid = (row chunks per sample, samples), wave per row.
template <int NCH>
__global__ __launch_bounds__(256) void gate_bwd_kernel(const bf16* dx, const bf16* br, const bf16* gate, int64_t ldgate,
                                                       bf16* dbr, float* dgate, int64_t lddg, int64_t C, int64_t rps,
                                                       int64_t rows_per_block) {
    __shared__ float red[4][64 * 8 * NCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t smp = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > rps) r1 = rps;
    float acc[NCH][8], gk[NCH][8];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = lane * 8 + j * 512;
        bf16x8 gv;
        if (c < C) gv = ld_bf16x8(gate + smp * ldgate + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            acc[j][e] = 0.f;
            gk[j][e] = c < C ? bf2f(gv[e]) : 0.f;
        }
    }
    for (int64_t lr = r0 + wave; lr < r1; lr += 4) {
        const int64_t row = smp * rps + lr;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = lane * 8 + j * 512;
            if (c < C) {
                const bf16x8 d = ld_bf16x8(dx + row * C + c), b = ld_bf16x8(br + row * C + c);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float dv = bf2f(d[e]);
                    acc[j][e] += dv * bf2f(b[e]);
                    o[e] = f2bf(dv * gk[j][e]);
                }
                st_bf16x8(dbr + row * C + c, o);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) red[wave][lane * 8 + j * 512 + e] = acc[j][e];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256)
        unsafeAtomicAdd(dgate + smp * lddg + c, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
}

__global__ __launch_bounds__(256) void act_fwd_kernel(const bf16* x, bf16* y, int64_t n8, int act) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const bf16x8 v = ld_bf16x8(x + i * 8);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(act_f(bf2f(v[e]), act));
        st_bf16x8(y + i * 8, o);
    }
}

// dx(bf16) = dy(f32) * act'(x)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* dy, const bf16* x, bf16* dx, int64_t n8, int act) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const bf16x8 v = ld_bf16x8(x + i * 8);
        const float4 d0 = *reinterpret_cast<const float4*>(dy + i * 8);
        const float4 d1 = *reinterpret_cast<const float4*>(dy + i * 8 + 4);
        const float d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(d[e] * dact_f(bf2f(v[e]), act));
        st_bf16x8(dx + i * 8, o);
    }
}

// out[c] += sum_rows x[row, c]; block = 256 columns x `rows_per_block` rows.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* x, int64_t ld, float* out, int64_t rows, int64_t C,
                                                     int64_t rows_per_block) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    float s = 0.f;
    for (int64_t r = r0; r < r1; ++r) s += (float)x[r * ld + c];
    unsafeAtomicAdd(out + c, s);
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* x, bf16* y, int64_t n8, const float* scale_ptr) {
    const float sc = scale_ptr ? *scale_ptr : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const float4 d0 = *reinterpret_cast<const float4*>(x + i * 8);
        const float4 d1 = *reinterpret_cast<const float4*>(x + i * 8 + 4);
        bf16x8 o;
        o[0] = f2bf(d0.x * sc); o[1] = f2bf(d0.y * sc); o[2] = f2bf(d0.z * sc); o[3] = f2bf(d0.w * sc);
        o[4] = f2bf(d1.x * sc); o[5] = f2bf(d1.y * sc); o[6] = f2bf(d1.z * sc); o[7] = f2bf(d1.w * sc);
        st_bf16x8(y + i * 8, o);
    }
}

// y = bf16(x); x = 0: the data-parallel exchange stages a bucket of the fp32 gradient accumulators as bf16 and hands the
// accumulators back cleared in the same pass (with a sharded optimiser step no later kernel visits the whole accumulator)
__global__ __launch_bounds__(256) void cast_f32_bf16_clear_kernel(float* x, bf16* y, int64_t n8) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const float4 d0 = *reinterpret_cast<const float4*>(x + i * 8);
        const float4 d1 = *reinterpret_cast<const float4*>(x + i * 8 + 4);
        bf16x8 o;
        o[0] = f2bf(d0.x); o[1] = f2bf(d0.y); o[2] = f2bf(d0.z); o[3] = f2bf(d0.w);
        o[4] = f2bf(d1.x); o[5] = f2bf(d1.y); o[6] = f2bf(d1.z); o[7] = f2bf(d1.w);
        st_bf16x8(y + i * 8, o);
        *reinterpret_cast<float4*>(x + i * 8) = z;
        *reinterpret_cast<float4*>(x + i * 8 + 4) = z;
    }
}

// out(bf16)[row, :] = in[row, :] * rowscale[row / rows_per_sample];  IN = _Float16 or float
template <typename IN>
__global__ __launch_bounds__(256) void cast_rows_kernel(const IN* x, bf16* y, int64_t rows, int64_t C,
                                                        const float* rowscale, int64_t rps) {
    const int64_t cpr = C / 8, total = rows * cpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cpr, c = (i % cpr) * 8;
        const float sc = rowscale ? rowscale[r / rps] : 1.f;
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf((float)x[r * C + c + e] * sc);
        st_bf16x8(y + r * C + c, o);
    }
}

// out[b, c] = mean_l y[b, l, c]
__global__ __launch_bounds__(256) void mean_tokens_kernel(const bf16* y, bf16* out, int64_t B, int64_t L, int64_t C) {
    const int64_t cpr = C / 8, total = B * cpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / cpr, c = (i % cpr) * 8;
        float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int64_t l = 0; l < L; ++l) {
            const bf16x8 v = ld_bf16x8(y + (b * L + l) * C + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += bf2f(v[e]);
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(s[e] / (float)L);
        st_bf16x8(out + b * C + c, o);
    }
}

// dy(f32)[b, l, c] += dpool[b, c] / L
__global__ __launch_bounds__(256) void mean_tokens_bwd_kernel(const bf16* dpool, float* dy, int64_t B, int64_t L, int64_t C) {
    const int64_t cpr = C / 8, total = B * L * cpr;
    const float inv = 1.f / (float)L;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / cpr, c = (i % cpr) * 8, b = row / L;
        const bf16x8 d = ld_bf16x8(dpool + b * C + c);
        float* p = dy + row * C + c;
        float4 a0 = *reinterpret_cast<float4*>(p), a1 = *reinterpret_cast<float4*>(p + 4);
        a0.x += bf2f(d[0]) * inv; a0.y += bf2f(d[1]) * inv; a0.z += bf2f(d[2]) * inv; a0.w += bf2f(d[3]) * inv;
        a1.x += bf2f(d[4]) * inv; a1.y += bf2f(d[5]) * inv; a1.z += bf2f(d[6]) * inv; a1.w += bf2f(d[7]) * inv;
        *reinterpret_cast<float4*>(p) = a0;
        *reinterpret_cast<float4*>(p + 4) = a1;
    }
}

// y(bf16) = a(bf16) + b(bf16), optional f32 addend
__global__ __launch_bounds__(256) void add_bf16_kernel(const bf16* a, const bf16* b, bf16* y, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const bf16x8 u = ld_bf16x8(a + i * 8), v = ld_bf16x8(b + i * 8);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(u[e]) + bf2f(v[e]));
        st_bf16x8(y + i * 8, o);
    }
}

}  // namespace

extern "C" int md_swiglu_fwd(const void* h12, int64_t ldh, void* a, int64_t lda, int64_t M, int64_t f, hipStream_t st) {
    if (!h12 || !a || M <= 0 || f <= 0 || f % 8 || ldh % 8 || lda % 8) return MD_BAD_ARG;
    const int64_t rpb = swiglu_rows_per_block(M, f);
    const dim3 grid((unsigned)((f / 8 + 63) / 64), (unsigned)((M + rpb - 1) / rpb));
    if (grid.y > 65535) return MD_BAD_ARG;
    hipLaunchKernelGGL(swiglu_fwd_kernel, grid, dim3(256), 0, st, (const bf16*)h12, ldh, (bf16*)a, lda, M, f, rpb);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_swiglu_bwd(const void* da, int64_t ldda, const void* h12, int64_t ldh, void* dh12, int64_t lddh, int64_t M,
                             int64_t f, hipStream_t st) {
    if (!da || !h12 || !dh12 || M <= 0 || f <= 0 || f % 8 || ldh % 8 || ldda % 8 || lddh % 8) return MD_BAD_ARG;
    const int64_t rpb = swiglu_rows_per_block(M, f);
    const dim3 grid((unsigned)((f / 8 + 63) / 64), (unsigned)((M + rpb - 1) / rpb));
    if (grid.y > 65535) return MD_BAD_ARG;
    hipLaunchKernelGGL(swiglu_bwd_kernel, grid, dim3(256), 0, st, (const bf16*)da, ldda, (const bf16*)h12, ldh, (bf16*)dh12, lddh, M, f,
                       rpb);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_gate_bwd(const void* dx, const void* br, const void* gate, int64_t ldgate, void* dbr, float* dgate,
                           int64_t lddg, int64_t rows, int64_t C, int64_t rows_per_sample, int64_t rows_per_block,
                           hipStream_t st) {
    if (!dx || !br || !gate || !dbr || !dgate || rows <= 0 || C <= 0 || C % 8 || C > 2048 || rows_per_sample <= 0 ||
        rows % rows_per_sample || rows_per_block <= 0 || ldgate % 8)
        return MD_BAD_ARG;
    dim3 grid((unsigned)((rows_per_sample + rows_per_block - 1) / rows_per_block), (unsigned)(rows / rows_per_sample));
#define GB(N) hipLaunchKernelGGL(gate_bwd_kernel<N>, grid, dim3(256), 0, st, (const bf16*)dx, (const bf16*)br, \
                                 (const bf16*)gate, ldgate, (bf16*)dbr, dgate, lddg, C, rows_per_sample, rows_per_block)
    if (C <= 512) GB(1); else if (C <= 1024) GB(2); else GB(4);
#undef GB
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_act_fwd(const void* x, void* y, int64_t n, int32_t act, hipStream_t st) {
    if (!x || !y || n <= 0 || n % 8) return MD_BAD_ARG;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, st, (const bf16*)x, (bf16*)y, n / 8, act);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_act_bwd(const float* dy, const void* x, void* dx, int64_t n, int32_t act, hipStream_t st) {
    if (!dy || !x || !dx || n <= 0 || n % 8) return MD_BAD_ARG;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, st, dy, (const bf16*)x, (bf16*)dx, n / 8, act);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_colsum(const void* x, int32_t x_is_f32, int64_t ld, float* out, int64_t rows, int64_t C, hipStream_t st) {
    if (!x || !out || rows <= 0 || C <= 0) return MD_BAD_ARG;
    int64_t rpb = 128;                       // rows per workgroup; short inputs (adaLN bias grads: 256 rows) get more,
    while (rpb > 8 && ((C + 255) / 256) * ((rows + rpb - 1) / rpb) < 512) rpb /= 2;   // smaller chunks to fill the chip
    dim3 grid((unsigned)((C + 255) / 256), (unsigned)((rows + rpb - 1) / rpb));
    if (x_is_f32)
        hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, st, (const float*)x, ld, out, rows, C, rpb);
    else
        hipLaunchKernelGGL(colsum_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)x, ld, out, rows, C, rpb);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_cast_f32_bf16(const float* x, void* y, int64_t n, const float* scale_ptr, hipStream_t st) {
    if (!x || !y || n <= 0 || n % 8) return MD_BAD_ARG;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, st, x, (bf16*)y, n / 8, scale_ptr);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_cast_f32_bf16_clear(float* x, void* y, int64_t n, hipStream_t st) {
    if (!x || !y || n <= 0 || n % 8) return MD_BAD_ARG;
    hipLaunchKernelGGL(cast_f32_bf16_clear_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, st, x, (bf16*)y, n / 8);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_cast_rows_bf16(const void* x, int32_t x_dtype /*0 = f16, 1 = f32*/, void* y, int64_t rows, int64_t C,
                                 const float* rowscale, int64_t rows_per_sample, hipStream_t st) {
    if (!x || !y || rows <= 0 || C <= 0 || C % 8 || (rowscale && rows_per_sample <= 0)) return MD_BAD_ARG;
    if (x_dtype == 0)
        hipLaunchKernelGGL(cast_rows_kernel<_Float16>, dim3(ew_grid(rows * C / 8)), dim3(256), 0, st, (const _Float16*)x,
                           (bf16*)y, rows, C, rowscale, rows_per_sample);
    else
        hipLaunchKernelGGL(cast_rows_kernel<float>, dim3(ew_grid(rows * C / 8)), dim3(256), 0, st, (const float*)x,
                           (bf16*)y, rows, C, rowscale, rows_per_sample);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_mean_tokens(const void* y, void* out, int64_t B, int64_t L, int64_t C, hipStream_t st) {
    if (!y || !out || B <= 0 || L <= 0 || C <= 0 || C % 8) return MD_BAD_ARG;
    hipLaunchKernelGGL(mean_tokens_kernel, dim3(ew_grid(B * C / 8)), dim3(256), 0, st, (const bf16*)y, (bf16*)out, B, L, C);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_mean_tokens_bwd(const void* dpool, float* dy, int64_t B, int64_t L, int64_t C, hipStream_t st) {
    if (!dpool || !dy || B <= 0 || L <= 0 || C <= 0 || C % 8) return MD_BAD_ARG;
    hipLaunchKernelGGL(mean_tokens_bwd_kernel, dim3(ew_grid(B * L * C / 8)), dim3(256), 0, st, (const bf16*)dpool, dy, B, L,
                       C);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_add_bf16(const void* a, const void* b, void* y, int64_t n, hipStream_t st) {
    if (!a || !b || !y || n <= 0 || n % 8) return MD_BAD_ARG;
    hipLaunchKernelGGL(add_bf16_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, st, (const bf16*)a, (const bf16*)b, (bf16*)y,
                       n / 8);
    MD_LAUNCH_CHECK();
    return 0;
}

// Zero-fill of an activation / accumulator region (the engine's only "initialise" operation: fp32 gradient accumulators, the
// scatter target of the un-masking backward).  hipMemsetAsync on the caller's stream: capturable, no at::native fill kernels.
extern "C" int md_fill_zero(void* p, int64_t bytes, hipStream_t stream) {
    if (!p || bytes < 0) return MD_BAD_ARG;
    if (bytes == 0) return 0;
    return (int)hipMemsetAsync(p, 0, (size_t)bytes, stream);
}
