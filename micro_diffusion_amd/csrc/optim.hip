// Per-step O(params) tail of the training step on the flat parameter buffers:
//   md_sumsq + md_sumsq_finish   sum of squared gradients (global L2 norm for clipping), DETERMINISTIC: a fixed grid writes one
//                                partial per workgroup (no atomics), a single workgroup adds the partials in a fixed order, so
//                                every data-parallel rank derives bit-identical clip coefficients from identical reduced gradients
//   md_adamw_step                clip (coef = min(1, max_norm / (||g|| + 1e-6)), read from the device-side sum: no host sync)
//                                + decoupled-weight-decay Adam + bf16 shadow-weight emit + gradient zeroing + optional EMA of the
//                                weights, one pass: 34 B / parameter (p, m, v read + write, g read + zero, shadow write; +8 with EMA)
// Gradients may be supplied as bf16 (the data-parallel exchange buffer) instead of the fp32 accumulators.
// Replaces clip_grad_norm_ (train.py:85-86), torch.optim.AdamW (train.py:39-43; configs/*.yaml optimizer) and the EMA algorithm
// named by configs/res_512_*.yaml:4-9 (diffusion.algorithms.ema.EMA: ema = s * ema + (1 - s) * p every batch after ema_start).
#include "md_common.h"
#include "../../include/microdit_hip.h"

namespace {

constexpr int SUMSQ_BLOCKS = MD_SUMSQ_PARTIALS;

// 16-byte non-temporal accesses (the builtins take clang vector types, not HIP's float4 struct)
__device__ __forceinline__ float4 nt_load4(const float* p) {
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void nt_store4(float* p, const float4& v) {
    f32x4 t;
    t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p));
}

template <bool BF16>
__global__ __launch_bounds__(256) void sumsq_kernel(const void* gv, int64_t n8, float* partial) {
    __shared__ float red[4];
    float s = 0.f;
    // 8 elements per thread and iteration: two float4 (fp32) or one 16-byte load (bf16)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        if (BF16) {
            const bf16x8 v = ld_bf16x8(reinterpret_cast<const bf16*>(gv) + i * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += bf2f(v[e]) * bf2f(v[e]);
        } else {
            const float* g = reinterpret_cast<const float*>(gv) + i * 8;
            const float4 a = nt_load4(g);
            const float4 b = nt_load4(g + 4);
            s += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w + b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = sum of `count` partials, always in the same order: thread t adds partials t, t + 256, ... and the 256 thread sums
// are combined by a fixed butterfly.
__global__ __launch_bounds__(256) void sumsq_finish_kernel(const float* partial, int64_t count, float* out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < count; i += 256) s += partial[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Range table of the sharded optimiser step (md_adamw_step_ranges): the rank's chunks of the flat buffers, packed back to back in
// the index space the kernel walks.  Packed element i of range j lives at flat element flat[j] + (i - start[j]).
constexpr int ADAMW_MAX_RANGES = MD_ADAMW_MAX_RANGES;
struct AdamWRanges {
    int n;
    int64_t start[ADAMW_MAX_RANGES + 1];     // start[n] = total packed elements
    int64_t flat[ADAMW_MAX_RANGES];
};

template <bool GBF16, int EMA, bool RANGES>
__global__ __launch_bounds__(256) void adamw_kernel(md_adamw_args a, AdamWRanges rg) {
    float coef = a.grad_scale;
    if (a.sumsq && a.max_norm > 0.f) {
        const float nrm = sqrtf(*reinterpret_cast<const float*>(a.sumsq)) * a.grad_scale;
        coef *= fminf(1.f, a.max_norm / (nrm + 1e-6f));
    }
    float* P = reinterpret_cast<float*>(a.p);
    float* G = reinterpret_cast<float*>(a.g);
    float* Mo = reinterpret_cast<float*>(a.m);
    float* Vo = reinterpret_cast<float*>(a.v);
    float* Em = reinterpret_cast<float*>(a.ema);
    bf16* S = reinterpret_cast<bf16*>(a.shadow);
    const bf16* Gb = reinterpret_cast<const bf16*>(a.g_bf16);
    const float decay = 1.f - a.lr * a.weight_decay;
    const float step_size = a.lr / a.bias_corr1;
    const float inv_sqrt_bc2 = 1.f / sqrtf(a.bias_corr2);
    const float b1 = a.beta1, b2 = a.beta2, eps = a.eps, es = a.ema_smoothing;
    const int64_t n4 = a.n / 4;
    // Every stream is touched exactly once: non-temporal loads / stores keep 40 GB of one-shot traffic from rotating through
    // the L2s and the Infinity Cache.
    for (int64_t ip = (int64_t)blockIdx.x * 256 + threadIdx.x; ip < n4; ip += (int64_t)gridDim.x * 256) {
        // RANGES: ip indexes the PACKED space (where the bf16 gradient and the bf16 weight output live); i is the same float4 in
        // the flat buffers (masters, moments, EMA, fp32 gradient).  Chunks are multiples of 64 elements: a float4 never straddles.
        int64_t i = ip;
        if (RANGES) {
            int lo = 0, hi = rg.n - 1;
            const int64_t e = ip * 4;
            while (lo < hi) {                      // last range whose start <= e
                const int mid = (lo + hi + 1) >> 1;
                if (rg.start[mid] <= e) lo = mid; else hi = mid - 1;
            }
            i = (rg.flat[lo] + (e - rg.start[lo])) >> 2;
        }
        float4 p = nt_load4(P + i * 4);
        float4 m = nt_load4(Mo + i * 4);
        float4 v = nt_load4(Vo + i * 4);
        float4 g;
        if (GBF16) {
            const bf16x4 gb = ld_bf16x4(Gb + ip * 4);
            g = make_float4(bf2f(gb[0]), bf2f(gb[1]), bf2f(gb[2]), bf2f(gb[3]));
        } else {
            g = nt_load4(G + i * 4);
        }
        float4 em;
        if (EMA == 2) em = nt_load4(Em + i * 4);
        float* pp = &p.x;
        float* gp = &g.x;
        float* mp = &m.x;
        float* vp = &v.x;
        float* ep = &em.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gg = gp[e] * coef;
            pp[e] *= decay;
            mp[e] = b1 * mp[e] + (1.f - b1) * gg;
            vp[e] = b2 * vp[e] + (1.f - b2) * gg * gg;
            pp[e] -= step_size * mp[e] / (sqrtf(vp[e]) * inv_sqrt_bc2 + eps);
            if (EMA == 2) ep[e] = es * ep[e] + (1.f - es) * pp[e];
        }
        nt_store4(P + i * 4, p);
        nt_store4(Mo + i * 4, m);
        nt_store4(Vo + i * 4, v);
        if (EMA == 1) nt_store4(Em + i * 4, p);        // ema_start: ema = weights
        if (EMA == 2) nt_store4(Em + i * 4, em);
        if (a.zero_grad) nt_store4(G + i * 4, make_float4(0.f, 0.f, 0.f, 0.f));
        if (S) {
            bf16x4 o;
            o[0] = f2bf(p.x); o[1] = f2bf(p.y); o[2] = f2bf(p.z); o[3] = f2bf(p.w);
            st_bf16x4(S + ip * 4, o);         // re-read by every GEMM of the next step: left cacheable
        }
    }
}

}  // namespace

extern "C" int md_sumsq(const void* g, int32_t g_is_bf16, int64_t n, float* partials, hipStream_t st) {
    if (!g || !partials || n <= 0 || n % 8) return MD_BAD_ARG;
    if (g_is_bf16) hipLaunchKernelGGL((sumsq_kernel<true>), dim3(SUMSQ_BLOCKS), dim3(256), 0, st, g, n / 8, partials);
    else hipLaunchKernelGGL((sumsq_kernel<false>), dim3(SUMSQ_BLOCKS), dim3(256), 0, st, g, n / 8, partials);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_sumsq_finish(const float* partials, int64_t count, float* out, hipStream_t st) {
    if (!partials || !out || count <= 0) return MD_BAD_ARG;
    hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(256), 0, st, partials, count, out);
    MD_LAUNCH_CHECK();
    return 0;
}

// Exact checksum of n 16-bit words (the bf16 shadow weights): out[0] += sum of the words, out[1] += sum of word * odd 32-bit
// multiplier of its index -- integer sums are exact and order-independent (bit-identical on identical data whatever the
// schedule), one flipped bf16 ulp, a sign flip or two swapped elements all change them.  n % 8 == 0, out zeroed by the caller.
__global__ __launch_bounds__(256) void checksum_u16_kernel(const uint4* x, int64_t n8, unsigned long long* out) {
    unsigned long long s0 = 0, s1 = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const uint4 v = x[i];
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned long long lo = w[e] & 0xffffu, hi = w[e] >> 16;
            const unsigned long long i0 = (unsigned long long)i * 8 + 2 * e, i1 = i0 + 1;
            s0 += lo + hi;
            s1 += lo * (((i0 * 0x9E3779B97F4A7C15ull) >> 32) | 1ull) + hi * (((i1 * 0x9E3779B97F4A7C15ull) >> 32) | 1ull);
        }
    }
    // wave-level integer reduction, then one atomic pair per wave
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s0 += __shfl_xor(s0, o, 64);
        s1 += __shfl_xor(s1, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(out, s0);
        atomicAdd(out + 1, s1);
    }
}

extern "C" int md_checksum_u16(const void* x, int64_t n, uint64_t* out2, hipStream_t st) {
    if (!x || !out2 || n <= 0 || n % 8 || ((uintptr_t)x & 15)) return MD_BAD_ARG;
    int64_t grid = (n / 8 + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(checksum_u16_kernel, dim3((unsigned)grid), dim3(256), 0, st, static_cast<const uint4*>(x), n / 8,
                       reinterpret_cast<unsigned long long*>(out2));
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_adamw_step(const md_adamw_args* a, hipStream_t st) {
    if (!a || !a->p || !a->g || !a->m || !a->v || a->n <= 0 || a->n % 4) return MD_BAD_ARG;
    if (a->ema_mode < 0 || a->ema_mode > 2 || (a->ema_mode && !a->ema)) return MD_BAD_ARG;
    // one 256-thread workgroup per 4 KiB of every stream and iteration; 8 workgroups resident per CU x 256 CUs x 4 rounds
    int64_t grid = (a->n / 4 + 255) / 256;
    if (grid > 8192) grid = 8192;
    const dim3 gd((unsigned)grid), bd(256);
    AdamWRanges none;
    none.n = 0;
#define ADAMW(GB, E) hipLaunchKernelGGL((adamw_kernel<GB, E, false>), gd, bd, 0, st, *a, none)
    if (a->g_bf16) {
        if (a->ema_mode == 0) ADAMW(true, 0); else if (a->ema_mode == 1) ADAMW(true, 1); else ADAMW(true, 2);
    } else {
        if (a->ema_mode == 0) ADAMW(false, 0); else if (a->ema_mode == 1) ADAMW(false, 1); else ADAMW(false, 2);
    }
#undef ADAMW
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_adamw_step_ranges(const md_adamw_args* a, const int64_t* flat_off, const int64_t* count, int32_t n_ranges,
                                    hipStream_t st) {
    if (!a || !a->p || !a->g || !a->m || !a->v || !a->g_bf16 || !flat_off || !count || n_ranges < 1 || n_ranges > ADAMW_MAX_RANGES)
        return MD_BAD_ARG;
    if (a->ema_mode < 0 || a->ema_mode > 2 || (a->ema_mode && !a->ema) || a->zero_grad) return MD_BAD_ARG;
    AdamWRanges rg;
    rg.n = n_ranges;
    int64_t tot = 0;
    for (int j = 0; j < n_ranges; ++j) {
        if (count[j] <= 0 || count[j] % 4 || flat_off[j] < 0 || flat_off[j] % 4) return MD_BAD_ARG;
        rg.start[j] = tot;
        rg.flat[j] = flat_off[j];
        tot += count[j];
    }
    rg.start[n_ranges] = tot;
    md_adamw_args b = *a;
    b.n = tot;                                   // the kernel walks the packed space
    int64_t grid = (tot / 4 + 255) / 256;
    if (grid > 8192) grid = 8192;
    const dim3 gd((unsigned)grid), bd(256);
#define ADAMWR(E) hipLaunchKernelGGL((adamw_kernel<true, E, true>), gd, bd, 0, st, b, rg)
    if (a->ema_mode == 0) ADAMWR(0); else if (a->ema_mode == 1) ADAMWR(1); else ADAMWR(2);
#undef ADAMWR
    MD_LAUNCH_CHECK();
    return 0;
}
