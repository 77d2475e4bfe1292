// Per-step O(params) tail of the training step, fused into two HBM passes over flat fp32 buffers:
//   md_sumsq          sum of squared gradients (global L2 norm for clipping)
//   md_adamw_step     clip (coef = min(1, max_norm / (||g|| + 1e-6)), read from the device-side sum: no host sync)
//                     + decoupled-weight-decay Adam + bf16 shadow-weight emit + optional gradient zeroing
// Replaces clip_grad_norm_ (train.py:85-86) and torch.optim.AdamW (train.py:39-43; configs/*.yaml optimizer).
#include "md_common.h"
#include "../../include/microdit_hip.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_kernel(const float* g, int64_t n4, float* out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = *reinterpret_cast<const float4*>(g + i * 4);
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

__global__ __launch_bounds__(256) void adamw_kernel(md_adamw_args a) {
    float coef = a.grad_scale;
    if (a.sumsq && a.max_norm > 0.f) {
        const float nrm = sqrtf(*reinterpret_cast<const float*>(a.sumsq)) * a.grad_scale;
        coef *= fminf(1.f, a.max_norm / (nrm + 1e-6f));
    }
    float* P = reinterpret_cast<float*>(a.p);
    float* G = reinterpret_cast<float*>(a.g);
    float* Mo = reinterpret_cast<float*>(a.m);
    float* Vo = reinterpret_cast<float*>(a.v);
    bf16* S = reinterpret_cast<bf16*>(a.shadow);
    const float decay = 1.f - a.lr * a.weight_decay;
    const float step_size = a.lr / a.bias_corr1;
    const float inv_sqrt_bc2 = 1.f / sqrtf(a.bias_corr2);
    const int64_t n4 = a.n / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 p = *reinterpret_cast<float4*>(P + i * 4);
        float4 g = *reinterpret_cast<float4*>(G + i * 4);
        float4 m = *reinterpret_cast<float4*>(Mo + i * 4);
        float4 v = *reinterpret_cast<float4*>(Vo + i * 4);
        float* pp = &p.x;
        float* gp = &g.x;
        float* mp = &m.x;
        float* vp = &v.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gg = gp[e] * coef;
            pp[e] *= decay;
            mp[e] = a.beta1 * mp[e] + (1.f - a.beta1) * gg;
            vp[e] = a.beta2 * vp[e] + (1.f - a.beta2) * gg * gg;
            pp[e] -= step_size * mp[e] / (sqrtf(vp[e]) * inv_sqrt_bc2 + a.eps);
        }
        *reinterpret_cast<float4*>(P + i * 4) = p;
        *reinterpret_cast<float4*>(Mo + i * 4) = m;
        *reinterpret_cast<float4*>(Vo + i * 4) = v;
        if (a.zero_grad) *reinterpret_cast<float4*>(G + i * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        if (S) {
            bf16x4 o;
            o[0] = f2bf(p.x); o[1] = f2bf(p.y); o[2] = f2bf(p.z); o[3] = f2bf(p.w);
            st_bf16x4(S + i * 4, o);
        }
    }
}

}  // namespace

extern "C" int md_sumsq(const float* g, int64_t n, float* out, hipStream_t st) {
    if (!g || !out || n <= 0 || n % 4) return MD_BAD_ARG;
    int64_t grid = (n / 4 + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)grid), dim3(256), 0, st, g, n / 4, out);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_adamw_step(const md_adamw_args* a, hipStream_t st) {
    if (!a || !a->p || !a->g || !a->m || !a->v || a->n <= 0 || a->n % 4) return MD_BAD_ARG;
    int64_t grid = (a->n / 4 + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)grid), dim3(256), 0, st, *a);
    MD_LAUNCH_CHECK();
    return 0;
}
