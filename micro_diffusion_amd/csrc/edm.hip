// Front / back end of the MicroDiT training step (all fp32 math, HBM-bound, tiny next to the transformer):
//   md_edm_prepare      sigma sampling + noise add + c_in/c_noise (model.py:154-164,182-188)
//   md_patchify         [B,C,H,W] f32 (* per-sample scale) -> bf16 patch rows ordered (c, ph, pw) = the im2col of
//                       timm PatchEmbed's stride-p conv (dit.py:312-314,479); the conv itself is a GEMM.
//   md_timestep_embed   sinusoidal features [cos | sin] (utils.py:266-281)
//   md_unpatchify       (+ unmask_tokens) token rows ordered (ph, pw, c) -> [B,C,H,W] f32 (utils.py:417-426,
//                       dit.py:566-575)
//   md_edm_loss         D = c_skip*xn + c_out*F, weighted SE, mean over the UN-MASKED patches per sample, batch
//                       mean (model.py:177,199-210) and its gradient w.r.t. the kept-token network output.
#include "md_common.h"
#include "../../include/microdit_hip.h"

namespace {

template <typename XT>
__global__ __launch_bounds__(256) void edm_prepare_kernel(const XT* x0, const float* eps, const float* rnd, float* xn,
                                                          float* x0_f32, float* sigma, float* cin, float* cnoise, int64_t B,
                                                          int64_t per, float p_mean, float p_std, float sd) {
    const int64_t total = B * per;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / per;
        const float s = expf(rnd[b] * p_std + p_mean);
        const float x = (float)x0[i];
        xn[i] = x + eps[i] * s;
        if (x0_f32) x0_f32[i] = x;      // the loss reads the clean latents as fp32 (model.py:104-110 casts the fp16 batch once)
        if (i % per == 0) {
            sigma[b] = s;
            cin[b] = 1.f / sqrtf(sd * sd + s * s);
            cnoise[b] = logf(s) * 0.25f;
        }
    }
}

// out[(b*T + ti*gw + tj), c*p*p + ph*p + pw] = x[b, c, ti*p + ph, tj*p + pw] * scale[b]
__global__ __launch_bounds__(256) void patchify_kernel(const float* x, const float* scale, bf16* out, int64_t B, int C,
                                                       int H, int W, int p) {
    const int gh = H / p, gw = W / p, pv = C * p * p;
    const int64_t total = B * gh * gw * pv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int e = (int)(i % pv);
        const int64_t tok = i / pv;
        const int tj = (int)(tok % gw), ti = (int)((tok / gw) % gh);
        const int64_t b = tok / ((int64_t)gw * gh);
        const int c = e / (p * p), ph = (e / p) % p, pw = e % p;
        const float v = x[((b * C + c) * H + ti * p + ph) * W + tj * p + pw];
        out[i] = f2bf(scale ? v * scale[b] : v);
    }
}

__global__ __launch_bounds__(256) void timestep_embed_kernel(const float* t, bf16* out, int64_t B, int dim) {
    const int half = dim / 2;
    const int64_t total = B * half;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / half;
        const int k = (int)(i % half);
        const float f = expf(-9.210340371976184f * (float)k / (float)half);  // ln(10000)
        const float a = t[b] * f;
        out[b * dim + k] = f2bf(cosf(a));
        out[b * dim + half + k] = f2bf(sinf(a));
    }
}

// image[b, c, ti*p+ph, tj*p+pw] = tok[row(b, ti*gw+tj), (ph*p + pw)*C + c];  masked tokens -> mask_token
__global__ __launch_bounds__(256) void unpatchify_kernel(const bf16* tok, const int32_t* ids_restore, int64_t Tk,
                                                         const float* mask_token, float* img, int64_t B, int C, int H,
                                                         int W, int p) {
    const int gh = H / p, gw = W / p, pv = C * p * p;
    const int64_t T = (int64_t)gh * gw, total = B * C * H * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int w = (int)(i % W), hrow = (int)((i / W) % H), c = (int)((i / ((int64_t)W * H)) % C);
        const int64_t b = i / ((int64_t)W * H * C);
        const int ti = hrow / p, ph = hrow % p, tj = w / p, pw = w % p;
        const int64_t t = (int64_t)ti * gw + tj;
        const int e = (ph * p + pw) * C + c;
        float v;
        if (ids_restore) {
            const int rank = ids_restore[b * T + t];
            v = rank < Tk ? bf2f(tok[(b * Tk + rank) * pv + e]) : (mask_token ? mask_token[e] : 0.f);
        } else {
            v = bf2f(tok[(b * T + t) * pv + e]);
        }
        img[i] = v;
    }
}

// One workgroup per sample.  Kept token j of sample b sits at grid position keep_rows[b*Tk + j] - b*T (or j when
// keep_rows is NULL = no masking).  loss_b = mean over kept patches of mean_{c,ph,pw} w * (D - x0)^2;
// dtok = gscale * d(mean_b loss_b)/dF for the kept tokens ([B*Tk, C*p*p], (ph, pw, c) order; f32, or bf16 for the
// training step, whose backward starts from bf16 anyway).  The batch mean is taken by edm_loss_finish_kernel in sample
// order (the first version added the per-sample terms with float atomics: the loss depended on the arrival order).
template <typename DT>
__global__ __launch_bounds__(256) void edm_loss_kernel(const bf16* tok, const int32_t* keep_rows, const float* xn,
                                                       const float* x0, const float* sigma, float* loss_per_sample,
                                                       DT* dtok, float gscale, int64_t B, int64_t Tk, int C, int H,
                                                       int W, int p, float sd) {
    __shared__ float red[4];
    const int gh = H / p, gw = W / p, pv = C * p * p;
    const int64_t T = (int64_t)gh * gw;
    const int64_t b = blockIdx.x;
    const float s = sigma[b];
    const float wgt = (s * s + sd * sd) / ((s * sd) * (s * sd));
    const float cskip = sd * sd / (s * s + sd * sd);
    const float cout = s * sd / sqrtf(s * s + sd * sd);
    const float norm = 1.f / ((float)pv * (float)Tk);
    const float gfac = 2.f * wgt * cout * norm / (float)B * gscale;
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < Tk * pv; i += 256) {
        const int64_t j = i / pv;
        const int e = (int)(i % pv);
        const int64_t t = keep_rows ? (int64_t)keep_rows[b * Tk + j] - b * T : j;
        const int ti = (int)(t / gw), tj = (int)(t % gw);
        const int c = e % C, pw = (e / C) % p, ph = e / (C * p);
        const int64_t pix = ((b * C + c) * H + ti * p + ph) * W + tj * p + pw;
        const float F = bf2f(tok[(b * Tk + j) * pv + e]);
        const float D = cskip * xn[pix] + cout * F;
        const float diff = D - x0[pix];
        acc += wgt * diff * diff;
        if (dtok) {
            if constexpr (sizeof(DT) == 4) dtok[(b * Tk + j) * pv + e] = gfac * diff;
            else dtok[(b * Tk + j) * pv + e] = f2bf(gfac * diff);
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss_per_sample[b] = (red[0] + red[1] + red[2] + red[3]) * norm;
}

// loss_mean = (1 / B) sum_b loss_per_sample[b] in a fixed order (one wave: lane-strided partial sums, then the wave
// reduction); optionally loss_accum += accum_weight * loss_mean (the rank-mean loss of a microbatched step).
__global__ __launch_bounds__(64) void edm_loss_finish_kernel(const float* loss_per_sample, float* loss_mean, float* loss_accum,
                                                             float accum_weight, int64_t B) {
    float acc = 0.f;
    for (int64_t b = threadIdx.x; b < B; b += 64) acc += loss_per_sample[b];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) {
        const float m = acc / (float)B;
        *loss_mean = m;
        if (loss_accum) *loss_accum += accum_weight * m;
    }
}

inline int egrid(int64_t work) {
    int64_t g = (work + 255) / 256;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return (int)g;
}

// ---------------------------------------------------------------------------------------------------------------------
// Sampler (model.py:231-297): the arithmetic around each network evaluation of the Heun loop as two fused kernels, with the
// reference's precisions: fp64 sampler state, fp32 preconditioning (model.py:144-179) and fp32 classifier-free-guidance
// combine (dit.py:542-550).
//   sampler_input : net_in = float(c_in(sigma) * float(x)), written once, or twice when the batch is doubled for guidance
//   heun_update   : F = Fu + cfg * (Fc - Fu);  D = c_skip * float(x_in) + c_out * F;  d = (x_in - D) / sigma;
//                   first half-step : d_cur = d, x_next = x_hat + (t_next - t_hat) * d
//                   second half-step: x_next = x_hat + (t_next - t_hat) * (0.5 d_cur + 0.5 d)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sampler_input_kernel(const double* x, float* out, int64_t n, float sigma, float sigma_data, int dup) {
    const float c_in = 1.f / sqrtf(sigma_data * sigma_data + sigma * sigma);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = c_in * (float)x[i];
        out[i] = v;
        if (dup) out[n + i] = v;
    }
}

__global__ __launch_bounds__(256) void heun_update_kernel(const double* x_hat, const double* x_in, const float* F, double* d_cur,
                                                          double* x_next, int64_t n, float cfg, int has_uncond, double t_in, double t_hat,
                                                          double t_next, float sigma_data, int second) {
    const float sg = (float)t_in;
    const float den = sg * sg + sigma_data * sigma_data;
    const float c_skip = sigma_data * sigma_data / den;
    const float c_out = sg * sigma_data / sqrtf(den);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double xi = x_in[i];
        float f = F[i];
        if (has_uncond) {
            const float fu = F[n + i];
            f = fu + cfg * (f - fu);
        }
        const double D = (double)(c_skip * (float)xi + c_out * f);
        const double d = (xi - D) / t_in;
        const double xh = x_hat[i];
        if (!second) {
            d_cur[i] = d;
            x_next[i] = xh + (t_next - t_hat) * d;
        } else {
            x_next[i] = xh + (t_next - t_hat) * (0.5 * d_cur[i] + 0.5 * d);
        }
    }
}

}  // namespace

extern "C" int md_edm_prepare(const float* x0, const float* eps, const float* rnd, float* xn, float* sigma, float* cin,
                              float* cnoise, int64_t B, int64_t per_sample, float p_mean, float p_std, float sigma_data,
                              hipStream_t st) {
    if (!x0 || !eps || !rnd || !xn || !sigma || !cin || !cnoise || B <= 0 || per_sample <= 0) return MD_BAD_ARG;
    hipLaunchKernelGGL(edm_prepare_kernel<float>, dim3(egrid(B * per_sample)), dim3(256), 0, st, x0, eps, rnd, xn, (float*)nullptr,
                       sigma, cin, cnoise, B, per_sample, p_mean, p_std, sigma_data);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_edm_prepare_f16(const void* x0_f16, const float* eps, const float* rnd, float* xn, float* x0_f32, float* sigma,
                                  float* cin, float* cnoise, int64_t B, int64_t per_sample, float p_mean, float p_std,
                                  float sigma_data, hipStream_t st) {
    if (!x0_f16 || !eps || !rnd || !xn || !x0_f32 || !sigma || !cin || !cnoise || B <= 0 || per_sample <= 0) return MD_BAD_ARG;
    hipLaunchKernelGGL(edm_prepare_kernel<_Float16>, dim3(egrid(B * per_sample)), dim3(256), 0, st, (const _Float16*)x0_f16, eps,
                       rnd, xn, x0_f32, sigma, cin, cnoise, B, per_sample, p_mean, p_std, sigma_data);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_patchify(const float* x, const float* scale, void* out, int64_t B, int32_t C, int32_t H, int32_t W,
                           int32_t p, hipStream_t st) {
    if (!x || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || p <= 0 || H % p || W % p) return MD_BAD_ARG;
    hipLaunchKernelGGL(patchify_kernel, dim3(egrid(B * C * H * W)), dim3(256), 0, st, x, scale, (bf16*)out, B, C, H, W, p);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_timestep_embed(const float* t, void* out, int64_t B, int32_t dim, hipStream_t st) {
    if (!t || !out || B <= 0 || dim <= 0 || dim % 2) return MD_BAD_ARG;
    hipLaunchKernelGGL(timestep_embed_kernel, dim3(egrid(B * dim / 2)), dim3(256), 0, st, t, (bf16*)out, B, dim);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_unpatchify(const void* tok, const int32_t* ids_restore, int64_t Tk, const float* mask_token, float* img,
                             int64_t B, int32_t C, int32_t H, int32_t W, int32_t p, hipStream_t st) {
    if (!tok || !img || B <= 0 || C <= 0 || H % p || W % p || (ids_restore && Tk <= 0)) return MD_BAD_ARG;
    hipLaunchKernelGGL(unpatchify_kernel, dim3(egrid(B * C * H * W)), dim3(256), 0, st, (const bf16*)tok, ids_restore, Tk,
                       mask_token, img, B, C, H, W, p);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_edm_loss(const void* tok, const int32_t* keep_rows, const float* xn, const float* x0, const float* sigma,
                           float* loss_per_sample, float* loss_mean, float* dtok, int64_t B, int64_t Tk, int32_t C,
                           int32_t H, int32_t W, int32_t p, float sigma_data, hipStream_t st) {
    if (!tok || !xn || !x0 || !sigma || !loss_per_sample || !loss_mean || B <= 0 || Tk <= 0 || H % p || W % p)
        return MD_BAD_ARG;
    hipLaunchKernelGGL(edm_loss_kernel<float>, dim3((unsigned)B), dim3(256), 0, st, (const bf16*)tok, keep_rows, xn, x0, sigma,
                       loss_per_sample, dtok, 1.f, B, Tk, C, H, W, p, sigma_data);
    hipLaunchKernelGGL(edm_loss_finish_kernel, dim3(1), dim3(64), 0, st, loss_per_sample, loss_mean, (float*)nullptr, 0.f, B);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_edm_loss_train(const void* tok, const int32_t* keep_rows, const float* xn, const float* x0, const float* sigma,
                                 float* loss_per_sample, float* loss_mean, void* dtok_bf16, float grad_scale, float* loss_accum,
                                 float accum_weight, int64_t B, int64_t Tk, int32_t C, int32_t H, int32_t W, int32_t p,
                                 float sigma_data, hipStream_t st) {
    if (!tok || !xn || !x0 || !sigma || !loss_per_sample || !loss_mean || !dtok_bf16 || B <= 0 || Tk <= 0 || H % p || W % p)
        return MD_BAD_ARG;
    hipLaunchKernelGGL(edm_loss_kernel<bf16>, dim3((unsigned)B), dim3(256), 0, st, (const bf16*)tok, keep_rows, xn, x0, sigma,
                       loss_per_sample, (bf16*)dtok_bf16, grad_scale, B, Tk, C, H, W, p, sigma_data);
    hipLaunchKernelGGL(edm_loss_finish_kernel, dim3(1), dim3(64), 0, st, loss_per_sample, loss_mean, loss_accum, accum_weight, B);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_edm_sampler_input(const double* x, float* out, int64_t n, float sigma, float sigma_data, int32_t duplicate,
                                    hipStream_t st) {
    if (!x || !out || n <= 0) return MD_BAD_ARG;
    hipLaunchKernelGGL(sampler_input_kernel, dim3(egrid(n)), dim3(256), 0, st, x, out, n, sigma, sigma_data, duplicate);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_edm_heun_update(const double* x_hat, const double* x_in, const float* F, double* d_cur, double* x_next, int64_t n,
                                  float cfg, int32_t has_uncond, double t_in, double t_hat, double t_next, float sigma_data,
                                  int32_t second, hipStream_t st) {
    if (!x_hat || !x_in || !F || !d_cur || !x_next || n <= 0 || t_in <= 0) return MD_BAD_ARG;
    hipLaunchKernelGGL(heun_update_kernel, dim3(egrid(n)), dim3(256), 0, st, x_hat, x_in, F, d_cur, x_next, n, cfg, has_uncond, t_in,
                       t_hat, t_next, sigma_data, second);
    MD_LAUNCH_CHECK();
    return 0;
}
