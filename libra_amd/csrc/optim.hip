// Fused AdamW step on a flat parameter range (data-parallel optimizer of SURVEY §8e / BASELINE configs[3],[4]) — HBM-bound:
// one pass, per element 14 B read (bf16 grad + fp32 master, m, v) and 14 B written (master, m, v, bf16 parameter).
// Replaces torch.optim.AdamW / DeepSpeed's fused Adam under the reference's recipes (libra_pretrain.yaml:83-91,
// deepspeed_configs/ZeRO-2.json): decoupled weight decay, bias-corrected moments, fp32 master weights, the bf16 working
// copy re-rounded from the master every step.
#include "hip_common.hpp"
#include "../../include/libra_hip.h"

namespace libra {

struct AdamArgs {
    float lr, beta1, beta2, eps, wd, inv_bc1, inv_sqrt_bc2, gscale;
    const float* gnorm_sq;     // device scalar: squared global gradient norm (null = no clipping)
    float max_norm;
};

__device__ __forceinline__ void adam1(float g, float& p, float& m, float& v, const AdamArgs& a) {
    g *= a.gscale;
    p -= a.lr * a.wd * p;                                   // decoupled decay (torch.optim.AdamW order: decay first)
    m = a.beta1 * m + (1.f - a.beta1) * g;
    v = a.beta2 * v + (1.f - a.beta2) * g * g;
    const float denom = sqrtf(v) * a.inv_sqrt_bc2 + a.eps;
    p -= a.lr * a.inv_bc1 * (m / denom);
}

// a thread owns 8 consecutive elements: one 16-byte bf16 load + six 16-byte fp32 loads in flight before the first use
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ master, float* __restrict__ m, float* __restrict__ v,
                                                    const bf16_t* __restrict__ grad, bf16_t* __restrict__ param, long n,
                                                    AdamArgs a) {
    const long i8 = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i8 >= n) return;
    if (a.gnorm_sq) {
        // torch.nn.utils.clip_grad_norm_ (HF Trainer `max_grad_norm`, DeepSpeed `gradient_clipping`): coef = max / (norm + 1e-6),
        // applied only when < 1.  The norm is a device scalar: no host read between the backward and the update.  *gnorm_sq is the
        // squared norm of the UNSCALED bucket contents; the gradients actually applied are grad_scale * grad, so their norm is
        // |grad_scale| * sqrt(*gnorm_sq) - what clip_grad_norm_ would see after the unscaling
        const float coef = a.max_norm / (fabsf(a.gscale) * sqrtf(*a.gnorm_sq) + 1e-6f);
        a.gscale *= coef < 1.f ? coef : 1.f;
    }
    if (i8 + 8 <= n) {
        const u32x4 gq = *(const u32x4*)(grad + i8);
        f32x4 p0 = *(const f32x4*)(master + i8), p1 = *(const f32x4*)(master + i8 + 4);
        f32x4 m0 = *(const f32x4*)(m + i8), m1 = *(const f32x4*)(m + i8 + 4);
        f32x4 v0 = *(const f32x4*)(v + i8), v1 = *(const f32x4*)(v + i8 + 4);
        float g[8], o[8];
        unpack8(gq, g);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float pp = p0[e], mm = m0[e], vv = v0[e];
            adam1(g[e], pp, mm, vv, a);
            p0[e] = pp; m0[e] = mm; v0[e] = vv; o[e] = pp;
            pp = p1[e]; mm = m1[e]; vv = v1[e];
            adam1(g[e + 4], pp, mm, vv, a);
            p1[e] = pp; m1[e] = mm; v1[e] = vv; o[e + 4] = pp;
        }
        *(f32x4*)(master + i8) = p0; *(f32x4*)(master + i8 + 4) = p1;
        *(f32x4*)(m + i8) = m0; *(f32x4*)(m + i8 + 4) = m1;
        *(f32x4*)(v + i8) = v0; *(f32x4*)(v + i8 + 4) = v1;
        *(u32x4*)(param + i8) = pack8(o);
    } else {
        for (long i = i8; i < n; ++i) {
            float pp = master[i], mm = m[i], vv = v[i];
            adam1(bf2f(grad[i]), pp, mm, vv, a);
            master[i] = pp; m[i] = mm; v[i] = vv; param[i] = f2bf(pp);
        }
    }
}

// Sum of squares of a bf16 range, deterministic: per-block partials (fixed partition, fixed in-block order), then ONE block
// folds the partials in index order into out[0] (+= when accumulate).  HBM-bound: 2 B / element.
constexpr int SUMSQ_ELEMS_PER_BLOCK = 256 * 8 * 8;
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const bf16_t* __restrict__ x, long n, float* __restrict__ part) {
    __shared__ float red[4];
    const long base = (long)blockIdx.x * SUMSQ_ELEMS_PER_BLOCK;
    float acc = 0.f;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const long i8 = base + ((long)it * 256 + threadIdx.x) * 8;
        if (i8 + 8 <= n) {
            float g[8];
            unpack8(*(const u32x4*)(x + i8), g);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += g[e] * g[e];
        } else {
            for (long i = i8; i < n; ++i) { const float g = bf2f(x[i]); acc += g * g; }
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ part, long nblk, float* __restrict__ out,
                                                          int accumulate) {
    __shared__ float red[4];
    float acc = 0.f;
    for (long i = threadIdx.x; i < nblk; i += 256) acc += part[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (red[0] + red[1]) + (red[2] + red[3]);
        out[0] = accumulate ? out[0] + t : t;
    }
}

}  // namespace libra

using namespace libra;

extern "C" size_t libra_sumsq_workspace_bytes(int64_t n) {
    return n <= 0 ? 0 : (size_t)((n + SUMSQ_ELEMS_PER_BLOCK - 1) / SUMSQ_ELEMS_PER_BLOCK) * sizeof(float);
}

extern "C" int libra_sumsq_bf16(const void* x, int64_t n, float* out, int accumulate, float* workspace, size_t workspace_bytes,
                                void* stream) {
    if (!out) return LIBRA_ERR_ALIGN;
    if (n < 0) return LIBRA_ERR_SHAPE;
    const long nblk = n > 0 ? (n + SUMSQ_ELEMS_PER_BLOCK - 1) / SUMSQ_ELEMS_PER_BLOCK : 0;
    if (n > 0 && (!x || ((uintptr_t)x & 15) || !workspace || workspace_bytes < (size_t)nblk * sizeof(float))) return LIBRA_ERR_ALIGN;
    if (nblk > 0x7fffffffL) return LIBRA_ERR_SHAPE;
    if (nblk)
        hipLaunchKernelGGL(sumsq_partial_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (long)n,
                           workspace);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, workspace, nblk, out, accumulate);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}

extern "C" int libra_adamw_step(float* master, float* m, float* v, const void* grad, void* param, int64_t n, float lr,
                                float beta1, float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2,
                                float grad_scale, const float* grad_norm_sq, float max_grad_norm, void* stream) {
    if (n <= 0) return LIBRA_OK;
    if (grad_norm_sq && !(max_grad_norm > 0.f)) return LIBRA_ERR_SHAPE;
    if (!master || !m || !v || !grad || !param) return LIBRA_ERR_ALIGN;
    if ((((uintptr_t)master | (uintptr_t)m | (uintptr_t)v | (uintptr_t)grad | (uintptr_t)param) & 15)) return LIBRA_ERR_ALIGN;
    if (!(bias_corr1 > 0.f) || !(bias_corr2 > 0.f)) return LIBRA_ERR_SHAPE;
    AdamArgs a{lr, beta1, beta2, eps, weight_decay, 1.f / bias_corr1, 1.f / sqrtf(bias_corr2), grad_scale, grad_norm_sq, max_grad_norm};
    const long blocks = (n + 2047) / 2048;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, master, m, v,
                       (const bf16_t*)grad, (bf16_t*)param, (long)n, a);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}
