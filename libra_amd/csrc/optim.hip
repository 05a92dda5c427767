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

}  // namespace libra

using namespace libra;

extern "C" int libra_adamw_step(float* master, float* m, float* v, const void* grad, void* param, int64_t n, float lr,
                                float beta1, float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2,
                                float grad_scale, void* stream) {
    if (n <= 0) return LIBRA_OK;
    if (!master || !m || !v || !grad || !param) return LIBRA_ERR_ALIGN;
    if ((((uintptr_t)master | (uintptr_t)m | (uintptr_t)v | (uintptr_t)grad | (uintptr_t)param) & 15)) return LIBRA_ERR_ALIGN;
    if (!(bias_corr1 > 0.f) || !(bias_corr2 > 0.f)) return LIBRA_ERR_SHAPE;
    AdamArgs a{lr, beta1, beta2, eps, weight_decay, 1.f / bias_corr1, 1.f / sqrtf(bias_corr2), grad_scale};
    const long blocks = (n + 2047) / 2048;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, master, m, v,
                       (const bf16_t*)grad, (bf16_t*)param, (long)n, a);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}
