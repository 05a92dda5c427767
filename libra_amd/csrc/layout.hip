// Layout / data-movement kernels (all HBM-bound): patch im2col / col2im,
// vision-tower feature select (+backward), fp32->bf16, bf16 add.
#include "hip_common.hpp"
#include "../../include/libra_hip.h"

namespace libra {

// ---- im2col for non-overlapping PxP patches. One thread per 8 output columns.
__global__ __launch_bounds__(256) void patch_im2col_kernel(const bf16_t* __restrict__ pix, bf16_t* __restrict__ cols,
                                                           int C, int H, int W, int P, int Kpad, long total8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total8) return;
    const int k8 = Kpad >> 3;
    const long row = i / k8;
    const int kc = (int)(i - row * k8) * 8;
    const int gw = W / P, gh = H / P;
    const int np = gw * gh;
    const long b = row / np;
    const int pidx = (int)(row - b * np);
    const int gy = pidx / gw, gx = pidx - gy * gw;
    const int K = C * P * P;
    bf16_t v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = kc + e;
        bf16_t val = 0;
        if (k < K) {
            const int c = k / (P * P);
            const int rem = k - c * P * P;
            const int ky = rem / P, kx = rem - ky * P;
            val = pix[((b * C + c) * H + gy * P + ky) * (long)W + gx * P + kx];
        }
        v[e] = val;
    }
    *(u32x4*)(cols + row * Kpad + kc) = *(u32x4*)v;
}

// inverse index map (each pixel belongs to exactly one patch): dpix[b,c,y,x] = dcols[row(b,y/P,x/P), k(c,y%P,x%P)]
__global__ __launch_bounds__(256) void patch_col2im_kernel(const bf16_t* __restrict__ dcols, bf16_t* __restrict__ dpix,
                                                           int C, int H, int W, int P, int Kpad, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % W);
    const long t = i / W;
    const int y = (int)(t % H);
    const long t2 = t / H;
    const int c = (int)(t2 % C);
    const long b = t2 / C;
    const int gw = W / P, gh = H / P;
    const int gy = y / P, gx = x / P;
    bf16_t v = 0;
    if (gy < gh && gx < gw) {
        const long row = (b * gh + gy) * gw + gx;
        const int k = c * P * P + (y - gy * P) * P + (x - gx * P);
        v = dcols[row * Kpad + k];
    }
    dpix[i] = v;
}

// ---- feature select: feat[b*(T-1)+p, j*D + d] = hs_j[b, 1+p, d]
struct SelPtrs { const bf16_t* p[4]; };
struct SelPtrsW { bf16_t* p[4]; int acc[4]; };

__global__ __launch_bounds__(256) void feature_select_kernel(SelPtrs hs, int n_sel, bf16_t* __restrict__ feat, int T,
                                                             int D, long total8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total8) return;
    const int d8 = D >> 3;
    const long per_row = (long)n_sel * d8;
    const long row = i / per_row;
    const int rem = (int)(i - row * per_row);
    const int j = rem / d8;
    const int dc = (rem - j * d8) * 8;
    const long b = row / (T - 1);
    const int p = (int)(row - b * (T - 1));
    const u32x4 v = *(const u32x4*)(hs.p[j] + ((b * T + 1 + p) * (long)D + dc));
    *(u32x4*)(feat + row * ((long)n_sel * D) + (long)j * D + dc) = v;
}

__global__ __launch_bounds__(256) void feature_select_bwd_kernel(const bf16_t* __restrict__ dfeat, SelPtrsW dhs,
                                                                 int n_sel, int T, int D, long total8) {
    // one thread per (b, t, j, 8 d) of the OUTPUT so CLS rows get their zeros too
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total8) return;
    const int d8 = D >> 3;
    const long per_row = (long)n_sel * d8;
    const long row = i / per_row;                  // row over B*T
    const int rem = (int)(i - row * per_row);
    const int j = rem / d8;
    const int dc = (rem - j * d8) * 8;
    const long b = row / T;
    const int t = (int)(row - b * T);
    bf16_t* dst = dhs.p[j] + row * (long)D + dc;
    float g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = 0.f;
    if (t > 0) unpack8(*(const u32x4*)(dfeat + (b * (T - 1) + t - 1) * ((long)n_sel * D) + (long)j * D + dc), g);
    if (dhs.acc[j]) {
        float o[8];
        unpack8(*(const u32x4*)dst, o);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] += o[e];
    }
    *(u32x4*)dst = pack8(g);
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = f2bf(in[i]);
}

__global__ __launch_bounds__(256) void add_bf16_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                       bf16_t* __restrict__ y, long n) {
    const long i8 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i8 >= n) return;
    if (i8 + 8 <= n) {
        float x[8], z[8];
        unpack8(*(const u32x4*)(a + i8), x);
        unpack8(*(const u32x4*)(b + i8), z);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] += z[e];
        *(u32x4*)(y + i8) = pack8(x);
    } else {
        for (long i = i8; i < n; ++i) y[i] = f2bf(bf2f(a[i]) + bf2f(b[i]));
    }
}

static inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
static inline int launched() { return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH; }

}  // namespace libra

using namespace libra;

extern "C" int libra_hip_abi_version(void) { return 13; }

extern "C" int libra_patch_im2col(const void* pixel, void* cols, int64_t B, int64_t C, int64_t H, int64_t W,
                                  int64_t P, int64_t Kpad, void* stream) {
    if (B <= 0) return LIBRA_OK;
    if (P <= 0 || H % P || W % P || Kpad < C * P * P || (Kpad % 64)) return LIBRA_ERR_SHAPE;
    if (!pixel || !cols || !al16(cols)) return LIBRA_ERR_ALIGN;
    const long total8 = B * (H / P) * (W / P) * (Kpad / 8);
    hipLaunchKernelGGL(patch_im2col_kernel, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)pixel, (bf16_t*)cols, (int)C, (int)H, (int)W, (int)P, (int)Kpad, total8);
    return launched();
}

extern "C" int libra_patch_col2im(const void* dcols, void* dpixel, int64_t B, int64_t C, int64_t H, int64_t W,
                                  int64_t P, int64_t Kpad, void* stream) {
    if (B <= 0) return LIBRA_OK;
    if (P <= 0 || H % P || W % P || Kpad < C * P * P) return LIBRA_ERR_SHAPE;
    if (!dcols || !dpixel) return LIBRA_ERR_ALIGN;
    const long total = B * C * H * W;
    hipLaunchKernelGGL(patch_col2im_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)dcols, (bf16_t*)dpixel, (int)C, (int)H, (int)W, (int)P, (int)Kpad, total);
    return launched();
}

extern "C" int libra_feature_select(const void* const* hs, int64_t n_sel, void* feat, int64_t B, int64_t T,
                                    int64_t D, void* stream) {
    if (B <= 0) return LIBRA_OK;
    if (n_sel < 1 || n_sel > 4 || T < 2 || (D % 8)) return LIBRA_ERR_SHAPE;
    if (!hs || !feat || !al16(feat)) return LIBRA_ERR_ALIGN;
    SelPtrs s;
    for (int j = 0; j < 4; ++j) {
        s.p[j] = j < n_sel ? (const bf16_t*)hs[j] : nullptr;
        if (j < n_sel && (!hs[j] || !al16(hs[j]))) return LIBRA_ERR_ALIGN;
    }
    const long total8 = B * (T - 1) * n_sel * (D / 8);
    hipLaunchKernelGGL(feature_select_kernel, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       s, (int)n_sel, (bf16_t*)feat, (int)T, (int)D, total8);
    return launched();
}

extern "C" int libra_feature_select_bwd(const void* dfeat, void* const* dhs, const int* accumulate, int64_t n_sel,
                                        int64_t B, int64_t T, int64_t D, void* stream) {
    if (B <= 0) return LIBRA_OK;
    if (n_sel < 1 || n_sel > 4 || T < 2 || (D % 8)) return LIBRA_ERR_SHAPE;
    if (!dfeat || !dhs || !al16(dfeat)) return LIBRA_ERR_ALIGN;
    SelPtrsW s;
    for (int j = 0; j < 4; ++j) {
        s.p[j] = j < n_sel ? (bf16_t*)dhs[j] : nullptr;
        s.acc[j] = (j < n_sel && accumulate) ? accumulate[j] : 0;
        if (j < n_sel && (!dhs[j] || !al16(dhs[j]))) return LIBRA_ERR_ALIGN;
    }
    const long total8 = B * T * n_sel * (D / 8);
    hipLaunchKernelGGL(feature_select_bwd_kernel, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (const bf16_t*)dfeat, s, (int)n_sel, (int)T, (int)D, total8);
    return launched();
}

extern "C" int libra_f32_to_bf16(const float* in, void* out, int64_t n, void* stream) {
    if (n <= 0) return LIBRA_OK;
    if (!in || !out) return LIBRA_ERR_ALIGN;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in,
                       (bf16_t*)out, (long)n);
    return launched();
}

extern "C" int libra_add_bf16(const void* a, const void* b, void* y, int64_t n, void* stream) {
    if (n <= 0) return LIBRA_OK;
    if (!a || !b || !y || !al16(a) || !al16(b) || !al16(y)) return LIBRA_ERR_ALIGN;
    const long nt = (n + 7) / 8;
    hipLaunchKernelGGL(add_bf16_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y, (long)n);
    return launched();
}
