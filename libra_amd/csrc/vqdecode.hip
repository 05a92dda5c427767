// Row kernels of the VQ image decoder (LFQ.indices_to_codes -> post_quant_conv -> taming Decoder;
// /root/reference/libra/models/libra/taming/modules/quantization/lookup_free_quantization.py:129-158,
// /root/reference/libra/models/libra/taming/modules/diffusionmodules/model.py:28-232, :474-588) - all HBM-bound.
// Activations are NHWC ([B, H, W, C] bf16, i.e. [pixels, channels] row-major): every 1x1 conv is a plain bf16 MFMA GEMM and a
// 3x3 conv is ONE GEMM over K = 9*C on a gathered operand built by `im2col3x3` below, which also applies what precedes the
// conv in the reference - GroupNorm(32) + swish with the reference's bf16 rounding points, and the nearest-neighbour upsample -
// so neither the normalised nor the upsampled activation is ever materialised.
#include "hip_common.hpp"
#include "../../include/libra_hip.h"

namespace libra {

// ---- LFQ codes: codes[m, q*nbits + j] = bit (nbits-1-j) of indices[m, q] ? +1 : -1; columns >= Q*nbits are zero ----
__global__ __launch_bounds__(256) void lfq_codes_kernel(const long long* __restrict__ idx, bf16_t* __restrict__ codes, long M, int Q,
                                                        int nbits, long ldc) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int c8 = (int)(ldc >> 3);
    if (i >= M * c8) return;
    const long m = i / c8;
    const int c0 = (int)(i - m * c8) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = c0 + e;
        if (c < Q * nbits) {
            const int q = c / nbits, j = c - q * nbits;
            v[e] = ((idx[m * Q + q] >> (nbits - 1 - j)) & 1) ? 1.f : -1.f;
        } else v[e] = 0.f;
    }
    *(u32x4*)(codes + m * ldc + c0) = pack8(v);
}

// ---- GroupNorm statistics, stage 1: per (image, strip of pixels) the per-channel sum and sum of squares ----
// thread = 8 consecutive channels of a pixel (16-byte loads, coalesced over the row); the 256 threads cover 2048 / C pixels
// per step; the threads of a block that share a channel chunk are summed through LDS.  part: [B][strips][2][C] fp32.
__global__ __launch_bounds__(256) void gn_partial_kernel(const bf16_t* __restrict__ x, long HW, int C, int strips, float* __restrict__ part) {
    __shared__ float red[2][256][8];
    const int b = blockIdx.y, strip = blockIdx.x;
    const int cpr = C >> 3;                                   // 16-byte chunks per pixel
    const int chunk = threadIdx.x % cpr, prow = threadIdx.x / cpr, rows_per_step = 256 / cpr;
    const long per = (HW + strips - 1) / strips;
    const long p0 = (long)strip * per, p1 = min(HW, p0 + per);
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
    if (prow < rows_per_step) {
        const bf16_t* base = x + ((long)b * HW) * C + chunk * 8;
        for (long p = p0 + prow; p < p1; p += rows_per_step) {
            float v[8];
            unpack8(*(const u32x4*)(base + p * C), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[e] += v[e]; q[e] = fmaf(v[e], v[e], q[e]); }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][threadIdx.x][e] = s[e]; red[1][threadIdx.x][e] = q[e]; }
    __syncthreads();
    if (threadIdx.x < cpr) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = 0.f, c = 0.f;
            for (int r = 0; r < rows_per_step; ++r) { a += red[0][r * cpr + threadIdx.x][e]; c += red[1][r * cpr + threadIdx.x][e]; }
            float* dst = part + (((long)b * strips + strip) * 2) * C + threadIdx.x * 8 + e;
            dst[0] = a; dst[C] = c;
        }
    }
}
// stage 2: one block per image, one thread per channel: group mean / rstd -> the per-channel affine y = x * scale + shift
__global__ __launch_bounds__(1024) void gn_final_kernel(const float* __restrict__ part, int strips, int C, int G, long HW, float eps,
                                                        const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta,
                                                        float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ float gs[1024], gq[1024];
    const int b = blockIdx.x, c = threadIdx.x;
    float s = 0.f, q = 0.f;
    if (c < C)
        for (int t = 0; t < strips; ++t) {
            const float* src = part + (((long)b * strips + t) * 2) * C + c;
            s += src[0]; q += src[C];
        }
    gs[c] = s; gq[c] = q;
    __syncthreads();
    if (c < C) {
        const int cg = C / G, g0 = (c / cg) * cg;
        float a = 0.f, d = 0.f;
        for (int j = 0; j < cg; ++j) { a += gs[g0 + j]; d += gq[g0 + j]; }
        const float n = (float)HW * cg;
        const float mean = a / n;
        const float var = fmaxf(d / n - mean * mean, 0.f);
        const float rstd = rsqrtf(var + eps);
        const float w = bf2f(gamma[c]) * rstd;
        scale[(long)b * C + c] = w;
        shift[(long)b * C + c] = bf2f(beta[c]) - mean * w;
    }
}

// ---- gathered conv operand ----
// out[(b, y, x), tap * C + c] = f( in[b, sy(y + dy), sx(x + dx), c] ) for the taps (dy, dx) of a ksize x ksize window (zero outside
// the image), where (sy, sx) = nearest-neighbour source of the UPSAMPLED pixel (PyTorch: min(int(floorf(dst * inv_scale)), n - 1))
// and f = identity, or GroupNorm affine (scale / shift per (b, c), rounded to bf16 like the reference's GroupNorm output) followed
// - when `swish` - by bf16(y * bf16(sigmoid(y))).  Columns >= taps * C up to ldo are zero (GEMM K granule).
struct Im2colArgs {
    const bf16_t* x; bf16_t* out; const float* scale; const float* shift;
    int B, Hs, Ws, C, H, W, ksize, swish;
    long ldo; float inv_h, inv_w;
};
__global__ __launch_bounds__(256) void im2col_kernel(const Im2colArgs p) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int c8 = (int)(p.ldo >> 3), cpr = p.C >> 3;
    const long M = (long)p.B * p.H * p.W;
    if (i >= M * c8) return;
    const long m = i / c8;
    const int j = (int)(i - m * c8);
    const int tap = j / cpr, c0 = (j - tap * cpr) * 8;
    u32x4 o = u32x4{0, 0, 0, 0};
    if (tap < p.ksize * p.ksize) {
        const int b = (int)(m / ((long)p.H * p.W));
        const int rem = (int)(m - (long)b * p.H * p.W);
        const int y = rem / p.W, x = rem - y * p.W;
        const int half = p.ksize >> 1;
        const int yy = y + tap / p.ksize - half, xx = x + tap % p.ksize - half;
        if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) {
            const int sy = min((int)floorf(yy * p.inv_h), p.Hs - 1), sx = min((int)floorf(xx * p.inv_w), p.Ws - 1);
            const u32x4 raw = *(const u32x4*)(p.x + (((long)b * p.Hs + sy) * p.Ws + sx) * p.C + c0);
            if (p.scale) {
                float v[8];
                unpack8(raw, v);
                const float* sc = p.scale + (long)b * p.C + c0;
                const float* sh = p.shift + (long)b * p.C + c0;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float t = bf2f(f2bf(fmaf(v[e], sc[e], sh[e])));
                    if (p.swish) t = t * bf2f(f2bf(1.0f / (1.0f + __expf(-t))));
                    v[e] = t;
                }
                o = pack8(v);
            } else o = raw;
        }
    }
    *(u32x4*)(p.out + m * p.ldo + (long)j * 8) = o;
}

// ---- row softmax of the spatial attention: x <- softmax(bf16(x * scale)) over the first `cols` columns, zeros up to ld ----
// one wave per row; both bf16 rounding points of the reference (bmm output * c^-0.5, then F.softmax in fp32 -> bf16) are kept
__global__ __launch_bounds__(256) void softmax_rows_kernel(bf16_t* __restrict__ x, long rows, int cols, long ld, float scale) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    bf16_t* r = x + row * ld;
    float mx = -INFINITY;
    for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, bf2f(f2bf(bf2f(r[c]) * scale)));
    mx = wave_max(mx);
    float sum = 0.f;
    for (int c = lane; c < cols; c += 64) sum += __expf(bf2f(f2bf(bf2f(r[c]) * scale)) - mx);
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int c = lane; c < (int)ld; c += 64)
        r[c] = c < cols ? f2bf(__expf(bf2f(f2bf(bf2f(r[c]) * scale)) - mx) * inv) : (bf16_t)0;
}

}  // namespace libra

using namespace libra;

extern "C" int libra_lfq_codes(const int64_t* indices, void* codes, int64_t M, int64_t Q, int64_t nbits, int64_t ldc, void* stream) {
    if (M <= 0) return LIBRA_OK;
    if (Q <= 0 || nbits <= 0 || nbits > 30 || ldc < Q * nbits || (ldc % 8)) return LIBRA_ERR_SHAPE;
    if (!indices || !codes || ((uintptr_t)codes & 15)) return LIBRA_ERR_ALIGN;
    const long n = M * (ldc / 8);
    hipLaunchKernelGGL(lfq_codes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const long long*)indices, (bf16_t*)codes, (long)M, (int)Q, (int)nbits, (long)ldc);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}

static int gn_strips(int64_t HW) { return (int)(HW >= 4096 ? 64 : (HW >= 256 ? 16 : 1)); }

extern "C" size_t libra_groupnorm_workspace_bytes(int64_t B, int64_t HW, int64_t C) {
    return (B > 0 && HW > 0 && C > 0) ? (size_t)B * gn_strips(HW) * 2 * C * sizeof(float) : 0;
}

extern "C" int libra_groupnorm_affine(const void* x, const void* gamma, const void* beta, float* scale, float* shift, int64_t B,
                                      int64_t HW, int64_t C, int64_t G, float eps, void* workspace, size_t workspace_bytes,
                                      void* stream) {
    if (B <= 0 || HW <= 0) return LIBRA_OK;
    if (C <= 0 || (C % 8) || C > 1024 || G <= 0 || (C % G) || B > 65535) return LIBRA_ERR_SHAPE;
    if (!x || !gamma || !beta || !scale || !shift || !workspace || ((uintptr_t)x & 15)) return LIBRA_ERR_ALIGN;
    if (workspace_bytes < libra_groupnorm_workspace_bytes(B, HW, C)) return LIBRA_ERR_ALIGN;
    const int strips = gn_strips(HW);
    hipLaunchKernelGGL(gn_partial_kernel, dim3((unsigned)strips, (unsigned)B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       (long)HW, (int)C, strips, (float*)workspace);
    if (hipGetLastError() != hipSuccess) return LIBRA_ERR_LAUNCH;
    hipLaunchKernelGGL(gn_final_kernel, dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream, (const float*)workspace, strips, (int)C,
                       (int)G, (long)HW, eps, (const bf16_t*)gamma, (const bf16_t*)beta, scale, shift);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}

extern "C" int libra_conv_gather(const void* x, void* out, int64_t ldo, const float* scale, const float* shift, int swish, int64_t B,
                                 int64_t Hs, int64_t Ws, int64_t C, int64_t H, int64_t W, int64_t ksize, float inv_scale_h,
                                 float inv_scale_w, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return LIBRA_OK;
    if (Hs <= 0 || Ws <= 0 || C <= 0 || (C % 8) || (ksize != 1 && ksize != 3) || ldo < ksize * ksize * C || (ldo % 8))
        return LIBRA_ERR_SHAPE;
    if (!x || !out || (((uintptr_t)x | (uintptr_t)out) & 15) || ((scale == nullptr) != (shift == nullptr))) return LIBRA_ERR_ALIGN;
    Im2colArgs a;
    a.x = (const bf16_t*)x; a.out = (bf16_t*)out; a.scale = scale; a.shift = shift;
    a.B = (int)B; a.Hs = (int)Hs; a.Ws = (int)Ws; a.C = (int)C; a.H = (int)H; a.W = (int)W; a.ksize = (int)ksize; a.swish = swish;
    a.ldo = ldo; a.inv_h = inv_scale_h; a.inv_w = inv_scale_w;
    const long n = B * H * W * (ldo / 8);
    if ((n + 255) / 256 > 0x7fffffffL) return LIBRA_ERR_SHAPE;
    hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}

extern "C" int libra_softmax_rows(void* x, int64_t rows, int64_t cols, int64_t ld, float scale, void* stream) {
    if (rows <= 0 || cols <= 0) return LIBRA_OK;
    if (ld < cols) return LIBRA_ERR_SHAPE;
    if (!x) return LIBRA_ERR_ALIGN;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, (long)rows,
                       (int)cols, (long)ld, scale);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}
