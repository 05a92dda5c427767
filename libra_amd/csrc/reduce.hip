// Column sums of a bf16 matrix (bias gradients: db[n] = sum_m dY[m,n]) — HBM-bound, deterministic two-stage.
// Stage 1: grid (column groups of 2048, row strips); a thread owns 8 adjacent columns (16-byte loads, coalesced
// across the 256 threads) and walks its strip of rows; partial sums [strip][cols] fp32 go to the workspace.
// Stage 2: out[n] += sum over strips.
#include "hip_common.hpp"
#include "../../include/libra_hip.h"

namespace libra {

constexpr int CS_STRIPS = 256;

__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16_t* __restrict__ x, long ld, long rows, int cols,
                                                             float* __restrict__ part) {
    const int c8 = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (c8 >= cols) return;
    const long per = (rows + gridDim.y - 1) / gridDim.y;
    const long r0 = (long)blockIdx.y * per, r1 = min(rows, r0 + per);
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    const bool full = c8 + 8 <= cols;
    long r = r0;
    if (full) {
        // 8 independent 16-byte loads in flight per thread: a one-load-at-a-time loop is latency bound (2.6 TB/s)
        for (; r + 8 <= r1; r += 8) {
            u32x4 q[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) q[j] = *(const u32x4*)(x + (r + j) * ld + c8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v[8];
                unpack8(q[j], v);
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += v[e];
            }
        }
    }
    for (; r < r1; ++r) {
        float v[8];
        if (full) unpack8(*(const u32x4*)(x + r * ld + c8), v);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (c8 + e < cols) ? bf2f(x[r * ld + c8 + e]) : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += v[e];
    }
    float* dst = part + (long)blockIdx.y * cols + c8;
#pragma unroll
    for (int e = 0; e < 8; ++e) if (c8 + e < cols) dst[e] = s[e];
}

// 32 columns x 32 strip groups per block: every thread adds strips/32 partials, then one LDS reduction
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ part, int strips, int cols,
                                                            float* __restrict__ out) {
    __shared__ float red[32][33];
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + c;
    float s = 0.f;
    if (col < cols)
        for (int p = g; p < strips; p += 32) s += part[(long)p * cols + col];
    red[g][c] = s;
    __syncthreads();
    if (g == 0 && col < cols) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) t += red[i][c];
        out[col] += t;
    }
}

}  // namespace libra

using namespace libra;

extern "C" size_t libra_colsum_workspace_bytes(int64_t rows, int64_t cols) {
    return (rows > 0 && cols > 0) ? (size_t)CS_STRIPS * cols * sizeof(float) : 0;
}

extern "C" int libra_colsum_bf16(const void* x, int64_t ld, int64_t rows, int64_t cols, float* out, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    if (rows <= 0 || cols <= 0) return LIBRA_OK;
    if (ld < cols || (ld % 8)) return LIBRA_ERR_SHAPE;
    if (!x || !out || !workspace || (((uintptr_t)x | (uintptr_t)workspace) & 15)) return LIBRA_ERR_ALIGN;
    if (workspace_bytes < libra_colsum_workspace_bytes(rows, cols)) return LIBRA_ERR_ALIGN;
    const int strips = (int)(rows < CS_STRIPS ? rows : CS_STRIPS);
    dim3 grid((unsigned)((cols + 2047) / 2048), (unsigned)strips);
    hipLaunchKernelGGL(colsum_partial_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (long)ld,
                       (long)rows, (int)cols, (float*)workspace);
    if (hipGetLastError() != hipSuccess) return LIBRA_ERR_LAUNCH;
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((cols + 31) / 32)), dim3(1024), 0, (hipStream_t)stream,
                       (const float*)workspace, strips, (int)cols, out);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}
