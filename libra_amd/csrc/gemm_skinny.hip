// Skinny-M "NT" GEMM for the generation step:  C[M, N] = A[M, K] . B[N, K]^T (+ residual), M <= 16 rows (one new token per
// sequence), B = a weight matrix that is read exactly once.  HBM-bound by construction (2 N K bytes of weights against 2 M N K
// FLOP at M <= 16), so no MFMA tile: one wave owns NC output columns, every lane a 16-byte K chunk per pass (the wave sweeps
// 512 reduction elements per pass); the A chunks of all rows are fetched once per pass (L2-resident: the whole A is a few tens of
// KB) and reused against the NC weight rows, products by v_dot2_f32_bf16 into fp32, and the M x NC lane-partial sums are finished
// with the register-only 16-value wave reduction.  Same contract as the tiled kernels for what the decode step uses: optional row
// gather on A, row scatter on C, fused residual (added in fp32 before the single bf16 rounding).
#include "hip_common.hpp"
#include "../../include/libra_hip.h"

namespace libra {

struct SkinnyArgs {
    const bf16_t* A; const bf16_t* B; bf16_t* C; const bf16_t* resid;
    const int* a_rows; const int* c_rows;
    long lda, ldb, ldc, ldr;
    int M, N, K;
};

typedef __bf16 bf16x2_sk __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot8(const u32x4 a, const u32x4 b, float c) {
    // (element first, cast second: __builtin_bit_cast applied directly to a vector element `a[i]` is miscompiled by this hipcc -
    // every iteration reads element 0 and the 16-byte loads shrink to 4 bytes)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned x = a[i], y = b[i];
        c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_sk, x), __builtin_bit_cast(bf16x2_sk, y), c, false);
    }
    return c;
}

// MR = rows held per lane (8 or 16), NC = columns per wave (MR * NC = 64 accumulators)
template <int MR, int NC>
__global__ __launch_bounds__(64) void gemm_skinny_kernel(const SkinnyArgs p) {
    const int lane = threadIdx.x;
    const int n0 = blockIdx.x * NC;
    long arow[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        const int mm = m < p.M ? m : 0;
        arow[m] = (long)(p.a_rows ? p.a_rows[mm] : mm) * p.lda;
    }
    float acc[NC][MR];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int m = 0; m < MR; ++m) acc[c][m] = 0.f;
    // Weight rows are streamed once by exactly one wave: non-temporal loads (they should not displace A in the caches), and the
    // chunk of the NEXT 512 reduction elements is requested before the dot products of the current one - the round-2 loop issued
    // its loads, waited for all of them and only then computed: the memory pipe idled through every compute phase (3.3 TB/s).
    const bf16_t* wrow[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int n = n0 + c < p.N ? n0 + c : p.N - 1;                        // clamped columns are computed and dropped
        wrow[c] = p.B + (long)n * p.ldb;
    }
    u32x4 w[NC];
    int k0 = lane * 8;
    if (k0 < p.K) {
#pragma unroll
        for (int c = 0; c < NC; ++c) w[c] = __builtin_nontemporal_load((const u32x4*)(wrow[c] + k0));
    }
    for (; k0 < p.K; k0 += 512) {
        u32x4 a[MR], wn[NC];
#pragma unroll
        for (int m = 0; m < MR; ++m) a[m] = *(const u32x4*)(p.A + arow[m] + k0);
        const int k1 = k0 + 512 < p.K ? k0 + 512 : k0;                        // (the last pass re-requests its own chunk: no branch)
#pragma unroll
        for (int c = 0; c < NC; ++c) wn[c] = __builtin_nontemporal_load((const u32x4*)(wrow[c] + k1));
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int m = 0; m < MR; ++m) acc[c][m] = dot8(w[c], a[m], acc[c][m]);
#pragma unroll
        for (int c = 0; c < NC; ++c) w[c] = wn[c];
    }
    // 16 sums at a time: v[16] = (MR = 8: two columns x 8 rows | MR = 16: one column x 16 rows); afterwards lane row rho
    // (lane >> 4) holds the totals of v[e + 4 rho] in rs[e]
    constexpr int CPG = 16 / MR;                                              // columns per group of 16 sums
#pragma unroll
    for (int c = 0; c < NC; c += CPG) {
        float v[16], rs[4];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = acc[c + i / MR][i % MR];
        wave_sum16(v, rs);
        if ((lane & 15) == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int idx = e + 4 * (lane >> 4);
                const int n = n0 + c + idx / MR, m = idx % MR;
                if (m < p.M && n < p.N) {
                    const long crow = p.c_rows ? p.c_rows[m] : m;
                    float r = rs[e];
                    if (p.resid) r += bf2f(p.resid[crow * p.ldr + n]);
                    p.C[crow * p.ldc + n] = f2bf(r);
                }
            }
        }
    }
}

}  // namespace libra

using namespace libra;

// Internal launcher (declared in gemm_bf16.hip).  Arguments were validated there: K % 64 == 0, 16-byte aligned A / B, M <= 16.
extern "C" int libra_gemm_skinny_launch_(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                                         int64_t N, int64_t K, const void* resid, int64_t ldr, const int32_t* a_rows,
                                         const int32_t* c_rows, void* stream) {
    SkinnyArgs p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = (bf16_t*)C; p.resid = (const bf16_t*)resid;
    p.a_rows = a_rows; p.c_rows = c_rows; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr;
    p.M = (int)M; p.N = (int)N; p.K = (int)K;
    if (M <= 8)
        hipLaunchKernelGGL((gemm_skinny_kernel<8, 8>), dim3((unsigned)((N + 7) / 8)), dim3(64), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((gemm_skinny_kernel<16, 4>), dim3((unsigned)((N + 3) / 4)), dim3(64), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}
