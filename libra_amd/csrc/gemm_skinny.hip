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
    int I;                          // SW instantiation: B = [gate rows 0..I) ; up rows I..2I), C[M, I] = silu(A.gate^T) * (A.up^T)
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

// MR = rows held per lane (8 or 16), NC = columns per workgroup (MR * NC = 64 accumulators per lane), KS = waves per workgroup:
// the waves of a workgroup take every KS-th 512-element pass of the SAME NC columns and their sums are folded through LDS.
// [Round 3: one wave per 8 columns gave N / 8 waves - 512 for the o / down projections, two per CU - each walking K serially
//  with an exposed L2 round trip for its A chunk per pass: 1.6-2.8 TB/s on the N = 4096 shapes.  Splitting K over the waves of a
//  workgroup multiplies the loads in flight by KS and shortens every wave's serial chain to K / (512 KS) passes.]
// SW (the MLP's first half in one launch, LlamaMLP act_fn(gate) * up, modeling_llama.py:199-201): a workgroup owns NC / 2 output
// columns and streams the gate row AND the up row of each; the epilogue applies swiglu_kernel's exact arithmetic (both products
// rounded to bf16 first, bf16(silu(g)) * u) - bit-identical to the GEMM + libra_swiglu pair, one launch and one round trip of the
// [M, 2I] intermediate less per generated token and layer.
template <int MR, int NC, int KS, bool SW = false>
__global__ __launch_bounds__(64 * KS) void gemm_skinny_kernel(const SkinnyArgs p) {
    __shared__ float red[KS > 1 ? KS : 1][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int HALF = NC / 2;
    const int n0 = blockIdx.x * (SW ? HALF : NC);
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
        int n = n0 + c < p.N ? n0 + c : p.N - 1;                              // clamped columns are computed and dropped
        if constexpr (SW) {
            const int col = n0 + (c % HALF) < p.I ? n0 + (c % HALF) : p.I - 1;
            n = c < HALF ? col : p.I + col;
        }
        wrow[c] = p.B + (long)n * p.ldb;
    }
    u32x4 w[NC];
    int k0 = (wave * 64 + lane) * 8;
    constexpr int KSTEP = 512 * KS;
    if (k0 < p.K) {
#pragma unroll
        for (int c = 0; c < NC; ++c) w[c] = __builtin_nontemporal_load((const u32x4*)(wrow[c] + k0));
    }
    for (; k0 < p.K; k0 += KSTEP) {
        u32x4 a[MR], wn[NC];
#pragma unroll
        for (int m = 0; m < MR; ++m) a[m] = *(const u32x4*)(p.A + arow[m] + k0);
        const int k1 = k0 + KSTEP < p.K ? k0 + KSTEP : k0;                    // (the last pass re-requests its own chunk: no branch)
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
    // (lane >> 4) holds the totals of v[e + 4 rho] in rs[e].  With KS > 1 every wave parks its 64 lane-reduced sums in LDS
    // (slot = c * MR + m) and wave 0 folds the waves in index order before the single bf16 rounding.
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
                const int cc = c + idx / MR, m = idx % MR;
                if (KS > 1 || SW) {
                    red[wave][cc * MR + m] = rs[e];
                } else {
                    const int n = n0 + cc;
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
    if constexpr (SW) {
        __syncthreads();
        if (wave == 0 && lane < HALF * MR) {                                  // lane = c * MR + m, c < HALF: gate sum here, up sum 32 lanes on
            float g = 0.f, u = 0.f;
#pragma unroll
            for (int k = 0; k < KS; ++k) { g += red[k][lane]; u += red[k][lane + HALF * MR]; }
            const int n = n0 + lane / MR, m = lane % MR;
            if (m < p.M && n < p.I) {
                const long crow = p.c_rows ? p.c_rows[m] : m;
                const float gb = bf2f(f2bf(g)), ub = bf2f(f2bf(u));
                p.C[crow * p.ldc + n] = f2bf(bf2f(f2bf(gb / (1.0f + __expf(-gb)))) * ub);
            }
        }
    } else if (KS > 1) {
        __syncthreads();
        if (wave == 0) {                                                      // lane = c * MR + m
            float r = 0.f;
#pragma unroll
            for (int k = 0; k < KS; ++k) r += red[k][lane];
            const int n = n0 + lane / MR, m = lane % MR;
            if (m < p.M && n < p.N) {
                const long crow = p.c_rows ? p.c_rows[m] : m;
                if (p.resid) r += bf2f(p.resid[crow * p.ldr + n]);
                p.C[crow * p.ldc + n] = f2bf(r);
            }
        }
    }
}

}  // namespace libra

using namespace libra;

static int skinny_launch(SkinnyArgs& p, void* stream) {
    // waves per workgroup: enough workgroup-waves to cover the chip ~4 times over, never more K slices than 512-element passes
    const bool sw = p.I > 0;
    const long ncols = sw ? p.I : p.N;
    const long per = p.M <= 8 ? (sw ? 4 : 8) : (sw ? 2 : 4);                  // output columns per workgroup
    const long groups = (ncols + per - 1) / per;
    const long passes = (p.K + 511) / 512;
    int ks = 1;
    while (ks < 8 && groups * ks < 4096 && ks * 2 <= passes) ks *= 2;
    const dim3 grid((unsigned)groups);
#define LIBRA_SKINNY(MR_, NC_, KS_) \
    do { if (sw) hipLaunchKernelGGL((gemm_skinny_kernel<MR_, NC_, KS_, true>), grid, dim3(64 * KS_), 0, (hipStream_t)stream, p); \
         else hipLaunchKernelGGL((gemm_skinny_kernel<MR_, NC_, KS_, false>), grid, dim3(64 * KS_), 0, (hipStream_t)stream, p); } while (0)
    if (p.M <= 8) {
        if (ks == 1) LIBRA_SKINNY(8, 8, 1); else if (ks == 2) LIBRA_SKINNY(8, 8, 2); else if (ks == 4) LIBRA_SKINNY(8, 8, 4); else LIBRA_SKINNY(8, 8, 8);
    } else {
        if (ks == 1) LIBRA_SKINNY(16, 4, 1); else if (ks == 2) LIBRA_SKINNY(16, 4, 2); else if (ks == 4) LIBRA_SKINNY(16, 4, 4); else LIBRA_SKINNY(16, 4, 8);
    }
#undef LIBRA_SKINNY
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}

// Internal launcher (declared in gemm_bf16.hip).  Arguments were validated there: K % 64 == 0, 16-byte aligned A / B, M <= 16.
extern "C" int libra_gemm_skinny_launch_(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                                         int64_t N, int64_t K, const void* resid, int64_t ldr, const int32_t* a_rows,
                                         const int32_t* c_rows, void* stream) {
    SkinnyArgs p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = (bf16_t*)C; p.resid = (const bf16_t*)resid;
    p.a_rows = a_rows; p.c_rows = c_rows; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr;
    p.M = (int)M; p.N = (int)N; p.K = (int)K; p.I = 0;
    return skinny_launch(p, stream);
}

extern "C" int libra_gemm_swiglu_skinny(const void* A, int64_t lda, const void* W_gate_up, int64_t ldw, void* Y, int64_t ldy,
                                        int64_t M, int64_t I, int64_t K, const int32_t* a_rows, int64_t a_phys_rows, void* stream) {
    if (M <= 0 || I <= 0) return LIBRA_OK;
    if (!A || !W_gate_up || !Y || M > 16 || K <= 0 || (K % 64) || I > (1 << 29)) return LIBRA_ERR_SHAPE;
    if ((lda % 8) || (ldw % 8) || lda < K || ldw < K || ldy < I) return LIBRA_ERR_SHAPE;
    if (a_rows && a_phys_rows <= 0) return LIBRA_ERR_SHAPE;
    if ((a_rows ? a_phys_rows : M) * lda >= (1LL << 31) || 2 * I * ldw >= (1LL << 31)) return LIBRA_ERR_SHAPE;
    if ((((uintptr_t)A | (uintptr_t)W_gate_up) & 15) || ((uintptr_t)Y & 1)) return LIBRA_ERR_ALIGN;
    SkinnyArgs p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)W_gate_up; p.C = (bf16_t*)Y; p.resid = nullptr;
    p.a_rows = a_rows; p.c_rows = nullptr; p.lda = lda; p.ldb = ldw; p.ldc = ldy; p.ldr = 0;
    p.M = (int)M; p.N = (int)(2 * I); p.K = (int)K; p.I = (int)I;
    return skinny_launch(p, stream);
}
