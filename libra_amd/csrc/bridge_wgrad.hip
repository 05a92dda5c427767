// Weight gradients of the rank-8 bridges as ONE streaming pass per operand (HBM-bound) instead of N = 8 GEMMs over compacted copies.
//
// The bridges (LibraLinear with rank 8, modeling_libra.py:150-189, used at :310-340): kb = B_k[m] (A_k[m] h), vb = B_v[m] (A_v[m] h),
// m = the token's modality.  Their weight gradients are sums of outer products over the tokens of one modality,
//     dB_k[m][c][j] = sum_{t: m_t = m} dkb[t][c] * t_k[t][j]            (weight_B [H, 8];  x = dkb, coef = t_k)
//     dA_k[m][j][c] = sum_{t: m_t = m} dt_k[t][j] * h[t][c]             (weight_A [8, H];  x = h,   coef = dt_k)
// i.e. out[m][j][c] = sum_t [m_t = m] coef[t][j] * x[t][c] with j < 8 or 16: far too skinny for an MFMA tile.  Round 2 ran them as
// M = 4096, N = 8 GEMMs (21 TFLOP/s) on row-compacted copies of dkb / dvb / h - 4 ms of GEMM, ~4 ms of copies and 200 launches per
// step.  Here a lane owns COLS adjacent columns of x and NC x 2 fp32 accumulators and walks a sub-chunk of tokens (x is read exactly
// once, coalesced), the four waves of a workgroup are folded through LDS, and a second small kernel folds the per-chunk partials in
// a fixed order: deterministic, no atomics.  [First version, one thread per column group and 128 chunks with per-lane coefficient
// loads: 98 / 113 us per pass (1.4 TB/s) + 32 us for the fold - slower than what it replaced; profiles/r03_rank_outer.txt]
#include "hip_common.hpp"
#include "../../include/libra_hip.h"

namespace libra {

constexpr int RW_THREADS = 256;
constexpr int RW_TOK = 64;                   // tokens per wave
constexpr int RW_CHUNK = 4 * RW_TOK;         // tokens per workgroup (4 waves walk 4 consecutive sub-chunks of the same columns)

// One workgroup = a 256-token chunk x (64 lanes x COLS) columns.  The chunk's coefficient rows and modality flags are staged in
// LDS once (a token's coefficients are then a broadcast ds_read, not a vector memory load per lane); each wave streams its 64
// tokens with U x-loads in flight per lane, accumulates NC x COLS x 2 (modality) fp32 sums, and the four waves are folded through
// LDS in a fixed order before ONE partial per workgroup is written: partials ws[chunk][m][j][c].
// MODS = 2: both modalities' sums; MODS = 1: only modality `only` is wanted (half the accumulators: twice the columns per lane)
template <int NC, int COLS, int MODS>
__global__ __launch_bounds__(RW_THREADS) void rank_outer_partial_kernel(const bf16_t* __restrict__ x, long ldx,
                                                                        const bf16_t* __restrict__ coef, long ldc,
                                                                        const unsigned char* __restrict__ flag, long N, int C,
                                                                        float* __restrict__ ws, int only) {
    constexpr int ACC = MODS * NC * COLS;                      // accumulators per lane
    __shared__ __attribute__((aligned(16))) char lds[2 * 64 * ACC * 4 > RW_CHUNK * (NC * 2 + 4) ? 2 * 64 * ACC * 4 : RW_CHUNK * (NC * 2 + 4)];
    bf16_t* scoef = (bf16_t*)lds;                              // [RW_CHUNK][NC]
    unsigned* sflag = (unsigned*)(lds + RW_CHUNK * NC * 2);    // [RW_CHUNK]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c0 = (blockIdx.x * 64 + lane) * COLS;
    const long tb = (long)blockIdx.y * RW_CHUNK;
    {   // stage: thread <-> token of the chunk
        const long t = tb + tid;
        u32x4 cv[NC / 8];
#pragma unroll
        for (int j8 = 0; j8 < NC / 8; ++j8) cv[j8] = t < N ? *(const u32x4*)(coef + t * ldc + j8 * 8) : u32x4{0, 0, 0, 0};
        const unsigned f = (t < N && flag) ? flag[t] : 0u;
#pragma unroll
        for (int j8 = 0; j8 < NC / 8; ++j8) *(u32x4*)(scoef + tid * NC + j8 * 8) = cv[j8];
        sflag[tid] = f;
    }
    __syncthreads();
    float al[NC][COLS], av[MODS == 2 ? NC : 1][COLS];        // (MODS = 1: `al` carries the wanted modality, `av` is unused)
#pragma unroll
    for (int j = 0; j < NC; ++j)
#pragma unroll
        for (int e = 0; e < COLS; ++e) { al[j][e] = 0.f; if (MODS == 2) av[j][e] = 0.f; }
    const bool col_ok = c0 < C;
    const long t0 = tb + wave * RW_TOK;
    long t1 = t0 + RW_TOK; t1 = t1 < N ? t1 : N;
    constexpr int U = 8;                                        // x loads in flight per lane
    if (col_ok) {
        for (long t = t0; t < t1; t += U) {
            float xv[U][COLS];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long tt = t + u < t1 ? t + u : t1 - 1;
                if constexpr (COLS == 4) {
                    const u32x2 v = *(const u32x2*)(x + tt * ldx + c0);
                    xv[u][0] = __uint_as_float(v[0] << 16); xv[u][1] = __uint_as_float(v[0] & 0xffff0000u);
                    xv[u][2] = __uint_as_float(v[1] << 16); xv[u][3] = __uint_as_float(v[1] & 0xffff0000u);
                } else {
                    const unsigned v = *(const unsigned*)(x + tt * ldx + c0);
                    xv[u][0] = __uint_as_float(v << 16); xv[u][1] = __uint_as_float(v & 0xffff0000u);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (t + u < t1) {                               // (wave-uniform)
                    const int ti = (int)(t + u - tb);
                    float cf[NC];
#pragma unroll
                    for (int j8 = 0; j8 < NC / 8; ++j8) unpack8(*(const u32x4*)(scoef + ti * NC + j8 * 8), cf + j8 * 8);   // broadcast read
                    const bool vis = __builtin_amdgcn_readfirstlane((int)sflag[ti]) != 0;
                    if (MODS == 1) {
                        if ((vis ? 1 : 0) == only) {
#pragma unroll
                            for (int j = 0; j < NC; ++j)
#pragma unroll
                                for (int e = 0; e < COLS; ++e) al[j][e] = __builtin_fmaf(cf[j], xv[u][e], al[j][e]);
                        }
                    } else if (vis) {
#pragma unroll
                        for (int j = 0; j < NC; ++j)
#pragma unroll
                            for (int e = 0; e < COLS; ++e) av[j][e] = __builtin_fmaf(cf[j], xv[u][e], av[j][e]);
                    } else {
#pragma unroll
                        for (int j = 0; j < NC; ++j)
#pragma unroll
                            for (int e = 0; e < COLS; ++e) al[j][e] = __builtin_fmaf(cf[j], xv[u][e], al[j][e]);
                    }
                }
            }
        }
    }
    // ---- fold the four waves in a fixed order: (2,3) -> (0,1), then 1 -> 0; slot layout [acc index][lane]: conflict-free
    float* red = (float*)lds;
    auto put = [&](int slot) {
        float* r = red + slot * 64 * ACC + lane;
#pragma unroll
        for (int j = 0; j < NC; ++j)
#pragma unroll
            for (int e = 0; e < COLS; ++e) { r[(j * COLS + e) * 64] = al[j][e]; if (MODS == 2) r[((NC + j) * COLS + e) * 64] = av[j][e]; }
    };
    auto add = [&](int slot) {
        const float* r = red + slot * 64 * ACC + lane;
#pragma unroll
        for (int j = 0; j < NC; ++j)
#pragma unroll
            for (int e = 0; e < COLS; ++e) { al[j][e] += r[(j * COLS + e) * 64]; if (MODS == 2) av[j][e] += r[((NC + j) * COLS + e) * 64]; }
    };
    __syncthreads();                                            // everyone is done with the staged coefficients
    if (wave >= 2) put(wave - 2);
    __syncthreads();
    if (wave < 2) add(wave);
    __syncthreads();
    if (wave == 1) put(0);
    __syncthreads();
    if (wave == 0 && col_ok) {
        add(0);
        float* w = ws + (long)blockIdx.y * 2 * NC * C + (MODS == 1 ? (long)only * NC * C : 0);
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            if constexpr (COLS == 4) {
                *(f32x4*)(w + (long)j * C + c0) = f32x4{al[j][0], al[j][1], al[j][2], al[j][3]};
                if (MODS == 2) *(f32x4*)(w + (long)(NC + j) * C + c0) = f32x4{av[j][0], av[j][1], av[j][2], av[j][3]};
            } else {
                *(float2*)(w + (long)j * C + c0) = float2{al[j][0], al[j][1]};
                if (MODS == 2) *(float2*)(w + (long)(NC + j) * C + c0) = float2{av[j][0], av[j][1]};
            }
        }
    }
}

// out_m[j][c] (or [c][j] when transposed) = bf16( sum_chunk ws[chunk][m][j][c] ): a workgroup = 64 outputs x 4 chunk groups; each
// thread folds every 4th chunk in index order, the four group sums are added in a fixed order (deterministic)
__global__ __launch_bounds__(256) void rank_outer_final_kernel(const float* __restrict__ ws, int chunks, int NC, int C,
                                                               bf16_t* __restrict__ out_l, bf16_t* __restrict__ out_v, long ldo,
                                                               int transpose) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + lane;                  // over (m, j, c)
    const long per = (long)NC * C;
    const bool wanted = i < 2 * per && ((i < per ? out_l : out_v) != nullptr);       // (an unwanted half was never written)
    float s = 0.f;
    if (wanted)
        for (int k = grp; k < chunks; k += 4) s += ws[(long)k * 2 * per + i];
    red[grp][lane] = s;
    __syncthreads();
    if (grp != 0 || i >= 2 * per) return;
    s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    const int m = (int)(i / per);
    const long jc = i - m * per;
    const int j = (int)(jc / C), c = (int)(jc - (long)j * C);
    bf16_t* o = m ? out_v : out_l;
    if (!o) return;
    if (transpose) o[(long)c * ldo + j] = f2bf(s);
    else o[(long)j * ldo + c] = f2bf(s);
}

static inline int rw_chunks(long N, long C) {
    (void)C;
    const long chunks = (N + RW_CHUNK - 1) / RW_CHUNK;
    return (int)(chunks < 1 ? 1 : chunks);
}

}  // namespace libra

using namespace libra;

extern "C" size_t libra_rank_outer_wgrad_workspace_bytes(int64_t N, int64_t C, int64_t ncoef) {
    if (N <= 0 || C <= 0 || ncoef <= 0) return 0;
    return (size_t)rw_chunks(N, C) * 2 * (size_t)ncoef * (size_t)C * sizeof(float);
}

extern "C" int libra_rank_outer_wgrad(const void* x, int64_t ldx, const void* coef, int64_t ldcoef, int64_t ncoef,
                                      const uint8_t* flag, void* out_l, void* out_v, int64_t ldo, int transpose_out, int64_t N,
                                      int64_t C, float* workspace, size_t workspace_bytes, void* stream) {
    if ((ncoef != 8 && ncoef != 16) || C <= 0 || (C % 8) || ldx < C || (ldx % 2) || ldcoef < ncoef || (ldcoef % 8) || N < 0)
        return LIBRA_ERR_SHAPE;
    if (ldo < (transpose_out ? ncoef : C)) return LIBRA_ERR_SHAPE;
    if (!out_l && !out_v) return LIBRA_OK;
    if (!workspace || ((uintptr_t)workspace & 15) || workspace_bytes < libra_rank_outer_wgrad_workspace_bytes(N > 0 ? N : 1, C, ncoef))
        return LIBRA_ERR_ALIGN;
    if (N > 0 && (!x || !coef || ((uintptr_t)x & 7) || ((uintptr_t)coef & 15))) return LIBRA_ERR_ALIGN;
    const int chunks = N > 0 ? rw_chunks(N, C) : 0;
    if (chunks > 65535) return LIBRA_ERR_SHAPE;
    if (N > 0) {
        if (ncoef == 8) {
            if ((C % 4) || (ldx % 4)) return LIBRA_ERR_SHAPE;
            const dim3 grid((unsigned)((C / 4 + 63) / 64), (unsigned)chunks);
            hipLaunchKernelGGL((rank_outer_partial_kernel<8, 4, 2>), grid, dim3(RW_THREADS), 0, (hipStream_t)stream, (const bf16_t*)x,
                               (long)ldx, (const bf16_t*)coef, (long)ldcoef, flag, (long)N, (int)C, workspace, 0);
        } else if ((out_l == nullptr) != (out_v == nullptr) && (C % 4) == 0 && (ldx % 4) == 0) {
            // one modality wanted (the frozen-language recipe's bridge A gradients): half the accumulators, 4 columns per lane
            const dim3 grid((unsigned)((C / 4 + 63) / 64), (unsigned)chunks);
            hipLaunchKernelGGL((rank_outer_partial_kernel<16, 4, 1>), grid, dim3(RW_THREADS), 0, (hipStream_t)stream, (const bf16_t*)x,
                               (long)ldx, (const bf16_t*)coef, (long)ldcoef, flag, (long)N, (int)C, workspace, out_v ? 1 : 0);
        } else {
            const dim3 grid((unsigned)((C / 2 + 63) / 64), (unsigned)chunks);
            hipLaunchKernelGGL((rank_outer_partial_kernel<16, 2, 2>), grid, dim3(RW_THREADS), 0, (hipStream_t)stream, (const bf16_t*)x,
                               (long)ldx, (const bf16_t*)coef, (long)ldcoef, flag, (long)N, (int)C, workspace, 0);
        }
        if (hipGetLastError() != hipSuccess) return LIBRA_ERR_LAUNCH;
    }
    const long tot = 2L * ncoef * C;
    hipLaunchKernelGGL(rank_outer_final_kernel, dim3((unsigned)((tot + 63) / 64)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, chunks, (int)ncoef, (int)C, (bf16_t*)out_l, (bf16_t*)out_v, (long)ldo, transpose_out);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}
