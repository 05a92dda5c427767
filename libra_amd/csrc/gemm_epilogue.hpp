// Pieces shared by the two MFMA-tiled bf16 GEMM structures whose waves own a 128 x 64 output block (gemm_bf16_256.hip: 256 x 256
// tiles, 8 waves, one workgroup per CU; gemm_bf16_w.hip: 256 x 128 tiles, 4 waves, two workgroups per CU): the argument block and
// the fused epilogue of one wave.
#pragma once
#include <type_traits>
#include "hip_common.hpp"
#include "../../include/libra_hip.h"

namespace libra {

struct Gemm256Args {
    const bf16_t* A; const bf16_t* B; bf16_t* C;
    const bf16_t* bias; const bf16_t* resid; const bf16_t* aux; bf16_t* preact;
    const int* a_rows; const int* c_rows;     // optional row gather (A, N-type only) / scatter (C, resid, aux, preact)
    long lda, ldb, ldc, ldr, ldaux, ldpre;
    int M, N, K;
    int tiles_m, tiles_n;
    float alpha; int alpha_cols;
    int flags;
    float* slab; int splitk;          // split-K: raw fp32 partial tiles to slab[split][M][N] (no epilogue)
    const bf16_t* Ag[3]; const bf16_t* Bg[3]; bf16_t* Cg[3];   // grouped launch: operands of groups 1..3 (blockIdx.z)
};

__device__ __forceinline__ float qgelu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x)); }   // (v_rcp_f32: 1 ulp, the result is rounded to bf16)
__device__ __forceinline__ float qgelu_grad(float x) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x));
    return s * (1.0f + 1.702f * x * (1.0f - s));
}

#define LIBRA_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define LIBRA_VMCNT_N(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define LIBRA_LGKMCNT0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// ---- epilogue of ONE wave: its 128 x 64 block (acc[i][j] = 32 x 32 MFMA accumulators, i = 32-row slab, j = 32-column half) goes
// through a private 8 KiB LDS region `ct` one 32 x 64 fp32 slab at a time and leaves as row-contiguous 16-byte bf16 stores with the
// fused operands applied (bias, column scale, quick-GELU (+ pre-activation store), GELU', residual; or raw fp32 split-K slabs).
// m0w / n0w: first output row / column of the wave's block; ky: the K slice of a split-K launch (slab index).  Two instantiations of the same code: INTERIOR (the whole
// workgroup tile lies inside C and N is a multiple of 8 - every per-lane bound test and every scalar tail path folds away; > 98 %
// of the tiles of the hot shapes) and the generic edge version.  The choice is wave-uniform (it comes from blockIdx).
// (NJ / J0: the accumulator array may be wider than the 64 columns handled here - gemm_bf16_x.hip's waves own 128 x 128 and call
// this once per 64-column half, J0 = 0 / 2.)
template <bool IN, int NJ = 2, int J0 = 0>
__device__ __forceinline__ void gemm_wave_epilogue(const Gemm256Args& p, const f32x16 (&acc)[4][NJ], bf16_t* Cp, float* ct,
                                                   const int m0w, const int n0w, const int lane, const int ky) {
    const int l31 = lane & 31, fk = lane >> 5;
    const int cg = lane & 7;                                   // 8-column group within the wave's 64 columns
    const int gn = n0w + cg * 8;
    const bool want_aux = (p.flags & LIBRA_GEMM_MUL_QGELU_GRAD) && !p.slab, want_res = (p.flags & LIBRA_GEMM_RESIDUAL) && !p.slab;
    // one prefetched fused operand per slab: aux when MUL_QGELU_GRAD is on, else the residual (both at once - no caller
    // does that - loads the residual late); keeping both in flight spilled registers
    struct Extras { int om[4]; u32x4 ext[4]; };
    const bool ncol_ok = IN || gn < p.N;
    const bool full8 = IN || gn + 8 <= p.N;
    float bias[8], cs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { bias[e] = 0.f; cs[e] = (gn + e < p.alpha_cols) ? p.alpha : 1.0f; }
    if ((p.flags & LIBRA_GEMM_BIAS) && ncol_ok) {
        if (full8) unpack8(*(const u32x4*)(p.bias + gn), bias);
        else for (int e = 0; e < 8 && gn + e < p.N; ++e) bias[e] = bf2f(p.bias[gn + e]);
    }
    // The fused operands (row map, aux, residual) of 32-row slab i+1 are fetched before slab i is processed: the
    // epilogue was a chain of 16 dependent global-load round trips per wave (a third of the launch at K = 1024).
    auto fetch = [&](const int i, Extras& x) {
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int gm = m0w + i * 32 + pass * 8 + (lane >> 3);
            const bool ok = IN || (gm < p.M && ncol_ok);
            x.om[pass] = (ok && p.c_rows) ? p.c_rows[gm] : gm;
        }
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int gm = m0w + i * 32 + pass * 8 + (lane >> 3);
            const bool ok = IN || (gm < p.M && full8);
            x.ext[pass] = u32x4{0, 0, 0, 0};
            if (want_aux && ok) x.ext[pass] = *(const u32x4*)(p.aux + (long)x.om[pass] * p.ldaux + gn);
            else if (want_res && ok) x.ext[pass] = *(const u32x4*)(p.resid + (long)x.om[pass] * p.ldr + gn);
        }
    };
    auto epi = [&](const f32x16& c0, const f32x16& c1, const int i, const Extras& ex) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ct[((r & 3) + 8 * (r >> 2) + 4 * fk) * 64 + j * 32 + l31] = (j == 0 ? c0 : c1)[r];
        // same-wave LDS write -> read (in-order per wave)
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int row = pass * 8 + (lane >> 3);
            const int gm = m0w + i * 32 + row;
            if (IN || (gm < p.M && ncol_ok)) {
                const int om = ex.om[pass];
                float v[8];
                const f32x4 lo = *(const f32x4*)(ct + row * 64 + cg * 8);
                const f32x4 hi = *(const f32x4*)(ct + row * 64 + cg * 8 + 4);
                if (p.slab) {
                    float* sd = p.slab + ((long)ky * p.M + gm) * p.N + gn;   // (ky = K slice) slabs are never row-mapped
                    if (full8) { *(f32x4*)sd = lo; *(f32x4*)(sd + 4) = hi; }
                    else for (int e = 0; e < 8 && gn + e < p.N; ++e) sd[e] = e < 4 ? lo[e] : hi[e - 4];
                    continue;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (v[e] + bias[e]) * cs[e];
                if (p.flags & LIBRA_GEMM_STORE_PREACT) {
                    bf16_t* pd = p.preact + (long)om * p.ldpre + gn;
                    if (full8) *(u32x4*)pd = pack8(v);
                    else for (int e = 0; e < 8 && gn + e < p.N; ++e) pd[e] = f2bf(v[e]);
                }
                if (p.flags & LIBRA_GEMM_QUICK_GELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = qgelu(bf2f(f2bf(v[e])));
                }
                if (p.flags & LIBRA_GEMM_MUL_QGELU_GRAD) {
                    float x[8];
                    if (full8) unpack8(ex.ext[pass], x);
                    else for (int e = 0; e < 8; ++e) x[e] = (gn + e < p.N) ? bf2f(p.aux[(long)om * p.ldaux + gn + e]) : 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= qgelu_grad(x[e]);
                }
                if (p.flags & LIBRA_GEMM_RESIDUAL) {
                    float x[8];
                    if (full8) unpack8(want_aux ? *(const u32x4*)(p.resid + (long)om * p.ldr + gn) : ex.ext[pass], x);
                    else for (int e = 0; e < 8; ++e) x[e] = (gn + e < p.N) ? bf2f(p.resid[(long)om * p.ldr + gn + e]) : 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += x[e];
                }
                bf16_t* dst = Cp + (long)om * p.ldc + gn;
                if (full8) *(u32x4*)dst = pack8(v);
                else for (int e = 0; e < 8 && gn + e < p.N; ++e) dst[e] = f2bf(v[e]);
            }
        }
    };
    Extras e0, e1;
    fetch(0, e0);
    fetch(1, e1);
    epi(acc[0][J0], acc[0][J0 + 1], 0, e0);
    fetch(2, e0);
    epi(acc[1][J0], acc[1][J0 + 1], 1, e1);
    fetch(3, e1);
    epi(acc[2][J0], acc[2][J0 + 1], 2, e0);
    epi(acc[3][J0], acc[3][J0 + 1], 3, e1);
}

}  // namespace libra
