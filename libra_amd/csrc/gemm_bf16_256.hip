// bf16 "NT" GEMM, 256x256x64 tile, 8-phase software pipeline (gfx950).  Same contract and epilogue as
// gemm_bf16.hip's 128x128 kernel; used for problems large enough to fill the chip with 256^2 tiles.
//
// Why a second structure: the 128^2 / one-barrier-per-K-tile kernel moves 15.6 B of L2->LDS traffic per
// kFLOP and drains its load queue at every barrier; it tops out near 0.9 PF/s.  Here
//   * a workgroup is 8 wave64 (2 x 4), each wave owns a 128x64 output block (8 MFMA 32x32 accumulators =
//     128 VGPRs) -> 7.8 B/kFLOP;
//   * a K tile is FOUR 128x64 half-tiles (A_lo, A_hi, B_lo, B_hi; 16 KiB each, the same source-swizzled
//     lane-linear LDS image as the small kernel), two K-tile buffers = 128 KiB LDS, 1 workgroup / CU;
//   * every K tile is processed in 4 phases = the 4 (64x32) quadrants of the wave's block, each phase
//     {LDS->register fragment reads -> s_barrier -> 8 MFMA with ONE half-tile of direct-to-LDS prefetch (2 pieces per
//     wave) issued after the 2nd and the 6th of them -> s_barrier}; fragments are reused across quadrants
//     (A0,B0 | B1 | A1 | -) so a phase reads 12/4/8/0 ds_read_b128.  [Round 2: a direct-to-LDS piece stalls the issuing
//     wave 100-185 cycles next to ds_reads but ~50 among MFMAs; with the pieces next to the reads the "read" interval of
//     one wave group outlasted the 256-cycle MFMA interval of the other - moving them took the step GEMMs from 7.43 to
//     7.10 ms (positions 0|4: no gain, 3|7: half; phase 4's two pieces back in its read-free load part: 7.25), a steady-state loop body without tail tests and one LDS address-space
//     cast per kernel instead of per piece to 6.83 ms (experiments/tools/gemm_big.py; sq8k 1265 -> 1394 TFLOP/s).  Also tried: a PERSISTENT
//     workgroup looping over output tiles with the next tile's first K tile staged under the epilogue - 7.16 ms (spills in the
//     per-tile part, and static tile assignment loses the dispatcher's load balancing); a 4-barrier K tile (16 MFMAs per
//     interval) cannot keep the lagging wave group's prefetch distance - not built.]
//   * the two wave groups (rows 0-127 / 128-255) run the phase sequence staggered by ONE barrier, so on
//     every SIMD one wave issues LDS/VMEM work while its partner owns the matrix pipe (s_setprio 1);
//   * the prefetch stream never drains: half-tiles are issued in the order A_lo(t+1) A_hi(t+1) B_lo(t+2)
//     B_hi(t+2) and the only wait is a counted `s_waitcnt vmcnt(2)` once per K tile (phase 4), i.e. one
//     half-tile stays in flight across that barrier and B_hi(t+2) follows right behind it.  A buffer is re-staged only after the reads of it
//     were retired before a barrier every wave has passed (B: lgkmcnt(0) before phase 2's barrier).
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include "gemm256_body.hpp"

#ifndef LIBRA_GEMM256_PERSIST
#define LIBRA_GEMM256_PERSIST 1
#endif
namespace libra {

// PERS: persistent workgroups (round 5) - the launcher starts one workgroup per CU and workgroup w walks the tiles w, w + P, w + 2 P, ...
// in the order the hardware would have dispatched them (so the lock-step L2 sharing of a wave of tiles is kept); every per-lane
// constant is re-derived per tile from an opaque copy of the thread id (hoisted out of the tile loop they cost the registers that
// made the round-2 persistent attempt spill).
template <bool AT, bool BT, bool PERS>
__global__ __launch_bounds__(G256_THREADS, 2) void gemm_bf16_nt_256_kernel(const Gemm256Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int ntiles = p.tiles_m * p.tiles_n;
#pragma unroll 1
    for (int bid = blockIdx.x; bid < ntiles; bid += PERS ? (int)gridDim.x : ntiles) {
    // grouped launch (same shapes / strides / row maps, different operands): blockIdx.z picks the group
    const bf16_t* Ap = p.A; const bf16_t* Bp = p.B; bf16_t* Cp = p.C;
    {   // (constant indices + selects: a dynamically indexed kernel-argument array would be copied to scratch)
        const int g = blockIdx.z;
        if (g == 1) { Ap = p.Ag[0]; Bp = p.Bg[0]; Cp = p.Cg[0]; }
        else if (g == 2) { Ap = p.Ag[1]; Bp = p.Bg[1]; Cp = p.Cg[1]; }
        else if (g == 3) { Ap = p.Ag[2]; Bp = p.Bg[2]; Cp = p.Cg[2]; }
    }
    gemm256_tile<AT, BT>(p, Ap, Bp, Cp, bid, (int)blockIdx.y, smem, tid0, wave, wr, wc);      // gemm256_body.hpp
    if (PERS) __syncthreads();                                 // the next tile's staging overwrites the epilogue's LDS slabs
    }   // tile loop
}

// out[row(m)][n] = bf16( sum_s slab[s][m][n] (+ resid[row(m)][n]) ), row(m) = c_rows ? c_rows[m] : m; 8 elements per thread
// (deterministic split-K second stage; the routed form serves the decoder's text / vision GEMMs with few output rows)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, int S, long MN, int N, bf16_t* __restrict__ C,
                                                            long ldc, const int* __restrict__ c_rows, const bf16_t* __restrict__ resid,
                                                            long ldr) {
    const long i8 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i8 >= MN) return;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    for (int s = 0; s < S; ++s) {
        const f32x4 a = *(const f32x4*)(slab + (long)s * MN + i8);
        const f32x4 b = *(const f32x4*)(slab + (long)s * MN + i8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += a[e]; v[4 + e] += b[e]; }
    }
    long m = i8 / N;
    const int n = (int)(i8 - m * N);
    if (c_rows) m = c_rows[m];
    if (resid) {
        float r[8];
        unpack8(*(const u32x4*)(resid + m * ldr + n), r);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += r[e];
    }
    *(u32x4*)(C + m * ldc + n) = pack8(v);
}

}  // namespace libra

using namespace libra;

// Second stage of a K-sliced launch (also used by gemm_bf16_multi.hip for its split problems): slab[S][M][N] fp32 -> bf16 C.
extern "C" int libra_splitk_reduce_launch_(const float* slab, int S, int64_t M, int64_t N, void* C, int64_t ldc, const int* c_rows,
                                           const void* resid, int64_t ldr, void* stream) {
    const long MN = (long)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((MN / 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       slab, S, MN, (int)N, (bf16_t*)C, (long)ldc, c_rows, (const bf16_t*)resid, (long)ldr);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}

// Internal launcher (declared in gemm_bf16.hip): returns LIBRA_OK / LIBRA_ERR_LAUNCH. Arguments were validated.
extern "C" int libra_gemm256_launch_(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                                     int64_t M, int64_t N, int64_t K, const void* bias, const void* resid,
                                     int64_t ldr, const void* aux, int64_t ldaux, void* preact, int64_t ldpre,
                                     float alpha, int64_t alpha_cols, int flags, float* slab, int splitk,
                                     const int* a_rows, const int* c_rows, void* stream, int groups,
                                     const void* const* Ag, const void* const* Bg, void* const* Cg) {
    Gemm256Args p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = (bf16_t*)C;
    for (int g = 0; g < 3; ++g) {
        const bool on = g + 1 < groups;
        p.Ag[g] = on ? (const bf16_t*)Ag[g] : nullptr; p.Bg[g] = on ? (const bf16_t*)Bg[g] : nullptr; p.Cg[g] = on ? (bf16_t*)Cg[g] : nullptr;
    }
    p.bias = (const bf16_t*)bias; p.resid = (const bf16_t*)resid; p.aux = (const bf16_t*)aux; p.preact = (bf16_t*)preact;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.ldaux = ldaux; p.ldpre = ldpre;
    p.M = (int)M; p.N = (int)N; p.K = (int)K;
    p.tiles_m = (int)((M + 255) / 256); p.tiles_n = (int)((N + 255) / 256);
    p.alpha = alpha; p.alpha_cols = (int)alpha_cols; p.flags = flags;
    p.slab = slab; p.splitk = splitk < 1 ? 1 : splitk;
    p.a_rows = a_rows; p.c_rows = c_rows;
    const int at = (flags & LIBRA_GEMM_A_T) ? 1 : 0, bt = (flags & LIBRA_GEMM_B_T) ? 1 : 0;
    long nblk = (long)p.tiles_m * p.tiles_n;
    // persistent form: plain launches (no split-K, no groups) with more tiles than CUs
    const bool pers = LIBRA_GEMM256_PERSIST && p.splitk == 1 && groups <= 1 && nblk > cu_count();
    void (*kern)(const Gemm256Args) =
        pers ? (at ? (bt ? gemm_bf16_nt_256_kernel<true, true, true> : gemm_bf16_nt_256_kernel<true, false, true>)
                   : (bt ? gemm_bf16_nt_256_kernel<false, true, true> : gemm_bf16_nt_256_kernel<false, false, true>))
             : (at ? (bt ? gemm_bf16_nt_256_kernel<true, true, false> : gemm_bf16_nt_256_kernel<true, false, false>)
                   : (bt ? gemm_bf16_nt_256_kernel<false, true, false> : gemm_bf16_nt_256_kernel<false, false, false>));
    static std::atomic<bool> attr_set[8];           // zero-initialised; idempotent call, atomic so concurrent first launches do not race
    if (!attr_set[(pers ? 4 : 0) + at * 2 + bt]) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS);
        attr_set[(pers ? 4 : 0) + at * 2 + bt] = true;
    }
    if (pers) nblk = persistent_grid(nblk, 1);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)p.splitk, (unsigned)(groups < 1 ? 1 : groups)), dim3(G256_THREADS), G256_LDS,
                       (hipStream_t)stream, p);
    if (hipGetLastError() != hipSuccess) return LIBRA_ERR_LAUNCH;
    if (slab)
        return libra_splitk_reduce_launch_(slab, p.splitk, M, N, C, ldc, c_rows, (flags & LIBRA_GEMM_RESIDUAL) ? resid : nullptr, ldr, stream);
    return LIBRA_OK;
}
