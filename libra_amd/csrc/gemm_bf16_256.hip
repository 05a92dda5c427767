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
#include "hip_common.hpp"
#include "gemm_tiles.hpp"
#include "gemm_epilogue.hpp"

#ifndef TO256
#define TO256 4, 8, 4, 2
#endif
#ifndef LIBRA_GEMM256_PERSIST
#define LIBRA_GEMM256_PERSIST 1
#endif
namespace libra {

constexpr int HB = 16384;               // one 128x64 half-tile
constexpr int KTB = 4 * HB;             // one K tile: A_lo A_hi B_lo B_hi
constexpr int G256_LDS = 2 * KTB;       // 128 KiB
constexpr int G256_THREADS = 512;

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
    int tid = tid0;
    asm volatile("" : "+v"(tid));          // (also in the one-tile form: 17-29 VGPR spills around its K loop become 0-2)
    const int lane = tid & 63;
    const int l31 = lane & 31, fk = lane >> 5;

    const TileRC trc = tile_order<TO256>(bid, p.tiles_m, p.tiles_n);
    const int tm = trc.tm, tn = trc.tn;
    const int m0 = tm * 256, n0 = tn * 256;
    // grouped launch (same shapes / strides / row maps, different operands): blockIdx.z picks the group
    const bf16_t* Ap = p.A; const bf16_t* Bp = p.B; bf16_t* Cp = p.C;
    {   // (constant indices + selects: a dynamically indexed kernel-argument array would be copied to scratch)
        const int g = blockIdx.z;
        if (g == 1) { Ap = p.Ag[0]; Bp = p.Bg[0]; Cp = p.Cg[0]; }
        else if (g == 2) { Ap = p.Ag[1]; Bp = p.Bg[1]; Cp = p.Cg[1]; }
        else if (g == 3) { Ap = p.Ag[2]; Bp = p.Bg[2]; Cp = p.Cg[2]; }
    }

    // ---- per-lane source offsets (elements) of this wave's 2 x 1-KiB pieces of every half-tile type
    unsigned srcA[2][2], srcB[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            srcA[h][j] = stage_src<AT>(wave * 2 + j, lane, m0 + h * 128, p.M, p.lda, p.a_rows);
            srcB[h][j] = stage_src<BT>(wave * 2 + j, lane, n0 + h * 128, p.N, p.ldb);
        }
    const int ldst = wave * 2048;          // this wave's byte offset inside any half-tile (2 x 1 KiB pieces)
    const long kstepA = ktile_stride<AT>(p.lda), kstepB = ktile_stride<BT>(p.ldb);

    auto stageA = [&](int h, int kt) {
        char* dst = smem + (kt & 1) * KTB + h * HB + ldst;
        const bf16_t* base = Ap + kt * kstepA;
        glds16(base + srcA[h][0], dst);
        glds16(base + srcA[h][1], dst + 1024);
    };
    auto stageB = [&](int h, int kt) {
        char* dst = smem + (kt & 1) * KTB + (2 + h) * HB + ldst;
        const bf16_t* base = Bp + kt * kstepB;
        glds16(base + srcB[h][0], dst);
        glds16(base + srcB[h][1], dst + 1024);
    };
    const unsigned lds0 = (unsigned)(unsigned long)(LIBRA_LDS char*)smem;       // (one address-space cast, not one per piece)
    // wave-uniform K-tile base in an SGPR pair + the loop-invariant per-lane byte offset: no per-piece 64-bit VALU address
    unsigned boA[2][2], boB[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) { boA[h][j] = 2u * srcA[h][j]; boB[h][j] = 2u * srcB[h][j]; }
    auto pieceA = [&](int h, int kt, int j) { glds16_off_at(Ap + kt * kstepA, boA[h][j], lds0 + (unsigned)((kt & 1) * KTB + h * HB + ldst + j * 1024)); };
    auto pieceB = [&](int h, int kt, int j) { glds16_off_at(Bp + kt * kstepB, boB[h][j], lds0 + (unsigned)((kt & 1) * KTB + (2 + h) * HB + ldst + j * 1024)); };

    const FragAddr fa = make_frag_addr(lane);
    const int aoff = wr * HB;                                  // A half of this wave group
    const int boff = (2 + (wc >> 1)) * HB;                     // B half of this wave
    const int toA[4] = {frag_toff<AT>(lane, 0), frag_toff<AT>(lane, 1), frag_toff<AT>(lane, 2), frag_toff<AT>(lane, 3)};
    const int toB[2] = {frag_toff<BT>(lane, (wc & 1) * 2), frag_toff<BT>(lane, (wc & 1) * 2 + 1)};

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk_all = p.K >> 6;
    const int kt0 = (int)((long)nk_all * blockIdx.y / p.splitk);
    const int nk = (int)((long)nk_all * (blockIdx.y + 1) / p.splitk);      // this split's K tiles are [kt0, nk)

    // ---- prologue: K tile 0 complete, B halves of K tile 1 in flight
    stageA(0, kt0); stageA(1, kt0); stageB(0, kt0); stageB(1, kt0);
    if (kt0 + 1 < nk) { stageB(0, kt0 + 1); stageB(1, kt0 + 1); LIBRA_VMCNT(4); } else { LIBRA_VMCNT(0); }
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();                 // stagger the second wave group by one barrier

    bf16x8 a[2][4], b0[4], b1[4];
    // one K tile; STEADY = K tiles kt+1 and kt+2 exist (every iteration but the last two): no tests between the MFMAs
    auto ktile = [&](const int kt, auto steady) {
        constexpr bool STEADY = decltype(steady)::value;
        const bool has1 = STEADY || kt + 1 < nk, has2 = STEADY || kt + 2 < nk;
        const char* buf = smem + (kt & 1) * KTB;
        const char* sa = buf + aoff;
        const char* sb = buf + boff;
        // ================= phase 1: read B0, A0; prefetch A_lo(kt+1); quadrant (0,0) =================
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) b0[ks] = load_frag<BT>(sb, fa, toB[0], ks);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) a[i][ks] = load_frag<AT>(sa, fa, toA[i], ks);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][ks], b0[ks], acc[i][0], 0, 0, 0);
                if (has1) {
                    if (ks * 2 + i == 1) { __builtin_amdgcn_sched_barrier(0); pieceA(0, kt + 1, 0); __builtin_amdgcn_sched_barrier(0); }
                    if (ks * 2 + i == 5) { __builtin_amdgcn_sched_barrier(0); pieceA(0, kt + 1, 1); __builtin_amdgcn_sched_barrier(0); }
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ================= phase 2: read B1; prefetch A_hi(kt+1); quadrant (0,1) =================
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) b1[ks] = load_frag<BT>(sb, fa, toB[1], ks);
        LIBRA_LGKMCNT0();            // all B reads of this K tile retired before the barrier: B may be re-staged next phase
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][ks], b1[ks], acc[i][1], 0, 0, 0);
                if (has1) {
                    if (ks * 2 + i == 1) { __builtin_amdgcn_sched_barrier(0); pieceA(1, kt + 1, 0); __builtin_amdgcn_sched_barrier(0); }
                    if (ks * 2 + i == 5) { __builtin_amdgcn_sched_barrier(0); pieceA(1, kt + 1, 1); __builtin_amdgcn_sched_barrier(0); }
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ================= phase 3: read A1; prefetch B_lo(kt+2); quadrant (1,1) =================
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) a[i][ks] = load_frag<AT>(sa, fa, toA[2 + i], ks);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc[2 + i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][ks], b1[ks], acc[2 + i][1], 0, 0, 0);
                if (has2) {
                    if (ks * 2 + i == 1) { __builtin_amdgcn_sched_barrier(0); pieceB(0, kt + 2, 0); __builtin_amdgcn_sched_barrier(0); }
                    if (ks * 2 + i == 5) { __builtin_amdgcn_sched_barrier(0); pieceB(0, kt + 2, 1); __builtin_amdgcn_sched_barrier(0); }
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ================= phase 4: prefetch B_hi(kt+2); counted wait for K tile kt+1; quadrant (1,0) =================
        // (B_hi(kt+2) is issued after this wait, between the MFMAs below: only B_lo(kt+2) may stay in flight across the barrier)
        if (has2) { LIBRA_VMCNT(2); } else { LIBRA_VMCNT(0); }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc[2 + i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][ks], b0[ks], acc[2 + i][0], 0, 0, 0);
                if (has2) {
                    if (ks * 2 + i == 1) { __builtin_amdgcn_sched_barrier(0); pieceB(1, kt + 2, 0); __builtin_amdgcn_sched_barrier(0); }
                    if (ks * 2 + i == 5) { __builtin_amdgcn_sched_barrier(0); pieceB(1, kt + 2, 1); __builtin_amdgcn_sched_barrier(0); }
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };
    int kt = kt0;
    for (; kt + 2 < nk; ++kt) ktile(kt, std::true_type{});
    for (; kt < nk; ++kt) ktile(kt, std::false_type{});
    if (wr == 0) __builtin_amdgcn_s_barrier();                 // re-align the two groups
    __syncthreads();

    // ---- epilogue: each wave round-trips its own 32x64 fp32 slabs through a private 8 KiB LDS region ----
    // Two instantiations of the same code: INTERIOR (the whole 256x256 tile lies inside C and N is a multiple of 8 - every
    // per-lane bound test and every scalar tail path folds away; > 98 % of the tiles of the hot shapes) and the generic
    // edge version.  The choice is wave-uniform (m0 / n0 come from blockIdx).
    float* ct = (float*)(smem + wave * 8192);
    if (m0 + 256 <= p.M && n0 + 256 <= p.N) gemm_wave_epilogue<true>(p, acc, Cp, ct, m0 + wr * 128, n0 + wc * 64, lane);
    else gemm_wave_epilogue<false>(p, acc, Cp, ct, m0 + wr * 128, n0 + wc * 64, lane);
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
    if (slab) {
        const long MN = (long)M * N;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((MN / 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           slab, p.splitk, MN, (int)N, (bf16_t*)C, (long)ldc, c_rows, (flags & LIBRA_GEMM_RESIDUAL) ? (const bf16_t*)resid : nullptr,
                           (long)ldr);
        if (hipGetLastError() != hipSuccess) return LIBRA_ERR_LAUNCH;
    }
    return LIBRA_OK;
}
