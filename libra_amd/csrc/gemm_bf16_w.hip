// bf16 "NT" GEMM, 256x128x64 tile, FOUR waves, TWO workgroups per CU (gfx950).  Same contract, operand modes, row maps, grouped /
// split-K launches and fused epilogue as gemm_bf16_256.hip (the argument block and the per-wave epilogue are shared:
// gemm_epilogue.hpp); chosen by the planner in gemm_bf16.hip for the shapes where one 128-KiB workgroup per CU loses.
//
// Why a third structure.  The 256^2 / 8-wave kernel owns a CU: nothing runs beside its ~9 us epilogue (a quarter of a K = 1024
// tile), a launch of 1.1-1.9 waves of tiles pays whole waves, and problems with < 256 tiles leave CUs empty.  The 128^2 kernel does
// run two workgroups per CU but its waves own 64 x 64 (one LDS fragment read per MFMA) behind a drain-and-barrier per K tile:
// ~0.9 PF at best.  Here a wave owns the SAME 128 x 64 block as in the 256^2 kernel (24 fragment reads per 32 MFMAs), a workgroup
// is one wave per SIMD, and the second workgroup of the CU - an independent tile, never in phase for long - fills the matrix pipe
// while this one reads fragments, waits at its barrier or runs its epilogue.  Half-size tiles also halve the quantisation step
// of a partly filled last wave (512 slots instead of 256).
//
//   * LDS: five 16-KiB units (80 KiB; 2 x 80 = the CU's 160 KiB): A is double buffered as A1st / A2nd of the even / odd K tile -
//     A1st holds the FIRST 64 rows of both 128-row wave blocks (tile rows 0-63, 128-191), A2nd the second 64 (the split is made
//     by the source addressing, gemm_tiles.hpp stage_src<.., SPLIT>), so that A2nd is first read in the SECOND half of a K tile;
//     B (128 lines) is SINGLE buffered and released in the middle of the K tile: a wave reads ALL its B fragments and its
//     A1st fragments at the top of the tile, and after barrier #1 (every wave's B reads retired) the unit takes B(t+1).
//   * K tile = 4 quadrants of the wave's block, 8 MFMAs each: (A0,B0) (A0,B1) | read A1 | (A1,B1) (A1,B0).  The 12 direct-to-LDS
//     pieces of K tile t+1 are issued BETWEEN the MFMAs of quadrants 1-3: A1st(t+1), then - behind barrier #1 - B(t+1), then
//     A2nd(t+1).  The loop never drains: barrier #2 (end of tile t) is preceded by `vmcnt(4)` = everything but the four youngest
//     pieces (A2nd(t+1)) has landed, barrier #1 of tile t+1 by `vmcnt(4)` again = A2nd(t+1) has landed, A1st(t+2) may fly.
//   * hazards: RAW - a piece is waited for by its issuing wave before a barrier every reader passes before its read (A1st,
//     B: barrier #2; A2nd: barrier #1 of the tile that reads it); WAR - A(t+1) goes to the buffers last read in tile t-1 (before
//     barrier #2 of t-1), B(t+1) is issued after barrier #1 of tile t, which every wave reaches with `lgkmcnt(0)` (its B(t)
//     fragments are in registers).  (First version, one drain per tile with A split lo / hi: 13-15 % behind the 256^2 kernel on
//     the text shapes, profiles/r04_gemm_experiments.)
#include <atomic>
#include <type_traits>
#include "hip_common.hpp"
#include "gemm_tiles.hpp"
#include "gemm_epilogue.hpp"

#ifndef TOW
#define TOW 4, 16, 4, 2
#endif
namespace libra {

constexpr int WU = 16384;               // one unit: 128 lines x 64 k
constexpr int GW_LDS = 5 * WU;          // A[2][lo,hi] + B = 80 KiB
constexpr int GW_THREADS = 256;
constexpr int GW_BN = 128;

template <bool AT, bool BT>
__global__ __launch_bounds__(GW_THREADS, 2) void gemm_bf16_nt_w_kernel(const Gemm256Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const TileRC trc = tile_order<TOW>(blockIdx.x, p.tiles_m, p.tiles_n);
    const int tm = trc.tm, tn = trc.tn;
    const int m0 = tm * 256, n0 = tn * GW_BN;
    const bf16_t* Ap = p.A; const bf16_t* Bp = p.B; bf16_t* Cp = p.C;
    {   // grouped launch: blockIdx.z picks the group (constant indices + selects: no scratch copy of the argument arrays)
        const int g = blockIdx.z;
        if (g == 1) { Ap = p.Ag[0]; Bp = p.Bg[0]; Cp = p.Cg[0]; }
        else if (g == 2) { Ap = p.Ag[1]; Bp = p.Bg[1]; Cp = p.Cg[1]; }
        else if (g == 3) { Ap = p.Ag[2]; Bp = p.Bg[2]; Cp = p.Cg[2]; }
    }

    // ---- per-lane source offsets (elements) of this wave's 4 x 1-KiB pieces of every unit type
    unsigned srcA[2][4], srcB[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        srcA[0][j] = stage_src<AT, true>(wave * 4 + j, lane, m0, p.M, p.lda, p.a_rows);          // A1st: rows 0-63, 128-191
        srcA[1][j] = stage_src<AT, true>(wave * 4 + j, lane, m0 + 64, p.M, p.lda, p.a_rows);     // A2nd: rows 64-127, 192-255
        srcB[j] = stage_src<BT>(wave * 4 + j, lane, n0, p.N, p.ldb);
    }
    const long kstepA = ktile_stride<AT>(p.lda), kstepB = ktile_stride<BT>(p.ldb);
    const unsigned lds0 = (unsigned)(unsigned long)(LIBRA_LDS char*)smem;       // (one address-space cast, not one per piece)
    const unsigned ldst = lds0 + (unsigned)(wave * 4096);                        // this wave's 4 pieces inside any unit
    // unit map: A1st / A2nd (h = 0 / 1) of K tile kt at ((kt & 1) * 2 + h) * WU, B at 4 * WU
    // (wave-uniform K-tile base in an SGPR pair + the loop-invariant per-lane byte offset: no per-piece 64-bit VALU address)
    unsigned boA[2][4], boB[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { boA[0][j] = 2u * srcA[0][j]; boA[1][j] = 2u * srcA[1][j]; boB[j] = 2u * srcB[j]; }
    auto pieceA = [&](int h, int kt, int j) { glds16_off_at(Ap + kt * kstepA, boA[h][j], ldst + (unsigned)(((kt & 1) * 2 + h) * WU + j * 1024)); };
    auto pieceB = [&](int kt, int j) { glds16_off_at(Bp + kt * kstepB, boB[j], ldst + (unsigned)(4 * WU + j * 1024)); };

    const FragAddr fa = make_frag_addr(lane);
    const int toA[2] = {frag_toff<AT>(lane, wr * 2), frag_toff<AT>(lane, wr * 2 + 1)};      // this wave's 64 lines inside either A unit
    const int toB[2] = {frag_toff<BT>(lane, wc * 2), frag_toff<BT>(lane, wc * 2 + 1)};

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

    // ---- prologue: K tile kt0 complete
#pragma unroll
    for (int j = 0; j < 4; ++j) { pieceA(0, kt0, j); pieceA(1, kt0, j); pieceB(kt0, j); }
    LIBRA_VMCNT(0);
    __builtin_amdgcn_s_barrier();

    bf16x8 a[2][4], b0[4], b1[4];
    // one K tile; STEADY = K tile kt+1 exists (every iteration but the last): no tests between the MFMAs
    auto ktile = [&](const int kt, auto steady) {
        constexpr bool STEADY = decltype(steady)::value;
        const char* sa = smem + (kt & 1) * 2 * WU;                  // A1st; A2nd at + WU
        const char* sb = smem + 4 * WU;
        // ---- top: all B fragments of this tile + the A1st fragments
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) b0[ks] = load_frag<BT>(sb, fa, toB[0], ks);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) a[i][ks] = load_frag<AT>(sa, fa, toA[i], ks);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) b1[ks] = load_frag<BT>(sb, fa, toB[1], ks);
        __builtin_amdgcn_sched_barrier(0);
        // ================= quadrant (0,0): A1st(kt+1) =================
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][ks], b0[ks], acc[i][0], 0, 0, 0);
                if (STEADY && ((ks * 2 + i) & 1)) {
                    __builtin_amdgcn_sched_barrier(0); pieceA(0, kt + 1, (ks * 2 + i) >> 1); __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        LIBRA_LGKMCNT0();            // this wave's B fragments of the tile are in registers,
        if (STEADY) { LIBRA_VMCNT(4); } else { LIBRA_VMCNT(0); }     // its A2nd(kt) pieces have landed (A1st(kt+1) may fly) ...
        __builtin_amdgcn_s_barrier();   // #1 ... and every other wave's too: the B unit may take B(kt+1), A2nd(kt) may be read
        __builtin_amdgcn_sched_barrier(0);
        // ================= quadrant (0,1): B(kt+1) =================
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][ks], b1[ks], acc[i][1], 0, 0, 0);
                if (STEADY && ((ks * 2 + i) & 1)) {
                    __builtin_amdgcn_sched_barrier(0); pieceB(kt + 1, (ks * 2 + i) >> 1); __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        // ---- A2nd fragments (the registers of the A1st fragments are free now)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) a[i][ks] = load_frag<AT>(sa + WU, fa, toA[i], ks);
        __builtin_amdgcn_sched_barrier(0);
        // ================= quadrant (1,1): A2nd(kt+1) =================
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc[2 + i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][ks], b1[ks], acc[2 + i][1], 0, 0, 0);
                if (STEADY && ((ks * 2 + i) & 1)) {
                    __builtin_amdgcn_sched_barrier(0); pieceA(1, kt + 1, (ks * 2 + i) >> 1); __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // ================= quadrant (1,0): no pieces =================
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[2 + i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][ks], b0[ks], acc[2 + i][0], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (STEADY) { LIBRA_VMCNT(4); }   // A1st(kt+1) and B(kt+1) have landed (this wave's pieces; the four A2nd(kt+1) pieces fly on) ...
        __builtin_amdgcn_s_barrier();   // #2 ... everywhere, and nobody reads the A units of tile kt any more
        __builtin_amdgcn_sched_barrier(0);
    };
    int kt = kt0;
    for (; kt + 1 < nk; ++kt) ktile(kt, std::true_type{});
    for (; kt < nk; ++kt) ktile(kt, std::false_type{});

    // ---- epilogue (gemm_epilogue.hpp): each wave round-trips its own 32x64 fp32 slabs through a private 8 KiB LDS region
    float* ct = (float*)(smem + wave * 8192);
    if (m0 + 256 <= p.M && n0 + GW_BN <= p.N) gemm_wave_epilogue<true>(p, acc, Cp, ct, m0 + wr * 128, n0 + wc * 64, lane, (int)blockIdx.y);
    else gemm_wave_epilogue<false>(p, acc, Cp, ct, m0 + wr * 128, n0 + wc * 64, lane, (int)blockIdx.y);
}

}  // namespace libra

using namespace libra;

// Internal launcher (declared in gemm_bf16.hip), same argument list as libra_gemm256_launch_.  Arguments were validated.
// The split-K slab reduction, when there is one, is launched by the caller (libra_gemm256_launch_ owns that kernel).
extern "C" int libra_gemmw_launch_(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
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
    p.tiles_m = (int)((M + 255) / 256); p.tiles_n = (int)((N + GW_BN - 1) / GW_BN);
    p.alpha = alpha; p.alpha_cols = (int)alpha_cols; p.flags = flags;
    p.slab = slab; p.splitk = splitk < 1 ? 1 : splitk;
    p.a_rows = a_rows; p.c_rows = c_rows;
    const int at = (flags & LIBRA_GEMM_A_T) ? 1 : 0, bt = (flags & LIBRA_GEMM_B_T) ? 1 : 0;
    void (*kern)(const Gemm256Args) =
        at ? (bt ? gemm_bf16_nt_w_kernel<true, true> : gemm_bf16_nt_w_kernel<true, false>)
           : (bt ? gemm_bf16_nt_w_kernel<false, true> : gemm_bf16_nt_w_kernel<false, false>);
    static std::atomic<bool> attr_set[4];           // zero-initialised; idempotent call, atomic so concurrent first launches do not race
    if (!attr_set[at * 2 + bt]) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, GW_LDS);
        attr_set[at * 2 + bt] = true;
    }
    const long nblk = (long)p.tiles_m * p.tiles_n;
    if (nblk > 0x7fffffffL) return LIBRA_ERR_SHAPE;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)p.splitk, (unsigned)(groups < 1 ? 1 : groups)), dim3(GW_THREADS), GW_LDS,
                       (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}
