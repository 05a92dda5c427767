// One 256x256 output tile of the bf16 "NT" GEMM (8 waves, 8-phase pipeline): the body shared by gemm_bf16_nt_256_kernel
// (gemm_bf16_256.hip: one problem per launch) and gemm_bf16_multi_kernel (gemm_bf16_multi.hip: a tile list over several problems).
// See gemm_bf16_256.hip for the structure; nothing here depends on blockIdx except the split-K slice `ky` the caller passes.
#pragma once
#include <type_traits>
#include "hip_common.hpp"
#include "gemm_tiles.hpp"
#include "gemm_epilogue.hpp"

#ifndef TO256
#define TO256 4, 8, 4, 2
#endif
namespace libra {

constexpr int HB = 16384;               // one 128x64 half-tile
constexpr int KTB = 4 * HB;             // one K tile: A_lo A_hi B_lo B_hi
constexpr int G256_LDS = 2 * KTB;       // 128 KiB
constexpr int G256_THREADS = 512;

// bid: index of the tile in the problem's tile order; ky / p.splitk: K slice; Ap / Bp / Cp: the (group's) operands.
// Every per-lane constant is derived from an opaque copy of the thread id INSIDE (a persistent caller's tile loop must not
// hoist them: DESIGN rule 5).  Ends with every wave past its epilogue; the caller barriers before the LDS is staged again.
template <bool AT, bool BT>
__device__ __forceinline__ void gemm256_tile(const Gemm256Args& p, const bf16_t* Ap, const bf16_t* Bp, bf16_t* Cp, const int bid,
                                             const int ky, char* smem, const int tid0, const int wave, const int wr, const int wc) {
    int tid = tid0;
    asm volatile("" : "+v"(tid));          // (also in the one-tile form: 17-29 VGPR spills around its K loop become 0-2)
    const int lane = tid & 63;
    const int l31 = lane & 31, fk = lane >> 5;

    const TileRC trc = tile_order<TO256>(bid, p.tiles_m, p.tiles_n);
    const int tm = trc.tm, tn = trc.tn;
    const int m0 = tm * 256, n0 = tn * 256;
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
    const int kt0 = (int)((long)nk_all * ky / p.splitk);
    const int nk = (int)((long)nk_all * (ky + 1) / p.splitk);      // this split's K tiles are [kt0, nk)

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
    if (m0 + 256 <= p.M && n0 + 256 <= p.N) gemm_wave_epilogue<true>(p, acc, Cp, ct, m0 + wr * 128, n0 + wc * 64, lane, ky);
    else gemm_wave_epilogue<false>(p, acc, Cp, ct, m0 + wr * 128, n0 + wc * 64, lane, ky);
}

}  // namespace libra
