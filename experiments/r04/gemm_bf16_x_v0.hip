// bf16 "NT" GEMM, 256x256x64 tile, FOUR waves (one per SIMD), each wave owns 128 x 128 of the output (gfx950).  Same contract,
// operand modes, row maps, grouped / split-K launches and fused epilogue as gemm_bf16_256.hip (gemm_epilogue.hpp).
//
// Why: in the 8-wave kernel a wave's 128 x 64 block costs 24 LDS fragment reads per 32 MFMAs (0.75 per MFMA, LDS port 75 % busy
// under a saturated matrix pipe).  128 x 128 per wave is the largest block the register file holds - 16 accumulators = 256
// registers (the AGPR half), fragments in the VGPR half - and needs 32 reads per 64 MFMAs (0.5): a third less LDS traffic per
// flop.  The price: nothing overlaps a wave's own stalls (no partner wave on its SIMD), so the K loop is software pipelined
// inside the wave - the fragments of the NEXT half K tile are read, and the direct-to-LDS pieces of the K tile after the next
// are issued, one instruction at a time between the MFMAs of the current half.
//
//   * LDS: two K-tile buffers of four 16-KiB half-tiles (A_lo A_hi B_lo B_hi; gemm_tiles.hpp images) = 128 KiB, 1 workgroup / CU.
//   * a K tile = two halves of 32 MFMAs (k steps 0-1 | 2-3), two fragment register sets (16 fragments each).
//       half 0 of tile t: MFMAs on set 0; reads set 1 <- tile t, k steps 2-3.
//       middle:           lgkmcnt(0) (this wave no longer reads buffer t&1), vmcnt(0) (its pieces of tile t+1 have landed), s_barrier.
//       half 1 of tile t: MFMAs on set 1; reads set 0 <- tile t+1, k steps 0-1; issues the 16 pieces of tile t+2 into buffer t&1.
//   * prefetch distance: a piece is issued in half 1 of tile t-1 and waited for in the middle of tile t (0.5 - 1 K tile).
#include <atomic>
#include <type_traits>
#include "hip_common.hpp"
#include "gemm_tiles.hpp"
#include "gemm_epilogue.hpp"

#ifndef XK_SCHED
#define XK_SCHED 1
#endif
#ifndef TOX
#define TOX 4, 8, 4, 2
#endif
namespace libra {

constexpr int XHB = 16384;              // one 128x64 half-tile
constexpr int XKTB = 4 * XHB;           // one K tile: A_lo A_hi B_lo B_hi
constexpr int GX_LDS = 2 * XKTB;        // 128 KiB
constexpr int GX_THREADS = 256;

template <bool AT, bool BT>
__global__ __launch_bounds__(GX_THREADS, 1) void gemm_bf16_nt_x_kernel(const Gemm256Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const TileRC trc = tile_order<TOX>(blockIdx.x, p.tiles_m, p.tiles_n);
    const int m0 = trc.tm * 256, n0 = trc.tn * 256;
    const bf16_t* Ap = p.A; const bf16_t* Bp = p.B; bf16_t* Cp = p.C;
    {   // grouped launch: blockIdx.z picks the group (constant indices + selects: no scratch copy of the argument arrays)
        const int g = blockIdx.z;
        if (g == 1) { Ap = p.Ag[0]; Bp = p.Bg[0]; Cp = p.Cg[0]; }
        else if (g == 2) { Ap = p.Ag[1]; Bp = p.Bg[1]; Cp = p.Cg[1]; }
        else if (g == 3) { Ap = p.Ag[2]; Bp = p.Bg[2]; Cp = p.Cg[2]; }
    }

    const int nk_all = p.K >> 6;
    const int kt0 = (int)((long)nk_all * blockIdx.y / p.splitk);
    const int nk = (int)((long)nk_all * (blockIdx.y + 1) / p.splitk);      // this split's K tiles are [kt0, nk)

    // ---- per-lane source offsets (elements) of this wave's 4 x 1-KiB pieces of every half-tile type
    unsigned srcA[2][4], srcB[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            srcA[h][j] = stage_src<AT>(wave * 4 + j, lane, m0 + h * 128, p.M, p.lda, p.a_rows);
            srcB[h][j] = stage_src<BT>(wave * 4 + j, lane, n0 + h * 128, p.N, p.ldb);
        }
    const long kstepA = ktile_stride<AT>(p.lda), kstepB = ktile_stride<BT>(p.ldb);
    const unsigned lds0 = (unsigned)(unsigned long)(LIBRA_LDS char*)smem;       // (one address-space cast, not one per piece)
    const unsigned ldst = lds0 + (unsigned)(wave * 4096);                        // this wave's 4 pieces inside any half-tile
    // piece q (0..15) of K tile kt: half-tile q >> 2 (A_lo A_hi B_lo B_hi), piece q & 3 of this wave
    auto piece = [&](const int kt, const int q) {
#ifdef XK_NODMA       // timing-only build (tools/): no staging after the prologue - results wrong
        if (kt > kt0 + 1) return;
#endif
        const int h = q >> 2, j = q & 3;
        const unsigned dst = ldst + (unsigned)((kt & 1) * XKTB + h * XHB + j * 1024);
#ifdef XK_SAMEK       // timing-only build: every K tile re-reads the first one's operands (cache hits) - results wrong
        const int ks_ = kt0 + (kt & 1);
#else
        const int ks_ = kt;
#endif
        if (h < 2) glds16_at(Ap + ks_ * kstepA + srcA[h][j], dst);
        else glds16_at(Bp + ks_ * kstepB + srcB[h - 2][j], dst);
    };

    const FragAddr fa = make_frag_addr(lane);
    const int aoff = wr * XHB, boff = (2 + wc) * XHB;
    const int toA[4] = {frag_toff<AT>(lane, 0), frag_toff<AT>(lane, 1), frag_toff<AT>(lane, 2), frag_toff<AT>(lane, 3)};
    const int toB[4] = {frag_toff<BT>(lane, 0), frag_toff<BT>(lane, 1), frag_toff<BT>(lane, 2), frag_toff<BT>(lane, 3)};

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment sets: fa_[s][i][kk] / fb_[s][j][kk], s = half of the K tile (k steps 2 s + kk)
    bf16x8 fa_[2][4][2], fb_[2][4][2];
    // read number r (0..15) of fragment set s from K tile kt, in the order the MFMAs consume them
    auto read = [&](const int s, const int kt, const int r) {
#ifdef XK_NOREAD
        if (kt > kt0) return;
#endif
        const char* buf = smem + (kt & 1) * XKTB;
        const int kk = r >> 3, w = r & 7;
        if (w < 4) fa_[s][w][kk] = load_frag<AT>(buf + aoff, fa, toA[w], 2 * s + kk);
        else fb_[s][w - 4][kk] = load_frag<BT>(buf + boff, fa, toB[w - 4], 2 * s + kk);
    };

    // ---- prologue: K tile kt0 complete, K tile kt0+1 in flight, set 0 read
#pragma unroll
    for (int q = 0; q < 16; ++q) piece(kt0, q);
    if (kt0 + 1 < nk) {
#pragma unroll
        for (int q = 0; q < 16; ++q) piece(kt0 + 1, q);
        LIBRA_VMCNT(16);
    } else {
        LIBRA_VMCNT(0);
    }
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int r = 0; r < 16; ++r) read(0, kt0, r);

    // one K tile; STEADY = K tiles kt+1 and kt+2 exist: no tests between the MFMAs
    auto ktile = [&](const int kt, auto steady) {
        constexpr bool STEADY = decltype(steady)::value;
        const bool has1 = STEADY || kt + 1 < nk, has2 = STEADY || kt + 2 < nk;
        // ================= half 0: k steps 0-1; reads set 1 <- this tile's k steps 2-3 =================
#pragma unroll
        for (int n = 0; n < 32; ++n) {
            const int kk = n >> 4, i = (n >> 2) & 3, j = n & 3;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[0][i][kk], fb_[0][j][kk], acc[i][j], 0, 0, 0);
#if XK_SCHED == 0
            if (!(n & 1)) { __builtin_amdgcn_sched_barrier(0); read(1, kt, n >> 1); __builtin_amdgcn_sched_barrier(0); }
#else
            if (n < 16) { __builtin_amdgcn_sched_barrier(0); read(1, kt, n); __builtin_amdgcn_sched_barrier(0); }
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
        LIBRA_LGKMCNT0();               // set 1 is in registers: this wave no longer reads buffer kt & 1 ...
        LIBRA_VMCNT(0);                 // ... and its pieces of K tile kt+1 have landed
#ifndef XK_NOBAR
        __builtin_amdgcn_s_barrier();   // every wave's: buffer kt & 1 may take K tile kt+2, buffer (kt+1) & 1 may be read
#endif
        __builtin_amdgcn_sched_barrier(0);
        // ================= half 1: k steps 2-3; reads set 0 <- next tile's k steps 0-1; pieces of K tile kt+2 =================
#pragma unroll
        for (int n = 0; n < 32; ++n) {
            const int kk = n >> 4, i = (n >> 2) & 3, j = n & 3;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[1][i][kk], fb_[1][j][kk], acc[i][j], 0, 0, 0);
#if XK_SCHED == 0
            if (!(n & 1)) { if (has1) { __builtin_amdgcn_sched_barrier(0); read(0, kt + 1, n >> 1); __builtin_amdgcn_sched_barrier(0); } }
            else if (has2) { __builtin_amdgcn_sched_barrier(0); piece(kt + 2, n >> 1); __builtin_amdgcn_sched_barrier(0); }
#else
            if (n < 16) { if (has2) { __builtin_amdgcn_sched_barrier(0); piece(kt + 2, n); __builtin_amdgcn_sched_barrier(0); } }
            else if (has1) { __builtin_amdgcn_sched_barrier(0); read(0, kt + 1, n - 16); __builtin_amdgcn_sched_barrier(0); }
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    int kt = kt0;
    for (; kt + 2 < nk; ++kt) ktile(kt, std::true_type{});
    for (; kt < nk; ++kt) ktile(kt, std::false_type{});
    __syncthreads();                    // every wave is done with the K-tile buffers: the epilogue reuses them

    // ---- epilogue (gemm_epilogue.hpp): the wave's 128 x 128 block as two 128 x 64 halves through a private 8 KiB LDS region
    float* ct = (float*)(smem + wave * 8192);
    const int m0w = m0 + wr * 128, n0w = n0 + wc * 128;
    if (m0 + 256 <= p.M && n0 + 256 <= p.N) {
        gemm_wave_epilogue<true, 4, 0>(p, acc, Cp, ct, m0w, n0w, lane);
        gemm_wave_epilogue<true, 4, 2>(p, acc, Cp, ct, m0w, n0w + 64, lane);
    } else {
        gemm_wave_epilogue<false, 4, 0>(p, acc, Cp, ct, m0w, n0w, lane);
        gemm_wave_epilogue<false, 4, 2>(p, acc, Cp, ct, m0w, n0w + 64, lane);
    }
}

}  // namespace libra

using namespace libra;

// Internal launcher (declared in gemm_bf16.hip), same argument list as libra_gemm256_launch_.  Arguments were validated.
// The split-K slab reduction, when there is one, is launched by the caller (libra_gemm256_launch_ owns that kernel).
extern "C" int libra_gemmx_launch_(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
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
    void (*kern)(const Gemm256Args) =
        at ? (bt ? gemm_bf16_nt_x_kernel<true, true> : gemm_bf16_nt_x_kernel<true, false>)
           : (bt ? gemm_bf16_nt_x_kernel<false, true> : gemm_bf16_nt_x_kernel<false, false>);
    static std::atomic<bool> attr_set[4];           // zero-initialised; idempotent call, atomic so concurrent first launches do not race
    if (!attr_set[at * 2 + bt]) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, GX_LDS);
        attr_set[at * 2 + bt] = true;
    }
    const long nblk = (long)p.tiles_m * p.tiles_n;
    if (nblk > 0x7fffffffL) return LIBRA_ERR_SHAPE;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)p.splitk, (unsigned)(groups < 1 ? 1 : groups)), dim3(GX_THREADS), GX_LDS,
                       (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}
