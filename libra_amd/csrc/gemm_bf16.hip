// bf16 "NT" GEMM for gfx950:  C[M,N] = epilogue( A[M,K] · B[N,K]^T )      (fp32 accumulate)
//
// Both operands are K-contiguous (a torch Linear: x[M,K], weight[N,K]), which is the natural MFMA
// feed: every 32x32x16 fragment is one 16-byte LDS read per lane.  The same kernel serves the
// forward linears, dgrad (with a K-contiguous transposed weight copy) and wgrad (with transposed
// activation / gradient copies) — see libra_amd/ops.py.
//
// Structure (MI355X-first, not a CUDA tiling):
//   * 128x128x64 block tile, 256 threads = 4 wave64, each wave a 64x64 quadrant = 2x2 MFMA 32x32 tiles
//     (64 fp32 accumulators / lane), v_mfma_f32_32x32x16_bf16.
//   * operands go HBM -> LDS with 16-byte direct-to-LDS loads (global_load_lds_dwordx4, no VGPR
//     round trip).  The LDS image is lane-linear, so the bank swizzle is applied to the *source*
//     address and mirrored on the fragment read:  chunk' = chunk ^ ((row >> 1) & 7)  (128-B rows).
//     That makes every ds_read_b128 16-lane group hit 16 distinct 16-B slots (conflict free).
//   * double-buffered LDS (2 x 32 KiB), one barrier per K tile, next tile's loads in flight under the
//     current tile's 16 MFMAs per wave.
//   * block ids are remapped XCD-aware (8 private L2s) + grouped along M so a weight panel stays L2 hot.
//   * epilogue: accumulators -> LDS as a 128x128 fp32 tile (reusing the 64 KiB staging space), then
//     row-contiguous 16-byte bf16 stores with fused bias / column scale / quick-GELU / GELU-grad /
//     residual.
//
// Requirements (checked by the C entry point): K % 64 == 0, row strides % 8 == 0 elements, pointers
// 16-byte aligned.  M and N are arbitrary (tail rows are clamped on load and masked on store).
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include "hip_common.hpp"
#include "gemm_tiles.hpp"
#include "../../include/libra_hip.h"

#ifndef TO128
#define TO128 8, 8, 4, 2
#endif
namespace libra {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int GEMM_THREADS = 256;
constexpr int TILE_BYTES = BM * BK * 2;           // 16 KiB per operand tile
constexpr int GEMM_LDS = 4 * TILE_BYTES;          // A0 B0 A1 B1 = 64 KiB (== 128*128*4 epilogue tile)

struct GemmArgs {
    const bf16_t* A; const bf16_t* B; bf16_t* C;
    const bf16_t* bias; const bf16_t* resid; const bf16_t* aux; bf16_t* preact;
    const int* a_rows; const int* c_rows;     // optional row gather (A, N-type only) / scatter (C, resid, aux, preact)
    long lda, ldb, ldc, ldr, ldaux, ldpre;
    int M, N, K;
    int tiles_m, tiles_n;
    float alpha; int alpha_cols;
    int flags;
    const bf16_t* Ag[3]; const bf16_t* Bg[3]; bf16_t* Cg[3];   // grouped launch: operands of groups 1..3 (blockIdx.z)
};

__device__ __forceinline__ float quick_gelu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x)); }   // (v_rcp_f32: 1 ulp, the result is rounded to bf16)
__device__ __forceinline__ float quick_gelu_grad_f(float x) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x));
    return s * (1.0f + 1.702f * x * (1.0f - s));
}

template <bool AT, bool BT>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_bf16_nt_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- tile coordinates: one compact patch per wave of workgroups, one sub-patch per XCD (hip_common.hpp: tile_order) ----
    const TileRC trc = tile_order<TO128>(blockIdx.x, p.tiles_m, p.tiles_n);
    const int tm = trc.tm, tn = trc.tn;
    const int m0 = tm * BM, n0 = tn * BN;
    const bf16_t* Ap = p.A; const bf16_t* Bp = p.B; bf16_t* Cp = p.C;          // grouped launch: blockIdx.z picks the group
    {   // (constant indices + selects: a dynamically indexed kernel-argument array would be copied to scratch)
        const int g = blockIdx.z;
        if (g == 1) { Ap = p.Ag[0]; Bp = p.Bg[0]; Cp = p.Cg[0]; }
        else if (g == 2) { Ap = p.Ag[1]; Bp = p.Bg[1]; Cp = p.Cg[1]; }
        else if (g == 3) { Ap = p.Ag[2]; Bp = p.Bg[2]; Cp = p.Cg[2]; }
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K / BK;
    // per-lane source offsets of this wave's four 1-KiB pieces of each operand tile
    unsigned srcA[4], srcB[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        srcA[j] = stage_src<AT>(wave * 4 + j, lane, m0, p.M, p.lda, p.a_rows);
        srcB[j] = stage_src<BT>(wave * 4 + j, lane, n0, p.N, p.ldb);
    }
    const long kstepA = ktile_stride<AT>(p.lda), kstepB = ktile_stride<BT>(p.ldb);
    auto stage = [&](int buf, int kt) {
        char* da = smem + buf * 2 * TILE_BYTES + wave * 4096;
        const bf16_t* ga = Ap + kt * kstepA;
        const bf16_t* gb = Bp + kt * kstepB;
        // (wave-uniform K-tile base + per-lane 32-bit byte offset: no 64-bit VALU address per piece)
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16_off(ga, 2u * srcA[j], da + j * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16_off(gb, 2u * srcB[j], da + TILE_BYTES + j * 1024);
    };
    stage(0, 0);
    const FragAddr fa = make_frag_addr(lane);
    const int toA[2] = {frag_toff<AT>(lane, wm * 2), frag_toff<AT>(lane, wm * 2 + 1)};
    const int toB[2] = {frag_toff<BT>(lane, wn * 2), frag_toff<BT>(lane, wn * 2 + 1)};

    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();            // tile kt landed everywhere; everyone is done reading the other buffer
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* sa = smem + cur * 2 * TILE_BYTES;
        const char* sb = sa + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = load_frag<AT>(sa, fa, toA[i], ks);
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[j] = load_frag<BT>(sb, fa, toB[j], ks);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: acc -> LDS fp32 [128][128] -> fused ops -> 16-byte row-contiguous stores ----
    __syncthreads();
    float* ct = (float*)smem;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // 32x32 C/D layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
                const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = wn * 64 + j * 32 + (lane & 31);
                ct[row * BN + col] = acc[i][j][r];
            }
    __syncthreads();

    // Two instantiations of the same epilogue code: INTERIOR (the whole 128x128 tile lies inside C: every per-lane bound test
    // and scalar tail path folds away) and the generic edge version; the choice is wave-uniform.
    auto run = [&](auto interior) {
        constexpr bool IN = decltype(interior)::value;
        const int cgrp = tid & 15;               // 8-column group owned by this thread (fixed across rows)
        const int gn = n0 + cgrp * 8;
        if (!IN && gn >= p.N) return;
        const bool full8 = IN || (gn + 8 <= p.N);
        float bias[8];
    #pragma unroll
        for (int e = 0; e < 8; ++e) bias[e] = 0.f;
        if (p.flags & LIBRA_GEMM_BIAS) {
            if (full8) {
                unpack8(*(const u32x4*)(p.bias + gn), bias);
            } else {
                for (int e = 0; e < 8 && gn + e < p.N; ++e) bias[e] = bf2f(p.bias[gn + e]);
            }
        }
        float cs[8];
    #pragma unroll
        for (int e = 0; e < 8; ++e) cs[e] = (gn + e < p.alpha_cols) ? p.alpha : 1.0f;

        // fused operands (row map, aux, residual) of all 8 row passes are fetched up front: one round trip, not eight
        const bool want_aux = (p.flags & LIBRA_GEMM_MUL_QGELU_GRAD) != 0, want_res = (p.flags & LIBRA_GEMM_RESIDUAL) != 0;
        int oms[8];
        u32x4 xaux[8], xres[8];
    #pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int gm = m0 + pass * 16 + (tid >> 4);
            oms[pass] = ((IN || gm < p.M) && p.c_rows) ? p.c_rows[gm] : gm;
        }
    #pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int gm = m0 + pass * 16 + (tid >> 4);
            const bool ok = IN || (gm < p.M && full8);
            xaux[pass] = u32x4{0, 0, 0, 0}; xres[pass] = u32x4{0, 0, 0, 0};
            if (want_aux && ok) xaux[pass] = *(const u32x4*)(p.aux + (long)oms[pass] * p.ldaux + gn);
            if (want_res && ok) xres[pass] = *(const u32x4*)(p.resid + (long)oms[pass] * p.ldr + gn);
        }
    #pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int row = pass * 16 + (tid >> 4);
            const int gm = m0 + row;
            if (!IN && gm >= p.M) break;
            const int om = oms[pass];
            float v[8];
            const f32x4 lo = *(const f32x4*)(ct + row * BN + cgrp * 8);
            const f32x4 hi = *(const f32x4*)(ct + row * BN + cgrp * 8 + 4);
    #pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
    #pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (v[e] + bias[e]) * cs[e];
            if (p.flags & LIBRA_GEMM_STORE_PREACT) {
                // the reference rounds the Linear output to bf16 before the activation sees it
                bf16_t* pd = p.preact + (long)om * p.ldpre + gn;
                if (full8) *(u32x4*)pd = pack8(v);
                else for (int e = 0; e < 8 && gn + e < p.N; ++e) pd[e] = f2bf(v[e]);
            }
            if (p.flags & LIBRA_GEMM_QUICK_GELU) {
    #pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = quick_gelu_f(bf2f(f2bf(v[e])));
            }
            if (p.flags & LIBRA_GEMM_MUL_QGELU_GRAD) {
                float a[8];
                if (full8) unpack8(xaux[pass], a);
                else for (int e = 0; e < 8; ++e) a[e] = (gn + e < p.N) ? bf2f(p.aux[(long)om * p.ldaux + gn + e]) : 0.f;
    #pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= quick_gelu_grad_f(a[e]);
            }
            if (p.flags & LIBRA_GEMM_RESIDUAL) {
                float a[8];
                if (full8) unpack8(xres[pass], a);
                else for (int e = 0; e < 8; ++e) a[e] = (gn + e < p.N) ? bf2f(p.resid[(long)om * p.ldr + gn + e]) : 0.f;
    #pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += a[e];
            }
            bf16_t* dst = Cp + (long)om * p.ldc + gn;
            if (full8) {
                *(u32x4*)dst = pack8(v);
            } else {
                for (int e = 0; e < 8 && gn + e < p.N; ++e) dst[e] = f2bf(v[e]);
            }
        }
    };
    if (m0 + BM <= p.M && n0 + BN <= p.N) run(std::true_type{});
    else run(std::false_type{});
}

}  // namespace libra

using namespace libra;

extern "C" int libra_gemm256_launch_(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                                     int64_t M, int64_t N, int64_t K, const void* bias, const void* resid,
                                     int64_t ldr, const void* aux, int64_t ldaux, void* preact, int64_t ldpre,
                                     float alpha, int64_t alpha_cols, int flags, float* slab, int splitk,
                                     const int* a_rows, const int* c_rows, void* stream, int groups,
                                     const void* const* Ag, const void* const* Bg, void* const* Cg);

extern "C" int libra_gemm_bf16_nt_routed(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                                         int64_t M, int64_t N, int64_t K, const void* bias, const void* resid,
                                         int64_t ldr, const void* aux, int64_t ldaux, void* preact, int64_t ldpre,
                                         float alpha, int64_t alpha_cols, int flags, const int32_t* a_rows,
                                         int64_t a_phys_rows, const int32_t* c_rows, void* stream);

extern "C" int libra_gemmw_launch_(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                                   int64_t M, int64_t N, int64_t K, const void* bias, const void* resid,
                                   int64_t ldr, const void* aux, int64_t ldaux, void* preact, int64_t ldpre,
                                   float alpha, int64_t alpha_cols, int flags, float* slab, int splitk,
                                   const int* a_rows, const int* c_rows, void* stream, int groups,
                                   const void* const* Ag, const void* const* Bg, void* const* Cg);

// Tile-structure choice (speed only).  Three structures share one contract:
//   256  gemm_bf16_256.hip  256 x 256 tiles, 8 waves, ONE workgroup / CU (256 slots): ~1.5x faster per FLOP than 128 on full waves of
//        workgroups, but a partly filled last wave costs a whole one and nothing overlaps a tile's ~9 us epilogue;
//   W    gemm_bf16_w.hip    256 x 128 tiles, 4 waves, TWO workgroups / CU (512 slots): the same per-wave block as 256, half the
//        quantisation step, one workgroup's epilogue / barrier waits under the other's MFMAs;
//   128  this file          128 x 128 tiles, 4 waves, two workgroups / CU: small problems.
// The plan gives the leading `rows_big` output rows (whole waves of 256^2 tiles) to the 256 kernel and the remaining rows - the
// would-be partial wave - to `rest` as a second launch: e.g. M=18464, N=1024 (ViT-L proj / fc2 at bs 32) = 73 x 4 tiles = 1.14
// waves becomes one full wave + the rest instead of two waves.
// Costs are microseconds fitted on MI355X (tools/gemm_sweep.py, profiles/r04_gemm_tile_fit.txt); kt = K / 64.
enum { KIND_128 = 128, KIND_256 = 256, KIND_W = 1 };
struct TilePlan { int64_t rows_big; int rest; };        // rows [0, rows_big) on the 256 kernel, the others on `rest` (KIND_128 / KIND_W)

static double cost256(double tiles, double kt) { return 8.6 + ceil(tiles / 256.0) * (1.48 * kt + 5.3); }   // (re-fitted round 2: DMA between MFMAs)
static double cost128(double tiles, double kt) {
    const double full = floor(tiles / 512.0), rest = tiles - full * 512.0;
    return full * (5.8 + 1.245 * kt) + (rest <= 0 ? 0.0 : rest <= 256.0 ? 4.8 + 0.606 * kt      // <= 1 block / CU: it owns the CU
                                                                          : 5.8 + 1.245 * kt);
}
// 256 x 128 tiles, two per CU: a round of 512 tiles shares each CU's matrix pipe between two workgroups (CW_PAIR us per K tile for
// the pair: 1.17 PFLOP/s in the loop against the 256^2 kernel's 1.45 - two unsynchronised workgroups do not interleave as well as
// its two staggered wave groups, but one's epilogue does run under the other's MFMAs); a last round of <= 256 tiles has the CUs to
// itself (CW_LONE per K tile).  Fitted on profiles/r04_gemm_experiments/sweep_w_v2.txt: 11760x4096x{4096,11008}, 4624x1024x{4096,
// 11008}, 4624x4096x1024; checked on 11760x22016x4096 (1854 predicted / 1870 us) and 18464x1024x4096 (177 / 166).
constexpr double CW_FIX = 4.7, CW_PAIR = 1.84, CW_LONE = 0.79, CW_EPI = 2.0;
static double costw(double tiles, double kt) {
    const double full = floor(tiles / 512.0), rest = tiles - full * 512.0;
    return CW_FIX + full * (CW_PAIR * kt + CW_EPI) + (rest <= 0 ? 0.0 : rest <= 256.0 ? CW_LONE * kt + CW_EPI : CW_PAIR * kt + CW_EPI);
}
static TilePlan plan_tiles(int64_t M, int64_t N, int64_t K, int64_t groups, int force) {
    if (force == LIBRA_GEMM_TILE_128) return {0, KIND_128};        // caller-chosen structure (libra_gemm_bf16_nt_tile: tests, tools)
    if (force == LIBRA_GEMM_TILE_256) return M >= 256 && N >= 256 ? TilePlan{M, KIND_128} : TilePlan{0, KIND_128};
    if (force == LIBRA_GEMM_TILE_W) return {0, KIND_W};
    if (K < 256) return {0, KIND_128};
    const double kt = (double)K / 64.0;
    const int64_t tm = (M + 255) / 256, tn = (N + 255) / 256 * groups, tn128 = (N + 127) / 128 * groups;   // per tile row, all groups
    auto rest_cost = [&](int64_t rows, int& kind) {           // cheapest two-per-CU structure for `rows` output rows
        const double c128 = cost128((double)((rows + 127) / 128) * tn128, kt), cw = costw((double)((rows + 255) / 256) * tn128, kt);
        kind = cw < c128 ? KIND_W : KIND_128;
        return cw < c128 ? cw : c128;
    };
    TilePlan best{0, KIND_128};
    double bc = rest_cost(M, best.rest);                       // everything on one two-per-CU structure
    if (M >= 256 && N >= 256) {
        const double all256 = cost256((double)(tm * tn), kt);
        if (all256 <= bc) { bc = all256; best = {M, KIND_128}; }
        // j full waves on the big kernel, then the rest.  Only worth a second launch (~10 us of boundary + cold start)
        // when the partial wave is a large share of the problem, i.e. for few waves.
        for (int64_t j = (tm * tn) / 256; j >= 1 && j <= 2; --j) {
            const int64_t r = (256 * j) / tn;
            if (r <= 0 || r >= tm) continue;
            int kind;
            const double c = cost256((double)(r * tn), kt) + rest_cost(M - r * 256, kind) + 10.0;
            if (c < 0.95 * bc) { bc = c; best = {r * 256, kind}; }
        }
    }
    return best;
}

// ---- split-K (wgrad-shaped problems: small M,N, very long K): the 256^2 kernel over K slices + a deterministic
// fp32 slab reduction.  plan() returns the number of slices (1 = not worth it).
extern "C" int libra_gemm_splitk_plan(int64_t M, int64_t N, int64_t K) {
    // (skinny outputs included: the rank-8 bridge weight gradients are [4096, 8] and [64, 4096] with K = 4.6k-11.8k tokens -
    //  as ordinary launches they occupy 16-32 workgroups; sliced over K they fill the chip and run at HBM speed)
    if (M < 8 || N < 8 || K < 4096 || (N % 8)) return 1;          // (M % 8 only matters for a reduction-major A: the entry point checks it)
    const long tiles = ((M + 255) / 256) * ((N + 255) / 256);
    if (tiles > 128) return 1;
    long s = 256 / tiles;
    const long max_s = K / 64 / 8;                               // at least 8 K tiles per slice
    if (s > max_s) s = max_s;
    if (s > 32) s = 32;
    if (s < 2) return 1;
    // a problem that fills most of the chip with 128^2 tiles and has a moderate K runs faster unsplit: the fp32 slabs and their
    // reduction (2 s M N 4 bytes through HBM) cost more than the partial wave ([4096 x 1024] x K 4672: 53 us unsplit, 58 sliced 4x;
    // at K 18496 sliced wins 140 : 180)
    const double kt = (double)K / 64.0, tiles128 = (double)(((M + 127) / 128) * ((N + 127) / 128));
    if (tiles128 >= 192.0) {
        const double split_us = 8.6 + 1.48 * kt / (double)s + 5.3 + 2.0 * (double)s * (double)M * (double)N * 4.0 / 4.0e6;
        if (cost128(tiles128, kt) < 0.8 * split_us) return 1;     // (0.8: the model is of the plain 128^2 launch; with row maps and a
                                                                   //  residual in its epilogue it measured 157 us where the model says 109,
                                                                   //  the sliced launch reads the residual in its reduction stage)
    }
    return (int)s;
}
extern "C" size_t libra_gemm_splitk_workspace_bytes(int64_t M, int64_t N, int64_t splits) {
    return splits > 1 ? (size_t)splits * M * N * sizeof(float) : 0;
}
extern "C" int libra_gemm_bf16_nt_splitk(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                                         int64_t M, int64_t N, int64_t K, int64_t splits, int flags, void* workspace,
                                         size_t workspace_bytes, void* stream) {
    if (M <= 0 || N <= 0) return LIBRA_OK;
    if (!A || !B || !C || K <= 0 || (K % BK) || splits < 2 || splits > K / BK) return LIBRA_ERR_SHAPE;
    if (flags & ~(LIBRA_GEMM_A_T | LIBRA_GEMM_B_T)) return LIBRA_ERR_SHAPE;
    const int at = (flags & LIBRA_GEMM_A_T) ? 1 : 0, bt = (flags & LIBRA_GEMM_B_T) ? 1 : 0;
    if ((lda % 8) || (ldb % 8) || (ldc % 8) || (N % 8) || ldc < N) return LIBRA_ERR_SHAPE;
    if (at ? (lda < M || (M % 8) || K * lda >= (1LL << 31)) : (lda < K || M * lda >= (1LL << 31))) return LIBRA_ERR_SHAPE;
    if (bt ? (ldb < N || K * ldb >= (1LL << 31)) : (ldb < K || N * ldb >= (1LL << 31))) return LIBRA_ERR_SHAPE;
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)workspace) & 15) return LIBRA_ERR_ALIGN;
    if (!workspace || workspace_bytes < libra_gemm_splitk_workspace_bytes(M, N, splits)) return LIBRA_ERR_ALIGN;
    return libra_gemm256_launch_(A, lda, B, ldb, C, ldc, M, N, K, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, 1.0f, 0, flags,
                                 (float*)workspace, (int)splits, nullptr, nullptr, stream, 1, nullptr, nullptr, nullptr);
}

// The routed form: row gather on A (K-contiguous A only), row scatter on C, optional residual read at the scattered row - what the
// decoder's text-stream projections need when a step has FEW text rows (libra_pretrain.yaml:19: 700-token sequences, M = 976 text
// rows per step: 64-172 tiles of 256^2 on 256 CUs unsplit).
extern "C" int libra_gemm_bf16_nt_splitk_routed(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                                                int64_t M, int64_t N, int64_t K, int64_t splits, int flags, const void* resid,
                                                int64_t ldr, const int32_t* a_rows, int64_t a_phys_rows, const int32_t* c_rows,
                                                void* workspace, size_t workspace_bytes, void* stream) {
    if (M <= 0 || N <= 0) return LIBRA_OK;
    if (!A || !B || !C || K <= 0 || (K % BK) || splits < 2 || splits > K / BK) return LIBRA_ERR_SHAPE;
    if (flags & ~(LIBRA_GEMM_B_T | LIBRA_GEMM_RESIDUAL)) return LIBRA_ERR_SHAPE;     // (a gathered A is K-contiguous)
    const int bt = (flags & LIBRA_GEMM_B_T) ? 1 : 0;
    if ((lda % 8) || (ldb % 8) || (ldc % 8) || (N % 8) || ldc < N || lda < K) return LIBRA_ERR_SHAPE;
    if (a_rows && a_phys_rows <= 0) return LIBRA_ERR_SHAPE;
    const int64_t arows = a_rows ? a_phys_rows : M;
    if (arows * lda >= (1LL << 31)) return LIBRA_ERR_SHAPE;
    if (bt ? (ldb < N || K * ldb >= (1LL << 31)) : (ldb < K || N * ldb >= (1LL << 31))) return LIBRA_ERR_SHAPE;
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)workspace) & 15) return LIBRA_ERR_ALIGN;
    if ((flags & LIBRA_GEMM_RESIDUAL) && (!resid || (ldr % 8) || ((uintptr_t)resid & 15))) return LIBRA_ERR_ALIGN;
    if (!workspace || workspace_bytes < libra_gemm_splitk_workspace_bytes(M, N, splits)) return LIBRA_ERR_ALIGN;
    return libra_gemm256_launch_(A, lda, B, ldb, C, ldc, M, N, K, nullptr, resid, ldr, nullptr, 0, nullptr, 0, 1.0f, 0, flags,
                                 (float*)workspace, (int)splits, a_rows, c_rows, stream, 1, nullptr, nullptr, nullptr);
}

extern "C" int libra_gemm_bf16_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                                  int64_t M, int64_t N, int64_t K, const void* bias, const void* resid,
                                  int64_t ldr, const void* aux, int64_t ldaux, void* preact, int64_t ldpre,
                                  float alpha, int64_t alpha_cols, int flags, void* stream) {
    return libra_gemm_bf16_nt_routed(A, lda, B, ldb, C, ldc, M, N, K, bias, resid, ldr, aux, ldaux, preact, ldpre, alpha,
                                     alpha_cols, flags, nullptr, M, nullptr, stream);
}

// Shared body of the plain / routed / grouped entry points.  `groups` problems of identical shape, strides and row maps;
// A/B/C = group 0, Ag/Bg/Cg = groups 1.. (fused epilogue operands only with groups == 1).
extern "C" int libra_gemm_skinny_launch_(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                                         int64_t N, int64_t K, const void* resid, int64_t ldr, const int32_t* a_rows,
                                         const int32_t* c_rows, void* stream);

static int gemm_run(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N,
                    int64_t K, const void* bias, const void* resid, int64_t ldr, const void* aux, int64_t ldaux, void* preact,
                    int64_t ldpre, float alpha, int64_t alpha_cols, int flags, const int32_t* a_rows, int64_t a_phys_rows,
                    const int32_t* c_rows, void* stream, int groups, const void* const* Ag_in, const void* const* Bg_in,
                    void* const* Cg_in, int tile = LIBRA_GEMM_TILE_AUTO) {
    if (M <= 0 || N <= 0) return LIBRA_OK;                       // empty problem: nothing to do
    if (!A || !B || !C || K <= 0 || (K % BK) != 0 || tile < LIBRA_GEMM_TILE_AUTO || tile > LIBRA_GEMM_TILE_W) return LIBRA_ERR_SHAPE;
    const int at = (flags & LIBRA_GEMM_A_T) ? 1 : 0, bt = (flags & LIBRA_GEMM_B_T) ? 1 : 0;
    if ((lda % 8) || (ldb % 8) || ldc < N) return LIBRA_ERR_SHAPE;
    if (a_rows && (at || a_phys_rows <= 0)) return LIBRA_ERR_SHAPE;               // row gather: K-contiguous A only
    const int64_t arows = a_rows ? a_phys_rows : M;
    if (at ? (lda < M || (M % 8) || K * lda >= (1LL << 31)) : (lda < K || arows * lda >= (1LL << 31))) return LIBRA_ERR_SHAPE;
    if (bt ? (ldb < N || (N % 8) || K * ldb >= (1LL << 31)) : (ldb < K || N * ldb >= (1LL << 31))) return LIBRA_ERR_SHAPE;
    if (((uintptr_t)A | (uintptr_t)B) & 15) return LIBRA_ERR_ALIGN;
    const bool vec_ok = (ldc % 8 == 0) && (((uintptr_t)C & 15) == 0);
    if (!vec_ok) return LIBRA_ERR_ALIGN;
    if ((flags & LIBRA_GEMM_BIAS) && (!bias || ((uintptr_t)bias & 15))) return LIBRA_ERR_ALIGN;
    if ((flags & LIBRA_GEMM_RESIDUAL) && (!resid || (ldr % 8) || ((uintptr_t)resid & 15))) return LIBRA_ERR_ALIGN;
    if ((flags & LIBRA_GEMM_MUL_QGELU_GRAD) && (!aux || (ldaux % 8) || ((uintptr_t)aux & 15))) return LIBRA_ERR_ALIGN;
    if ((flags & LIBRA_GEMM_STORE_PREACT) && (!preact || (ldpre % 8) || ldpre < N || ((uintptr_t)preact & 15))) return LIBRA_ERR_ALIGN;
    if (M > (1 << 30) || N > (1 << 30) || K > (1 << 30)) return LIBRA_ERR_SHAPE;
    const void* Ag[3] = {nullptr, nullptr, nullptr}; const void* Bg[3] = {nullptr, nullptr, nullptr};
    void* Cg[3] = {nullptr, nullptr, nullptr};
    for (int g = 0; g + 1 < groups; ++g) {
        Ag[g] = Ag_in[g]; Bg[g] = Bg_in[g]; Cg[g] = Cg_in[g];
        if (!Ag[g] || !Bg[g] || !Cg[g] || (((uintptr_t)Ag[g] | (uintptr_t)Bg[g] | (uintptr_t)Cg[g]) & 15)) return LIBRA_ERR_ALIGN;
    }

    // ---- M <= 16 (the generation step: one token per sequence): weights are read once, HBM-bound -> the skinny kernel ----
    if (groups == 1 && M <= 16 && !at && !bt && (flags & ~LIBRA_GEMM_RESIDUAL) == 0 && alpha_cols == 0 && (K % 8) == 0)
        return libra_gemm_skinny_launch_(A, lda, B, ldb, C, ldc, M, N, K, (flags & LIBRA_GEMM_RESIDUAL) ? resid : nullptr, ldr,
                                         a_rows, c_rows, stream);

    // ---- leading rows on the 256^2 kernel, the rest (if any) on a two-per-CU structure: W here, 128^2 below ----
    const TilePlan plan = plan_tiles(M, N, K, groups, tile);
    const int64_t rows256 = plan.rows_big;
    if (rows256 > 0) {
        const int rc = libra_gemm256_launch_(A, lda, B, ldb, C, ldc, rows256, N, K, bias, resid, ldr, aux, ldaux, preact, ldpre,
                                             alpha, alpha_cols, flags, nullptr, 1, a_rows, c_rows, stream, groups, Ag, Bg, Cg);
        if (rc != LIBRA_OK || rows256 >= M) return rc;
        // remaining output rows [rows256, M): gathered / scattered operands advance their row maps, the others their base
        const int64_t m0 = rows256;
        if (a_rows) a_rows += m0;
        else {
            A = (const bf16_t*)A + (at ? m0 : m0 * lda);
            for (int g = 0; g + 1 < groups; ++g) Ag[g] = (const bf16_t*)Ag[g] + (at ? m0 : m0 * lda);
        }
        if (c_rows) c_rows += m0;
        else {
            C = (bf16_t*)C + m0 * ldc;
            for (int g = 0; g + 1 < groups; ++g) Cg[g] = (bf16_t*)Cg[g] + m0 * ldc;
            if (resid) resid = (const bf16_t*)resid + m0 * ldr;
            if (aux) aux = (const bf16_t*)aux + m0 * ldaux;
            if (preact) preact = (bf16_t*)preact + m0 * ldpre;
        }
        M -= m0;
    }
    if (plan.rest == KIND_W)
        return libra_gemmw_launch_(A, lda, B, ldb, C, ldc, M, N, K, bias, resid, ldr, aux, ldaux, preact, ldpre, alpha, alpha_cols,
                                   flags, nullptr, 1, a_rows, c_rows, stream, groups, Ag, Bg, Cg);
    GemmArgs p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = (bf16_t*)C;
    p.bias = (const bf16_t*)bias; p.resid = (const bf16_t*)resid; p.aux = (const bf16_t*)aux; p.preact = (bf16_t*)preact;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.ldaux = ldaux; p.ldpre = ldpre;
    p.M = (int)M; p.N = (int)N; p.K = (int)K;
    p.tiles_m = (int)((M + BM - 1) / BM); p.tiles_n = (int)((N + BN - 1) / BN);
    p.alpha = alpha; p.alpha_cols = (int)alpha_cols; p.flags = flags;
    p.a_rows = a_rows; p.c_rows = c_rows;
    for (int g = 0; g < 3; ++g) { p.Ag[g] = (const bf16_t*)Ag[g]; p.Bg[g] = (const bf16_t*)Bg[g]; p.Cg[g] = (bf16_t*)Cg[g]; }

    const long nblk = (long)p.tiles_m * p.tiles_n;
    if (nblk > 0x7fffffffL) return LIBRA_ERR_SHAPE;
    void (*kern)(const GemmArgs) = at ? (bt ? gemm_bf16_nt_kernel<true, true> : gemm_bf16_nt_kernel<true, false>)
                                      : (bt ? gemm_bf16_nt_kernel<false, true> : gemm_bf16_nt_kernel<false, false>);
    static std::atomic<bool> attr_set[4];           // zero-initialised; idempotent call, atomic so concurrent first launches do not race
    if (!attr_set[at * 2 + bt]) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        attr_set[at * 2 + bt] = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, 1, (unsigned)groups), dim3(GEMM_THREADS), GEMM_LDS, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}

extern "C" int libra_gemm_bf16_nt_routed(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                                         int64_t M, int64_t N, int64_t K, const void* bias, const void* resid,
                                         int64_t ldr, const void* aux, int64_t ldaux, void* preact, int64_t ldpre,
                                         float alpha, int64_t alpha_cols, int flags, const int32_t* a_rows,
                                         int64_t a_phys_rows, const int32_t* c_rows, void* stream) {
    return gemm_run(A, lda, B, ldb, C, ldc, M, N, K, bias, resid, ldr, aux, ldaux, preact, ldpre, alpha, alpha_cols, flags,
                    a_rows, a_phys_rows, c_rows, stream, 1, nullptr, nullptr, nullptr);
}

extern "C" int libra_gemm_bf16_nt_tile(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                                       int64_t M, int64_t N, int64_t K, const void* bias, const void* resid,
                                       int64_t ldr, const void* aux, int64_t ldaux, void* preact, int64_t ldpre,
                                       float alpha, int64_t alpha_cols, int flags, const int32_t* a_rows,
                                       int64_t a_phys_rows, const int32_t* c_rows, int tile, void* stream) {
    return gemm_run(A, lda, B, ldb, C, ldc, M, N, K, bias, resid, ldr, aux, ldaux, preact, ldpre, alpha, alpha_cols, flags,
                    a_rows, a_phys_rows, c_rows, stream, 1, nullptr, nullptr, nullptr, tile);
}

extern "C" int libra_gemm_bf16_nt_grouped(const void* const* A, int64_t lda, const void* const* B, int64_t ldb, void* const* C,
                                          int64_t ldc, int64_t groups, int64_t M, int64_t N, int64_t K, float alpha,
                                          int64_t alpha_cols, int flags, const int32_t* a_rows, int64_t a_phys_rows,
                                          const int32_t* c_rows, int tile, void* stream) {
    if (groups <= 0) return LIBRA_OK;
    if (groups > 4 || !A || !B || !C) return LIBRA_ERR_SHAPE;
    if (flags & ~(LIBRA_GEMM_A_T | LIBRA_GEMM_B_T)) return LIBRA_ERR_SHAPE;          // no fused epilogue operands in a grouped launch
    return gemm_run(A[0], lda, B[0], ldb, C[0], ldc, M, N, K, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, alpha, alpha_cols,
                    flags, a_rows, a_phys_rows, c_rows, stream, (int)groups, A + 1, B + 1, C + 1, tile);
}
