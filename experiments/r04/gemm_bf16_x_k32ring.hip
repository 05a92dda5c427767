// bf16 "NT" GEMM, 256x256 tile, FOUR waves (one per SIMD), each wave owns 128 x 128 of the output, K staged in 32-deep slices
// through a five-slot LDS ring (gfx950).  Same contract, operand modes, row maps, grouped / split-K launches and fused epilogue
// as gemm_bf16_256.hip (gemm_epilogue.hpp).
//
// Why.  (1) 128 x 128 per wave is the largest block the register file holds - 16 accumulators = 256 registers (the AGPR half),
// fragments in the VGPR half - and needs 32 LDS fragment reads per 64 MFMAs against the 8-wave kernel's 48.  (2) Measured on the
// first version of this kernel (64-deep K tiles, two buffers; tools/r04_experiments/gemm_bf16_x_v0.hip, profiles/
// r04_gemm_experiments/x_anatomy.md): with the LDS reads removed it gains 1-3 %, with the direct-to-LDS staging removed 16-19 % -
// half of that comes back when the staging only ever hits L2.  The staging, not the fragment traffic, is what the main loop waits
// for: bursts of pieces stall the issuing wave past its MFMA (one piece per MFMA: -5 %), and with two 64-KiB buffers a piece has
// only 0.5-1 K tile to arrive.  So: 32-deep slices of 32 KiB, FIVE slots = all 160 KiB of the CU - at any time one slice is being
// read into registers and up to four are landed or in flight (2 K tiles of prefetch distance instead of 0.5-1), and the 8 pieces
// a wave issues per slice are spread one per 4 MFMAs over the whole loop.
//
//   * slice u lives in slot u % 5 as four 8-KiB units A_lo A_hi B_lo B_hi (128 lines x 32 k; N-type: 64-byte rows, 16-byte chunk
//     c of line r at position c ^ ((r >> 2) & 3) - the 16 lanes of a ds_read_b128 service group cover all 64 banks; T-type: the
//     first 32 k-rows of gemm_tiles.hpp's image, same addressing).
//   * phase u (32 MFMAs on fragment set u & 1, registers only): reads set (u+1) & 1 <- slice u+1, issues the 8 pieces of slice
//     u+5 into slot u % 5 (every wave finished reading slice u before the barrier that ended phase u-1).
//     end of phase: lgkmcnt(0) (this wave's reads of slice u+1 retired), vmcnt(24) (its pieces of slice u+2 have landed; slices
//     u+3, u+4, u+5 fly on), s_barrier.
#include <atomic>
#include <type_traits>
#include "hip_common.hpp"
#include "gemm_tiles.hpp"
#include "gemm_epilogue.hpp"

#ifndef TOX
#define TOX 4, 8, 4, 2
#endif
#ifndef XK_STAGGER
#define XK_STAGGER 1
#endif
#ifndef XK_NS
#define XK_NS 5
#endif
namespace libra {

constexpr int XU = 8192;                // one unit: 128 lines x 32 k
constexpr int XSL = 4 * XU;             // one slice: A_lo A_hi B_lo B_hi
constexpr int XNS = XK_NS;              // ring slots
constexpr int GX_LDS = XNS * XSL;       // 160 KiB
constexpr int GX_THREADS = 256;

// ---- 32-deep unit images (see above).  N-type piece pc (0..7) = 16 lines x 64 B: line = 16 pc + lane / 4, LDS position lane % 4.
template <bool T>
__device__ __forceinline__ unsigned stage_src32(int pc, int lane, int line0, int nlines, long ld, const int* __restrict__ rows = nullptr) {
    if constexpr (T) return stage_src<true>(pc, lane, line0, nlines, ld);          // 4 k-rows x 256 B per piece: rows 0..31
    const int r = pc * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((r >> 2) & 3);
    int g = line0 + r;
    g = g < nlines ? g : nlines - 1;                 // clamp the tail (masked at the store)
    if (rows) g = rows[g];                           // routed gather: logical line -> physical row
    return (unsigned)g * (unsigned)ld + c * 8;
}
template <bool T>
__device__ __forceinline__ int frag_toff32(int lane, int t) { if constexpr (T) return frag_toff<true>(lane, t); else return t * 2048; }
// fragment of the 32-line block with offset `toff`, k step ks (0..1) of the slice; koff32[ks] = N-type per-lane byte offset
template <bool T>
__device__ __forceinline__ bf16x8 load_frag32(const char* unit, const FragAddr& f, const int (&koff32)[2], int toff, int ks) {
    if constexpr (T) return load_frag<true>(unit, f, toff, ks);
    else return *(const bf16x8*)(unit + toff + koff32[ks]);
}

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
    const int u0 = 2 * (int)((long)nk_all * blockIdx.y / p.splitk);
    const int nu = 2 * (int)((long)nk_all * (blockIdx.y + 1) / p.splitk);      // this split's 32-deep slices are [u0, nu), both even

    // ---- per-lane source BYTE offsets of this wave's 2 x 1-KiB pieces of every unit type (pieces 2 wave, 2 wave + 1)
    unsigned srcA[2][2], srcB[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            srcA[h][j] = 2u * stage_src32<AT>(wave * 2 + j, lane, m0 + h * 128, p.M, p.lda, p.a_rows);
            srcB[h][j] = 2u * stage_src32<BT>(wave * 2 + j, lane, n0 + h * 128, p.N, p.ldb);
        }
    const long kstepA = AT ? 64 * p.lda : 64, kstepB = BT ? 64 * p.ldb : 64;     // BYTES per slice (32 reduction steps)
    const unsigned lds0 = (unsigned)(unsigned long)(LIBRA_LDS char*)smem;       // (one address-space cast, not one per piece)
    const unsigned ldst = lds0 + (unsigned)(wave * 2048);                        // this wave's 2 pieces inside any unit
    // piece q (0..7) of slice u into slot `slot`: unit q >> 1 (A_lo A_hi B_lo B_hi), piece q & 1 of this wave
    auto piece = [&](const int u, const int slot, const int q) {
#ifdef XK_NODMA       // timing-only build (tools/): no staging after the prologue - results wrong
        if (u >= u0 + XNS) return;
#endif
        const int h = q >> 1, j = q & 1;
        const unsigned dst = ldst + (unsigned)(slot * XSL + h * XU + j * 1024);
        if (h < 2) glds16_off_at((const char*)Ap + u * kstepA, srcA[h][j], dst);
        else glds16_off_at((const char*)Bp + u * kstepB, srcB[h - 2][j], dst);
    };

    const FragAddr fa = make_frag_addr(lane);
    int koff32[2];
    {
        const int l31 = lane & 31, fk = lane >> 5;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) koff32[ks] = l31 * 64 + (((2 * ks + fk) ^ ((l31 >> 2) & 3)) << 4);
    }
    const int aoff = wr * XU, boff = (2 + wc) * XU;
    const int toA[4] = {frag_toff32<AT>(lane, 0), frag_toff32<AT>(lane, 1), frag_toff32<AT>(lane, 2), frag_toff32<AT>(lane, 3)};
    const int toB[4] = {frag_toff32<BT>(lane, 0), frag_toff32<BT>(lane, 1), frag_toff32<BT>(lane, 2), frag_toff32<BT>(lane, 3)};

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment sets: fa_[s][i][kk] / fb_[s][j][kk], s = slice parity, kk = k step of the slice
    bf16x8 fa_[2][4][2], fb_[2][4][2];
    // read number r (0..15) of fragment set s from the slice in slot `slot`, in the order the MFMAs consume them
    auto read = [&](const int s, const int slot, const int r) {
        const char* buf = smem + slot * XSL;
        const int kk = r >> 3, w = r & 7;
        if (w < 4) fa_[s][w][kk] = load_frag32<AT>(buf + aoff, fa, koff32, toA[w], kk);
        else fb_[s][w - 4][kk] = load_frag32<BT>(buf + boff, fa, koff32, toB[w - 4], kk);
    };

    // ---- prologue: the first XNS slices requested, the first two landed, set 0 <- slice u0
    {
        int issued = 0;
#pragma unroll
        for (int t = 0; t < XNS; ++t)
            if (u0 + t < nu) {
#pragma unroll
                for (int q = 0; q < 8; ++q) piece(u0 + t, t, q);
                ++issued;
            }
        if (issued == XNS) { LIBRA_VMCNT_N((XNS - 2) * 8); } else { LIBRA_VMCNT(0); }
    }
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int r = 0; r < 16; ++r) read(0, 0, r);
    LIBRA_LGKMCNT0();
    __builtin_amdgcn_s_barrier();       // every wave has slice u0 in registers: slot 0 may be overwritten

    int sw = 0, sr = 1;                 // slot of the slice in registers (free: takes slice u + XNS) / of the slice read next
    // one phase = one slice; P = its fragment set; STEADY = slice u + XNS exists: no tests between the MFMAs
    auto phase = [&](const int u, auto par, auto steady, auto wtag) {
        constexpr int P = decltype(par)::value;
        constexpr int W = XK_STAGGER ? decltype(wtag)::value : 1;
        constexpr bool STEADY = decltype(steady)::value;
        const bool has_r = STEADY || u + 1 < nu, has_w = STEADY || u + XNS < nu;
#pragma unroll
        for (int n = 0; n < 32; ++n) {
            const int kk = n >> 4, i = (n >> 2) & 3, j = n & 3;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[P][i][kk], fb_[P][j][kk], acc[i][j], 0, 0, 0);
            // after MFMA n: this wave's staging slot is n % 4 == W (one wave of the CU per slot: the vector-memory path takes ~16
            // cycles per 1-KiB piece and four waves at the same slot would stall each other past their MFMAs); the 16 fragment
            // reads take the first 16 of the other 24 slots
            if ((n & 3) == W) { if (has_w) { __builtin_amdgcn_sched_barrier(0); piece(u + XNS, sw, n >> 2); __builtin_amdgcn_sched_barrier(0); } }
            else {
                const int r = n - (n >> 2) - ((n & 3) > W ? 1 : 0);        // index among the non-staging slots
                if (r < 16 && has_r) { __builtin_amdgcn_sched_barrier(0); read(P ^ 1, sr, r); __builtin_amdgcn_sched_barrier(0); }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        LIBRA_LGKMCNT0();               // this wave's reads of slice u+1 have retired ...
        if (STEADY) { LIBRA_VMCNT_N((XNS - 2) * 8); } else { LIBRA_VMCNT(0); }     // ... its pieces of slice u+2 have landed (u+3 .. u+XNS fly on)
#ifndef XK_NOBAR
        __builtin_amdgcn_s_barrier();
#endif
        __builtin_amdgcn_sched_barrier(0);
        sw = sr;
        sr = sr + 1 == XNS ? 0 : sr + 1;
    };
    auto loop = [&](auto wtag) {
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        int u = u0;
        for (; u + 1 + XNS < nu; u += 2) { phase(u, I0{}, std::true_type{}, wtag); phase(u + 1, I1{}, std::true_type{}, wtag); }
        for (; u < nu; u += 2) { phase(u, I0{}, std::false_type{}, wtag); phase(u + 1, I1{}, std::false_type{}, wtag); }
    };
    if (!XK_STAGGER || wave == 0) loop(std::integral_constant<int, 0>{});        // (one copy of the loop per staging slot)
    else if (wave == 1) loop(std::integral_constant<int, 1>{});
    else if (wave == 2) loop(std::integral_constant<int, 2>{});
    else loop(std::integral_constant<int, 3>{});
    __syncthreads();                    // every wave is done with the ring: the epilogue reuses it

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
