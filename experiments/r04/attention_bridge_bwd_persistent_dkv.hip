// Backward of the fused routed-bridge causal attention (gfx950, head_dim 128).
//
//   P_ij   = exp(scale * q_i.k^x_j - L_i)           x = "same" if m_i == m_j else "cross";  L = forward log-sum-exp
//   dV^x_j = sum_{i: x(i,j)=x} P_ij dO_i            dP_ij = dO_i . v^x_j            D_i = dO_i . O_i
//   dS_ij  = P_ij (dP_ij - D_i)                     dQ_i = scale sum_j dS_ij k^x_j  dK^x_j = scale sum_{i: x(i,j)=x} dS_ij q_i
// (the four operand gradients dK_same, dK_cross, dV_same, dV_cross are folded back onto k, kb, v, vb by
//  libra_rope_bridge_bwd).  Deterministic: two passes, no atomics.
//
//   dq pass  : forward-like (lane <-> query, 128 queries / workgroup, 32-key tiles, same variant skipping);
//              K and V tiles are staged once in the reduction-major image and read BOTH ways: 16-byte row reads
//              for S^T = K Q^T and dP^T = V dO^T, LDS transpose reads for dQ^T += K^T dS^T.
//   dkv pass : lane <-> key.  Workgroup = 64 keys, 4 waves = 2 key halves x 2 ROLES: a "dV wave" recomputes P
//              (S = Q K^T) and accumulates dV_same / dV_cross, a "dK wave" recomputes P and dP = dO V^T and
//              accumulates dK_same / dK_cross (two [128 d x 32 keys] accumulators = 128 VGPRs per wave, no
//              first-stage work duplicated for the same output).  The workgroup's K/V operand tiles stay
//              resident in LDS (64 KiB); Q / dO tiles of 32 queries stream through a double buffer in BOTH
//              images (row image for the first-stage A operand, reduction-major image for the transpose reads).
#include <atomic>
#include <type_traits>
#include "hip_common.hpp"
#include "gemm_tiles.hpp"
#include "../../include/libra_hip.h"

namespace libra {

constexpr int D128 = 128;
constexpr float LOG2E = 1.4426950408889634f;

struct BridgeBwdArgs {
    const bf16_t* q; long ldq;
    const bf16_t* k_same; long ldk; const bf16_t* k_cross; long ldkc;
    const bf16_t* v_same; long ldv; const bf16_t* v_cross; long ldvc;
    const bf16_t* dout; long ldo;
    const unsigned char* flag; const int* kv_len;
    const float* lse; float* delta;                    // [B,H,S]; delta = sum_d dO.O is WRITTEN by the dQ pass and read by the dK/dV pass
    const bf16_t* out; const bf16_t* out_lo; long ldout;   // attention output (+ its rounding residual, or null): D = dO . (O + O_lo)
    bf16_t* dq; long lddq;
    bf16_t* dk_same; bf16_t* dk_cross; bf16_t* dv_same; bf16_t* dv_cross; long ldg;   // [B*S, H*128] each
    int B, S, H, n_t;
    float sl2, scale;
    int* err;                                          // sticky device-side error word (or null): bit 0 = a dK/dV P hand-over timed out
};

// Reduction-major image of a [rows][128 d] tile: 256-byte rows, 16-byte chunk c of row r stored at position
// c ^ tswz(r).  tswz mixes (r&3) into the chunk's high bits (what the transpose read ds_read_b64_tr_b16 needs to be
// conflict free) and (r>>2)&3 into its low bits, which also spreads the 16 rows of a ds_read_b128 lane group over
// all 64 banks: one image serves BOTH the row-fragment reads (A/B operand with k = d) and the transposed reads
// (A operand with k = rows).  [With only the (r&3) term, row reads were 4-way conflicted: 65 % of the dQ pass's LDS
// cycles in the round-1 PMC profile.]
__device__ __forceinline__ int tswz(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }

// [64 rows][128 d] image, 16 pieces of 1 KiB over 8 waves; base is wave-uniform, ld_b = row stride in bytes
__device__ __forceinline__ void stage_t64(const bf16_t* __restrict__ base, unsigned ld_b, int row0, int nrows, char* dst,
                                          int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pc = wave * 2 + j;
        const int r = pc * 4 + (lane >> 4);
        const int c = (lane & 15) ^ tswz(r);
        int row = row0 + r; row = row < nrows ? row : nrows - 1;
        glds16_off(base, (unsigned)row * ld_b + (unsigned)(c * 16), dst + pc * 1024);
    }
}

// ================================================================================================
// dQ pass: 8 waves x 32 queries per workgroup, 64-key tiles (two 32-key halves per barrier), per variant one K image
// (row reads for S^T = K Q^T, transposed reads for dQ^T += K^T dS^T) and one V image (row reads for dP^T = V dO^T).
constexpr int DQ_VAR = 32768;                 // K tile 16 KiB + V tile 16 KiB (64 keys)
constexpr int DQ_STAGE_B = 2 * DQ_VAR;        // same + cross
constexpr int DQ_LDS_B = 2 * DQ_STAGE_B + 1024;
constexpr int DQ_BQ = 256;

__global__ __launch_bounds__(512, 1) void bridge_attn_bwd_dq_kernel(const BridgeBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* kmask = (unsigned*)(smem + 2 * DQ_STAGE_B);
    int* qpres = (int*)(kmask + 192);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fk = lane >> 5, l31 = lane & 31;
    const int nblk = p.B * p.H * p.n_t;
    const int L = xcd_remap(blockIdx.x, nblk);
    const int qt = p.n_t - 1 - (L % p.n_t);
    const int bh = L / p.n_t;
    const int h = bh % p.H, b = bh / p.H;
    const int S = p.S;
    const long tok0 = (long)b * S;
    const int len = p.kv_len ? p.kv_len[b] : S;
    const int q0w = qt * DQ_BQ + wave * 32;
    const bool active = q0w < S;
    int q = q0w + l31;
    const bool qin = q < S;
    q = qin ? q : S - 1;

    // every per-lane global operand of the prologue is requested before the first wait (one round trip instead of three: with one
    // workgroup per CU nothing else covers them): modality byte, Q and dO fragments, L and D of this lane's query
    // (the first K / V tile goes out first of all, both variants - see bridge_attn_fwd_kernel)
    {
        stage_t64(p.k_same + tok0 * p.ldk + h * D128, (unsigned)p.ldk * 2u, 0, S, smem, wave, lane);
        stage_t64(p.v_same + tok0 * p.ldv + h * D128, (unsigned)p.ldv * 2u, 0, S, smem + 16384, wave, lane);
        stage_t64(p.k_cross + tok0 * p.ldkc + h * D128, (unsigned)p.ldkc * 2u, 0, S, smem + DQ_VAR, wave, lane);
        stage_t64(p.v_cross + tok0 * p.ldvc + h * D128, (unsigned)p.ldvc * 2u, 0, S, smem + DQ_VAR + 16384, wave, lane);
    }
    const int q_vis_raw = p.flag[tok0 + q];
    bf16x8 qf[8], dof[8];
    {
        const bf16_t* qp = p.q + (tok0 + q) * p.ldq + h * D128 + fk * 8;
        const bf16_t* dp = p.dout + (tok0 + q) * p.ldo + h * D128 + fk * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) { qf[ks] = *(const bf16x8*)(qp + ks * 16); dof[ks] = *(const bf16x8*)(dp + ks * 16); }
    }
    const long sidx = ((long)b * p.H + h) * S + q;
    float nLq2 = -p.lse[sidx] * LOG2E;
    // D = sum_d dO . O of this lane's query (the softmax-backward row term) from the dO fragments already in flight + the O row:
    // this lane's 64 channels here, the other half one permlane swap away - the separate delta pass over dO, O and O_lo is gone;
    // the dK / dV pass reads what the fk = 0 lanes store below
    bf16x8 of[8], ol[8];
    {
        const bf16_t* op = p.out + (tok0 + q) * p.ldout + h * D128 + fk * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) of[ks] = *(const bf16x8*)(op + ks * 16);
        if (p.out_lo) {
            const bf16_t* lp = p.out_lo + (tok0 + q) * p.ldout + h * D128 + fk * 8;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) ol[ks] = *(const bf16x8*)(lp + ks * 16);
        }
    }
    modality_masks(p.flag + tok0, S, kmask, tid, 512);
    float Dq = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float o = bf2f((bf16_t)of[ks][e]);
            if (p.out_lo) o += bf2f((bf16_t)ol[ks][e]);
            Dq = __builtin_fmaf(o, bf2f((bf16_t)dof[ks][e]), Dq);
        }
    Dq = half_swap_sum(Dq);
    if (fk == 0 && qin) p.delta[sidx] = Dq;
    const bool q_vis = q_vis_raw != 0;
    if (tid < 2) qpres[tid] = 0;
    __syncthreads();
    if (__ballot(qin && fk == 0 && q_vis)) if (lane == 0) atomicOr(&qpres[1], 1);
    if (__ballot(qin && fk == 0 && !q_vis)) if (lane == 0) atomicOr(&qpres[0], 1);
    __syncthreads();
    const bool blkL = __builtin_amdgcn_readfirstlane(qpres[0]) != 0, blkV = __builtin_amdgcn_readfirstlane(qpres[1]) != 0;
    const bool wV = __ballot(q_vis && qin) != 0, wL = __ballot(!q_vis && qin) != 0;

#pragma unroll
    for (int ks = 0; ks < 8; ++ks) { pin(qf[ks]); pin(dof[ks]); }   // prologue loads have landed before any LDS-DMA is in flight
    pin(nLq2); pin(Dq);
    const bf16_t* ks_base = p.k_same + tok0 * p.ldk + h * D128;
    const bf16_t* kc_base = p.k_cross + tok0 * p.ldkc + h * D128;
    const bf16_t* vs_base = p.v_same + tok0 * p.ldv + h * D128;
    const bf16_t* vc_base = p.v_cross + tok0 * p.ldvc + h * D128;

    f32x16 dq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;

    int kend = (qt + 1) * DQ_BQ; kend = kend < S ? kend : S;
    const int nkt = (kend + 63) / 64;
    // modality content of `n` (32 or 64) keys starting at mask word w0, valid keys only (wave-uniform by construction)
    auto key_mods = [&](int w0, int n, bool& kV, bool& kL) {
        unsigned long long m = (unsigned)__builtin_amdgcn_readfirstlane((int)kmask[w0]);
        if (n == 64) m |= (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)kmask[w0 + 1]) << 32;
        int nvalid = S - w0 * 32; nvalid = nvalid > n ? n : nvalid;
        const unsigned long long full = nvalid >= 64 ? ~0ull : ((1ull << nvalid) - 1ull);
        kV = (m & full) != 0; kL = ((~m) & full) != 0;
    };
    auto stage = [&](int buf, int t) {
        bool kV, kL;
        key_mods(2 * t, 64, kV, kL);
        char* dst = smem + buf * DQ_STAGE_B;
        if ((blkL && kL) || (blkV && kV)) {
            stage_t64(ks_base, (unsigned)p.ldk * 2u, t * 64, S, dst, wave, lane);
            stage_t64(vs_base, (unsigned)p.ldv * 2u, t * 64, S, dst + 16384, wave, lane);
        }
        if ((blkL && kV) || (blkV && kL)) {
            stage_t64(kc_base, (unsigned)p.ldkc * 2u, t * 64, S, dst + DQ_VAR, wave, lane);
            stage_t64(vc_base, (unsigned)p.ldvc * 2u, t * 64, S, dst + DQ_VAR + 16384, wave, lane);
        }
    };
    int xr = 0, xt0 = 0, xt1 = 0;
    {
        const int pp = lane & 15, g16 = (lane >> 4) & 1;
        const int r1 = 4 * fk + (pp >> 2), lp = 2 * g16 + ((pp & 3) >> 1);
        xr = l31 * 256 + ((fk ^ tswz(l31)) << 4);
        xt0 = r1 * 256 + ((pp & 1) << 3) + ((lp ^ tswz(r1)) << 4);
        xt1 = r1 * 256 + ((pp & 1) << 3) + ((lp ^ tswz(r1 + 8)) << 4);
    }

    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int cur = kt & 1;
        if (kt + 1 < nkt) stage(cur ^ 1, kt + 1);
        if (!active) continue;
        const int kv0 = kt * 64;
        if (kv0 > q0w + 31) continue;
        asm volatile("" : "+v"(xr), "+v"(xt0), "+v"(xt1));
        auto rd_row = [&](const char* tile, int ks) -> bf16x8 {
            return *(const bf16x8*)(tile + (xr ^ (ks << 5)));
        };
        auto rd_tr = [&](const char* tile, int dt, int sx) -> bf16x8 {
            union { bf16x8 v; s16x4 h2[2]; } u;
            u.h2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(tile + sx * 4096 + (xt0 ^ (dt << 6))));
            u.h2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(tile + sx * 4096 + 2048 + (xt1 ^ (dt << 6))));
            return u.v;
        };
#pragma unroll 1
        for (int kh = 0; kh < 2; ++kh) {
            const int k0 = kv0 + kh * 32;
            if (k0 > q0w + 31 || k0 >= S) break;
            bool hV, hL;
            key_mods(2 * kt + kh, 32, hV, hL);
            const bool hsame = (wL && hL) || (wV && hV);
            const bool hcross = (wL && hV) || (wV && hL);
            const bool mixed = hsame && hcross;
            const char* skc = smem + cur * DQ_STAGE_B + DQ_VAR + kh * 8192;     // cross variant: K rows of this half (V at +16384)
            const char* img1 = hsame ? skc - DQ_VAR : skc;                      // primary variant
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_row(img1, ks), qf[ks], s, 0, 0, 0);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_row(img1 + 16384, ks), dof[ks], dp, 0, 0, 0);
            unsigned crossbits = 0;
            if (mixed) {                                            // both variants present: per-element select
                f32x16 t, u;
#pragma unroll
                for (int r = 0; r < 16; ++r) { t[r] = 0.f; u[r] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_row(skc, ks), qf[ks], t, 0, 0, 0);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) u = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_row(skc + 16384, ks), dof[ks], u, 0, 0, 0);
                const unsigned km = (unsigned)__builtin_amdgcn_readfirstlane((int)kmask[2 * kt + kh]);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kl = (r & 3) + 8 * (r >> 2) + 4 * fk;
                    const bool cr = (((km >> kl) & 1u) != 0) != q_vis;
                    s[r] = cr ? t[r] : s[r];
                    dp[r] = cr ? u[r] : dp[r];
                    crossbits |= (cr ? 1u : 0u) << r;
                }
            }
            // P = exp2(S*sl2 - L) (recomputed), dS^T = P (dP - D)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], p.sl2, nLq2));
            if (k0 + 31 > q0w || k0 + 32 > len) {                   // causal diagonal / padded keys inside this half
                const int qabs = q0w + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                    s[r] = (key <= qabs && key < len) ? s[r] : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] *= dp[r] - Dq;
#pragma unroll
            for (int sx = 0; sx < 2; ++sx) {
                union { bf16x8 v; unsigned u[4]; } pk, pk2;
#pragma unroll
                for (int j = 0; j < 4; ++j) pk.u[j] = pack2bf(s[8 * sx + 2 * j], s[8 * sx + 2 * j + 1]);
                if (mixed) {                                        // split dS by variant (bf16 pair masks)
                    const unsigned cr = crossbits >> (8 * sx);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned keep2 = (((cr >> (2 * j)) & 1u) ? 0xffffu : 0u) | (((cr >> (2 * j + 1)) & 1u) ? 0xffff0000u : 0u);
                        pk2.u[j] = pk.u[j] & keep2;
                        pk.u[j] &= ~keep2;
                    }
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_tr(img1, dt, sx), pk.v, dq[dt], 0, 0, 0);
                if (mixed) {
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_tr(skc, dt, sx), pk2.v, dq[dt], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();
    constexpr int OROW = 264;
    char* so = smem + wave * (32 * OROW);
    if (active) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = dt * 32 + 8 * g + 4 * fk;
                u32x2 w;
                w[0] = pack2bf(dq[dt][4 * g + 0] * p.scale, dq[dt][4 * g + 1] * p.scale);
                w[1] = pack2bf(dq[dt][4 * g + 2] * p.scale, dq[dt][4 * g + 3] * p.scale);
                *(u32x2*)(so + l31 * OROW + d * 2) = w;
            }
    }
    __syncthreads();
    if (active) {
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int r = pass * 4 + (lane >> 4);
            const int qq = q0w + r;
            if (qq < S) {
                const char* src = so + r * OROW + (lane & 15) * 16;
                const u32x2 a = *(const u32x2*)src;
                const u32x2 c2 = *(const u32x2*)(src + 8);
                u32x4 v;
                v[0] = a[0]; v[1] = a[1]; v[2] = c2[0]; v[3] = c2[1];
                *(u32x4*)(p.dq + (tok0 + qq) * p.lddq + h * D128 + (lane & 15) * 8) = v;
            }
        }
    }
}

// ================================================================================================
// dK / dV pass.  One workgroup owns 64 keys of one (sequence, head): their four operand tiles (K/V, same/cross) stay
// resident in LDS; 64-query tiles of Q and dO stream through a 2-deep ring (one reduction-major image each: row reads
// for S = Q K^T / dP = dO V^T, transposed reads for dV^T += dO^T P / dK^T += Q^T dS).  8 waves = 2 roles (waves 0-3
// accumulate dV, waves 4-7 dK: a workgroup's waves w and w+4 share a SIMD, so every SIMD carries one of each) x 2 key
// sub-blocks of 32 x 2 query halves of the streamed tile; the two query halves' partial sums meet in LDS at the end.
constexpr int KV_RES = 4 * 16384;             // resident K_same, K_cross, V_same, V_cross: [64 keys][128 d] each
constexpr int QD_STAGE = 2 * 16384 + 512;     // Q image, dO image (64 queries each), L[64], D[64]
// + the P hand-over slots of the four (dV wave, dK wave) pairs (2 slots x 2 KiB each) and their sequence words
constexpr int DKV_XP = KV_RES + 2 * QD_STAGE + 1024;
constexpr int DKV_LDS_B = DKV_XP + 4 * 2 * 2048 + 64;

// resident operand tile: two N-type [64 keys][64 d] sub-tiles (128-byte rows, chunk ^ ((row>>1)&7)), 16 KiB; 8 waves
__device__ __forceinline__ void stage_res64(const bf16_t* __restrict__ base, unsigned ld_b, int key0, int S, char* dst, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pc = wave * 2 + j;                 // 16 pieces of 1 KiB: sub-tile pc>>3, rows 8*(pc&7)..
        const int sub = pc >> 3, r = (pc & 7) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int key = key0 + r; key = key < S ? key : S - 1;
        glds16_off(base, (unsigned)key * ld_b + (unsigned)(sub * 128 + c * 16), dst + pc * 1024);
    }
}

// Lane-constant LDS addressing: every fragment address is a per-lane constant XOR a compile-time constant (one VALU op per
// read) instead of the swizzle arithmetic rebuilt per read (the round-1 PMC profile counted 12.4 VALU per MFMA in this kernel).
// The dV wave and the dK wave of a (key sub-block, query half) pair sit on the same SIMD and used to compute the SAME
// S = Q K^T block each (40 MFMAs per 32 x 32 block pair for 32 of arithmetic, and the dK wave - S, dP, dK - was the long pole of
// every iteration; round 2, A/B in profiles/r03_attn_dkv_shared_p_ab.txt).  Now the dV wave alone forms P (exp2, masks, variant select), hands the bf16-packed block to its partner
// through LDS (2 KiB, a sequence word; only the two waves of the pair synchronise - their control flow is identical - the
// workgroup barrier at the loop top covers slot reuse) and the dK wave computes dP = dO V^T meanwhile: 16 MFMAs per wave and
// iteration on both sides.  dS = bf16(P) (dP - D): P enters in bf16, as it does in the reference (softmax(..).to(q.dtype)).
__global__ __launch_bounds__(512, 1) void bridge_attn_bwd_dkv_kernel(const BridgeBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* res = smem;                                            // Ks, Kc, Vs, Vc
    char* qd = smem + KV_RES;
    unsigned* qmask = (unsigned*)(smem + KV_RES + 2 * QD_STAGE); // per 32 queries: bit i = query i is a vision token
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kw = wave & 1, qh = (wave >> 1) & 1;
    const bool role_dk = (wave >> 2) != 0;                       // waves 0-3: dV; waves 4-7: dK
    const int fk = lane >> 5, l31 = lane & 31;
    const int S = p.S;
    const int nqt = (S + 63) / 64;
    // ---- persistent workgroup: one per CU, walking (sequence, head, key block) items.  Round 3 measured 11 us of every one of the
    // 8 192 workgroups of a launch (0.35 of 1.25 ms) as prologue + epilogue that nothing overlapped - a 147-KiB workgroup owns its CU,
    // so dispatch, the resident tiles' round trip, the query-half exchange and the stores were serial, 32 times per CU.  Now the
    // NEXT item's four resident tiles are requested as soon as the query loop of the current item has ended (their LDS region is
    // free then: the exchange and the stores run in the streamed-tile region), and nothing is dispatched between items.
    // Item order (speed only): XCD x = block % 8 owns the (sequence, head) pairs bh = x (mod 8) and its workgroups walk them in
    // bh-major order, so the 32 CUs of an XCD stream the SAME Q / dO tiles at about the same time (one L2 fetch instead of 32);
    // the key block is rotated by 5 per pair so that every workgroup sees light and heavy key blocks alike.
    const int G8 = (int)gridDim.x >> 3;                          // workgroups per XCD (gridDim.x is a multiple of 8)
    const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    const int BH = p.B * p.H;
    const int bh_per_xcd = (BH + 7) >> 3;
    const int n_items = bh_per_xcd * p.n_t;                      // positions in this XCD's stream (some may map to bh >= BH)
    auto item_at = [&](int u, int& bh, int& ktile) -> bool {     // position u of this XCD's stream -> item; false = no such item
        if (u >= n_items) return false;
        const int bl = u / p.n_t;
        bh = xcd + 8 * bl;
        ktile = (u - bl * p.n_t + 5 * bl) % p.n_t;
        return bh < BH;
    };
    auto issue_res = [&](int bh, int ktile) {                    // the four resident operand tiles of an item (LDS-DMA, 8 pieces per wave)
        const int h = bh % p.H, b = bh / p.H;
        const long tok0 = (long)b * S;
        stage_res64(p.k_same + tok0 * p.ldk + h * D128, (unsigned)p.ldk * 2u, ktile * 64, S, res, wave, lane);
        stage_res64(p.k_cross + tok0 * p.ldkc + h * D128, (unsigned)p.ldkc * 2u, ktile * 64, S, res + 16384, wave, lane);
        stage_res64(p.v_same + tok0 * p.ldv + h * D128, (unsigned)p.ldv * 2u, ktile * 64, S, res + 32768, wave, lane);
        stage_res64(p.v_cross + tok0 * p.ldvc + h * D128, (unsigned)p.ldvc * 2u, ktile * 64, S, res + 49152, wave, lane);
    };
    // lane constants: row image (xr), resident image (xv), transposed reads (xt0 / xt1) - each read is then
    // `constant ^ (k-step or d-tile bits)`
    int xr = 0, xv = 0, xt0 = 0, xt1 = 0;
    {
        const int pp = lane & 15, g16 = (lane >> 4) & 1;
        const int r1 = 4 * fk + (pp >> 2), lp = 2 * g16 + ((pp & 3) >> 1);
        xr = l31 * 256 + ((fk ^ tswz(l31)) << 4);
        xv = l31 * 128 + ((fk ^ ((l31 >> 1) & 7)) << 4);
        xt0 = r1 * 256 + ((pp & 1) << 3) + ((lp ^ tswz(r1)) << 4);
        xt1 = r1 * 256 + ((pp & 1) << 3) + ((lp ^ tswz(r1 + 8)) << 4);
    }
    // this pair's two P slots and its sequence word (the number of P blocks published so far in the current item)
    char* xp = smem + DKV_XP + (qh * 2 + kw) * 4096;
    // (an LDS-space pointer: through a generic `volatile int*` the poll compiled to `flat_load_dword .. sc0 sc1` + `s_waitcnt vmcnt(0)`,
    //  which also drained the next tile's direct-to-LDS queue in every iteration of the dK waves)
    volatile LIBRA_LDS int* xseq = (volatile LIBRA_LDS int*)(LIBRA_LDS char*)(smem + DKV_XP + 4 * 4096) + (qh * 2 + kw);

    int u = slot, bh = 0, ktile = 0;
    bool have = false;
    for (; u < n_items && !(have = item_at(u, bh, ktile)); u += G8) {}
    if (!have) return;
    issue_res(bh, ktile);
    int mask_b = -1;                                             // the sequence whose query-modality masks sit in `qmask`
  for (;;) {                                                     // ---- one item per trip
    const int h = bh % p.H, b = bh / p.H;
    const long tok0 = (long)b * S;
    int len = p.kv_len ? p.kv_len[b] : S;
    len = len < S ? len : S;
    const int key0 = ktile * 64;
    const int kbase_w = key0 + kw * 32;
    int key = kbase_w + l31;
    const bool kin = key < S;
    key = kin ? key : S - 1;
    // One memory round trip for the item's prologue: this lane's key modality byte and the first Q / dO tile (LDS-DMA) are requested
    // first, the mask pass's own flag loads last - its wait then covers everything (the resident tiles are already on their way)
    int k_vis_i = p.flag[tok0 + key] != 0;
    const bf16_t* qbase = p.q + tok0 * p.ldq + h * D128;
    const bf16_t* dobase = p.dout + tok0 * p.ldo + h * D128;
    const float* lbase = p.lse + ((long)b * p.H + h) * S;
    const float* dbase = p.delta + ((long)b * p.H + h) * S;
    auto stage_q = [&](int buf, int t) {
        char* dst = qd + buf * QD_STAGE;
        stage_t64(qbase, (unsigned)p.ldq * 2u, t * 64, S, dst, wave, lane);
        stage_t64(dobase, (unsigned)p.ldo * 2u, t * 64, S, dst + 16384, wave, lane);
        if (wave < 2) {                                          // 64 fp32 each: one 4-byte direct-to-LDS op
            int qi = t * 64 + lane; qi = qi < S ? qi : S - 1;
            glds4((wave == 0 ? lbase : dbase) + qi, dst + 32768 + wave * 256);
        }
    };
    const int it0 = key0 / 64;                                   // first query tile that can see this key block
    int npass = 0;
    if (tid < 16) ((int*)(smem + DKV_XP + 4 * 4096))[tid] = 0;
    f32x16 acc_s[4], acc_c[4];                                   // dV (or dK) for the same / cross variant, [128 d x 32 keys]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_s[i][r] = 0.f; acc_c[i][r] = 0.f; }
    if (it0 < nqt) stage_q(0, it0);
    if (b != mask_b) {                                           // (the masks of a sequence serve all its heads and key blocks)
        modality_masks(p.flag + tok0, S, qmask, tid, 512);
        mask_b = b;
    }
    pin(k_vis_i);
    const bool k_vis = k_vis_i != 0;
    const bool wkV = __ballot(k_vis && kin) != 0, wkL = __ballot(!k_vis && kin) != 0;

    const char* rK = res + kw * 32 * 128;                         // this wave's 32 key rows inside each 64-row sub-tile
    for (int it = it0; it < nqt; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int cur = (it - it0) & 1;
        if (it + 1 < nqt) stage_q(cur ^ 1, it + 1);
        const int q0 = it * 64 + qh * 32;
        if (kbase_w >= S || q0 >= S || q0 + 31 < kbase_w) continue;   // no (query >= key) pair for this wave in the tile
        asm volatile("" : "+v"(xr), "+v"(xt0), "+v"(xt1), "+v"(xv));
        auto rd_row = [&](const char* tile, int ks) -> bf16x8 {
            return *(const bf16x8*)(tile + (xr ^ (ks << 5)));
        };
        auto rd_res = [&](const char* tile, int ks) -> bf16x8 {
            return *(const bf16x8*)(tile + (ks >> 2) * 8192 + (xv ^ ((ks & 3) << 5)));
        };
        auto rd_tr = [&](const char* tile, int dt, int sx) -> bf16x8 {
            union { bf16x8 v; s16x4 h2[2]; } u;
            u.h2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(tile + sx * 4096 + (xt0 ^ (dt << 6))));
            u.h2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(tile + sx * 4096 + 2048 + (xt1 ^ (dt << 6))));
            return u.v;
        };
        const char* sq = qd + cur * QD_STAGE + qh * 8192;         // this wave's 32 query rows of the Q image (dO at +16384)
        const float* sL = (const float*)(qd + cur * QD_STAGE + 32768) + qh * 32;
        const float* sD = sL + 64;
        const unsigned qm = (unsigned)__builtin_amdgcn_readfirstlane((int)qmask[2 * it + qh]);
        int nvalid = S - q0; nvalid = nvalid > 32 ? 32 : nvalid;
        const unsigned full = nvalid >= 32 ? 0xffffffffu : ((1u << nvalid) - 1u);
        const bool qV = (qm & full) != 0, qL = ((~qm) & full) != 0;
        const bool wsame = (qL && wkL) || (qV && wkV);
        const bool wcross = (qL && wkV) || (qV && wkL);
        const bool masked = q0 < kbase_w + 31 || q0 + 32 > S || kbase_w + 32 > len;

        // accumulator row r <-> query q0 + (r&3) + 8(r>>2) + 4fk ; column <-> this lane's key
        // S = Q K^T (both roles), dP = dO V^T (dK waves only): A = row fragments of the streamed tile, B = resident fragments
        auto score_s = [&](const char* rk, f32x16& s) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_row(sq, ks), rd_res(rk, ks), s, 0, 0, 0);
        };
        auto score_dp = [&](const char* rk, f32x16& dp) {
#pragma unroll
            for (int r = 0; r < 16; ++r) dp[r] = 0.f;
            if (role_dk) {
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_row(sq + 16384, ks), rd_res(rk + 32768, ks), dp, 0, 0, 0);
            }
        };
        // s <- P = exp2(S*sl2 - L), masked (dV waves; the dK waves apply (dP - D) to the bf16 P they are handed)
        auto finish = [&](f32x16& s) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ql = 8 * g + 4 * fk;
                const f32x4 Lv = *(const f32x4*)(sL + ql);
#pragma unroll
                for (int e = 0; e < 4; ++e) s[4 * g + e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[4 * g + e], p.sl2, -Lv[e] * LOG2E));
            }
            if (masked) {
                const int kabs = kbase_w + l31;
                const bool kok = kabs < len;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qa = q0 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                    s[r] = (qa >= kabs && qa < S && kok) ? s[r] : 0.f;
                }
            }
        };
        const char* st = role_dk ? sq : sq + 16384;               // Q^T fragments (dK) or dO^T fragments (dV)
        const bool mixed = wsame && wcross;
        // one pass per variant present (a tile pair with both modalities on either side - rare - pays S twice): every
        // accumulator set is touched from exactly one place, which keeps all 128 of them in registers
        auto pass = [&](const char* rk, bool cross, f32x16* acc) {
            union { bf16x8 v; unsigned u[4]; } pk[2];
            ++npass;
            char* slot = xp + (npass & 1) * 2048 + lane * 16;
            if (!role_dk) {                                       // producer: P (masked, variant-selected), bf16
                f32x16 s;
                score_s(rk, s);
                finish(s);
                if (mixed) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ql = (r & 3) + 8 * (r >> 2) + 4 * fk;
                        s[r] = ((((qm >> ql) & 1u) != 0) != k_vis) == cross ? s[r] : 0.f;
                    }
                }
#pragma unroll
                for (int sx = 0; sx < 2; ++sx)
#pragma unroll
                    for (int j = 0; j < 4; ++j) pk[sx].u[j] = pack2bf(s[8 * sx + 2 * j], s[8 * sx + 2 * j + 1]);
                *(bf16x8*)slot = pk[0].v;
                *(bf16x8*)(slot + 1024) = pk[1].v;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) *xseq = npass;                     // (LDS serves one wave's operations in order: data, then the word)
            } else {                                              // consumer: dP while P is being formed, then dS = P (dP - D)
                f32x16 dp;
                score_dp(rk, dp);
                // Bounded wait (a lost partner must not hang the GPU).  The two waves of a pair run the same control flow on the
                // same wave-uniform conditions, so the bound is never reached by design; if it ever is, the cold branch raises the
                // sticky error word of the launch (the host checks it once per backward) instead of silently using a stale P.
                int spins = 0;
                while (*xseq < npass) {
                    if (++spins >= (1 << 22)) {
                        if (lane == 0 && p.err) atomicOr(p.err, 1);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                pk[0].v = *(const bf16x8*)slot;
                pk[1].v = *(const bf16x8*)(slot + 1024);
#pragma unroll
                for (int sx = 0; sx < 2; ++sx)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r0 = 8 * sx + 2 * j;             // accumulator rows r0, r0 + 1 <-> queries 8 (r0 >> 2) + 4 fk + (r0 & 3), + 1
                        const float d0 = sD[8 * (r0 >> 2) + 4 * fk + (r0 & 3)], d1 = sD[8 * (r0 >> 2) + 4 * fk + (r0 & 3) + 1];
                        const float p0 = __uint_as_float(pk[sx].u[j] << 16), p1 = __uint_as_float(pk[sx].u[j] & 0xffff0000u);
                        pk[sx].u[j] = pack2bf(p0 * (dp[r0] - d0), p1 * (dp[r0 + 1] - d1));
                    }
            }
#pragma unroll
            for (int sx = 0; sx < 2; ++sx) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_tr(st, dt, sx), pk[sx].v, acc[dt], 0, 0, 0);
            }
        };
        if (wsame) pass(rK, false, acc_s);
        if (wcross) pass(rK + 16384, true, acc_c);
    }
    // ---- the item's query loop is over: nobody reads the resident tiles any more -> the NEXT item's go out now, under the exchange
    // and the stores of this one
    __syncthreads();
    int nbh = 0, nkt = 0;
    bool more = false;
    for (u += G8; u < n_items && !(more = item_at(u, nbh, nkt)); u += G8) {}
    if (more) issue_res(nbh, nkt);
    // ---- combine the two query halves' partial sums through LDS: the qh = 0 wave finishes (and stores) the "same" variant, the
    // qh = 1 wave the "cross" variant - each hands the other half of its sums over (a + b = b + a: the same bits as a one-sided sum).
    // Two rounds of two accumulators through the streamed-tile region (64 KiB per round; the resident region is being refilled).
    {
        float* xch = (float*)qd + ((wave >> 2) * 2 + kw) * 4096;     // 16 KiB per (role, key sub-block) pair and round
        float* gdst = xch + (qh == 1 ? 0 : 2048);
        const float* gsrc = xch + (qh == 0 ? 0 : 2048);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int i = 2 * rr + ii;
                    f32x4 a;
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = qh == 1 ? acc_s[i][4 * g + e] : acc_c[i][4 * g + e];
                    *(f32x4*)(gdst + ((ii * 4 + g) * 64 + lane) * 4) = a;
                }
            __syncthreads();
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int i = 2 * rr + ii;
                    const f32x4 a = *(const f32x4*)(gsrc + ((ii * 4 + g) * 64 + lane) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (qh == 0) acc_s[i][4 * g + e] += a[e];
                        else acc_c[i][4 * g + e] += a[e];
                    }
                }
            __syncthreads();
        }
    }
    // ---- store: each wave's [128 d x 32 keys] block, transposed through a private LDS region (32 rows x 264 B): waves 0-6 in the
    // streamed-tile region, wave 7 in the P hand-over slots (both idle now; the resident region belongs to the next item)
    if (kbase_w < S) {
        constexpr int OROW = 264;
        char* so = wave < 7 ? qd + wave * (32 * OROW) : smem + DKV_XP;
        auto store = [&](const f32x16* acc, float mul, bf16_t* dst) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = dt * 32 + 8 * g + 4 * fk;
                    u32x2 w;
                    w[0] = pack2bf(acc[dt][4 * g + 0] * mul, acc[dt][4 * g + 1] * mul);
                    w[1] = pack2bf(acc[dt][4 * g + 2] * mul, acc[dt][4 * g + 3] * mul);
                    *(u32x2*)(so + l31 * OROW + d * 2) = w;
                }
            // same-wave LDS write -> read: LDS ops of one wave execute in order and no other wave touches `so`
#pragma unroll
            for (int pass = 0; pass < 8; ++pass) {
                const int r = pass * 4 + (lane >> 4);
                const int kk = kbase_w + r;
                if (kk < S) {
                    const char* src = so + r * OROW + (lane & 15) * 16;
                    const u32x2 a = *(const u32x2*)src;
                    const u32x2 c2 = *(const u32x2*)(src + 8);
                    u32x4 v;
                    v[0] = a[0]; v[1] = a[1]; v[2] = c2[0]; v[3] = c2[1];
                    *(u32x4*)(dst + (tok0 + kk) * p.ldg + h * D128 + (lane & 15) * 8) = v;
                }
            }
        };
        if (qh == 0) { if (role_dk) store(acc_s, p.scale, p.dk_same); else store(acc_s, 1.0f, p.dv_same); }
        else { if (role_dk) store(acc_c, p.scale, p.dk_cross); else store(acc_c, 1.0f, p.dv_cross); }
    }
    if (!more) break;
    __syncthreads();                                             // the staging / hand-over regions are free for the next item
    bh = nbh; ktile = nkt;
  }
}

}  // namespace libra

using namespace libra;

extern "C" int libra_bridge_attn_bwd(const void* q, int64_t ldq, const void* k_same, int64_t ldk, const void* k_cross,
                                     int64_t ldkc, const void* v_same, int64_t ldv, const void* v_cross, int64_t ldvc,
                                     const void* out, const void* out_lo, int64_t ldout, const void* dout, int64_t lddo, const uint8_t* flag,
                                     const int32_t* kv_len, const float* lse, float* delta, void* dq, int64_t lddq,
                                     void* dk_same, void* dk_cross, void* dv_same, void* dv_cross, int64_t ldg, int64_t B,
                                     int64_t S, int64_t H, float scale, int32_t* err_word, void* stream) {
    if (B <= 0 || S <= 0) return LIBRA_OK;
    const int64_t HD = H * D128;
    if (ldq >= (1 << 18) || ldk >= (1 << 18) || ldkc >= (1 << 18) || ldv >= (1 << 18) || ldvc >= (1 << 18) || lddo >= (1 << 18))
        return LIBRA_ERR_SHAPE;                                    // 32-bit per-lane byte offsets in the tile loaders
    if (H <= 0 || S > 4096 || ldq < HD || ldk < HD || ldkc < HD || ldv < HD || ldvc < HD || ldout < HD || lddo < HD || lddq < HD || ldg < HD)
        return LIBRA_ERR_SHAPE;
    if ((ldq | ldk | ldkc | ldv | ldvc | ldout | lddo | lddq | ldg) % 8) return LIBRA_ERR_ALIGN;
    if (!q || !k_same || !k_cross || !v_same || !v_cross || !out || !dout || !flag || !lse || !delta || !dq || !dk_same ||
        !dk_cross || !dv_same || !dv_cross) return LIBRA_ERR_ALIGN;
    if (((uintptr_t)q | (uintptr_t)k_same | (uintptr_t)k_cross | (uintptr_t)v_same | (uintptr_t)v_cross | (uintptr_t)out |
         (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk_same | (uintptr_t)dk_cross | (uintptr_t)dv_same | (uintptr_t)dv_cross) & 15)
        return LIBRA_ERR_ALIGN;
    if (out_lo && ((uintptr_t)out_lo & 15)) return LIBRA_ERR_ALIGN;
    if ((uintptr_t)err_word & 3) return LIBRA_ERR_ALIGN;
    BridgeBwdArgs a;
    a.q = (const bf16_t*)q; a.ldq = ldq; a.k_same = (const bf16_t*)k_same; a.ldk = ldk; a.k_cross = (const bf16_t*)k_cross; a.ldkc = ldkc;
    a.v_same = (const bf16_t*)v_same; a.ldv = ldv; a.v_cross = (const bf16_t*)v_cross; a.ldvc = ldvc;
    a.dout = (const bf16_t*)dout; a.ldo = lddo; a.flag = flag; a.kv_len = kv_len; a.lse = lse; a.delta = delta;
    a.out = (const bf16_t*)out; a.out_lo = (const bf16_t*)out_lo; a.ldout = ldout;
    a.dq = (bf16_t*)dq; a.lddq = lddq; a.dk_same = (bf16_t*)dk_same; a.dk_cross = (bf16_t*)dk_cross;
    a.dv_same = (bf16_t*)dv_same; a.dv_cross = (bf16_t*)dv_cross; a.ldg = ldg;
    a.B = (int)B; a.S = (int)S; a.H = (int)H; a.scale = scale; a.sl2 = scale * LOG2E;
    a.err = err_word;
    static std::atomic<bool> attr_set{false};     // (idempotent call; atomic only so that concurrent first launches do not race on the flag)
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)bridge_attn_bwd_dq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DQ_LDS_B);
        (void)hipFuncSetAttribute((const void*)bridge_attn_bwd_dkv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DKV_LDS_B);
        attr_set = true;
    }
    a.n_t = (int)((S + DQ_BQ - 1) / DQ_BQ);
    long nblk = (long)B * H * a.n_t;
    if (nblk > 0x7fffffffL) return LIBRA_ERR_SHAPE;
    // (other dK/dV structures that were built and measured in round 2: profiles/r02_attn_bwd_anatomy.md)
    hipLaunchKernelGGL(bridge_attn_bwd_dq_kernel, dim3((unsigned)nblk), dim3(512), DQ_LDS_B, (hipStream_t)stream, a);
    if (hipGetLastError() != hipSuccess) return LIBRA_ERR_LAUNCH;
    a.n_t = (int)((S + 63) / 64);
    nblk = (long)B * H * a.n_t;
    if (nblk > 0x7fffffffL) return LIBRA_ERR_SHAPE;
    // persistent workgroups: one per CU (a 147-KiB workgroup owns its CU), a multiple of 8 so that block % 8 is the XCD of the
    // kernel's item order; fewer when the launch has fewer items than CUs
    static const int n_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        return n / 8 * 8;
    }();
    long grid = nblk < n_cu ? (nblk + 7) / 8 * 8 : n_cu;
    hipLaunchKernelGGL(bridge_attn_bwd_dkv_kernel, dim3((unsigned)grid), dim3(512), DKV_LDS_B, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}
