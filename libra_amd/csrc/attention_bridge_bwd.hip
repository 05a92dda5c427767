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
#include <cstdio>
#include <cstdlib>
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
    const float* lse; const float* delta;              // [B,H,S]
    bf16_t* dq; long lddq;
    bf16_t* dk_same; bf16_t* dk_cross; bf16_t* dv_same; bf16_t* dv_cross; long ldg;   // [B*S, H*128] each
    int B, S, H, n_t;
    float sl2, scale;
    int dbg;
    unsigned long long* trace;     // debug: per-phase s_memtime stamps of one workgroup (LIBRA_DKV_TRACE)
};

// Reduction-major image of a [rows][128 d] tile: 256-byte rows, 16-byte chunk c of row r stored at position
// c ^ tswz(r).  tswz mixes (r&3) into the chunk's high bits (what the transpose read ds_read_b64_tr_b16 needs to be
// conflict free) and (r>>2)&3 into its low bits, which also spreads the 16 rows of a ds_read_b128 lane group over
// all 64 banks: one image serves BOTH the row-fragment reads (A/B operand with k = d) and the transposed reads
// (A operand with k = rows).  [With only the (r&3) term, row reads were 4-way conflicted: 65 % of the dQ pass's LDS
// cycles in the round-1 PMC profile.]
__device__ __forceinline__ int tswz(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }

__device__ __forceinline__ void stage_t32(const bf16_t* __restrict__ base, long ld, int row0, int nrows, char* dst,
                                          int wave, int lane, int nwaves) {
    for (int pc = wave; pc < 8; pc += nwaves) {
        const int r = pc * 4 + (lane >> 4);
        const int c = (lane & 15) ^ tswz(r);
        int row = row0 + r; row = row < nrows ? row : nrows - 1;
        glds16(base + (long)row * ld + c * 8, dst + pc * 1024);
    }
}
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
// 16-byte row read (8 consecutive d of one row) from the reduction-major image
__device__ __forceinline__ bf16x8 nread_t(const char* tile, int row, int chunk) {
    return *(const bf16x8*)(tile + row * 256 + ((chunk ^ tswz(row)) << 4));
}
// transpose read: A-operand fragment X^T[d = 32*dt + l31][rows 16*sx + 4*fk + {0..3, 8..11}]
__device__ __forceinline__ bf16x8 tread_t(const char* tile, int lane, int dt, int sx) {
    const int pp = lane & 15, g16 = (lane >> 4) & 1, fk = lane >> 5;
    const int r1 = 16 * sx + 4 * fk + (pp >> 2);
    const int chunk = dt * 4 + 2 * g16 + ((pp & 3) >> 1);
    const char* a = tile + r1 * 256 + ((pp & 1) << 3);
    union { bf16x8 v; s16x4 h2[2]; } u;
    u.h2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(a + ((chunk ^ tswz(r1)) << 4)));
    u.h2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(a + 2048 + ((chunk ^ tswz(r1 + 8)) << 4)));
    return u.v;
}
__device__ __forceinline__ void split_pack(const f32x16& a, unsigned crossbits, int sx, bf16x8& same, bf16x8& cross) {
    union { bf16x8 v; unsigned u[4]; } ps, pc;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r0 = 8 * sx + 2 * j, r1 = r0 + 1;
        const bool c0 = (crossbits >> r0) & 1u, c1 = (crossbits >> r1) & 1u;
        ps.u[j] = pack2bf(c0 ? 0.f : a[r0], c1 ? 0.f : a[r1]);
        pc.u[j] = pack2bf(c0 ? a[r0] : 0.f, c1 ? a[r1] : 0.f);
    }
    same = ps.v; cross = pc.v;
}

// ================================================================================================
// dQ pass: 8 waves x 32 queries per workgroup, 64-key tiles (two 32-key halves per barrier), per variant one K image
// (row reads for S^T = K Q^T, transposed reads for dQ^T += K^T dS^T) and one V image (row reads for dP^T = V dO^T).
constexpr int DQ_VAR = 32768;                 // K tile 16 KiB + V tile 16 KiB (64 keys)
constexpr int DQ_STAGE_B = 2 * DQ_VAR;        // same + cross
constexpr int DQ_LDS_B = 2 * DQ_STAGE_B + 1024;
constexpr int DQ_BQ = 256;

template <bool CA>        // CA: lane-constant LDS addressing, see bridge_attn_bwd_dkv_kernel
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

    modality_masks(p.flag + tok0, S, kmask, tid, 512);
    const bool q_vis = p.flag[tok0 + q] != 0;
    if (tid < 2) qpres[tid] = 0;
    __syncthreads();
    if (__ballot(qin && fk == 0 && q_vis)) if (lane == 0) atomicOr(&qpres[1], 1);
    if (__ballot(qin && fk == 0 && !q_vis)) if (lane == 0) atomicOr(&qpres[0], 1);
    __syncthreads();
    const bool blkL = __builtin_amdgcn_readfirstlane(qpres[0]) != 0, blkV = __builtin_amdgcn_readfirstlane(qpres[1]) != 0;
    const bool wV = __ballot(q_vis && qin) != 0, wL = __ballot(!q_vis && qin) != 0;

    bf16x8 qf[8], dof[8];
    {
        const bf16_t* qp = p.q + (tok0 + q) * p.ldq + h * D128 + fk * 8;
        const bf16_t* dp = p.dout + (tok0 + q) * p.ldo + h * D128 + fk * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) { qf[ks] = *(const bf16x8*)(qp + ks * 16); dof[ks] = *(const bf16x8*)(dp + ks * 16); }
    }
    const long sidx = ((long)b * p.H + h) * S + q;
    float nLq2 = -p.lse[sidx] * LOG2E;
    float Dq = p.delta[sidx];
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
    stage(0, 0);
    int lane_o = lane;                                              // (see the forward kernel: per-tile recomputed LDS offsets)
    int xr = 0, xt0 = 0, xt1 = 0;
    if constexpr (CA) {
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
        asm volatile("" : "+v"(lane_o));
        if constexpr (CA) asm volatile("" : "+v"(xr), "+v"(xt0), "+v"(xt1));
        const int l31o = lane_o & 31, fko = lane_o >> 5;
        auto rd_row = [&](const char* tile, int ks) -> bf16x8 {
            if constexpr (CA) return *(const bf16x8*)(tile + (xr ^ (ks << 5)));
            else return nread_t(tile, l31o, 2 * ks + fko);
        };
        auto rd_tr = [&](const char* tile, int dt, int sx) -> bf16x8 {
            if constexpr (CA) {
                union { bf16x8 v; s16x4 h2[2]; } u;
                u.h2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(tile + sx * 4096 + (xt0 ^ (dt << 6))));
                u.h2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(tile + sx * 4096 + 2048 + (xt1 ^ (dt << 6))));
                return u.v;
            } else return tread_t(tile, lane_o, dt, sx);
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
constexpr int DKV_LDS_B = KV_RES + 2 * QD_STAGE + 1024;

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
// fragment (row, 8 consecutive d of k-step ks) of a row image whose sub-tiles are `sub_bytes` apart
__device__ __forceinline__ bf16x8 nfrag(const char* tile, int row, int ks, int fk, int sub_bytes) {
    const int sub = ks >> 2, c = (2 * (ks & 3) + fk) ^ ((row >> 1) & 7);
    return *(const bf16x8*)(tile + sub * sub_bytes + row * 128 + (c << 4));
}

// CA = lane-constant LDS addressing: every fragment address is a per-lane constant XOR a compile-time constant (one VALU op per
// read) instead of the swizzle arithmetic rebuilt per read (the round-1 PMC profile counted 12.4 VALU per MFMA in this kernel)
template <bool CA>
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
    const int nblk = p.B * p.H * p.n_t;
    const int L = xcd_remap(blockIdx.x, nblk);
    const int ktile = L % p.n_t;                                 // low key tiles see the most queries: they come first
    const int bh = L / p.n_t;
    const int h = bh % p.H, b = bh / p.H;
    const int S = p.S;
    const long tok0 = (long)b * S;
    int len = p.kv_len ? p.kv_len[b] : S;
    len = len < S ? len : S;
    const int key0 = ktile * 64;
    const int kbase_w = key0 + kw * 32;
    int key = kbase_w + l31;
    const bool kin = key < S;
    key = kin ? key : S - 1;
    int k_vis_i = p.flag[tok0 + key] != 0;
    pin(k_vis_i);
    const bool k_vis = k_vis_i != 0;
    const bool wkV = __ballot(k_vis && kin) != 0, wkL = __ballot(!k_vis && kin) != 0;

    modality_masks(p.flag + tok0, S, qmask, tid, 512);
    stage_res64(p.k_same + tok0 * p.ldk + h * D128, (unsigned)p.ldk * 2u, key0, S, res, wave, lane);
    stage_res64(p.k_cross + tok0 * p.ldkc + h * D128, (unsigned)p.ldkc * 2u, key0, S, res + 16384, wave, lane);
    stage_res64(p.v_same + tok0 * p.ldv + h * D128, (unsigned)p.ldv * 2u, key0, S, res + 32768, wave, lane);
    stage_res64(p.v_cross + tok0 * p.ldvc + h * D128, (unsigned)p.ldvc * 2u, key0, S, res + 49152, wave, lane);

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
    const int nqt = (S + 63) / 64;
    f32x16 acc_s[4], acc_c[4];                                   // dV (or dK) for the same / cross variant, [128 d x 32 keys]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_s[i][r] = 0.f; acc_c[i][r] = 0.f; }
    if (it0 < nqt) stage_q(0, it0);

    const char* rK = res + kw * 32 * 128;                         // this wave's 32 key rows inside each 64-row sub-tile
    int lane_o = lane;
    // lane constants of the CA variant (see bridge_attn_bwd_dkv2_kernel for the derivation)
    int xr = 0, xv = 0, xt0 = 0, xt1 = 0;
    if constexpr (CA) {
        const int pp = lane & 15, g16 = (lane >> 4) & 1;
        const int r1 = 4 * fk + (pp >> 2), lp = 2 * g16 + ((pp & 3) >> 1);
        xr = l31 * 256 + ((fk ^ tswz(l31)) << 4);
        xv = l31 * 128 + ((fk ^ ((l31 >> 1) & 7)) << 4);
        xt0 = r1 * 256 + ((pp & 1) << 3) + ((lp ^ tswz(r1)) << 4);
        xt1 = r1 * 256 + ((pp & 1) << 3) + ((lp ^ tswz(r1 + 8)) << 4);
    }
    for (int it = it0; it < nqt; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int cur = (it - it0) & 1;
        if (it + 1 < nqt) stage_q(cur ^ 1, it + 1);
        const int q0 = it * 64 + qh * 32;
        if (kbase_w >= S || q0 >= S || q0 + 31 < kbase_w) continue;   // no (query >= key) pair for this wave in the tile
        asm volatile("" : "+v"(lane_o));
        if constexpr (CA) asm volatile("" : "+v"(xr), "+v"(xt0), "+v"(xt1), "+v"(xv));
        const int l31o = lane_o & 31, fko = lane_o >> 5;
        auto rd_row = [&](const char* tile, int ks) -> bf16x8 {
            if constexpr (CA) return *(const bf16x8*)(tile + (xr ^ (ks << 5)));
            else return nread_t(tile, l31o, 2 * ks + fko);
        };
        auto rd_res = [&](const char* tile, int ks) -> bf16x8 {
            if constexpr (CA) return *(const bf16x8*)(tile + (ks >> 2) * 8192 + (xv ^ ((ks & 3) << 5)));
            else return nfrag(tile, l31o, ks, fko, 8192);
        };
        auto rd_tr = [&](const char* tile, int dt, int sx) -> bf16x8 {
            if constexpr (CA) {
                union { bf16x8 v; s16x4 h2[2]; } u;
                u.h2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(tile + sx * 4096 + (xt0 ^ (dt << 6))));
                u.h2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(tile + sx * 4096 + 2048 + (xt1 ^ (dt << 6))));
                return u.v;
            } else return tread_t(tile, lane_o, dt, sx);
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
        // s <- P = exp2(S*sl2 - L) (dV waves) or dS = P (dP - D) (dK waves)
        auto finish = [&](f32x16& s, const f32x16& dp) {
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
            if (role_dk) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 Dv = *(const f32x4*)(sD + 8 * g + 4 * fk);
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[4 * g + e] *= dp[4 * g + e] - Dv[e];
                }
            }
        };
        const char* st = role_dk ? sq : sq + 16384;               // Q^T fragments (dK) or dO^T fragments (dV)
        const bool mixed = wsame && wcross;
        // one pass per variant present (a tile pair with both modalities on either side - rare - pays S twice): every
        // accumulator set is touched from exactly one place, which keeps all 128 of them in registers
        auto pass = [&](const char* rk, bool cross, f32x16* acc) {
            f32x16 s, dp;
            score_s(rk, s);
            score_dp(rk, dp);
            finish(s, dp);
            if (mixed) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ql = (r & 3) + 8 * (r >> 2) + 4 * fk;
                    s[r] = ((((qm >> ql) & 1u) != 0) != k_vis) == cross ? s[r] : 0.f;
                }
            }
#pragma unroll
            for (int sx = 0; sx < 2; ++sx) {
                union { bf16x8 v; unsigned u[4]; } pk;
#pragma unroll
                for (int j = 0; j < 4; ++j) pk.u[j] = pack2bf(s[8 * sx + 2 * j], s[8 * sx + 2 * j + 1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_tr(st, dt, sx), pk.v, acc[dt], 0, 0, 0);
            }
        };
        if (wsame) pass(rK, false, acc_s);
        if (wcross) pass(rK + 16384, true, acc_c);
    }
    // ---- combine the two query halves' partial sums: waves with qh = 1 hand theirs over through LDS ----
    __syncthreads();
    {
        float* xch = (float*)smem + ((wave >> 2) * 2 + kw) * 8192;   // 32 KiB per (role, key sub-block) pair: [acc][reg quad][lane] x4
        if (qh == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 a, c;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { a[e] = acc_s[i][4 * g + e]; c[e] = acc_c[i][4 * g + e]; }
                    *(f32x4*)(xch + ((i * 4 + g) * 64 + lane) * 4) = a;
                    *(f32x4*)(xch + 4096 + ((i * 4 + g) * 64 + lane) * 4) = c;
                }
        }
        __syncthreads();
        if (qh == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 a = *(const f32x4*)(xch + ((i * 4 + g) * 64 + lane) * 4);
                    const f32x4 c = *(const f32x4*)(xch + 4096 + ((i * 4 + g) * 64 + lane) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { acc_s[i][4 * g + e] += a[e]; acc_c[i][4 * g + e] += c[e]; }
                }
        }
        __syncthreads();
    }
    if (qh != 0 || kbase_w >= S) return;
    // ---- store: each wave's two [128 d x 32 keys] blocks, transposed through a private LDS region (32 rows x 264 B)
    constexpr int OROW = 264;
    char* so = smem + wave * (32 * OROW);
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
    if (role_dk) { store(acc_s, p.scale, p.dk_same); store(acc_c, p.scale, p.dk_cross); }
    else { store(acc_s, 1.0f, p.dv_same); store(acc_c, 1.0f, p.dv_cross); }
}

// ================================================================================================
// dK / dV pass, lane-linear images + an explicit software pipeline (round 2; same decomposition and roles as the kernel above).
// What limited the kernel above (round-2 ISA read + round-1 PMC: matrix pipe busy 21 %, 5.7k cycles per 64-query iteration
// against 1.3k of MFMA work per SIMD): every fragment address was a swizzle XOR rebuilt per read (or, hoisted, one live VGPR
// per address: > 40 of them), and the compiler issued each fragment pair right before the MFMA that consumes it, so each of the
// 16-24 MFMAs of a pass waited a full LDS round trip.  Here
//   * the images are K-STEP MAJOR: [k-step of 16 d][fk][row][16 B].  A row fragment is lane (row, fk) -> base + 16 B * row: one
//     lane-constant base and the k-step in the instruction's immediate offset, conflict free without a swizzle (the 16 lanes of a
//     ds_read_b128 group hold 16 consecutive-in-banks rows).  The transposed reads of the streamed images (ds_read_b64_tr_b16:
//     4 rows x {2 k-steps} x {2 fk} per 32 lanes) are made conflict free by storing row r of the fk = 1 half at position r ^ 4
//     and padding the k-step slab to 1152 B: the four 64-byte pieces land on four different bank quarters.  Zero VALU per read.
//   * a pass is written as load batch / MFMA batch groups separated by scheduling fences: the reads of batch i+1 (4 k-steps) are
//     in flight under the MFMAs of batch i, the transposed fragments of the second stage and L / D under the last first-stage
//     MFMAs, and the wave's role is a compile-time constant of the pass body.
constexpr int QSLAB = 1152;                    // [fk][32 rows][16 B] = 1 KiB (one direct-to-LDS piece) + 128 B pad
constexpr int QHALF = 8 * QSLAB;               // 32 queries x 128 d
constexpr int QIMG = 2 * QHALF;                // 64 queries
constexpr int QD4_STAGE = 2 * QIMG + 512;      // Q image, dO image, L[64], D[64]
constexpr int DKV4_LDS_B = KV_RES + 2 * QD4_STAGE + 1024;

// streamed [64 rows][128 d] image: piece pc (16 x 1 KiB over 8 waves) = (row half pc >> 3, k-step pc & 7); lane = (fk, position)
__device__ __forceinline__ void stage_lin64(const bf16_t* __restrict__ base, unsigned ld_b, int row0, int nrows, char* dst, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pc = wave * 2 + j, half = pc >> 3, ks = pc & 7;
        const int f = lane >> 5, r = (lane & 31) ^ (f << 2);
        int row = row0 + half * 32 + r; row = row < nrows ? row : nrows - 1;
        glds16_off(base, (unsigned)row * ld_b + (unsigned)(ks * 32 + f * 16), dst + half * QHALF + ks * QSLAB);
    }
}
// resident [64 keys][128 d] tile: [k-step][fk][key][16 B], piece pc = (k-step pc >> 1, fk pc & 1), lane = key; 16 KiB
__device__ __forceinline__ void stage_lin_res(const bf16_t* __restrict__ base, unsigned ld_b, int key0, int S, char* dst, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pc = wave * 2 + j;
        int key = key0 + lane; key = key < S ? key : S - 1;
        glds16_off(base, (unsigned)key * ld_b + (unsigned)((pc >> 1) * 32 + (pc & 1) * 16), dst + pc * 1024);
    }
}

__global__ __launch_bounds__(512, 1) void bridge_attn_bwd_dkv_lin_kernel(const BridgeBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* res = smem;                                            // Ks, Kc, Vs, Vc
    char* qd = smem + KV_RES;
    unsigned* qmask = (unsigned*)(smem + KV_RES + 2 * QD4_STAGE); // per 32 queries: bit i = query i is a vision token
    if (p.dbg & 64) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kw = wave & 1, qh = (wave >> 1) & 1;
    const bool role_dk = (wave >> 2) != 0;                       // waves 0-3: dV; waves 4-7: dK
    const int fk = lane >> 5, l31 = lane & 31;
    const int nblk = p.B * p.H * p.n_t;
    const int L = xcd_remap(blockIdx.x, nblk);
    const int ktile = L % p.n_t;                                 // low key tiles see the most queries: they come first
    const int bh = L / p.n_t;
    const int h = bh % p.H, b = bh / p.H;
    const int S = p.S;
    const long tok0 = (long)b * S;
    int len = p.kv_len ? p.kv_len[b] : S;
    len = len < S ? len : S;
    const int key0 = ktile * 64;
    const int kbase_w = key0 + kw * 32;
    int key = kbase_w + l31;
    const bool kin = key < S;
    key = kin ? key : S - 1;
    int k_vis_i = p.flag[tok0 + key] != 0;
    pin(k_vis_i);
    const bool k_vis = k_vis_i != 0;
    const bool wkV = __ballot(k_vis && kin) != 0, wkL = __ballot(!k_vis && kin) != 0;

    modality_masks(p.flag + tok0, S, qmask, tid, 512);
    if (!(p.dbg & 256)) {
    stage_lin_res(p.k_same + tok0 * p.ldk + h * D128, (unsigned)p.ldk * 2u, key0, S, res, wave, lane);
    stage_lin_res(p.k_cross + tok0 * p.ldkc + h * D128, (unsigned)p.ldkc * 2u, key0, S, res + 16384, wave, lane);
    stage_lin_res(p.v_same + tok0 * p.ldv + h * D128, (unsigned)p.ldv * 2u, key0, S, res + 32768, wave, lane);
    stage_lin_res(p.v_cross + tok0 * p.ldvc + h * D128, (unsigned)p.ldvc * 2u, key0, S, res + 49152, wave, lane);
    }

    const bf16_t* qbase = p.q + tok0 * p.ldq + h * D128;
    const bf16_t* dobase = p.dout + tok0 * p.ldo + h * D128;
    const float* lbase = p.lse + ((long)b * p.H + h) * S;
    const float* dbase = p.delta + ((long)b * p.H + h) * S;
    auto stage_q = [&](int buf, int t) {
        char* dst = qd + buf * QD4_STAGE;
        stage_lin64(qbase, (unsigned)p.ldq * 2u, t * 64, S, dst, wave, lane);
        stage_lin64(dobase, (unsigned)p.ldo * 2u, t * 64, S, dst + QIMG, wave, lane);
        if (wave < 2) {                                          // 64 fp32 each: one 4-byte direct-to-LDS op
            int qi = t * 64 + lane; qi = qi < S ? qi : S - 1;
            glds4((wave == 0 ? lbase : dbase) + qi, dst + 2 * QIMG + wave * 256);
        }
    };
    const int it0 = key0 / 64;                                   // first query tile that can see this key block
    const int nqt = (p.dbg & 32) ? key0 / 64 : (S + 63) / 64;
    f32x16 acc_s[4], acc_c[4];                                   // dV (or dK) for the same / cross variant, [128 d x 32 keys]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_s[i][r] = 0.f; acc_c[i][r] = 0.f; }
    if (it0 < nqt) stage_q(0, it0);

    // lane constants: every fragment address below is one of these + a wave-uniform tile base + an immediate
    const int pp = lane & 15, g16 = (lane >> 4) & 1, fkp = (pp & 3) >> 1;
    int off_row = fk * 512 + ((l31 ^ (fk << 2)) << 4);                                   // row fragment of a streamed half image
    int off_res = fk * 1024 + ((kw * 32 + l31) << 4);                                    // key fragment of a resident tile
    int off_tr = g16 * QSLAB + fkp * 512 + (((4 * fk + (pp >> 2)) ^ (fkp << 2)) << 4) + ((pp & 1) << 3);   // transposed fragment
    auto run_loop = [&](auto role_tag) {
    constexpr bool DK = decltype(role_tag)::value;
    for (int it = it0; it < nqt; ++it) {
        if (!(p.dbg & 8)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(p.dbg & 16)) __syncthreads();
        const int cur = (it - it0) & 1;
        if (it + 1 < nqt && !((p.dbg & 1) && it > it0 + 1)) stage_q(cur ^ 1, it + 1);
        const int q0 = it * 64 + qh * 32;
        if (kbase_w >= S || q0 >= S || q0 + 31 < kbase_w) continue;   // no (query >= key) pair for this wave in the tile
        asm volatile("" : "+v"(off_row), "+v"(off_res), "+v"(off_tr));
        const char* stg = qd + cur * QD4_STAGE;
        const char* aq = stg + qh * QHALF + off_row;              // Q row fragments (dO at + QIMG)
        const char* tq = stg + qh * QHALF + off_tr + (DK ? 0 : QIMG);   // Q^T fragments (dK) or dO^T fragments (dV)
        const char* rres = res + off_res;
        const float* sL = (const float*)(stg + 2 * QIMG) + qh * 32 + 4 * fk;
        const unsigned qm = (unsigned)__builtin_amdgcn_readfirstlane((int)qmask[2 * it + qh]);
        int nvalid = S - q0; nvalid = nvalid > 32 ? 32 : nvalid;
        const unsigned full = nvalid >= 32 ? 0xffffffffu : ((1u << nvalid) - 1u);
        const bool qV = (qm & full) != 0, qL = ((~qm) & full) != 0;
        const bool wsame = (qL && wkL) || (qV && wkV);
        const bool wcross = (qL && wkV) || (qV && wkL);
        const bool masked = q0 < kbase_w + 31 || q0 + 32 > S || kbase_w + 32 > len;
        const bool mixed = wsame && wcross;

        // accumulator row r <-> query q0 + (r&3) + 8(r>>2) + 4fk ; column <-> this lane's key.  VOFF: resident variant (0 / 16384)
        auto pass = [&](const int voff, bool cross, f32x16* acc) {
            const char* rk = rres + voff;
#define LIBRA_SB() __builtin_amdgcn_sched_barrier(0)
#define RD_A(base, ks) (*(const bf16x8*)((base) + (ks) * QSLAB))
#define RD_B(base, ks) (*(const bf16x8*)((base) + (ks) * 2048))
#define RD_T(dt, sx, hi) __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(tq + (dt) * 2 * QSLAB + (sx) * 256 + (hi) * 128))
            // first stage as one stream of N (A, B) fragment pairs read PD pairs ahead of the MFMA that consumes them; the
            // transposed fragments of the second stage ride on the last steps.  Consecutive MFMAs never share an accumulator
            // (anything issued between two MFMAs on the SAME accumulator costs ~43 cycles, MI355X_MICROARCH "cycle constants"):
            // dK waves alternate S and dP k-steps, dV waves alternate two partial S sums.
            constexpr int N = DK ? 16 : 8, PD = 4;
            bf16x8 fa[N], fb[N];
            union TR { bf16x8 v; s16x4 h2[2]; } tr[8];
            f32x16 s, dp;                                            // dV waves: dp = the odd k-steps' partial S
            f32x4 Lq[4], Dq[4];
            auto load_pair = [&](const int i) {
                if constexpr (DK) {
                    if (i & 1) { fa[i] = RD_A(aq + QIMG, i >> 1); fb[i] = RD_B(rk + 32768, i >> 1); }
                    else { fa[i] = RD_A(aq, i >> 1); fb[i] = RD_B(rk, i >> 1); }
                } else { fa[i] = RD_A(aq, i); fb[i] = RD_B(rk, i); }
            };
#pragma unroll
            for (int i = 0; i < PD; ++i) load_pair(i);
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
            LIBRA_SB();
#pragma unroll
            for (int i = 0; i < N; ++i) {
                if (i + PD < N) load_pair(i + PD);
                else {                                               // the last PD steps: two transposed fragments each
                    const int j = (i + PD - N) * 2;
#pragma unroll
                    for (int e = j; e < j + 2; ++e) { tr[e].h2[0] = RD_T(e & 3, e >> 2, 0); tr[e].h2[1] = RD_T(e & 3, e >> 2, 1); }
                }
                if (i & 1) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[i], dp, 0, 0, 0);
                else s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[i], s, 0, 0, 0);
                LIBRA_SB();
            }
            if constexpr (!DK) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] += dp[r];
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) Lq[g] = *(const f32x4*)(sL + 8 * g);
            LIBRA_SB();
#undef RD_A
#undef RD_B
#undef RD_T
#undef LIBRA_SB
            if (!(p.dbg & 2)) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) s[4 * g + e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[4 * g + e], p.sl2, -Lq[g][e] * LOG2E));
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
            if (DK && !(p.dbg & 2)) {
#pragma unroll
                for (int g = 0; g < 4; ++g) Dq[g] = *(const f32x4*)(sL + 64 + 8 * g);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[4 * g + e] *= dp[4 * g + e] - Dq[g][e];
            }
            if (mixed) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ql = (r & 3) + 8 * (r >> 2) + 4 * fk;
                    s[r] = ((((qm >> ql) & 1u) != 0) != k_vis) == cross ? s[r] : 0.f;
                }
            }
#pragma unroll
            for (int sx = 0; sx < 2; ++sx) {
                union { bf16x8 v; unsigned u[4]; } pk;
#pragma unroll
                for (int j = 0; j < 4; ++j) pk.u[j] = pack2bf(s[8 * sx + 2 * j], s[8 * sx + 2 * j + 1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    if (!(p.dbg & 4) || (dt == 0 && sx == 0)) acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr[4 * sx + dt].v, pk.v, acc[dt], 0, 0, 0);
            }
        };
        if (wsame) pass(0, false, acc_s);
        if (wcross) pass(16384, true, acc_c);
    }
    };
    // the wave's role is fixed: the whole loop is instantiated per role, so every accumulator is written from one place
    if (role_dk) run_loop(std::true_type{}); else run_loop(std::false_type{});
    if (p.dbg & 128) return;
    // ---- combine the two query halves' partial sums: waves with qh = 1 hand theirs over through LDS ----
    __syncthreads();
    if (p.dbg & 512) { if (qh == 0 && lane == 0) p.dk_same[(tok0 + kbase_w) * p.ldg] = f2bf(acc_s[0][0] + acc_c[0][0]); return; }
    {
        float* xch = (float*)smem + ((wave >> 2) * 2 + kw) * 8192;   // 32 KiB per (role, key sub-block) pair: [acc][reg quad][lane] x4
        if (qh == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 a, c;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { a[e] = acc_s[i][4 * g + e]; c[e] = acc_c[i][4 * g + e]; }
                    *(f32x4*)(xch + ((i * 4 + g) * 64 + lane) * 4) = a;
                    *(f32x4*)(xch + 4096 + ((i * 4 + g) * 64 + lane) * 4) = c;
                }
        }
        __syncthreads();
        if (qh == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 a = *(const f32x4*)(xch + ((i * 4 + g) * 64 + lane) * 4);
                    const f32x4 c = *(const f32x4*)(xch + 4096 + ((i * 4 + g) * 64 + lane) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { acc_s[i][4 * g + e] += a[e]; acc_c[i][4 * g + e] += c[e]; }
                }
        }
        __syncthreads();
    }
    if (qh != 0 || kbase_w >= S) return;
    // ---- store: each wave's two [128 d x 32 keys] blocks, transposed through a private LDS region (32 rows x 264 B)
    constexpr int OROW = 264;
    char* so = smem + wave * (32 * OROW);
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
        for (int pass_i = 0; pass_i < 8; ++pass_i) {
            const int r = pass_i * 4 + (lane >> 4);
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
    if (role_dk) { store(acc_s, p.scale, p.dk_same); store(acc_c, p.scale, p.dk_cross); }
    else { store(acc_s, 1.0f, p.dv_same); store(acc_c, 1.0f, p.dv_cross); }
}

// ================================================================================================
// dK / dV pass, third structure: P handed from the dV waves to the dK waves, the two roles half an iteration apart.
// Anatomy of the kernel above on MI355X (tools/gpu_dkv_anatomy.sh, B=8 S=2048 H=32): of 1.38 ms, 0.98 is the query-tile loop -
// 3.9k cycles per 64-query iteration for 1.3k cycles of MFMA per SIMD - and it does not move when the VALU work is halved, the
// fragment reads are pipelined or the DMA / barrier are removed one at a time: all eight waves pass through the same phases
// (LDS reads -> first-stage MFMAs -> exp / dS arithmetic -> second-stage MFMAs) in lock step behind one barrier per iteration,
// so the LDS pipe, the matrix pipe and the VALU are each busy for a fraction of the iteration, one after the other.  Here
//   * S = Q K^T and P = exp2(..) are computed ONCE, by the dV wave of a (32 query x 32 key) block, and P (fp32, 4 KiB) is handed
//     to the block's dK wave through LDS: 4 matmuls per tile pair instead of 5, and a quarter fewer fragment reads;
//   * a wave's iteration is two half steps: stage 1 (fragment reads + 8 first-stage MFMAs + its arithmetic -> packed P or dS)
//     and stage 2 (8 second-stage MFMAs on operands already in registers); the dK waves run ONE HALF STEP BEHIND the dV waves
//     (two barriers per iteration), so on every SIMD - which carries one wave of each role - one wave's LDS / VALU phase overlaps
//     the other's MFMA phase, and the handed-over P is ready exactly when its consumer starts;
//   * a tile with both variants (modality boundary) runs its first stage twice and merges per element; the second stage adds
//     the same-variant and cross-variant parts of the packed operand into their own accumulators - every accumulator set is
//     written from one place.
constexpr int HO_PBUF = KV_RES + 2 * QD4_STAGE;          // 4 blocks x [4 quads][64 lanes][4 fp32]
constexpr int HO_MASK = HO_PBUF + 4 * 4096;
constexpr int DKV5_LDS_B = HO_MASK + 1024;

__global__ __launch_bounds__(512, 1) void bridge_attn_bwd_dkv_ho_kernel(const BridgeBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* res = smem;                                            // Ks, Kc, Vs, Vc: [k-step][fk][64 keys][16 B]
    char* qd = smem + KV_RES;
    unsigned* qmask = (unsigned*)(smem + HO_MASK);               // per 32 queries: bit i = query i is a vision token
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kw = wave & 1, qh = (wave >> 1) & 1;
    const bool role_dk = (wave >> 2) != 0;                       // waves 0-3: dV (and P); waves 4-7: dK
    const int fk = lane >> 5, l31 = lane & 31;
    const int nblk = p.B * p.H * p.n_t;
    const int L = xcd_remap(blockIdx.x, nblk);
    const int ktile = L % p.n_t;                                 // low key tiles see the most queries: they come first
    const int bh = L / p.n_t;
    const int h = bh % p.H, b = bh / p.H;
    const int S = p.S;
    const long tok0 = (long)b * S;
    int len = p.kv_len ? p.kv_len[b] : S;
    len = len < S ? len : S;
    const int key0 = ktile * 64;
    const int kbase_w = key0 + kw * 32;
    int key = kbase_w + l31;
    const bool kin = key < S;
    key = kin ? key : S - 1;
    int k_vis_i = p.flag[tok0 + key] != 0;
    pin(k_vis_i);
    const bool k_vis = k_vis_i != 0;
    const bool wkV = __ballot(k_vis && kin) != 0, wkL = __ballot(!k_vis && kin) != 0;

    modality_masks(p.flag + tok0, S, qmask, tid, 512);
    stage_lin_res(p.k_same + tok0 * p.ldk + h * D128, (unsigned)p.ldk * 2u, key0, S, res, wave, lane);
    stage_lin_res(p.k_cross + tok0 * p.ldkc + h * D128, (unsigned)p.ldkc * 2u, key0, S, res + 16384, wave, lane);
    stage_lin_res(p.v_same + tok0 * p.ldv + h * D128, (unsigned)p.ldv * 2u, key0, S, res + 32768, wave, lane);
    stage_lin_res(p.v_cross + tok0 * p.ldvc + h * D128, (unsigned)p.ldvc * 2u, key0, S, res + 49152, wave, lane);

    const bf16_t* qbase = p.q + tok0 * p.ldq + h * D128;
    const bf16_t* dobase = p.dout + tok0 * p.ldo + h * D128;
    const float* lbase = p.lse + ((long)b * p.H + h) * S;
    const float* dbase = p.delta + ((long)b * p.H + h) * S;
    // streamed tile t -> ring buffer `buf`: 16 + 16 pieces of 1 KiB (Q image, dO image) + L and D.  The per-lane part of a piece's
    // source offset is tile independent; the rest rides in the wave-uniform base.  Direct-to-LDS issue stalls the issuing wave ~100-200 cycles per piece, so the pieces are shared out by the slack of the two roles (see the loop).
    int lane_v = lane;
    auto stage_piece = [&](int buf, int t, int pc) {
        asm volatile("" : "+v"(lane_v));                         // rebuilt per call (6 VALU): kept live across the loop it was spilled
        const int sf = lane_v >> 5, sr = (lane_v & 31) ^ (sf << 2);
        char* dst = qd + buf * QD4_STAGE + (pc >> 3) * QHALF + (pc & 7) * QSLAB;
        const int row0 = t * 64 + (pc >> 3) * 32;
        int row = sr;
        long rbase = row0;
        if (row0 + 32 > S) { row = row0 + sr; row = row < S ? row : S - 1; rbase = 0; }      // ragged last tile: clamp the rows
        glds16_off(qbase + rbase * p.ldq + (pc & 7) * 16, (unsigned)row * ((unsigned)p.ldq * 2u) + (unsigned)(sf * 16), dst);
        glds16_off(dobase + rbase * p.ldo + (pc & 7) * 16, (unsigned)row * ((unsigned)p.ldo * 2u) + (unsigned)(sf * 16), dst + QIMG);
    };
    auto stage_ld = [&](int buf, int t, int which) {             // 64 fp32: one 4-byte direct-to-LDS op
        int qi = t * 64 + lane; qi = qi < S ? qi : S - 1;
        glds4_off(which == 0 ? lbase : dbase, (unsigned)qi * 4u, qd + buf * QD4_STAGE + 2 * QIMG + which * 256);
    };
    auto stage_q = [&](int buf, int t) {                         // the whole tile, all 8 waves (prologue)
        stage_piece(buf, t, wave * 2); stage_piece(buf, t, wave * 2 + 1);
        if (wave < 2) stage_ld(buf, t, wave);
    };
    const int it0 = key0 / 64;                                   // first query tile that can see this key block
    const int nqt = (S + 63) / 64;
    f32x16 acc_s[4], acc_c[4];                                   // dV (or dK) for the same / cross variant, [128 d x 32 keys]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_s[i][r] = 0.f; acc_c[i][r] = 0.f; }
    if (it0 < nqt) stage_q(0, it0);

    // lane constants (rebuilt from the lane id where they are used - a dozen VALU per tile - rather than held in registers across
    // the loop): every fragment address is one of these + a wave-uniform tile base + an immediate
    auto lane_offsets = [&](int ln, int& off_row, int& off_res, int& off_tr) {
        const int f = ln >> 5, l = ln & 31, pp = ln & 15, g16 = (ln >> 4) & 1, fkp = (pp & 3) >> 1;
        off_row = f * 512 + ((l ^ (f << 2)) << 4);                                       // row fragment of a streamed half image
        off_res = f * 1024 + ((kw * 32 + l) << 4);                                       // key fragment of a resident tile
        off_tr = g16 * QSLAB + fkp * 512 + (((4 * f + (pp >> 2)) ^ (fkp << 2)) << 4) + ((pp & 1) << 3);   // transposed fragment
    };
    const int pblk_off = HO_PBUF + (wave & 3) * 4096;             // this block's P: [quad g][lane][4 fp32]

    auto run_loop = [&](auto role_tag) {
        constexpr bool DK = decltype(role_tag)::value;
        // operands of the pending second stage (registers only): transposed fragments + the packed P / dS
        union TR { bf16x8 v; s16x4 h2[2]; } tr[8];
        union PK { bf16x8 v; unsigned u[4]; } pk[2];
        f32x16 x;                                                    // P (dV waves) / dP (dK waves) of the tile in flight
        bool pend = false, pend_s = false, pend_c = false;           // second stage pending (and for which variants)
        bool act = false, act_s = false, act_c = false;              // dK waves: tile whose dP is computed, dS still to come
        unsigned pend_qm = 0, act_qm = 0;
        auto stage2 = [&]() {
            const bool mixed = pend_s && pend_c;                     // modality boundary: each variant takes its own elements
            auto part = [&](const int sx, const bool cross) -> bf16x8 {
                if (!mixed) return pk[sx].v;
                PK m;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r0 = 8 * sx + 2 * j, r1 = r0 + 1;
                    const int ql0 = (r0 & 3) + 8 * (r0 >> 2) + 4 * fk, ql1 = (r1 & 3) + 8 * (r1 >> 2) + 4 * fk;
                    const bool x0 = ((((pend_qm >> ql0) & 1u) != 0) != k_vis) == cross, x1 = ((((pend_qm >> ql1) & 1u) != 0) != k_vis) == cross;
                    m.u[j] = pk[sx].u[j] & ((x0 ? 0xffffu : 0u) | (x1 ? 0xffff0000u : 0u));
                }
                return m.v;
            };
            if (pend_s) {
#pragma unroll
                for (int sx = 0; sx < 2; ++sx) {
                    const bf16x8 op = part(sx, false);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) acc_s[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr[4 * sx + dt].v, op, acc_s[dt], 0, 0, 0);
                }
            }
            if (pend_c) {
#pragma unroll
                for (int sx = 0; sx < 2; ++sx) {
                    const bf16x8 op = part(sx, true);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) acc_c[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr[4 * sx + dt].v, op, acc_c[dt], 0, 0, 0);
                }
            }
        };
        // first stage of tile `it`: x <- S (dV waves) / dP (dK waves), tr <- the transposed fragments of its second stage.
        // Sets act / act_s / act_c / act_qm.  `mid()` runs once, after the first variant's MFMAs are issued (direct-to-LDS issue).
        auto first = [&](const int it, auto&& mid) {
            const int cur = (it - it0) & 1;
            const int q0 = it * 64 + qh * 32;
            act = !(kbase_w >= S || q0 >= S || q0 + 31 < kbase_w);   // any (query >= key) pair for this wave in the tile?
            act_s = act_c = false;
            if (!act) { mid(); return; }
            asm volatile("" : "+v"(lane_v));
            int off_row, off_res, off_tr;
            lane_offsets(lane_v, off_row, off_res, off_tr);
            const char* stg = qd + cur * QD4_STAGE;
            const char* aq = stg + qh * QHALF + off_row + (DK ? QIMG : 0);        // dO rows (dK waves) / Q rows (dV waves)
            const char* tq = stg + qh * QHALF + off_tr + (DK ? 0 : QIMG);         // Q^T fragments (dK) / dO^T fragments (dV)
            const unsigned qm = (unsigned)__builtin_amdgcn_readfirstlane((int)qmask[2 * it + qh]);
            int nvalid = S - q0; nvalid = nvalid > 32 ? 32 : nvalid;
            const unsigned full = nvalid >= 32 ? 0xffffffffu : ((1u << nvalid) - 1u);
            const bool qV = (qm & full) != 0, qL = ((~qm) & full) != 0;
            act_s = (qL && wkL) || (qV && wkV);
            act_c = (qL && wkV) || (qV && wkL);
            act_qm = qm;
            const bool mixed = act_s && act_c;
#define LIBRA_SB() __builtin_amdgcn_sched_barrier(0)
#define RD_A(base, ks) (*(const bf16x8*)((base) + (ks) * QSLAB))
#define RD_B(base, ks) (*(const bf16x8*)((base) + (ks) * 2048))
#define RD_T(dt, sx, hi) __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(tq + (dt) * 2 * QSLAB + (sx) * 256 + (hi) * 128))
            // one variant: 8 k-steps, fragment pairs read PD steps ahead of their MFMA (dV waves: two partial sums, so consecutive
            // MFMAs do not share an accumulator - the dK waves have no registers to spare for that); with `with_tr` the transposed fragments of the second stage ride on the last steps
            auto mm = [&](const int v, const bool with_tr) -> f32x16 {
                const char* rk = res + off_res + v * 16384 + (DK ? 32768 : 0);    // K (dV waves) / V (dK waves) of variant v
                constexpr int PD = DK ? 3 : 4;
                bf16x8 fa[8], fb[8];
                f32x16 y0, y1;
#pragma unroll
                for (int i = 0; i < PD; ++i) { fa[i] = RD_A(aq, i); fb[i] = RD_B(rk, i); }
#pragma unroll
                for (int r = 0; r < 16; ++r) { y0[r] = 0.f; y1[r] = 0.f; }
                LIBRA_SB();
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (i + PD < 8) { fa[i + PD] = RD_A(aq, i + PD); fb[i + PD] = RD_B(rk, i + PD); }
                    else if (with_tr) {                              // the 8 transposed fragments over the last PD steps
                        const int k = i + PD - 8;
#pragma unroll
                        for (int e = k * 8 / PD; e < (k + 1) * 8 / PD; ++e) { tr[e].h2[0] = RD_T(e & 3, e >> 2, 0); tr[e].h2[1] = RD_T(e & 3, e >> 2, 1); }
                    }
                    if (!DK && (i & 1)) y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[i], y1, 0, 0, 0);
                    else y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[i], y0, 0, 0, 0);
                    LIBRA_SB();
                }
                if constexpr (!DK) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) y0[r] += y1[r];
                }
                return y0;
            };
            x = mm(act_s ? 0 : 1, !mixed);
            mid();
            if (mixed) {                                             // modality boundary: second variant, merged per element
                const f32x16 y = mm(1, true);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ql = (r & 3) + 8 * (r >> 2) + 4 * fk;
                    x[r] = ((((qm >> ql) & 1u) != 0) != k_vis) ? y[r] : x[r];
                }
            }
#undef RD_A
#undef RD_B
#undef RD_T
#undef LIBRA_SB
        };
        // arithmetic of tile `it` (accumulator row r <-> query q0 + (r&3) + 8(r>>2) + 4fk ; column <-> this lane's key):
        // dV waves: P = exp2(S * scale * log2e - L * log2e), masked, handed to the dK wave through LDS; dK waves: dS = P (dP - D);
        // then the packed second-stage operand
        auto finish = [&](const int it) {
            pend = act; pend_s = act_s; pend_c = act_c; pend_qm = act_qm;
            if (!act) return;
            const int cur = (it - it0) & 1;
            const int q0 = it * 64 + qh * 32;
            asm volatile("" : "+v"(lane_v));
            const float* sLD = (const float*)(qd + cur * QD4_STAGE + 2 * QIMG) + (DK ? 64 : 0) + qh * 32 + 4 * (lane_v >> 5);   // D (dK) / L (dV)
            float* pblk = (float*)(smem + pblk_off) + lane_v * 4;
            f32x4 ld[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) ld[g] = *(const f32x4*)(sLD + 8 * g);
            if constexpr (!DK) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[4 * g + e] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[4 * g + e], p.sl2, -ld[g][e] * LOG2E));
                const bool masked = q0 < kbase_w + 31 || q0 + 32 > S || kbase_w + 32 > len;
                if (masked) {
                    const int kabs = kbase_w + l31;
                    const bool kok = kabs < len;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int qa = q0 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                        x[r] = (qa >= kabs && qa < S && kok) ? x[r] : 0.f;
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = x[4 * g + e];
                    *(f32x4*)(pblk + g * 256) = o;
                }
            } else {
                f32x4 Pq[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) Pq[g] = *(const f32x4*)(pblk + g * 256);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[4 * g + e] = Pq[g][e] * (x[4 * g + e] - ld[g][e]);
            }
#pragma unroll
            for (int sx = 0; sx < 2; ++sx)
#pragma unroll
                for (int j = 0; j < 4; ++j) pk[sx].u[j] = pack2bf(x[8 * sx + 2 * j], x[8 * sx + 2 * j + 1]);
        };

        // Per iteration two half steps.  Half A: dV waves S(it) -> P(it) -> LDS;  dK waves second stage of tile it-1, then dP(it).
        //                               Half B: dV waves second stage of tile it;  dK waves dS(it) from the handed-over P.
        // Direct-to-LDS staging of tile it+1 (into the buffer tile it-1 used: its last readers finished before this barrier):
        // 3 of 4 pieces by the dK waves behind their 16 queued MFMAs, the rest + L, D by the dV waves behind their 8.
        const int w4 = wave & 3;
        const bool tracing = p.trace != nullptr && L == (p.dbg >> 16);
        auto stamp = [&](int it, int k) {
            if (tracing) {
                const unsigned long long t = __builtin_amdgcn_s_memtime();
                if (lane == 0) p.trace[((it - it0) * 8 + wave) * 8 + k] = t;
            }
        };
        for (int it = it0; it < nqt; ++it) {
            stamp(it, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stamp(it, 1);
            __syncthreads();                                         // tile `it` landed; all readers of tile it-1's buffer are done
            stamp(it, 2);
            const int nbuf = ((it - it0) & 1) ^ 1;
            const bool more = it + 1 < nqt;
            if constexpr (DK) {
                if (pend) stage2();
                stamp(it, 3);
                first(it, [&]() { if (more) { stage_piece(nbuf, it + 1, w4); stage_piece(nbuf, it + 1, w4 + 4); stage_piece(nbuf, it + 1, w4 + 8); } });
            } else {
                first(it, [&]() { if (more) { stage_piece(nbuf, it + 1, w4 + 12); if (w4 < 2) stage_ld(nbuf, it + 1, w4); } });
                stamp(it, 3);
                finish(it);
            }
            stamp(it, 4);
            __syncthreads();                                         // P of tile `it` is in LDS
            stamp(it, 5);
            if constexpr (DK) finish(it);
            else { if (pend) stage2(); }
            stamp(it, 6);
        }
        if constexpr (DK) { if (pend) stage2(); }
    };
    if (role_dk) run_loop(std::true_type{}); else run_loop(std::false_type{});

    // ---- combine the two query halves' partial sums: waves with qh = 1 hand theirs over through LDS ----
    __syncthreads();
    {
        float* xch = (float*)smem + ((wave >> 2) * 2 + kw) * 8192;   // 32 KiB per (role, key sub-block) pair: [acc][reg quad][lane] x4
        if (qh == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 a, c;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { a[e] = acc_s[i][4 * g + e]; c[e] = acc_c[i][4 * g + e]; }
                    *(f32x4*)(xch + ((i * 4 + g) * 64 + lane) * 4) = a;
                    *(f32x4*)(xch + 4096 + ((i * 4 + g) * 64 + lane) * 4) = c;
                }
        }
        __syncthreads();
        if (qh == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 a = *(const f32x4*)(xch + ((i * 4 + g) * 64 + lane) * 4);
                    const f32x4 c = *(const f32x4*)(xch + 4096 + ((i * 4 + g) * 64 + lane) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { acc_s[i][4 * g + e] += a[e]; acc_c[i][4 * g + e] += c[e]; }
                }
        }
        __syncthreads();
    }
    if (qh != 0 || kbase_w >= S) return;
    // ---- store: each wave's two [128 d x 32 keys] blocks, transposed through a private LDS region (32 rows x 264 B)
    constexpr int OROW = 264;
    char* so = smem + wave * (32 * OROW);
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
        for (int pass_i = 0; pass_i < 8; ++pass_i) {
            const int r = pass_i * 4 + (lane >> 4);
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
    if (role_dk) { store(acc_s, p.scale, p.dk_same); store(acc_c, p.scale, p.dk_cross); }
    else { store(acc_s, 1.0f, p.dv_same); store(acc_c, 1.0f, p.dv_cross); }
}

// delta[b,h,s] = sum_d dO * O   (16 lanes per (token, head), head_dim 128)
__global__ __launch_bounds__(256) void bridge_delta_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ o_lo, long ldo_,
                                                           const bf16_t* __restrict__ dout, long lddo, float* __restrict__ delta,
                                                           int S, int H, long total_chunks) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = i < total_chunks;
    const int cpr = H * 16;
    const long row = ok ? i / cpr : 0;
    const int ch = ok ? (int)(i - row * cpr) : 0;
    float s = 0.f;
    if (ok) {
        float a[8], g[8];
        unpack8(*(const u32x4*)(o + row * ldo_ + ch * 8), a);
        unpack8(*(const u32x4*)(dout + row * lddo + ch * 8), g);
        if (o_lo) {                                           // O to ~16 mantissa bits (see libra_bridge_attn_fwd: out_lo)
            float l[8];
            unpack8(*(const u32x4*)(o_lo + row * ldo_ + ch * 8), l);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += l[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) s += a[e] * g[e];
    }
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
    if (ok && (ch & 15) == 0) {
        const int hh = ch >> 4;
        const long bb = row / S;
        const int t = (int)(row - bb * S);
        delta[(bb * H + hh) * S + t] = s;
    }
}

}  // namespace libra

using namespace libra;

extern "C" int libra_bridge_attn_bwd(const void* q, int64_t ldq, const void* k_same, int64_t ldk, const void* k_cross,
                                     int64_t ldkc, const void* v_same, int64_t ldv, const void* v_cross, int64_t ldvc,
                                     const void* out, const void* out_lo, int64_t ldout, const void* dout, int64_t lddo, const uint8_t* flag,
                                     const int32_t* kv_len, const float* lse, float* delta, void* dq, int64_t lddq,
                                     void* dk_same, void* dk_cross, void* dv_same, void* dv_cross, int64_t ldg, int64_t B,
                                     int64_t S, int64_t H, float scale, void* stream) {
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
    const long rows = B * S;
    const long total = rows * H * 16;
    hipLaunchKernelGGL(bridge_delta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)out, (const bf16_t*)out_lo, (long)ldout, (const bf16_t*)dout, (long)lddo, delta, (int)S, (int)H, total);
    if (hipGetLastError() != hipSuccess) return LIBRA_ERR_LAUNCH;
    BridgeBwdArgs a;
    a.q = (const bf16_t*)q; a.ldq = ldq; a.k_same = (const bf16_t*)k_same; a.ldk = ldk; a.k_cross = (const bf16_t*)k_cross; a.ldkc = ldkc;
    a.v_same = (const bf16_t*)v_same; a.ldv = ldv; a.v_cross = (const bf16_t*)v_cross; a.ldvc = ldvc;
    a.dout = (const bf16_t*)dout; a.ldo = lddo; a.flag = flag; a.kv_len = kv_len; a.lse = lse; a.delta = delta;
    a.dq = (bf16_t*)dq; a.lddq = lddq; a.dk_same = (bf16_t*)dk_same; a.dk_cross = (bf16_t*)dk_cross;
    a.dv_same = (bf16_t*)dv_same; a.dv_cross = (bf16_t*)dv_cross; a.ldg = ldg;
    a.B = (int)B; a.S = (int)S; a.H = (int)H; a.scale = scale; a.sl2 = scale * LOG2E;
    { const char* e = getenv("LIBRA_DKV_DBG"); a.dbg = e ? atoi(e) : 0; }
    a.trace = nullptr;
    static unsigned long long* trace_buf = nullptr;
    static int trace_calls = 0;
    if (getenv("LIBRA_DKV_TRACE")) {
        if (!trace_buf) { (void)hipMalloc((void**)&trace_buf, 64 * 64 * 8); (void)hipMemset(trace_buf, 0, 64 * 64 * 8); }
        a.trace = trace_buf;
    }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)bridge_attn_bwd_dq_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, DQ_LDS_B);
        (void)hipFuncSetAttribute((const void*)bridge_attn_bwd_dq_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, DQ_LDS_B);
        (void)hipFuncSetAttribute((const void*)bridge_attn_bwd_dkv_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, DKV_LDS_B);
        (void)hipFuncSetAttribute((const void*)bridge_attn_bwd_dkv_lin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DKV4_LDS_B);
        (void)hipFuncSetAttribute((const void*)bridge_attn_bwd_dkv_ho_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DKV5_LDS_B);
        (void)hipFuncSetAttribute((const void*)bridge_attn_bwd_dkv_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, DKV_LDS_B);
        attr_set = true;
    }
    a.n_t = (int)((S + DQ_BQ - 1) / DQ_BQ);
    long nblk = (long)B * H * a.n_t;
    if (nblk > 0x7fffffffL) return LIBRA_ERR_SHAPE;
    // LIBRA_ATTN_DKV selects the backward structure (A/B measurements inside one box visit): 1 = round-1 kernels, 3 = the same with
    // lane-constant LDS addressing (dq and dkv), 4 = 3 + the lane-linear / pipelined dK/dV kernel; read once, never written again
    static const int dkv_structure = [] { const char* e = getenv("LIBRA_ATTN_DKV"); return e ? atoi(e) : 3; }();
    if (dkv_structure >= 3)
        hipLaunchKernelGGL(bridge_attn_bwd_dq_kernel<true>, dim3((unsigned)nblk), dim3(512), DQ_LDS_B, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(bridge_attn_bwd_dq_kernel<false>, dim3((unsigned)nblk), dim3(512), DQ_LDS_B, (hipStream_t)stream, a);
    if (hipGetLastError() != hipSuccess) return LIBRA_ERR_LAUNCH;
    a.n_t = (int)((S + 63) / 64);
    nblk = (long)B * H * a.n_t;
    if (nblk > 0x7fffffffL) return LIBRA_ERR_SHAPE;
    if (dkv_structure == 1)
        hipLaunchKernelGGL(bridge_attn_bwd_dkv_kernel<false>, dim3((unsigned)nblk), dim3(512), DKV_LDS_B, (hipStream_t)stream, a);
    else if (dkv_structure == 3)
        hipLaunchKernelGGL(bridge_attn_bwd_dkv_kernel<true>, dim3((unsigned)nblk), dim3(512), DKV_LDS_B, (hipStream_t)stream, a);
    else if (dkv_structure == 4)
        hipLaunchKernelGGL(bridge_attn_bwd_dkv_lin_kernel, dim3((unsigned)nblk), dim3(512), DKV4_LDS_B, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(bridge_attn_bwd_dkv_ho_kernel, dim3((unsigned)nblk), dim3(512), DKV5_LDS_B, (hipStream_t)stream, a);
    if (a.trace && ++trace_calls == 5) {                       // debug: dump the stamps of the traced workgroup once
        static unsigned long long hostbuf[64 * 64];
        (void)hipStreamSynchronize((hipStream_t)stream);
        (void)hipMemcpy(hostbuf, trace_buf, sizeof(hostbuf), hipMemcpyDeviceToHost);
        for (int it = 0; it < 32; ++it) {
            if (!hostbuf[(it * 8) * 8]) break;
            for (int w = 0; w < 8; ++w) {
                fprintf(stderr, "trace it %d wave %d:", it, w);
                for (int k = 0; k < 7; ++k) fprintf(stderr, " %llu", hostbuf[(it * 8 + w) * 8 + k] - hostbuf[0]);
                fprintf(stderr, "\n");
            }
        }
    }
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}
