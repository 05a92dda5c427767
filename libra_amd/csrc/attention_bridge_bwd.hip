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
};

// reduction-major image of a [32 rows][128 d] tile (256-byte rows, chunk ^ ((row&3)<<2)); 8 pieces of 1 KiB.
__device__ __forceinline__ void stage_t32(const bf16_t* __restrict__ base, long ld, int row0, int nrows, char* dst,
                                          int wave, int lane, int nwaves) {
    for (int pc = wave; pc < 8; pc += nwaves) {
        const int r = pc * 4 + (lane >> 4);
        const int c = (lane & 15) ^ ((r & 3) << 2);
        int row = row0 + r; row = row < nrows ? row : nrows - 1;
        glds16(base + (long)row * ld + c * 8, dst + pc * 1024);
    }
}
// 16-byte row read (8 consecutive d of one row) from the reduction-major image
__device__ __forceinline__ bf16x8 nread_t(const char* tile, int row, int chunk) {
    return *(const bf16x8*)(tile + row * 256 + ((chunk ^ ((row & 3) << 2)) << 4));
}
// transpose read: A-operand fragment X^T[d = 32*dt + l31][rows 16*sx + 4*fk + {0..3, 8..11}]
__device__ __forceinline__ bf16x8 tread_t(const char* tile, int lane, int dt, int sx) {
    const int pp = lane & 15, g16 = (lane >> 4) & 1, fk = lane >> 5;
    const int toff = (((((dt ^ (pp >> 2)) & 3) << 2) | (2 * g16 + ((pp & 3) >> 1))) << 4) + ((pp & 1) << 3);
    const char* a = tile + sx * 4096 + (4 * fk + (pp >> 2)) * 256 + toff;
    union { bf16x8 v; s16x4 h2[2]; } u;
    u.h2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(a));
    u.h2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(a + 2048));
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
// dQ pass
constexpr int DQ_VAR = 16384;                 // K tile 8 KiB + V tile 8 KiB (32 keys)
constexpr int DQ_STAGE_B = 2 * DQ_VAR;
constexpr int DQ_LDS_B = 2 * DQ_STAGE_B + 1024;

__global__ __launch_bounds__(256, 2) void bridge_attn_bwd_dq_kernel(const BridgeBwdArgs p) {
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
    const int q0w = qt * 128 + wave * 32;
    const bool active = q0w < S;
    int q = q0w + l31;
    const bool qin = q < S;
    q = qin ? q : S - 1;

    const int ntile_all = (S + 31) / 32;
    for (int t = wave; t < ntile_all; t += 4) {
        const int key = t * 32 + l31;
        const bool vis = (key < S) && (fk == 0) && p.flag[tok0 + key] != 0;
        const unsigned long long bal = __ballot(vis);
        if (lane == 0) kmask[t] = (unsigned)bal;
    }
    const bool q_vis = p.flag[tok0 + q] != 0;
    if (tid < 2) qpres[tid] = 0;
    __syncthreads();
    if (__ballot(qin && fk == 0 && q_vis)) if (lane == 0) atomicOr(&qpres[1], 1);
    if (__ballot(qin && fk == 0 && !q_vis)) if (lane == 0) atomicOr(&qpres[0], 1);
    __syncthreads();
    const bool blkL = qpres[0] != 0, blkV = qpres[1] != 0;
    const bool wV = __ballot(q_vis && qin) != 0, wL = __ballot(!q_vis && qin) != 0;

    bf16x8 qf[8], dof[8];
    {
        const bf16_t* qp = p.q + (tok0 + q) * p.ldq + h * D128 + fk * 8;
        const bf16_t* dp = p.dout + (tok0 + q) * p.ldo + h * D128 + fk * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) { qf[ks] = *(const bf16x8*)(qp + ks * 16); dof[ks] = *(const bf16x8*)(dp + ks * 16); }
    }
    const long sidx = ((long)b * p.H + h) * S + q;
    const float Lq2 = p.lse[sidx] * LOG2E;
    const float Dq = p.delta[sidx];
    const bf16_t* ks_base = p.k_same + tok0 * p.ldk + h * D128;
    const bf16_t* kc_base = p.k_cross + tok0 * p.ldkc + h * D128;
    const bf16_t* vs_base = p.v_same + tok0 * p.ldv + h * D128;
    const bf16_t* vc_base = p.v_cross + tok0 * p.ldvc + h * D128;

    f32x16 dq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;

    int kend = (qt + 1) * 128; kend = kend < S ? kend : S;
    const int nkt = (kend + 31) / 32;
    auto needs = [&](int t, bool bL, bool bV, bool& same, bool& cross) {
        const unsigned km = kmask[t];
        int nvalid = S - t * 32; nvalid = nvalid > 32 ? 32 : nvalid;
        const unsigned full = nvalid >= 32 ? 0xffffffffu : ((1u << nvalid) - 1u);
        const bool kV = (km & full) != 0, kL = ((~km) & full) != 0;
        same = (bL && kL) || (bV && kV);
        cross = (bL && kV) || (bV && kL);
    };
    auto stage = [&](int buf, int t) {
        bool same, cross;
        needs(t, blkL, blkV, same, cross);
        char* dst = smem + buf * DQ_STAGE_B;
        if (same) { stage_t32(ks_base, p.ldk, t * 32, S, dst, wave, lane, 4); stage_t32(vs_base, p.ldv, t * 32, S, dst + 8192, wave, lane, 4); }
        if (cross) { stage_t32(kc_base, p.ldkc, t * 32, S, dst + DQ_VAR, wave, lane, 4); stage_t32(vc_base, p.ldvc, t * 32, S, dst + DQ_VAR + 8192, wave, lane, 4); }
    };
    stage(0, 0);

    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int cur = kt & 1;
        if (kt + 1 < nkt) stage(cur ^ 1, kt + 1);
        if (!active) continue;
        const int kv0 = kt * 32;
        if (kv0 > q0w + 31) continue;
        bool wsame, wcross;
        needs(kt, wL, wV, wsame, wcross);
        const unsigned km = kmask[kt];
        const char* sks = smem + cur * DQ_STAGE_B;
        const char* skc = sks + DQ_VAR;
        f32x16 s_s, s_c, p_s, p_c;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s_s[r] = 0.f; s_c[r] = 0.f; p_s[r] = 0.f; p_c[r] = 0.f; }
        if (wsame) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) s_s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(nread_t(sks, l31, 2 * ks + fk), qf[ks], s_s, 0, 0, 0);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) p_s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(nread_t(sks + 8192, l31, 2 * ks + fk), dof[ks], p_s, 0, 0, 0);
        }
        if (wcross) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) s_c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(nread_t(skc, l31, 2 * ks + fk), qf[ks], s_c, 0, 0, 0);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) p_c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(nread_t(skc + 8192, l31, 2 * ks + fk), dof[ks], p_c, 0, 0, 0);
        }
        const int qabs = q0w + l31;
        unsigned crossbits = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kl = (r & 3) + 8 * (r >> 2) + 4 * fk;
            const int key = kv0 + kl;
            const bool cr = (((km >> kl) & 1u) != 0) != q_vis;
            const float sv = cr ? s_c[r] : s_s[r];
            const float dpv = cr ? p_c[r] : p_s[r];
            const float pr = (key <= qabs && key < len) ? __builtin_amdgcn_exp2f(sv * p.sl2 - Lq2) : 0.f;
            s_s[r] = pr * (dpv - Dq);                               // dS^T
            crossbits |= (cr ? 1u : 0u) << r;
        }
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
            bf16x8 ds_same, ds_cross;
            split_pack(s_s, crossbits, sx, ds_same, ds_cross);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                if (wsame) dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tread_t(sks, lane, dt, sx), ds_same, dq[dt], 0, 0, 0);
                if (wcross) dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tread_t(skc, lane, dt, sx), ds_cross, dq[dt], 0, 0, 0);
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
// dK / dV pass
constexpr int KV_RES = 4 * 16384;             // resident K_same, K_cross, V_same, V_cross: [64 keys][128 d] each
constexpr int QD_STAGE = 4 * 8192 + 256;      // Q row image, Q reduction-major image, dO row image, dO r-m image, L[32], D[32]
constexpr int DKV_LDS_B = KV_RES + 2 * QD_STAGE + 1024;

// resident operand tile: two N-type [64 keys][64 d] sub-tiles (128-byte rows, chunk ^ ((row>>1)&7)), 16 KiB
__device__ __forceinline__ void stage_res64(const bf16_t* __restrict__ base, long ld, int key0, int S, char* dst, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pc = wave * 4 + j;                 // 16 pieces of 1 KiB: sub-tile pc>>3, rows 8*(pc&7)..
        const int sub = pc >> 3, r = (pc & 7) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int key = key0 + r; key = key < S ? key : S - 1;
        glds16(base + (long)key * ld + sub * 64 + c * 8, dst + pc * 1024);
    }
}
// row image of a [32 rows][128 d] tile: two N-type [32][64] sub-tiles (4 KiB each); 8 pieces of 1 KiB over 4 waves
__device__ __forceinline__ void stage_n32(const bf16_t* __restrict__ base, long ld, int row0, int nrows, char* dst, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pc = wave * 2 + j;
        const int sub = pc >> 2, r = (pc & 3) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int row = row0 + r; row = row < nrows ? row : nrows - 1;
        glds16(base + (long)row * ld + sub * 64 + c * 8, dst + pc * 1024);
    }
}
// fragment (row, 8 consecutive d of k-step ks) of a row image whose sub-tiles are `sub_bytes` apart
__device__ __forceinline__ bf16x8 nfrag(const char* tile, int row, int ks, int fk, int sub_bytes) {
    const int sub = ks >> 2, c = (2 * (ks & 3) + fk) ^ ((row >> 1) & 7);
    return *(const bf16x8*)(tile + sub * sub_bytes + row * 128 + (c << 4));
}

__global__ __launch_bounds__(256, 1) void bridge_attn_bwd_dkv_kernel(const BridgeBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* res = smem;                                            // Ks, Kc, Vs, Vc
    char* qd = smem + KV_RES;
    unsigned* qmask = (unsigned*)(smem + KV_RES + 2 * QD_STAGE); // per 32-query tile: bit i = query i is a vision token
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kw = wave & 1;
    const bool role_dk = (wave >> 1) != 0;                       // waves 0,1: dV; waves 2,3: dK
    const int fk = lane >> 5, l31 = lane & 31;
    const int nblk = p.B * p.H * p.n_t;
    const int L = xcd_remap(blockIdx.x, nblk);
    const int ktile = L % p.n_t;                                 // low key tiles see the most queries: they come first
    const int bh = L / p.n_t;
    const int h = bh % p.H, b = bh / p.H;
    const int S = p.S;
    const long tok0 = (long)b * S;
    const int len = p.kv_len ? p.kv_len[b] : S;
    const int key0 = ktile * 64;
    const int kbase_w = key0 + kw * 32;
    int key = kbase_w + l31;
    const bool kin = key < S;
    key = kin ? key : S - 1;
    const bool k_vis = p.flag[tok0 + key] != 0;
    const bool wkV = __ballot(k_vis && kin) != 0, wkL = __ballot(!k_vis && kin) != 0;

    const int ntile_all = (S + 31) / 32;
    for (int t = wave; t < ntile_all; t += 4) {
        const int qq = t * 32 + l31;
        const bool vis = (qq < S) && (fk == 0) && p.flag[tok0 + qq] != 0;
        const unsigned long long bal = __ballot(vis);
        if (lane == 0) qmask[t] = (unsigned)bal;
    }
    stage_res64(p.k_same + tok0 * p.ldk + h * D128, p.ldk, key0, S, res, wave, lane);
    stage_res64(p.k_cross + tok0 * p.ldkc + h * D128, p.ldkc, key0, S, res + 16384, wave, lane);
    stage_res64(p.v_same + tok0 * p.ldv + h * D128, p.ldv, key0, S, res + 32768, wave, lane);
    stage_res64(p.v_cross + tok0 * p.ldvc + h * D128, p.ldvc, key0, S, res + 49152, wave, lane);

    const bf16_t* qbase = p.q + tok0 * p.ldq + h * D128;
    const bf16_t* dobase = p.dout + tok0 * p.ldo + h * D128;
    const float* lbase = p.lse + ((long)b * p.H + h) * S;
    const float* dbase = p.delta + ((long)b * p.H + h) * S;
    auto stage_q = [&](int buf, int t) {
        char* dst = qd + buf * QD_STAGE;
        stage_n32(qbase, p.ldq, t * 32, S, dst, wave, lane);                    // Q rows      (first-stage A operand)
        stage_t32(qbase, p.ldq, t * 32, S, dst + 8192, wave, lane, 4);          // Q, r-m image (Q^T fragments for dK)
        stage_n32(dobase, p.ldo, t * 32, S, dst + 16384, wave, lane);           // dO rows
        stage_t32(dobase, p.ldo, t * 32, S, dst + 24576, wave, lane, 4);        // dO, r-m image (dO^T fragments for dV)
        if (wave < 2 && lane < 32) {
            int qi = t * 32 + lane; qi = qi < S ? qi : S - 1;
            const float* src = (wave == 0 ? lbase : dbase) + qi;
            __builtin_amdgcn_global_load_lds((const LIBRA_GLB void*)src, (LIBRA_LDS void*)(dst + 32768 + wave * 128), 4, 0, 0);
        }
    };
    const int qt0 = key0 / 32;                                   // first query tile that can see this key block
    const int nqt = ntile_all;
    f32x16 acc_s[4], acc_c[4];                                   // dV (or dK) for the same / cross variant, [128 d x 32 keys]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_s[i][r] = 0.f; acc_c[i][r] = 0.f; }
    if (qt0 < nqt) stage_q(0, qt0);

    const char* rK = res + kw * 32 * 128;                         // this wave's 32 key rows inside each 64-row sub-tile
    for (int it = qt0; it < nqt; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int cur = (it - qt0) & 1;
        if (it + 1 < nqt) stage_q(cur ^ 1, it + 1);
        const int q0 = it * 32;
        if (kbase_w >= S || q0 + 31 < kbase_w) continue;          // no (query >= key) pair for this wave in the tile
        const char* sqn = qd + cur * QD_STAGE;
        const char* sqt = sqn + 8192;
        const char* sdn = sqn + 16384;
        const char* sdt = sqn + 24576;
        const float* sL = (const float*)(sqn + 32768);
        const float* sD = sL + 32;
        const unsigned qm = qmask[it];
        int nvalid = S - q0; nvalid = nvalid > 32 ? 32 : nvalid;
        const unsigned full = nvalid >= 32 ? 0xffffffffu : ((1u << nvalid) - 1u);
        const bool qV = (qm & full) != 0, qL = ((~qm) & full) != 0;
        const bool wsame = (qL && wkL) || (qV && wkV);
        const bool wcross = (qL && wkV) || (qV && wkL);
        f32x16 s_s, s_c, p_s, p_c;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s_s[r] = 0.f; s_c[r] = 0.f; p_s[r] = 0.f; p_c[r] = 0.f; }
        // S = Q K^T (both roles), dP = dO V^T (dK waves only): A = row image of the streamed tile, B = resident fragments
        if (wsame) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                s_s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(nfrag(sqn, l31, ks, fk, 4096), nfrag(rK, l31, ks, fk, 8192), s_s, 0, 0, 0);
            if (role_dk) {
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
                    p_s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(nfrag(sdn, l31, ks, fk, 4096), nfrag(rK + 32768, l31, ks, fk, 8192), p_s, 0, 0, 0);
            }
        }
        if (wcross) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                s_c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(nfrag(sqn, l31, ks, fk, 4096), nfrag(rK + 16384, l31, ks, fk, 8192), s_c, 0, 0, 0);
            if (role_dk) {
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
                    p_c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(nfrag(sdn, l31, ks, fk, 4096), nfrag(rK + 49152, l31, ks, fk, 8192), p_c, 0, 0, 0);
            }
        }
        // accumulator row r <-> query q0 + (r&3) + 8(r>>2) + 4fk ; column <-> this lane's key
        const int kabs = kbase_w + l31;
        unsigned crossbits = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ql = 8 * g + 4 * fk;
            const f32x4 Lv = *(const f32x4*)(sL + ql);
            const f32x4 Dv = *(const f32x4*)(sD + ql);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g + e;
                const int qa = q0 + ql + e;
                const bool cr = (((qm >> (ql + e)) & 1u) != 0) != k_vis;
                const float sv = cr ? s_c[r] : s_s[r];
                const float dpv = cr ? p_c[r] : p_s[r];
                const float pr = (qa >= kabs && qa < S && kabs < len && kin)
                                     ? __builtin_amdgcn_exp2f(sv * p.sl2 - Lv[e] * LOG2E) : 0.f;
                s_s[r] = role_dk ? pr * (dpv - Dv[e]) : pr;       // dS for the dK waves, P for the dV waves
                crossbits |= (cr ? 1u : 0u) << r;
            }
        }
        const char* st = role_dk ? sqt : sdt;                     // Q^T fragments (dK) or dO^T fragments (dV)
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
            bf16x8 bS, bC;
            split_pack(s_s, crossbits, sx, bS, bC);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 af = tread_t(st, lane, dt, sx);
                if (wsame) acc_s[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bS, acc_s[dt], 0, 0, 0);
                if (wcross) acc_c[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bC, acc_c[dt], 0, 0, 0);
            }
        }
    }
    // ---- store: each wave's two [128 d x 32 keys] blocks, transposed through a private LDS region (32 rows x 264 B)
    __syncthreads();
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

// delta[b,h,s] = sum_d dO * O   (16 lanes per (token, head), head_dim 128)
__global__ __launch_bounds__(256) void bridge_delta_kernel(const bf16_t* __restrict__ o, long ldo_, const bf16_t* __restrict__ dout,
                                                           long lddo, float* __restrict__ delta, int S, int H, long total_chunks) {
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
                                     const void* out, int64_t ldout, const void* dout, int64_t lddo, const uint8_t* flag,
                                     const int32_t* kv_len, const float* lse, float* delta, void* dq, int64_t lddq,
                                     void* dk_same, void* dk_cross, void* dv_same, void* dv_cross, int64_t ldg, int64_t B,
                                     int64_t S, int64_t H, float scale, void* stream) {
    if (B <= 0 || S <= 0) return LIBRA_OK;
    const int64_t HD = H * D128;
    if (H <= 0 || S > 4096 || ldq < HD || ldk < HD || ldkc < HD || ldv < HD || ldvc < HD || ldout < HD || lddo < HD || lddq < HD || ldg < HD)
        return LIBRA_ERR_SHAPE;
    if ((ldq | ldk | ldkc | ldv | ldvc | ldout | lddo | lddq | ldg) % 8) return LIBRA_ERR_ALIGN;
    if (!q || !k_same || !k_cross || !v_same || !v_cross || !out || !dout || !flag || !lse || !delta || !dq || !dk_same ||
        !dk_cross || !dv_same || !dv_cross) return LIBRA_ERR_ALIGN;
    if (((uintptr_t)q | (uintptr_t)k_same | (uintptr_t)k_cross | (uintptr_t)v_same | (uintptr_t)v_cross | (uintptr_t)out |
         (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk_same | (uintptr_t)dk_cross | (uintptr_t)dv_same | (uintptr_t)dv_cross) & 15)
        return LIBRA_ERR_ALIGN;
    const long rows = B * S;
    const long total = rows * H * 16;
    hipLaunchKernelGGL(bridge_delta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)out, (long)ldout, (const bf16_t*)dout, (long)lddo, delta, (int)S, (int)H, total);
    if (hipGetLastError() != hipSuccess) return LIBRA_ERR_LAUNCH;
    BridgeBwdArgs a;
    a.q = (const bf16_t*)q; a.ldq = ldq; a.k_same = (const bf16_t*)k_same; a.ldk = ldk; a.k_cross = (const bf16_t*)k_cross; a.ldkc = ldkc;
    a.v_same = (const bf16_t*)v_same; a.ldv = ldv; a.v_cross = (const bf16_t*)v_cross; a.ldvc = ldvc;
    a.dout = (const bf16_t*)dout; a.ldo = lddo; a.flag = flag; a.kv_len = kv_len; a.lse = lse; a.delta = delta;
    a.dq = (bf16_t*)dq; a.lddq = lddq; a.dk_same = (bf16_t*)dk_same; a.dk_cross = (bf16_t*)dk_cross;
    a.dv_same = (bf16_t*)dv_same; a.dv_cross = (bf16_t*)dv_cross; a.ldg = ldg;
    a.B = (int)B; a.S = (int)S; a.H = (int)H; a.scale = scale; a.sl2 = scale * LOG2E;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)bridge_attn_bwd_dq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DQ_LDS_B);
        (void)hipFuncSetAttribute((const void*)bridge_attn_bwd_dkv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DKV_LDS_B);
        attr_set = true;
    }
    a.n_t = (int)((S + 127) / 128);
    long nblk = (long)B * H * a.n_t;
    if (nblk > 0x7fffffffL) return LIBRA_ERR_SHAPE;
    hipLaunchKernelGGL(bridge_attn_bwd_dq_kernel, dim3((unsigned)nblk), dim3(256), DQ_LDS_B, (hipStream_t)stream, a);
    if (hipGetLastError() != hipSuccess) return LIBRA_ERR_LAUNCH;
    a.n_t = (int)((S + 63) / 64);
    nblk = (long)B * H * a.n_t;
    if (nblk > 0x7fffffffL) return LIBRA_ERR_SHAPE;
    hipLaunchKernelGGL(bridge_attn_bwd_dkv_kernel, dim3((unsigned)nblk), dim3(256), DKV_LDS_B, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}
