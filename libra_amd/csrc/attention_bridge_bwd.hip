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

#ifndef LIBRA_DKV_PERSIST       // 1: persistent dK/dV workgroups with the rotation schedule (measured SLOWER in round 5: see DESIGN)
#define LIBRA_DKV_PERSIST 0
#endif

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
// dQ pass, round-5 structure = the forward kernel's (attention_bridge.hip): 8 waves x 32 queries per workgroup, the work list is
// a list of UNITS = (64-key tile, operand variant) with a per-wave mode (skip / plain / masked) decided once in the prologue,
// a unit is a DS phase (VALU only: P = exp2(S sl2 - L), mask, dS = P (dP - D), pack) and an M phase
//     [dQ^T += K_u^T dS_u^T : 16 MFMAs, transposed reads]  [S_{u+1}^T = K_{u+1} Q^T, dP_{u+1}^T = V_{u+1} dO^T : 32 MFMAs, row reads]
// waves 0-3 and 4-7 (one of each per SIMD) run the sequence one phase apart, so a SIMD's matrix pipe works for one wave while the
// other does its dS arithmetic; a unit's K | V tile (32 KiB, one variant) travels through a ring of 4 stages, requested two units
// ahead from the DS phases only; persistent workgroups with the static rotation schedule.  The K tile is staged once in the
// reduction-major image and read BOTH ways (16-byte row reads for S, LDS transpose reads for dQ), V in the same image (row reads).
constexpr int DQ_BQ = 256;
constexpr int DQ_TILE = 16384;                // one [64 keys][128 d] operand tile
constexpr int DQ_STAGE_B = 2 * DQ_TILE;       // ring stage: K tile | V tile of one unit
constexpr int DQ_NSTAGE = 4;
constexpr int DQ_MASK_OFF = DQ_NSTAGE * DQ_STAGE_B;   // key-modality words (<= 130; 1 KiB)
constexpr int DQ_BLK_OFF = DQ_MASK_OFF + 1024;        // block-level unit sets (4 words)
constexpr int DQ_TAB_OFF = DQ_BLK_OFF + 64;           // per-wave unit tables: 8 x 128 x 2 B
constexpr int DQ_LDS_B = DQ_TAB_OFF + 8 * 256;

typedef unsigned long long u64;
__device__ __forceinline__ u64 bits_below64(int n) { return n >= 64 ? ~0ull : (n <= 0 ? 0ull : ((1ull << n) - 1ull)); }

__global__ __launch_bounds__(512, 2) void bridge_attn_bwd_dq_kernel(const BridgeBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* kmask = (unsigned*)(smem + DQ_MASK_OFF);
    unsigned* blk = (unsigned*)(smem + DQ_BLK_OFF);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int fk = lane >> 5, l31 = lane & 31;
    // persistent workgroups with the forward kernel's static rotation schedule (bridge_attn_fwd_kernel): item i = w + k P is
    // (sequence, head) i / n_t, query block (i + k) mod n_t
    const int nitems = p.B * p.H * p.n_t;
    const int P = (int)gridDim.x;
    const int w_id = xcd_remap(blockIdx.x, P);
#pragma unroll 1
    for (int step = 0, item = w_id; item < nitems; ++step, item += P) {
    const int qt = p.n_t - 1 - ((item % p.n_t + step) % p.n_t);
    const int bh = item / p.n_t;
    const int h = bh % p.H, b = bh / p.H;
    const int S = p.S;
    const long tok0 = (long)b * S;
    int len = p.kv_len ? p.kv_len[b] : S;
    len = len < S ? len : S;
    const int q0w = qt * DQ_BQ + wave * 32;
    const bool active = q0w < S;
    int q = q0w + l31;
    const bool qin = q < S;
    q = qin ? q : S - 1;
    int kend = (qt + 1) * DQ_BQ; kend = kend < S ? kend : S;
    const int nkt = (kend + 63) / 64;                               // <= 64 (S <= 4096)

    const bf16_t* ks_base = p.k_same + tok0 * p.ldk + h * D128;
    const bf16_t* kc_base = p.k_cross + tok0 * p.ldkc + h * D128;
    const bf16_t* vs_base = p.v_same + tok0 * p.ldv + h * D128;
    const bf16_t* vc_base = p.v_cross + tok0 * p.ldvc + h * D128;
    const unsigned lds0 = (unsigned)(unsigned long)(LIBRA_LDS char*)smem;

    // ---- direct-to-LDS pieces: a 16-KiB tile is 16 pieces of 1 KiB (4 rows x 256 B); wave w moves pieces 2w, 2w + 1 of the K and
    // of the V tile of a unit: rows 8w + (lane >> 4) and + 4, 16-byte chunk (lane & 15) ^ tswz(row).  tswz(row + 4) = tswz(row) ^ 1
    // here (row >> 2 is even), so the second piece's per-lane offset is the first one's +- 16 bytes and its row step goes into the
    // wave-uniform base: 4 offset registers + 1 for the four operands.
    const int r0 = wave * 8 + (lane >> 4);
    const int c0 = (lane & 15) ^ tswz(r0);
    const int d16 = (c0 & 1) ? -16 : 16;
    const unsigned ldkb_s = (unsigned)(p.ldk * 2), ldkb_c = (unsigned)(p.ldkc * 2), ldvb_s = (unsigned)(p.ldv * 2), ldvb_c = (unsigned)(p.ldvc * 2);
    const unsigned oKs = (unsigned)(lane >> 4) * ldkb_s + (unsigned)(c0 << 4), oKc = (unsigned)(lane >> 4) * ldkb_c + (unsigned)(c0 << 4);
    const unsigned oVs = (unsigned)(lane >> 4) * ldvb_s + (unsigned)(c0 << 4), oVc = (unsigned)(lane >> 4) * ldvb_c + (unsigned)(c0 << 4);
    auto stage_unit = [&](const int t, const int var, const int st) {
        const unsigned dst = lds0 + (unsigned)(st * DQ_STAGE_B + wave * 2048);
        const long ldk_ = var ? p.ldkc : p.ldk, ldv_ = var ? p.ldvc : p.ldv;
        const bf16_t* kbase = (var ? kc_base : ks_base) + ((long)t * 64 + wave * 8) * ldk_;
        const bf16_t* vbase = (var ? vc_base : vs_base) + ((long)t * 64 + wave * 8) * ldv_;
        if (t * 64 + 64 <= S) {
            const unsigned ok = var ? oKc : oKs, ov = var ? oVc : oVs;
            glds16_off_at(kbase, ok, dst); glds16_off_at(kbase + 4 * ldk_, ok + (unsigned)d16, dst + 1024);
            glds16_off_at(vbase, ov, dst + DQ_TILE); glds16_off_at(vbase + 4 * ldv_, ov + (unsigned)d16, dst + DQ_TILE + 1024);
            return;
        }
        // the sequence's last, ragged tile: rows clamped to the last token (their keys are masked); offsets from the tile's first row
        const int lim = S - 1 - t * 64;                             // >= 0: the tile holds at least one token
        const unsigned ldkb = var ? ldkb_c : ldkb_s, ldvb = var ? ldvb_c : ldvb_s;
        const bf16_t* kt_ = kbase - (long)(wave * 8) * ldk_;
        const bf16_t* vt_ = vbase - (long)(wave * 8) * ldv_;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int row = r0 + 4 * j;
            row = row < lim ? row : lim;
            const unsigned cc = (unsigned)((c0 ^ j) << 4);
            glds16_off_at(kt_, (unsigned)row * ldkb + cc, dst + j * 1024);
            glds16_off_at(vt_, (unsigned)row * ldvb + cc, dst + DQ_TILE + j * 1024);
        }
    };

    // ---- prologue: every per-lane global operand is requested before the first wait (one round trip; with one workgroup per CU
    // nothing else covers them); the first two stages are requested on the forward kernel's guess (tile 0, same), (tile 0, cross)
    stage_unit(0, 0, 0);
    stage_unit(0, 1, 1);
    const int q_vis_raw = p.flag[tok0 + q];
    bf16x8 qf[8], dof[8];
    {
        const bf16_t* qp = p.q + (tok0 + q) * p.ldq + h * D128 + fk * 8;
        const bf16_t* dp_ = p.dout + (tok0 + q) * p.ldo + h * D128 + fk * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) { qf[ks] = *(const bf16x8*)(qp + ks * 16); dof[ks] = *(const bf16x8*)(dp_ + ks * 16); }
    }
    const long sidx = ((long)b * p.H + h) * S + q;
    float nLq2 = -p.lse[sidx] * LOG2E;
    // D = sum_d dO . O of this lane's query (the softmax-backward row term) from the dO fragments already in flight + the O row:
    // this lane's 64 channels here, the other half one permlane swap away; the dK / dV pass reads what the fk = 0 lanes store
    float Dq = 0.f;
    {
        bf16x8 of[8], ol[8];
        const bf16_t* op = p.out + (tok0 + q) * p.ldout + h * D128 + fk * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) of[ks] = *(const bf16x8*)(op + ks * 16);
        if (p.out_lo) {
            const bf16_t* lp = p.out_lo + (tok0 + q) * p.ldout + h * D128 + fk * 8;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) ol[ks] = *(const bf16x8*)(lp + ks * 16);
        }
        modality_masks(p.flag + tok0, S, kmask, tid, 512);
        if (tid < 8) blk[tid] = 0;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float o = bf2f((bf16_t)of[ks][e]);
                if (p.out_lo) o += bf2f((bf16_t)ol[ks][e]);
                Dq = __builtin_fmaf(o, bf2f((bf16_t)dof[ks][e]), Dq);
            }
    }
    Dq = half_swap_sum(Dq);
    if (fk == 0 && qin) p.delta[sidx] = Dq;
    const bool q_vis = q_vis_raw != 0;
    __syncthreads();

    // ---- per-wave classification of every key tile, lane = tile (as the forward kernel; no left padding in the backward)
    const bool wV = __ballot(q_vis && qin) != 0, wL = __ballot(!q_vis && qin) != 0;
    unsigned m_same = 0, m_cross = 0;
    {
        const int kv0 = lane * 64;
        const u64 mm = (u64)kmask[2 * lane] | ((u64)kmask[2 * lane + 1] << 32);
        const u64 rng = bits_below64(len - kv0);
        const bool kV = (mm & rng) != 0, kL = (~mm & rng) != 0;
        const bool in = active && lane < nkt && kv0 <= q0w + 31;
        const bool wsame = in && ((wL && kL) || (wV && kV)), wcross = in && ((wL && kV) || (wV && kL));
        const bool full = kv0 + 63 <= q0w && kv0 + 64 <= len;
        const bool plain = full && !(wsame && wcross);
        m_same = wsame ? (plain ? 1u : 2u) : 0u;
        m_cross = wcross ? (plain ? 1u : 2u) : 0u;
        const u64 b_same = __ballot(wsame), b_cross = __ballot(wcross);
        if (lane == 0) {
            if ((unsigned)b_same) atomicOr(&blk[0], (unsigned)b_same);
            if ((unsigned)(b_same >> 32)) atomicOr(&blk[1], (unsigned)(b_same >> 32));
            if ((unsigned)b_cross) atomicOr(&blk[2], (unsigned)b_cross);
            if ((unsigned)(b_cross >> 32)) atomicOr(&blk[3], (unsigned)(b_cross >> 32));
        }
    }
    __syncthreads();
    const u64 same_blk = (u64)(unsigned)__builtin_amdgcn_readfirstlane((int)blk[0]) | ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)blk[1]) << 32);
    const u64 cross_blk = (u64)(unsigned)__builtin_amdgcn_readfirstlane((int)blk[2]) | ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)blk[3]) << 32);
    const int U = __popcll(same_blk) + __popcll(cross_blk);         // units of this workgroup (<= 128)
    unsigned tab0, tab1;                                            // lane i: entry of unit i / unit 64 + i (0 past the end)
    {
        unsigned short* tab = (unsigned short*)(smem + DQ_TAB_OFF) + wave * 128;
        if (lane < nkt) {
            const u64 below = bits_below64(lane);
            const int u0 = __popcll(same_blk & below) + __popcll(cross_blk & below);
            const bool hs = (same_blk >> lane) & 1ull, hc = (cross_blk >> lane) & 1ull;
            if (hs) tab[u0] = (unsigned short)(m_same | (unsigned)(lane << 3));
            if (hc) tab[u0 + (hs ? 1 : 0)] = (unsigned short)(m_cross | 4u | (unsigned)(lane << 3));
        }
        tab0 = lane < U ? tab[lane] : 0u;                           // (same wave, in-order LDS queue: no barrier)
        tab1 = lane + 64 < U ? tab[lane + 64] : 0u;
    }
    auto entry = [&](const int u) -> unsigned {                     // u wave-uniform; 0 past the end
        const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)tab0, u & 63), c = (unsigned)__builtin_amdgcn_readlane((int)tab1, u & 63);
        return u < 64 ? a : (u < 128 ? c : 0u);
    };
    {
        const unsigned e0 = entry(0), e1 = entry(1);
        const bool ok0 = U < 1 || (e0 >> 2) == 0u, ok1 = U < 2 || (e1 >> 2) == 1u;       // (tile 0, same) / (tile 0, cross)
        if (!ok0 || !ok1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (!ok0) stage_unit((int)(e0 >> 3), (int)((e0 >> 2) & 1u), 0);
            if (!ok1) stage_unit((int)(e1 >> 3), (int)((e1 >> 2) & 1u), 1);
        }
    }

    // ---- fragment addressing: per-lane constants XOR a compile-time constant
    int xr, xt0, xt1;
    {
        const int pp = lane & 15, g16 = (lane >> 4) & 1;
        const int r1 = 4 * fk + (pp >> 2), lp = 2 * g16 + ((pp & 3) >> 1);
        xr = l31 * 256 + ((fk ^ tswz(l31)) << 4);
        xt0 = r1 * 256 + ((pp & 1) << 3) + ((lp ^ tswz(r1)) << 4);
        xt1 = r1 * 256 + ((pp & 1) << 3) + ((lp ^ tswz(r1 + 8)) << 4) + 2048;
    }

    f32x16 dq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;
    f32x16 sA, sB, dA, dB;                                          // S^T and dP^T of the unit in flight: key halves 0 / 1
    union PK { bf16x8 v; unsigned u[4]; };
    PK pk[4];                                                       // dS^T of the unit in flight as the four 16-key B operands
    const int qabs = q0w + l31;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) { pin(qf[ks]); pin(dof[ks]); }   // prologue loads have landed before the loop's LDS-DMA traffic
    pin(nLq2); pin(Dq);

    union VA { bf16x8 v; s16x4 h2[2]; };
    // M phase.  Fragment i: 0-15 = K_u^T (16-key step i >> 2, 32-line d block i & 3), transposed reads; 16-47 = row reads of the next
    // unit, j = i - 16: k-step j >> 2, operand j & 3 = K half 0 / K half 1 / V half 0 / V half 1 (four accumulators in rotation).
    // ONE ring of NF fragments, each requested NF - 1 MFMAs ahead of its consumer.
    constexpr int NF = 4;
    auto m_phase = [&](auto dq_c, auto nx_c, const char* cur, const char* nxt) {
        constexpr bool DQ = decltype(dq_c)::value, NX = decltype(nx_c)::value;
        constexpr int N = (DQ ? 16 : 0) + (NX ? 32 : 0), I0 = DQ ? 0 : 16;
        bf16x8 F[NF];
        auto fread = [&](const int i) -> bf16x8 {
            if (i < 16) {
                const char* a = cur + (i >> 2) * 4096;
                VA t;
                t.h2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(a + (xt0 ^ ((i & 3) << 6))));
                t.h2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(a + (xt1 ^ ((i & 3) << 6))));
                return t.v;
            }
            const int j = i - 16, ks = j >> 2, w = j & 3;
            return *(const bf16x8*)(nxt + (w >> 1) * DQ_TILE + (w & 1) * 8192 + (xr ^ (ks << 5)));
        };
        if constexpr (N > 0) {
#pragma unroll
            for (int n = 0; n < NF; ++n) F[n] = fread(I0 + n);
            if constexpr (NX) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { sA[r] = 0.f; sB[r] = 0.f; dA[r] = 0.f; dB[r] = 0.f; }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < N; ++n) {
                const int i = I0 + n;
                if (i < 16) dq[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[n % NF], pk[i >> 2].v, dq[i & 3], 0, 0, 0);
                else {
                    const int j = i - 16, ks = j >> 2, w = j & 3;
                    if (w == 0) sA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[n % NF], qf[ks], sA, 0, 0, 0);
                    else if (w == 1) sB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[n % NF], qf[ks], sB, 0, 0, 0);
                    else if (w == 2) dA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[n % NF], dof[ks], dA, 0, 0, 0);
                    else dB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[n % NF], dof[ks], dB, 0, 0, 0);
                }
                if (n + NF < N) F[n % NF] = fread(i + NF);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // per-lane key mask of a MASKED unit: key <= query (causal), key < len (padding), and the pair's modality relation == the
    // unit's variant.  One 32-bit word per key half, shifted by 4 fk so that the bit positions below are compile-time.
    auto key_masks = [&](const int kt, const int var, unsigned& v0, unsigned& v1) {
        const int kv0 = kt * 64;
        const unsigned km0 = (unsigned)__builtin_amdgcn_readfirstlane((int)kmask[2 * kt]);
        const unsigned km1 = (unsigned)__builtin_amdgcn_readfirstlane((int)kmask[2 * kt + 1]);
        const unsigned flip = ~((q_vis ? ~0u : 0u) ^ (var ? ~0u : 0u));   // cross pair <=> key bit != query bit; wanted <=> cross == var
        const int hi = qabs < len - 1 ? qabs : len - 1;             // last valid key of this row
        const u64 rng = bits_below64(hi - kv0 + 1);
        v0 = ((km0 ^ flip) & (unsigned)rng) >> (4 * fk);
        v1 = ((km1 ^ flip) & (unsigned)(rng >> 32)) >> (4 * fk);
    };

    const u64 act0 = __ballot((tab0 & 3u) != 0), act1 = __ballot((tab1 & 3u) != 0);
    const int Uw = act1 ? 128 - (int)__builtin_clzll(act1) : (act0 ? 64 - (int)__builtin_clzll(act0) : 0);
    auto stage_of = [&](const int u) -> const char* { return smem + (u & (DQ_NSTAGE - 1)) * DQ_STAGE_B; };
    // DS phase of unit u (entry e): request stage u + 2 (entry e2), dS^T = P (dP - D) -> pk, wait for the stage the next M phase reads
    auto ds_phase = [&](const int u, const unsigned e, const unsigned e2) {
        const bool issue = u + 2 < U;
        if (issue) stage_unit((int)(e2 >> 3), (int)((e2 >> 2) & 1u), (u + 2) & (DQ_NSTAGE - 1));
        float a[16], c[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            a[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sA[r], p.sl2, nLq2));
            c[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sB[r], p.sl2, nLq2));
        }
        if ((e & 3u) != 1u) {                                       // masked unit (or a skipped one inside the wave's range: empty key set)
            unsigned v0 = 0u, v1 = 0u;
            if ((e & 3u) == 2u) key_masks((int)(e >> 3), (int)((e >> 2) & 1u), v0, v1);
            const float zero = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int bpos = (r & 3) + 8 * (r >> 2);            // local key of accumulator row r (minus 4 fk)
                const u64 k0 = __builtin_amdgcn_ballot_w64(((v0 >> bpos) & 1u) != 0), k1 = __builtin_amdgcn_ballot_w64(((v1 >> bpos) & 1u) != 0);
                asm volatile("v_cndmask_b32 %0, %2, %0, %1" : "+v"(a[r]) : "s"(k0), "v"(zero));
                asm volatile("v_cndmask_b32 %0, %2, %0, %1" : "+v"(c[r]) : "s"(k1), "v"(zero));
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { a[r] *= dA[r] - Dq; c[r] *= dB[r] - Dq; }
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r0_ = 8 * (st & 1) + 2 * j;
                pk[st].u[j] = st < 2 ? pack2bf(a[r0_], a[r0_ + 1]) : pack2bf(c[r0_], c[r0_ + 1]);
            }
#pragma unroll
        for (int st = 0; st < 4; ++st) pin(pk[st].v);               // HERE: keep the dS arithmetic out of the M phase's MFMA stream
        if (grp == 0) { if (issue) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto m_end = [&]() {
        if (grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // group 1: its pieces of stage u + 2, requested in its DS_u
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                // stages 0 and 1 landed
    unsigned e_cur = entry(0), e_nxt = entry(1), e_dma = entry(2);
    if (grp == 1) __builtin_amdgcn_s_barrier();
    if (Uw > 0) m_phase(std::false_type{}, std::true_type{}, nullptr, stage_of(0));
    __builtin_amdgcn_s_barrier();
    int u = 0;
    for (; u + 1 < Uw; ++u) {
        ds_phase(u, e_cur, e_dma);
        __builtin_amdgcn_s_setprio(1);
        m_phase(std::true_type{}, std::true_type{}, stage_of(u), stage_of(u + 1));
        __builtin_amdgcn_s_setprio(0);
        m_end();
        e_cur = e_nxt; e_nxt = e_dma; e_dma = entry(u + 3);
    }
    if (u < Uw) {                                                   // this wave's last unit: nothing to prepare
        ds_phase(u, e_cur, e_dma);
        __builtin_amdgcn_s_setprio(1);
        m_phase(std::true_type{}, std::false_type{}, stage_of(u), nullptr);
        __builtin_amdgcn_s_setprio(0);
        m_end();
        e_cur = e_nxt; e_nxt = e_dma; e_dma = entry(u + 3);
        ++u;
    }
    for (; u < U; ++u) {                                            // units above this wave's diagonal: staging duty only
        const bool issue = u + 2 < U;
        if (issue) stage_unit((int)(e_dma >> 3), (int)((e_dma >> 2) & 1u), (u + 2) & (DQ_NSTAGE - 1));
        if (grp == 0) { if (issue) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();
        m_end();
        e_cur = e_nxt; e_nxt = e_dma; e_dma = entry(u + 3);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();                     // re-align the two groups

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
    __syncthreads();                                                // the next item's staging overwrites the output rows' LDS
    }   // persistent item loop
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
    // Persistent workgroups, static schedule (as in bridge_attn_fwd_kernel): the launcher starts P workgroups (one per CU, P a
    // multiple of n_t); workgroup w handles the items i = w + k P, k = 0, 1, ...: (sequence, head) i / n_t and key tile
    // (i + k) mod n_t.  A key tile's weight is the number of query tiles at or below it (1 .. n_t): the rotation by k hands every
    // workgroup every weight once per n_t steps, so the CUs finish together without a queue, and a CU no longer waits ~7 us for
    // the dispatch of each of its (8192 / 256 =) 32 one-per-CU workgroups.  An XCD's 32 workgroups stream the Q / dO tiles of the
    // same few (sequence, head) pairs at a time through that XCD's L2.
    const int nitems = p.B * p.H * p.n_t;
    int npass = 0;
    if (tid < 16) ((int*)(smem + DKV_XP + 4 * 4096))[tid] = 0;   // the pairs' sequence words count on across items
#if LIBRA_DKV_PERSIST
    const int P = (int)gridDim.x;
    const int w_id = xcd_remap(blockIdx.x, P);
#pragma unroll 1
    for (int step = 0, item = w_id; item < nitems; ++step, item += P) {
    const int ktile = (item % p.n_t + step) % p.n_t;
#else
    {
    const int item = xcd_remap(blockIdx.x, nitems);
    const int ktile = item % p.n_t;                              // low key tiles see the most queries: they come first
#endif
    const int bh = item / p.n_t;
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
    // One memory round trip for the whole prologue (it was three in series, with one workgroup per CU and nothing to cover them):
    // this lane's key modality byte, the four resident operand tiles and the first Q / dO tile (LDS-DMA) are requested first, the
    // mask pass's own flag loads last - its wait then covers everything.
    int k_vis_i = p.flag[tok0 + key] != 0;
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
    // this pair's two P slots and its sequence word (the number of P blocks published so far)
    char* xp = smem + DKV_XP + (qh * 2 + kw) * 4096;
    // (an LDS-space pointer: through a generic `volatile int*` the poll compiled to `flat_load_dword .. sc0 sc1` + `s_waitcnt vmcnt(0)`,
    //  which also drained the next tile's direct-to-LDS queue in every iteration of the dK waves)
    volatile LIBRA_LDS int* xseq = (volatile LIBRA_LDS int*)(LIBRA_LDS char*)(smem + DKV_XP + 4 * 4096) + (qh * 2 + kw);
    f32x16 acc_s[4], acc_c[4];                                   // dV (or dK) for the same / cross variant, [128 d x 32 keys]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_s[i][r] = 0.f; acc_c[i][r] = 0.f; }
    if (it0 < nqt) stage_q(0, it0);
    modality_masks(p.flag + tok0, S, qmask, tid, 512);
    pin(k_vis_i);
    const bool k_vis = k_vis_i != 0;
    const bool wkV = __ballot(k_vis && kin) != 0, wkL = __ballot(!k_vis && kin) != 0;

    const char* rK = res + kw * 32 * 128;                         // this wave's 32 key rows inside each 64-row sub-tile
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
    // ---- combine the two query halves' partial sums through LDS: the qh = 0 wave finishes (and stores) the "same" variant, the
    // qh = 1 wave the "cross" variant - each hands the other half of its sums over (a + b = b + a: the same bits as a one-sided sum)
    __syncthreads();
    {
        float* xch = (float*)smem + ((wave >> 2) * 2 + kw) * 8192;   // 32 KiB per (role, key sub-block) pair: [acc][reg quad][lane] x4
        float* gdst = xch + (qh == 1 ? 0 : 4096);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 a;
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = qh == 1 ? acc_s[i][4 * g + e] : acc_c[i][4 * g + e];
                *(f32x4*)(gdst + ((i * 4 + g) * 64 + lane) * 4) = a;
            }
        __syncthreads();
        const float* gsrc = xch + (qh == 0 ? 0 : 4096);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 a = *(const f32x4*)(gsrc + ((i * 4 + g) * 64 + lane) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (qh == 0) acc_s[i][4 * g + e] += a[e];
                    else acc_c[i][4 * g + e] += a[e];
                }
            }
        __syncthreads();
    }
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
    if (kbase_w < S) {
        if (qh == 0) { if (role_dk) store(acc_s, p.scale, p.dk_same); else store(acc_s, 1.0f, p.dv_same); }
        else { if (role_dk) store(acc_c, p.scale, p.dk_cross); else store(acc_c, 1.0f, p.dv_cross); }
    }
    __syncthreads();                                             // the next item's staging overwrites the store rows' LDS
    }   // persistent item loop
}

}  // namespace libra

using namespace libra;

// persistent grid: one workgroup per CU, rounded down to a multiple of the rotation period (the kernels' static schedules need
// it), at least one period, at most one workgroup per item
static long persistent_grid(long nitems, int period) {
    static std::atomic<int> n_cu{0};              // (benign race: every thread stores the same value)
    if (!n_cu) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        n_cu = cus;
    }
    long nblk = (long)n_cu / period * period;
    if (nblk < period) nblk = period;
    return nblk > nitems ? nitems : nblk;
}

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
    hipLaunchKernelGGL(bridge_attn_bwd_dq_kernel, dim3((unsigned)persistent_grid(nblk, a.n_t)), dim3(512), DQ_LDS_B, (hipStream_t)stream, a);
    if (hipGetLastError() != hipSuccess) return LIBRA_ERR_LAUNCH;
    a.n_t = (int)((S + 63) / 64);
    nblk = (long)B * H * a.n_t;
    if (nblk > 0x7fffffffL) return LIBRA_ERR_SHAPE;
#if LIBRA_DKV_PERSIST
    nblk = persistent_grid(nblk, a.n_t);
#endif
    hipLaunchKernelGGL(bridge_attn_bwd_dkv_kernel, dim3((unsigned)nblk), dim3(512), DKV_LDS_B, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}
