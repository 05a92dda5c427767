// Backward of the fused routed-bridge causal attention (gfx950, head_dim 128).
//
//   P_ij   = exp(scale * q_i.k^x_j - L_i)           x = "same" if m_i == m_j else "cross";  L = forward log-sum-exp
//   dV^x_j = sum_{i: x(i,j)=x} P_ij dO_i            dP_ij = dO_i . v^x_j            D_i = dO_i . O_i
//   dS_ij  = P_ij (dP_ij - D_i)                     dQ_i = scale sum_j dS_ij k^x_j  dK^x_j = scale sum_{i: x(i,j)=x} dS_ij q_i
// (the four operand gradients dK_same, dK_cross, dV_same, dV_cross are folded back onto k, kb, v, vb by
//  libra_rope_bridge_bwd).  Deterministic: two passes, no atomics.
//
//   dq pass  : lane <-> query, 256 queries / workgroup; work list of units (64-key tile, operand variant), two wave groups one
//              phase apart (the forward kernel's structure); writes D = dO . O for the second pass.
//   dkv pass : lane <-> key; item = (128-key block, operand variant), units = 64-query tiles; the dV waves and the dK waves are
//              the two groups, K / V fragments in registers, P handed over through LDS one barrier later.
#include <atomic>
#include <type_traits>
#include "hip_common.hpp"
#include "../../include/libra_hip.h"

#ifndef LIBRA_DKV_GUESS_CROSS   // A/B switches of the dK/dV pass (compile time; the shipped values are the defaults)
#define LIBRA_DKV_GUESS_CROSS 0
#endif
#ifndef LIBRA_DKV_NF
#define LIBRA_DKV_NF 6
#endif
#ifndef LIBRA_DKV_DBG
#define LIBRA_DKV_DBG 0
#endif
#ifndef LIBRA_DKV6_PERSIST      // 1: persistent dK/dV workgroups with the rotation schedule (-0.13 ms per layer: profiles/r05_attn/dkv_switches_ab.txt)
#define LIBRA_DKV6_PERSIST 1
#endif
#if LIBRA_DKV6_PERSIST
#define KV6_NEXT_ITEM continue
#else
#define KV6_NEXT_ITEM return
#endif
#ifndef LIBRA_DKV_ROWPRE        // 1: a unit's L / D rows are read from LDS one phase early (-0.03 ms in the one-item-per-workgroup build, +0.05 ms in the
                                // persistent one: its 25 registers push lane constants into scratch around the item loop)
#define LIBRA_DKV_ROWPRE 0
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
    int* err;                                          // sticky device-side error word (or null): unused since round 5 (no in-kernel wait left)
};

// Reduction-major image of a [rows][128 d] tile: 256-byte rows, 16-byte chunk c of row r stored at position
// c ^ tswz(r).  tswz mixes (r&3) into the chunk's high bits (what the transpose read ds_read_b64_tr_b16 needs to be
// conflict free) and (r>>2)&3 into its low bits, which also spreads the 16 rows of a ds_read_b128 lane group over
// all 64 banks: one image serves BOTH the row-fragment reads (A/B operand with k = d) and the transposed reads
// (A operand with k = rows).  [With only the (r&3) term, row reads were 4-way conflicted: 65 % of the dQ pass's LDS
// cycles in the round-1 PMC profile.]
__device__ __forceinline__ int tswz(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }

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
    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int grp = wave >> 2;
    // persistent workgroups with the forward kernel's static rotation schedule (bridge_attn_fwd_kernel): item i = w + k P is
    // (sequence, head) i / n_t, query block (i + k) mod n_t
    const int nitems = p.B * p.H * p.n_t;
    const int P = (int)gridDim.x;
    const int w_id = xcd_remap(blockIdx.x, P);
#pragma unroll 1
    for (int step = 0, item = w_id; item < nitems; ++step, item += P) {
    int tid = tid0;                                                 // opaque per item: lane constants are re-derived inside the item, not
    asm volatile("" : "+v"(tid));                                   // hoisted out of the persistent loop and held across it (see the dK/dV pass)
    const int lane = tid & 63;
    const int fk = lane >> 5, l31 = lane & 31;
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
// dK / dV pass, round-5 structure ("dkv6").  Item = (sequence, head, 128-key block, operand VARIANT): one variant per item halves
// the accumulators (a wave carries ONE [128 d x 32 keys] block), so a wave also keeps its 32 keys' K (or V) fragments in
// REGISTERS for the whole item - nothing is resident in LDS, the two query halves of the old structure (and their exchange
// epilogue) are gone, and every first-stage MFMA reads ONE operand from LDS instead of two.  8 waves = 4 key sub-blocks x 2 ROLES,
// and the two roles ARE the forward kernel's two wave groups, one phase apart (waves w and w + 4 share a SIMD):
//     dV wave (group 0):  SM_u : P_u = exp2(S_u sl2 - L) masked -> bf16 registers + the pair's LDS slot
//                         M_u  : dV^T += dO_u^T P_u  (16 MFMAs, transposed reads)   S_{u+1} = Q_{u+1} K^T  (16 MFMAs, row reads)
//     dK wave (group 1):  DS_u : dS_u = P_u (dP_u - D)   (P_u from the slot: one barrier after it was written)
//                         M'_u : dK^T += Q_u^T dS_u (16)                            dP_{u+1} = dO_{u+1} V^T (16)
// A unit u is a 64-query tile holding at least one (query, key) pair of the item's variant for some wave; per wave it is skipped,
// PLAIN (no per-element test) or MASKED (per-lane 64-bit query mask: causal, sequence end, pair kind), decided once in the
// prologue.  A unit's Q | dO tile (+ its L and D rows) travels through a ring of 4 stages exactly as in the forward kernel.
constexpr int KV6_TILE = 16384;                       // [64 queries][128 d] image
constexpr int KV6_STAGE = 2 * KV6_TILE + 512;         // Q image | dO image | L[64] | D[64]
constexpr int KV6_NSTAGE = 4;
constexpr int KV6_SLOT_OFF = KV6_NSTAGE * KV6_STAGE;  // P hand-over: 4 pairs x 4 KiB
constexpr int KV6_MASK_OFF = KV6_SLOT_OFF + 4 * 4096; // query-modality words (<= 130; 1 KiB)
constexpr int KV6_BLK_OFF = KV6_MASK_OFF + 1024;      // block-level unit set (2 words)
constexpr int KV6_TAB_OFF = KV6_BLK_OFF + 64;         // per-wave unit tables: 8 x 64 x 2 B
#if LIBRA_DKV_DBG & 128          // timing-only anatomy build: cycle stamps of one heavy item (key block 5, "same"), dumped over the start of dk_same
constexpr int KV6_STAMP_OFF = KV6_TAB_OFF + 8 * 128;
constexpr int KV6_LDS_B = KV6_STAMP_OFF + 8 * 1024;
#define KSTAMP() do { if (dbg_item) { const unsigned t_ = (unsigned)__builtin_readcyclecounter(); if (lane == 0 && n_stamp < 256) ((unsigned*)(smem + KV6_STAMP_OFF))[wave * 256 + n_stamp] = t_; ++n_stamp; } } while (0)
#else
constexpr int KV6_LDS_B = KV6_TAB_OFF + 8 * 128;
#define KSTAMP() ((void)0)
#endif
constexpr int KV6_KEYS = 128;

__global__ __launch_bounds__(512, 2) void bridge_attn_bwd_dkv6_kernel(const BridgeBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* qmask = (unsigned*)(smem + KV6_MASK_OFF);
    unsigned* blk = (unsigned*)(smem + KV6_BLK_OFF);
    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int role = wave >> 2;                                   // 0: dV wave (group 0), 1: dK wave (group 1, one phase behind)
    const int ksub = wave & 3;
    const int n_kb = p.n_t;                                       // 128-key blocks per sequence
    const int nitems = p.B * p.H * n_kb * 2;
#if LIBRA_DKV6_PERSIST
    // persistent workgroups, static rotation (the forward kernel's schedule): item i = w + k P is (sequence, head) i / (2 n_kb) and
    // (block, variant) index (i + k) mod 2 n_kb - every workgroup meets every (block, variant) weight once per 2 n_kb steps
    const int P = (int)gridDim.x;
    const int per_bh = 2 * n_kb;
    const int w_id = xcd_remap(blockIdx.x, P);
#pragma unroll 1
    for (int step = 0, item0 = w_id; item0 < nitems; ++step, item0 += P) {
    const int item = (item0 / per_bh) * per_bh + (item0 % per_bh + step) % per_bh;
#else
    {
    const int item = xcd_remap(blockIdx.x, nitems);
#endif
    // (the thread id is made opaque per item: every lane constant below is then re-derived inside the item instead of being
    //  hoisted out of the persistent loop and kept in registers across it - ~50 VGPRs, which is what made the first persistent
    //  build spill inside its unit loop)
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int fk = lane >> 5, l31 = lane & 31;
    const int var = item & 1;
    const int kb = (item >> 1) % n_kb;                            // low key blocks see the most queries: they come first
#if LIBRA_DKV_DBG & 128
    const bool dbg_item = item == 10;
    int n_stamp = 0;
    const unsigned long long t_item0 = __builtin_readcyclecounter();
#endif
    const int bh = (item >> 1) / n_kb;
    const int h = bh % p.H, b = bh / p.H;
    const int S = p.S;
    const long tok0 = (long)b * S;
    int len = p.kv_len ? p.kv_len[b] : S;
    len = len < S ? len : S;
    const int kbase_w = kb * KV6_KEYS + ksub * 32;
    int key = kbase_w + l31;
    const bool kin = key < S;
    key = kin ? key : S - 1;
    const int nqt = (S + 63) / 64;
    const int t_first = (kb * KV6_KEYS) / 64;                     // first query tile that can see the block

    const bf16_t* qbase = p.q + tok0 * p.ldq + h * D128;
    const bf16_t* dobase = p.dout + tok0 * p.ldo + h * D128;
    const float* lbase = p.lse + ((long)b * p.H + h) * S;
    const float* dbase = p.delta + ((long)b * p.H + h) * S;
    const unsigned lds0 = (unsigned)(unsigned long)(LIBRA_LDS char*)smem;

    // ---- direct-to-LDS pieces of a stage: wave w moves pieces 2w, 2w + 1 (rows 8w + (lane >> 4), + 4) of the Q and of the dO image;
    // waves 4 and 5 (group 1: it waits with vmcnt(0), no counting) also move the tile's 64 L and 64 D values
    const int r0 = wave * 8 + (lane >> 4);
    const int c0 = (lane & 15) ^ tswz(r0);
    const int d16 = (c0 & 1) ? -16 : 16;
    const unsigned ldqb = (unsigned)(p.ldq * 2), ldob = (unsigned)(p.ldo * 2);
    const unsigned oQ = (unsigned)(lane >> 4) * ldqb + (unsigned)(c0 << 4), oO = (unsigned)(lane >> 4) * ldob + (unsigned)(c0 << 4);
    auto stage_tile = [&](const int t, const int st) {
        const unsigned dst = lds0 + (unsigned)(st * KV6_STAGE + wave * 2048);
        if (t * 64 + 64 <= S) {
            const bf16_t* qb = qbase + ((long)t * 64 + wave * 8) * p.ldq;
            const bf16_t* ob = dobase + ((long)t * 64 + wave * 8) * p.ldo;
            glds16_off_at(qb, oQ, dst); glds16_off_at(qb + 4 * p.ldq, oQ + (unsigned)d16, dst + 1024);
            glds16_off_at(ob, oO, dst + KV6_TILE); glds16_off_at(ob + 4 * p.ldo, oO + (unsigned)d16, dst + KV6_TILE + 1024);
        } else {                                                  // ragged last tile: rows clamped to the last token (masked)
            const int lim = S - 1 - t * 64;
            const bf16_t* qb = qbase + (long)t * 64 * p.ldq;
            const bf16_t* ob = dobase + (long)t * 64 * p.ldo;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                int row = r0 + 4 * j;
                row = row < lim ? row : lim;
                const unsigned cc = (unsigned)((c0 ^ j) << 4);
                glds16_off_at(qb, (unsigned)row * ldqb + cc, dst + j * 1024);
                glds16_off_at(ob, (unsigned)row * ldob + cc, dst + KV6_TILE + j * 1024);
            }
        }
        if (wave == 4 || wave == 5) {
            int qi = t * 64 + lane; qi = qi < S ? qi : S - 1;
            glds4((wave == 4 ? lbase : dbase) + qi, smem + st * KV6_STAGE + 2 * KV6_TILE + (wave - 4) * 256);
        }
    };

    // ---- prologue: the first two stages on a guess (the first two tiles that can see the block), this lane's key modality and
    // its 8 B-operand fragments (K rows for a dV wave, V rows for a dK wave, of the item's variant), the query modality words
    // (guessed for the "same" variant only: a "cross" item's first pair is usually far from the block's diagonal - text queries
    //  behind an image - or it has none at all, and then nothing is in flight when it finds that out)
    const bool guess = var == 0 || LIBRA_DKV_GUESS_CROSS;
    if (guess && t_first < nqt) stage_tile(t_first, 0);
    if (guess && t_first + 1 < nqt) stage_tile(t_first + 1, 1);
    int k_vis_i = p.flag[tok0 + key] != 0;
    bf16x8 bfr[8];
    {
        const bf16_t* src = role == 0 ? (var ? p.k_cross : p.k_same) : (var ? p.v_cross : p.v_same);
        const long ld = role == 0 ? (var ? p.ldkc : p.ldk) : (var ? p.ldvc : p.ldv);
        const bf16_t* kp = src + (tok0 + key) * ld + h * D128 + fk * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) bfr[ks] = *(const bf16x8*)(kp + ks * 16);
    }
    modality_masks(p.flag + tok0, S, qmask, tid, 512);
    if (tid < 2) blk[tid] = 0;
    pin(k_vis_i);
    const bool k_vis = k_vis_i != 0;
    const bool kval = kin && (kbase_w + l31) < len;               // this lane's key exists and is not padding
    const bool wkV = __ballot(k_vis && kval) != 0, wkL = __ballot(!k_vis && kval) != 0;
    __syncthreads();

    // ---- per-wave classification of every query tile, lane = tile
    unsigned mode = 0;
    {
        const int q0 = lane * 64;
        const u64 qm = (u64)qmask[2 * lane] | ((u64)qmask[2 * lane + 1] << 32);
        const u64 rng = bits_below64(S - q0);
        const bool qV = (qm & rng) != 0, qL = (~qm & rng) != 0;
        const bool wsame = (qL && wkL) || (qV && wkV), wcross = (qL && wkV) || (qV && wkL);
        const bool in = lane < nqt && q0 + 63 >= kbase_w && kbase_w < len;
        const bool has = in && (var ? wcross : wsame);
        const bool full = q0 >= kbase_w + 31 && q0 + 64 <= S && kbase_w + 32 <= len;
        const bool plain = full && !(wsame && wcross);
        mode = has ? (plain ? 1u : 2u) : 0u;
        const u64 bset = __ballot(has);
        if (lane == 0) {
            if ((unsigned)bset) atomicOr(&blk[0], (unsigned)bset);
            if ((unsigned)(bset >> 32)) atomicOr(&blk[1], (unsigned)(bset >> 32));
        }
    }
    __syncthreads();
    const u64 uset = (u64)(unsigned)__builtin_amdgcn_readfirstlane((int)blk[0]) | ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)blk[1]) << 32);
    const int U = __popcll(uset);                                 // units of this item (<= 64)

    // ---- store of a wave's [128 d x 32 keys] block, transposed through a private LDS region (32 rows x 264 B)
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    constexpr int OROW = 264;
    auto store_out = [&]() {
        char* so = smem + wave * (32 * OROW);
        bf16_t* dst = role == 0 ? (var ? p.dv_cross : p.dv_same) : (var ? p.dk_cross : p.dk_same);
        const float mul = role == 0 ? 1.0f : p.scale;
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
    if (U == 0) {                                                 // no pair of this variant: the block's gradients are zero
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                          // everybody's guessed pieces have landed: LDS is free
        store_out();
        __syncthreads();                                          // (persistent: the next item's staging overwrites the store rows)
        KV6_NEXT_ITEM;
    }

    unsigned tab0;                                                // lane i: entry of unit i = mode | tile << 3 (0 past the end)
    {
        unsigned short* tab = (unsigned short*)(smem + KV6_TAB_OFF) + wave * 64;
        if ((uset >> lane) & 1ull) tab[__popcll(uset & bits_below64(lane))] = (unsigned short)(mode | (unsigned)(lane << 3));
        tab0 = lane < U ? tab[lane] : 0u;                         // (same wave, in-order LDS queue: no barrier)
    }
    auto entry = [&](const int u) -> unsigned {                   // u wave-uniform; 0 past the end
        const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)tab0, u & 63);
        return u < 64 ? a : 0u;
    };
    {
        const unsigned e0 = entry(0), e1 = entry(1);
        const bool ok0 = guess && (int)(e0 >> 3) == t_first, ok1 = U < 2 || (guess && (int)(e1 >> 3) == t_first + 1);
        if (!ok0 || !ok1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (!ok0) stage_tile((int)(e0 >> 3), 0);
            if (!ok1 && U >= 2) stage_tile((int)(e1 >> 3), 1);
        }
    }

    // ---- fragment addressing (lane constants XOR compile-time constants), role offsets (wave-uniform)
    int xr, xt0, xt1;
    {
        const int pp = lane & 15, g16 = (lane >> 4) & 1;
        const int r1 = 4 * fk + (pp >> 2), lp = 2 * g16 + ((pp & 3) >> 1);
        xr = l31 * 256 + ((fk ^ tswz(l31)) << 4);
        xt0 = r1 * 256 + ((pp & 1) << 3) + ((lp ^ tswz(r1)) << 4);
        xt1 = r1 * 256 + ((pp & 1) << 3) + ((lp ^ tswz(r1 + 8)) << 4) + 2048;
    }
    const int tr_off = role == 0 ? KV6_TILE : 0;                  // dV: dO^T fragments; dK: Q^T fragments
    const int row_off = role == 0 ? 0 : KV6_TILE;                 // dV: Q rows (S); dK: dO rows (dP)
    char* slot = smem + KV6_SLOT_OFF + ksub * 4096 + lane * 16;

    f32x16 sA, sB;                                                // S (dV wave) or dP (dK wave) of the unit in flight: query halves 0 / 1
    union PK { bf16x8 v; unsigned u[4]; };
    PK pk[4];                                                     // P (dV wave) / dS (dK wave) as the four 16-query B operands
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) pin(bfr[ks]);                  // prologue loads have landed before the loop's LDS-DMA traffic

    union VA { bf16x8 v; s16x4 h2[2]; };
    constexpr int NF = LIBRA_DKV_NF;
    auto m_phase = [&](auto ac_c, auto nx_c, const char* cur, const char* nxt) {
        constexpr bool AC = decltype(ac_c)::value, NX = decltype(nx_c)::value;
        constexpr int N = (AC ? 16 : 0) + (NX ? 16 : 0), I0 = AC ? 0 : 16;
        bf16x8 F[NF];
        auto fread = [&](const int i) -> bf16x8 {
            if (i < 16) {
                const char* a = cur + tr_off + (i >> 2) * 4096;
                VA t;
                t.h2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(a + (xt0 ^ ((i & 3) << 6))));
                t.h2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(a + (xt1 ^ ((i & 3) << 6))));
                return t.v;
            }
            const int j = i - 16, ks = j >> 1, hh = j & 1;
            return *(const bf16x8*)(nxt + row_off + hh * 8192 + (xr ^ (ks << 5)));
        };
        if constexpr (N > 0) {
#pragma unroll
            for (int n = 0; n < NF; ++n) F[n] = fread(I0 + n);
            if constexpr (NX) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { sA[r] = 0.f; sB[r] = 0.f; }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < N; ++n) {
                const int i = I0 + n;
                if (i < 16) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[n % NF], pk[i >> 2].v, acc[i & 3], 0, 0, 0);
                else if (i & 1) sB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[n % NF], bfr[(i - 16) >> 1], sB, 0, 0, 0);
                else sA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[n % NF], bfr[(i - 16) >> 1], sA, 0, 0, 0);
                if (n + NF < N) F[n % NF] = fread(i + NF);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // per-lane query mask of a MASKED unit: query >= key (causal), query < S, key < len, pair kind == the item's variant.
    // One 32-bit word per query half, shifted by 4 fk so that the bit positions below are compile-time.
    const int kabs = kbase_w + l31;
    auto query_masks = [&](const int qt, unsigned& v0, unsigned& v1) {
        const int q0 = qt * 64;
        const unsigned qm0 = (unsigned)__builtin_amdgcn_readfirstlane((int)qmask[2 * qt]);
        const unsigned qm1 = (unsigned)__builtin_amdgcn_readfirstlane((int)qmask[2 * qt + 1]);
        const unsigned flip = ~((k_vis ? ~0u : 0u) ^ (var ? ~0u : 0u));   // cross pair <=> query bit != key bit; wanted <=> cross == var
        u64 rng = bits_below64(S - q0) & ~bits_below64(kabs - q0);
        rng = kval ? rng : 0ull;
        v0 = ((qm0 ^ flip) & (unsigned)rng) >> (4 * fk);
        v1 = ((qm1 ^ flip) & (unsigned)(rng >> 32)) >> (4 * fk);
    };
    auto stage_of = [&](const int u) -> const char* { return smem + (u & (KV6_NSTAGE - 1)) * KV6_STAGE; };

    const u64 act = __ballot((tab0 & 3u) != 0);
    const int Uw = act ? 64 - (int)__builtin_clzll(act) : 0;
    f32x4 rowv[8];                                                // L (dV wave) / D (dK wave) of the next unit's 64 queries, this lane's rows
    auto load_rows = [&](const int u_) {
        const float* sv = (const float*)(stage_of(u_) + 2 * KV6_TILE) + role * 64;
#pragma unroll
        for (int g = 0; g < 8; ++g) rowv[g] = *(const f32x4*)(sv + 8 * g + 4 * fk);
    };
    // VALU phase of unit u (entry e): request stage u + 2 (entry e2), then
    //   dV wave: P = exp2(S sl2 - L) masked -> pk and the slot;      dK wave: dS = P (dP - D) -> pk, P read from the slot
    auto v_phase = [&](const int u, const unsigned e, const unsigned e2) {
        KSTAMP();                                                 // [0] V phase start
        const bool issue = u + 2 < U;
        if (issue) stage_tile((int)(e2 >> 3), (u + 2) & (KV6_NSTAGE - 1));
        const float* sL = (const float*)(stage_of(u) + 2 * KV6_TILE);
        if (role == 0) {
            float a[16], c[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#if LIBRA_DKV_ROWPRE
                const f32x4 La = rowv[g], Lc = rowv[4 + g];
#else
                const f32x4 La = *(const f32x4*)(sL + 8 * g + 4 * fk), Lc = *(const f32x4*)(sL + 32 + 8 * g + 4 * fk);
#endif
#pragma unroll
                for (int e_ = 0; e_ < 4; ++e_) {
                    a[4 * g + e_] = __builtin_amdgcn_exp2f(__builtin_fmaf(sA[4 * g + e_], p.sl2, -La[e_] * LOG2E));
                    c[4 * g + e_] = __builtin_amdgcn_exp2f(__builtin_fmaf(sB[4 * g + e_], p.sl2, -Lc[e_] * LOG2E));
                }
            }
            if ((e & 3u) != 1u) {                                 // masked unit (or a skipped one inside the wave's range: empty pair set)
                unsigned v0 = 0u, v1 = 0u;
                if ((e & 3u) == 2u) query_masks((int)(e >> 3), v0, v1);
                const float zero = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int bpos = (r & 3) + 8 * (r >> 2);      // local query of accumulator row r (minus 4 fk)
                    const u64 k0 = __builtin_amdgcn_ballot_w64(((v0 >> bpos) & 1u) != 0), k1 = __builtin_amdgcn_ballot_w64(((v1 >> bpos) & 1u) != 0);
                    asm volatile("v_cndmask_b32 %0, %2, %0, %1" : "+v"(a[r]) : "s"(k0), "v"(zero));
                    asm volatile("v_cndmask_b32 %0, %2, %0, %1" : "+v"(c[r]) : "s"(k1), "v"(zero));
                }
            }
#pragma unroll
            for (int st = 0; st < 4; ++st) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r0_ = 8 * (st & 1) + 2 * j;
                    pk[st].u[j] = st < 2 ? pack2bf(a[r0_], a[r0_ + 1]) : pack2bf(c[r0_], c[r0_ + 1]);
                }
                *(bf16x8*)(slot + st * 1024) = pk[st].v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the slot is written before the barrier that releases its reader
            KSTAMP();                                             // [1] arithmetic + slot write done (dV wave)
            if (issue) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            const float* sD = sL + 64;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                PK pv;
                pv.v = *(const bf16x8*)(slot + st * 1024);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r0_ = 8 * (st & 1) + 2 * j;         // accumulator rows r0, r0 + 1 <-> queries 32 (st >> 1) + 8 (r0 >> 2) + 4 fk + (r0 & 3), + 1
                    const int ql = 32 * (st >> 1) + 8 * (r0_ >> 2) + 4 * fk + (r0_ & 3);
#if LIBRA_DKV_ROWPRE
                    const float d0 = rowv[2 * st + (j >> 1)][2 * (j & 1)], d1 = rowv[2 * st + (j >> 1)][2 * (j & 1) + 1];   // = sD[ql], sD[ql + 1]
#else
                    const float d0 = sD[ql], d1 = sD[ql + 1];
#endif
                    const float p0 = __uint_as_float(pv.u[j] << 16), p1 = __uint_as_float(pv.u[j] & 0xffff0000u);
                    const float x0 = st < 2 ? sA[r0_] : sB[r0_], x1 = st < 2 ? sA[r0_ + 1] : sB[r0_ + 1];
                    pk[st].u[j] = pack2bf(p0 * (x0 - d0), p1 * (x1 - d1));
                }
            }
        }
#pragma unroll
        for (int st = 0; st < 4; ++st) pin(pk[st].v);             // HERE: keep this phase's arithmetic out of the M phase's MFMA stream
        if (role == 1) KSTAMP();                                  // [1] arithmetic done (dK wave)
        KSTAMP();                                                 // [2] staging wait done
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        KSTAMP();                                                 // [3] barrier passed = M start
    };
    auto m_end = [&]() {
        KSTAMP();                                                 // [4] MFMAs issued
        if (role == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // group 1: its pieces of stage u + 2, requested in its DS_u
        KSTAMP();                                                 // [5] staging wait done
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                              // stages 0 and 1 landed
    unsigned e_cur = entry(0), e_nxt = entry(1), e_dma = entry(2);
    if (role == 1) __builtin_amdgcn_s_barrier();
#if LIBRA_DKV_ROWPRE
    if (Uw > 0) load_rows(0);
#endif
    if (Uw > 0) m_phase(std::false_type{}, std::true_type{}, nullptr, stage_of(0));
    __builtin_amdgcn_s_barrier();
    int u = 0;
    for (; u + 1 < Uw; ++u) {
        v_phase(u, e_cur, e_dma);
        __builtin_amdgcn_s_setprio(1);
#if LIBRA_DKV_ROWPRE
        load_rows(u + 1);                                         // stage u + 1 has landed (this M phase reads its rows)
#endif
        m_phase(std::true_type{}, std::true_type{}, stage_of(u), stage_of(u + 1));
        __builtin_amdgcn_s_setprio(0);
        m_end();
        e_cur = e_nxt; e_nxt = e_dma; e_dma = entry(u + 3);
    }
    if (u < Uw) {                                                 // this wave's last unit: nothing to prepare
        v_phase(u, e_cur, e_dma);
        __builtin_amdgcn_s_setprio(1);
        m_phase(std::true_type{}, std::false_type{}, stage_of(u), nullptr);
        __builtin_amdgcn_s_setprio(0);
        m_end();
        e_cur = e_nxt; e_nxt = e_dma; e_dma = entry(u + 3);
        ++u;
    }
    for (; u < U; ++u) {                                          // units past this wave's last pair: staging duty only
        const bool issue = u + 2 < U;
        if (issue) stage_tile((int)(e_dma >> 3), (u + 2) & (KV6_NSTAGE - 1));
        if (role == 0) { if (issue) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();
        m_end();
        e_cur = e_nxt; e_nxt = e_dma; e_dma = entry(u + 3);
    }
    if (role == 0) __builtin_amdgcn_s_barrier();                  // re-align the two groups
    __syncthreads();
#if LIBRA_DKV_DBG & 128
    if (dbg_item) {
        unsigned* dump = (unsigned*)p.dk_same;
        for (int i = tid; i < 2048; i += 512) dump[i] = ((unsigned*)(smem + KV6_STAMP_OFF))[i];
        if (tid == 0) { dump[2048] = (unsigned)U; dump[2058] = (unsigned)(__builtin_readcyclecounter() - t_item0); }
        if (lane == 0) dump[2049 + wave] = (unsigned)Uw;
    }
    return;
#endif
    store_out();
    __syncthreads();                                              // (persistent: the next item's staging overwrites the store rows)
    }   // item loop
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
        (void)hipFuncSetAttribute((const void*)bridge_attn_bwd_dkv6_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, KV6_LDS_B);
        attr_set = true;
    }
    a.n_t = (int)((S + DQ_BQ - 1) / DQ_BQ);
    long nblk = (long)B * H * a.n_t;
    if (nblk > 0x7fffffffL) return LIBRA_ERR_SHAPE;
    // (other dK/dV structures that were built and measured in round 2: profiles/r02_attn_bwd_anatomy.md)
    hipLaunchKernelGGL(bridge_attn_bwd_dq_kernel, dim3((unsigned)persistent_grid(nblk, a.n_t)), dim3(512), DQ_LDS_B, (hipStream_t)stream, a);
    if (hipGetLastError() != hipSuccess) return LIBRA_ERR_LAUNCH;
    a.n_t = (int)((S + KV6_KEYS - 1) / KV6_KEYS);
    nblk = (long)B * H * a.n_t * 2;
    if (nblk > 0x7fffffffL) return LIBRA_ERR_SHAPE;
#if LIBRA_DKV6_PERSIST
    nblk = persistent_grid(nblk, 2 * a.n_t);
#endif
    hipLaunchKernelGGL(bridge_attn_bwd_dkv6_kernel, dim3((unsigned)nblk), dim3(512), KV6_LDS_B, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}
