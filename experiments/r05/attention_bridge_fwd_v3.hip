// Fused routed-"bridge" causal flash attention (forward) for Libra's decoder layers, gfx950, head_dim 128.
//
// Reference semantics (LibraAttention.forward + attn_with_bridge, modeling_libra.py:267-414), closed form:
//     S_ij = q_i . (k_j + [m_i != m_j] kb_j) / sqrt(d) + causal/padding mask,   P = softmax_fp32(S)
//     O_i  = sum_j P_ij (v_j + [m_i != m_j] vb_j)
// where m is the per-token modality flag.  The reference evaluates this with TWO full QK^T and TWO full PV
// products and ~6 materialised [B,H,S,S] tensors (its own "TODO: make it more efficient", :288).  Here the
// caller provides the four operands K_same = rope(k), K_cross = rope(k + kb), V_same = v, V_cross = v + vb
// (libra_rope_bridge) and a workgroup of 8 waves x 32 query rows streams 64-key tiles through LDS.
//
// Round-5 structure ("two wave groups half a tile apart").  Rounds 2-4 ran the 8 waves in lock step (one barrier per tile):
// on every SIMD both resident waves multiplied at the same time and both did their softmax at the same time, so the matrix
// pipe idled through every softmax (23.7 % busy, 10.3 VALU + 6 SALU per MFMA by PMC) whatever the staging primitive,
// occupancy shape or fragment schedule was (six structures within +-5 %).  Now:
//   * the work list of a workgroup is a list of UNITS = (key tile, operand variant pass).  A wave whose 32 rows meet keys of
//     one modality combination only - the overwhelmingly common case - has ONE unit per tile; only waves that really
//     straddle a modality boundary visit a tile twice (same-variant pass, cross-variant pass: disjoint (row, key) sets are
//     two valid online-softmax steps).  A unit is skipped, PLAIN (no per-element test of any kind: interior tile, one
//     variant) or MASKED (causal diagonal / padding / modality selected by a per-lane 64-bit key mask) - decided once per
//     (wave, unit) in the prologue, lane-parallel, and kept in two registers (v_readlane per unit: no loads, no mask code
//     and ~0 SALU in the steady state);
//   * a unit is two phases: SM (online softmax of S_u: VALU only) and M = [O += V_u^T P_u ; S_{u+1} = K_{u+1} Q^T]
//     (32 MFMAs + their LDS fragment reads, software-pipelined one step ahead, fragment addresses = lane constant + immediate);
//   * waves 0-3 and waves 4-7 (one of each per SIMD) run this sequence ONE PHASE APART (the second group passes one extra
//     s_barrier at the start): on every SIMD one wave owns the matrix pipe while its partner does its softmax on the VALU;
//   * K/V tiles arrive by direct-to-LDS loads into a 2-slot K ring and a 2-slot V ring (slot = both variants): stage
//     (V_{t+1}, K_{t+2}) is requested in the phase after the last read of the slots it overwrites and waited for (vmcnt(0) by
//     every wave, then the phase barrier) one phase before its first read - two phases in flight, nothing else ever waits.
#include <atomic>
#include <type_traits>
#include "hip_common.hpp"
#include "attention_bridge_args.hpp"
#include "../../include/libra_hip.h"

#ifndef LIBRA_ATTN_PRIO
#define LIBRA_ATTN_PRIO 1
#endif
#ifndef LIBRA_ATTN_DBG          // timing-only anatomy builds (results wrong): 1 no in-loop staging, 2 no softmax arithmetic,
#define LIBRA_ATTN_DBG 0        // 4 no P.V product, 8 no Q.K product, 16 no main loop, 32 every unit plain, 64 no barriers in the loop
#endif

#if LIBRA_ATTN_DBG & 64
#define LOOP_BARRIER() ((void)0)
#else
#define LOOP_BARRIER() __builtin_amdgcn_s_barrier()
#endif

namespace libra {

constexpr int BD = 128;            // head dim
constexpr int BQ = 256;            // query rows per workgroup (8 waves x 32)
constexpr int BKV = 64;            // keys per tile (two 32-key halves)
constexpr int TILE_B = BKV * BD * 2;          // one operand tile, 16 KiB
constexpr int SLOT_B = 2 * TILE_B;            // ring slot: same variant | cross variant
constexpr int KRING = 0;                      // 2 slots
constexpr int VRING = 2 * SLOT_B;             // 2 slots
constexpr int MASK_OFF = 4 * SLOT_B;          // key-modality words of the sequence (<= 130 words; 1 KiB reserved)
constexpr int BLK_OFF = MASK_OFF + 1024;      // block-level tile sets (6 words)
constexpr int TAB_OFF = BLK_OFF + 64;         // per-wave unit tables: 8 x 128 x 2 B
#if LIBRA_ATTN_DBG & 128          // + cycle stamps of workgroup 0 (heaviest block of sequence 0 / head 0), dumped over the start of out_lo
constexpr int STAMP_OFF = TAB_OFF + 8 * 256;
constexpr int BR_LDS = STAMP_OFF + 8 * 1024;
#define STAMP() do { if (dbg_blk) { const unsigned t_ = (unsigned)__builtin_readcyclecounter(); if (lane == 0 && n_stamp < 256) ((unsigned*)(smem + STAMP_OFF))[wave * 256 + n_stamp] = t_; ++n_stamp; } } while (0)
#else
constexpr int BR_LDS = TAB_OFF + 8 * 256;
#define STAMP() ((void)0)
#endif

// K tile image: four N-type [32 keys][64 d] sub-tiles (128-byte rows, chunk ^ ((row>>1)&7)), 4 KiB each, ordered
//               (key half, d half).  V tile image: T-type [64 keys][128 d] (256-byte rows, chunk ^ ((row&3)<<2)), 16 KiB.
// Unit table entry: bits 0-1 mode (0 skip, 1 plain, 2 masked), bit 2 operand variant (1 = cross), bit 3 last unit of its tile,
//                   bits 4.. key tile.
typedef unsigned long long u64;
__device__ __forceinline__ u64 bits_below(int n) { return n >= 64 ? ~0ull : (n <= 0 ? 0ull : ((1ull << n) - 1ull)); }

__global__ __launch_bounds__(512, 2) void bridge_attn_fwd_kernel(const BridgeArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* kmask = (unsigned*)(smem + MASK_OFF);               // per 32 keys: bit j = key j is a vision token
    unsigned* blk = (unsigned*)(smem + BLK_OFF);                  // [0,1] tiles with a second pass, [2,3] same needed, [4,5] cross needed
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                                    // waves w and w + 4 share a SIMD: one of each group
    const int fk = lane >> 5, l31 = lane & 31;

    const int nblk = p.B * p.H * p.n_qt;
    const int L = xcd_remap(blockIdx.x, nblk);
    const int qt = p.n_qt - 1 - (L % p.n_qt);                      // heaviest (most key tiles) first
    const int bh = L / p.n_qt;
    const int h = bh % p.H, b = bh / p.H;
    const int S = p.S;
    const long tok0 = (long)b * S;
    const int len = p.kv_len ? p.kv_len[b] : S;
    const int start = p.kv_start ? p.kv_start[b] : 0;
    const int q0w = qt * BQ + wave * 32;
    const bool active = q0w < S;
    int q = q0w + l31;
    q = q < S ? q : S - 1;
    // causal: keys 0 .. min(S, (qt+1)*BQ) - 1
    int kend = (qt + 1) * BQ; kend = kend < S ? kend : S;
    const int nkt = (kend + BKV - 1) / BKV;                        // <= 64 (S <= 4096)

    const bf16_t* ks_base = p.k_same + tok0 * p.ldk + h * BD;
    const bf16_t* kc_base = p.k_cross + tok0 * p.ldkc + h * BD;
    const bf16_t* vs_base = p.v_same + tok0 * p.ldv + h * BD;
    const bf16_t* vc_base = p.v_cross + tok0 * p.ldvc + h * BD;
    const unsigned lds0 = (unsigned)(unsigned long)(LIBRA_LDS char*)smem;
#if LIBRA_ATTN_DBG & 128
    const bool dbg_blk = blockIdx.x == 0;
    int n_stamp = 0;
#endif

    // ---- direct-to-LDS pieces.  A 16-KiB tile is 16 pieces of 1 KiB; wave w moves pieces 2w, 2w+1 of every tile.
    // K piece pc -> sub-tile pc>>2, rows 8*(pc&3)..+7 (8 rows x 128 B); V piece pc -> rows 4*pc..+3 (4 rows x 256 B)
    const int stK = (wave * 2) >> 2;
    const int rK = ((wave * 2) & 3) * 8 + (lane >> 3);             // row inside the sub-tile (piece 1: + 8)
    const int rowK = (stK >> 1) * 32 + rK;                         // key row inside the tile
    const unsigned colK0 = (unsigned)((stK & 1) * 128 + (((lane & 7) ^ ((rK >> 1) & 7)) << 4));
    const unsigned colK1 = colK0 ^ 64u;                            // row + 8 flips bit 2 of the chunk swizzle
    const int rV = wave * 8 + (lane >> 4);                         // (piece 1: + 4, same chunk)
    const unsigned colV = (unsigned)(((lane & 15) ^ ((rV & 3) << 2)) << 4);
    // piece (operand, variant, j) of key tile t: rows past the end of the sequence are clamped (masked later)
    auto piece_t = [&](const bool isK, const int var, const int j, const int t) {
        const int lim = S - 1 - t * BKV;
        int row = isK ? rowK + 8 * j : rV + 4 * j;
        row = row < lim ? row : lim;
        const long ld = isK ? (var ? p.ldkc : p.ldk) : (var ? p.ldvc : p.ldv);
        const bf16_t* base = (isK ? (var ? kc_base : ks_base) : (var ? vc_base : vs_base)) + (long)t * BKV * ld;
        const unsigned voff = (unsigned)row * (unsigned)(ld * 2) + (isK ? (j ? colK1 : colK0) : colV);
        glds16_off_at(base, voff, lds0 + (unsigned)((isK ? KRING : VRING) + (t & 1) * SLOT_B + var * TILE_B + (wave * 2 + j) * 1024));
    };
    // in-loop pieces of the stage issued at the last unit of key tile kt: slot i = 4*variant + 2*isK + j; V of tile kt+1, K of kt+2
    auto piece = [&](const int i, const int kt) { piece_t((i >> 1) & 1, i >> 2, i & 1, kt + 1 + ((i >> 1) & 1)); };

    // ---- prologue: everything the first phases need is REQUESTED before the first wait (one workgroup per CU: nothing else
    // covers a prologue's serial round trips).  K_0, V_0 and K_1 go out in both variants: which of them the tiles need is only
    // known after the mask pass and two barriers.
#pragma unroll
    for (int var = 0; var < 2; ++var)
#pragma unroll
        for (int j = 0; j < 2; ++j) { piece_t(true, var, j, 0); piece_t(false, var, j, 0); }
    if (nkt > 1) {
#pragma unroll
        for (int var = 0; var < 2; ++var)
#pragma unroll
            for (int j = 0; j < 2; ++j) piece_t(true, var, j, 1);
    }
    const int q_vis_raw = p.flag[tok0 + q];
    bf16x8 qf[8];                                                   // lane (q = l31, half fk) holds Q[q][16*ks + 8*fk .. +8]
    {
        const bf16_t* qp = p.q + (tok0 + q) * p.ldq + h * BD + fk * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
    }
    modality_masks(p.flag + tok0, S, kmask, tid, 512);
    if (tid < 8) blk[tid] = 0;
    const bool q_vis = q_vis_raw != 0;
    __syncthreads();

    // ---- per-wave classification of every key tile, lane = tile
    const bool wV = __ballot(q_vis && (q0w + l31) < S) != 0;        // this wave's query modalities
    const bool wL = __ballot(!q_vis && (q0w + l31) < S) != 0;
    unsigned e0 = 0, e1 = 0;                                        // this lane's tile: entries of pass 0 / pass 1
    {
        const int kv0 = lane * BKV;
        const u64 mm = (u64)kmask[2 * lane] | ((u64)kmask[2 * lane + 1] << 32);
        const u64 rng = bits_below(len - kv0) & ~bits_below(start - kv0);       // valid keys of the tile
        const bool kV = (mm & rng) != 0, kL = (~mm & rng) != 0;
        const bool wsame = (wL && kL) || (wV && kV), wcross = (wL && kV) || (wV && kL);
        const bool in = active && lane < nkt && kv0 <= q0w + 31 && (wsame || wcross);
        const bool both = in && wsame && wcross;
        const bool full = kv0 + BKV - 1 <= q0w && kv0 >= start && kv0 + BKV <= len;
        const u64 b_sec = __ballot(both), b_same = __ballot(in && wsame), b_cross = __ballot(in && wcross);
        if (lane == 0) {
            if ((unsigned)b_sec) atomicOr(&blk[0], (unsigned)b_sec);
            if ((unsigned)(b_sec >> 32)) atomicOr(&blk[1], (unsigned)(b_sec >> 32));
            if ((unsigned)b_same) atomicOr(&blk[2], (unsigned)b_same);
            if ((unsigned)(b_same >> 32)) atomicOr(&blk[3], (unsigned)(b_same >> 32));
            if ((unsigned)b_cross) atomicOr(&blk[4], (unsigned)b_cross);
            if ((unsigned)(b_cross >> 32)) atomicOr(&blk[5], (unsigned)(b_cross >> 32));
        }
        e0 = (unsigned)((!in ? 0 : ((full && !both) ? 1 : 2)) | ((in && !wsame) ? 4 : 0) | (lane << 4));
        e1 = (unsigned)((both ? 2 : 0) | 4 | 8 | (lane << 4));
    }
    __syncthreads();
    const u64 sec_blk = (u64)(unsigned)__builtin_amdgcn_readfirstlane((int)blk[0]) | ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)blk[1]) << 32);
    const u64 same_blk = (u64)(unsigned)__builtin_amdgcn_readfirstlane((int)blk[2]) | ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)blk[3]) << 32);
    const u64 cross_blk = (u64)(unsigned)__builtin_amdgcn_readfirstlane((int)blk[4]) | ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)blk[5]) << 32);
#if LIBRA_ATTN_DBG & 16
    const int U = 0;
#else
    const int U = nkt + __popcll(sec_blk);                          // units of this workgroup (<= 128)
#endif
    unsigned tab0, tab1;                                            // lane i: entry of unit i / unit 64 + i (0 past the end)
    {
        unsigned short* tab = (unsigned short*)(smem + TAB_OFF) + wave * 128;
        if (lane < nkt) {
            const int u0 = lane + __popcll(sec_blk & bits_below(lane));
            const bool sec = (sec_blk >> lane) & 1ull;
            tab[u0] = (unsigned short)(e0 | (sec ? 0u : 8u));
            if (sec) tab[u0 + 1] = (unsigned short)e1;
        }
        tab0 = lane < U ? tab[lane] : 0u;                           // (same wave, in-order LDS queue: no barrier)
        tab1 = lane + 64 < U ? tab[lane + 64] : 0u;
    }
    auto entry = [&](const int u) -> unsigned {                     // u wave-uniform
        const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)tab0, u & 63), c = (unsigned)__builtin_amdgcn_readlane((int)tab1, u & 63);
        return u < 64 ? a : c;
    };
    // stage requested at the last unit of tile kt: bit i = piece slot i wanted (V of tile kt + 1, K of tile kt + 2)
    // (a tile NO wave needs - keys before a left-padded sequence's start - is still staged in its same variant: a wave multiplies
    //  P = 0 into it, and 0 x whatever-was-in-LDS could be NaN)
    const u64 same_st = same_blk | ~cross_blk;
    auto dma_mask = [&](const int kt) -> unsigned {
        unsigned dm = 0;
        if (kt + 1 < nkt) dm |= (((same_st >> (kt + 1)) & 1ull) ? 0x03u : 0u) | (((cross_blk >> (kt + 1)) & 1ull) ? 0x30u : 0u);
        if (kt + 2 < nkt) dm |= (((same_st >> (kt + 2)) & 1ull) ? 0x0cu : 0u) | (((cross_blk >> (kt + 2)) & 1ull) ? 0xc0u : 0u);
#if LIBRA_ATTN_DBG & 1
        dm = 0;
#endif
        return dm;
    };
    auto issue_all = [&](const unsigned dm, const int kt) {
#pragma unroll
        for (int i = 0; i < 8; ++i) if (dm & (1u << i)) piece(i, kt);
    };

    // ---- fragment addressing: lane constants; tile base, k-step and key half are uniform / immediate
    int kb[4], vb[4];
    {
        const int kswz = (l31 >> 1) & 7;
#pragma unroll
        for (int j = 0; j < 4; ++j) kb[j] = l31 * 128 + (((2 * j + fk) ^ kswz) << 4);
        const int pp = lane & 15, g16 = (lane >> 4) & 1;
        const int vrow = (4 * fk + (pp >> 2)) * 256 + ((pp & 1) << 3);       // keys 4fk + (p>>2), 2nd read +8
        const int tlo = (2 * g16 + ((pp & 3) >> 1)) << 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) vb[dt] = vrow + ((((dt ^ (pp >> 2)) & 3) << 6) | tlo);   // 32-line block dt of the 128-line (d) tile
    }

    f32x16 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 sA, sB;                                                  // S^T of the unit in flight: key halves 0 / 1
    union PK { bf16x8 v; unsigned u[4]; };
    PK pk[4];                                                       // P^T of the unit in flight as the four 16-key B operands
    const int qabs = q0w + l31;

#pragma unroll
    for (int ks = 0; ks < 8; ++ks) pin(qf[ks]);                     // Q has landed before the loop's LDS-DMA traffic starts

    union VA { bf16x8 v; s16x4 h2[2]; };
    // M phase: [O^T += V^T P^T of this unit: MFMAs 0-15 = (k-step st, 32-line d block dt)] [S^T = K Q^T of the next unit: MFMAs
    // 16-31 = (k-step ks, key half)].  ONE ring of NF operand fragments serves both products: the fragment of MFMA n + NF is
    // requested right after MFMA n has issued (its registers are free then), i.e. every LDS read runs NF - 1 MFMAs (~160 cycles)
    // ahead of its consumer and the phase holds 24 fragment registers instead of 64.  k-step st of P.V consumes accumulator
    // regs 8(st&1)..+7 of half st>>1 = local keys 32(st>>1) + 16(st&1) + 4fk + {0..3, 8..11}.  The two S accumulators alternate:
    // eight back-to-back MFMAs on ONE accumulator are a dependent chain that runs at half rate.
    constexpr int NF = 6;
    auto m_phase = [&](auto pv_c, auto qk_c, const char* vimg, const char* kimg, const unsigned dmA, const int kt) {
        constexpr bool PV = decltype(pv_c)::value && !(LIBRA_ATTN_DBG & 4), QK = decltype(qk_c)::value && !(LIBRA_ATTN_DBG & 8);
        if constexpr (!PV && decltype(pv_c)::value) { if (dmA) issue_all(dmA, kt); }
        if constexpr (!QK && decltype(qk_c)::value) { asm volatile("" : "+v"(sA), "+v"(sB)); }
        constexpr int N = (PV ? 16 : 0) + (QK ? 16 : 0), I0 = PV ? 0 : 16;
        bf16x8 F[NF];
        auto fread = [&](const int i) -> bf16x8 {
            if (i < 16) {
                const char* a = vimg + (i >> 2) * 4096 + vb[i & 3];
                VA t;
                t.h2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(a));
                t.h2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(a + 2048));
                return t.v;
            }
            const int ks = (i - 16) >> 1, hh = (i - 16) & 1;
            return *(const bf16x8*)(kimg + kb[ks & 3] + hh * 8192 + (ks >> 2) * 4096);
        };
#pragma unroll
        for (int n = 0; n < NF; ++n) F[n] = fread(I0 + n);
        if constexpr (QK) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { sA[r] = 0.f; sB[r] = 0.f; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < N; ++n) {
            const int i = I0 + n;
            if (i < 16) o[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[n % NF], pk[i >> 2].v, o[i & 3], 0, 0, 0);
            else if (i & 1) sB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[n % NF], qf[(i - 16) >> 1], sB, 0, 0, 0);
            else sA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[n % NF], qf[(i - 16) >> 1], sA, 0, 0, 0);
            if (n + NF < N) F[n % NF] = fread(i + NF);
            if (decltype(pv_c)::value && PV && i < 16 && (i & 1)) {                          // group 0's staging pieces ride between the P.V MFMAs
                __builtin_amdgcn_sched_barrier(0);
                if (dmA & (1u << (i >> 1))) piece(i >> 1, kt);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // per-element key mask of a MASKED unit: key <= query (causal), start <= key < len (padding), and the pair's modality
    // relation == the unit's variant.  One 64-bit word per lane, tested with compile-time bit positions.
    auto apply_mask = [&](const int kt, const int var) {
        const int kv0 = kt * BKV;
        const unsigned km0 = (unsigned)__builtin_amdgcn_readfirstlane((int)kmask[2 * kt]);
        const unsigned km1 = (unsigned)__builtin_amdgcn_readfirstlane((int)kmask[2 * kt + 1]);
        // cross pair <=> key bit != query bit; wanted <=> cross == var  =>  valid = ~(km ^ QV ^ VAR)
        const unsigned flip = ~((q_vis ? ~0u : 0u) ^ (var ? ~0u : 0u));
        // left padding: keys before `start` are masked for real queries; a padding QUERY row keeps them (its output is never used)
        const int lo = qabs < start ? 0 : start;
        int hi = qabs < len - 1 ? qabs : len - 1;                   // last valid key of this row
        const u64 rng = bits_below(hi - kv0 + 1) & ~bits_below(lo - kv0);
        const unsigned v0 = ((km0 ^ flip) & (unsigned)rng) >> (4 * fk);
        const unsigned v1 = ((km1 ^ flip) & (unsigned)(rng >> 32)) >> (4 * fk);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int bpos = (r & 3) + 8 * (r >> 2);                // local key of accumulator row r (minus 4 fk)
            sA[r] = ((v0 >> bpos) & 1u) ? sA[r] : -INFINITY;
            sB[r] = ((v1 >> bpos) & 1u) ? sB[r] : -INFINITY;
        }
    };
    // online softmax of (sA, sB) -> pk; the running max only advances when a tile exceeds it by 2^DEFER_THR
    auto softmax = [&]() {
#if LIBRA_ATTN_DBG & 2
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r0 = 8 * (st & 1) + 2 * j;
                pk[st].u[j] = st < 2 ? pack2bf(sA[r0], sA[r0 + 1]) : pack2bf(sB[r0], sB[r0 + 1]);
            }
        return;
#endif
        float tmax = max3f(sA[0], sA[1], sB[0]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) tmax = max3f(tmax, sA[r], sA[r + 1]);
#pragma unroll
        for (int r = 1; r < 15; r += 2) tmax = max3f(tmax, sB[r], sB[r + 1]);
        tmax = fmaxf(tmax, sB[15]);
        tmax = half_swap_max(tmax * p.sl2);
        const float m_new = fmaxf(m_run, tmax);
        if (__any(m_new > m_run + DEFER_THR)) {                     // wave-uniform; the first unit always lands here
            // a row that has seen no key yet has m_run = m_new = -inf: exp2(-inf - -inf) = NaN would poison o and l for good
            const float alpha = m_new == -INFINITY ? 1.f : __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
            m_run = m_new;
        }
        const float nm = m_run == -INFINITY ? 0.f : -m_run;         // (a row with no visible key yet stays at exactly 0)
        float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sA[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sA[r], p.sl2, nm));
            sB[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sB[r], p.sl2, nm));
            ps0 += sA[r];
            ps1 += sB[r];
        }
        l_run += ps0 + ps1;
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r0 = 8 * (st & 1) + 2 * j;
                pk[st].u[j] = st < 2 ? pack2bf(sA[r0], sA[r0 + 1]) : pack2bf(sB[r0], sB[r0 + 1]);
            }
    };
    // ---- main loop.  Global phase g: group 0 does SM_u at g = 2u and M_u at g = 2u + 1, group 1 one phase later.
    // A wave computes units [0, Uw) - Uw = one past its last non-skipped unit - with ONE code path (a skipped unit inside that
    // range, i.e. another wave's second pass, multiplies P = 0 into an operand tile the workgroup did load); the remaining units
    // [Uw, U) (tiles above this wave's diagonal) only keep the staging and barrier protocol going.
    const u64 act0 = __ballot((tab0 & 3u) != 0), act1 = __ballot((tab1 & 3u) != 0);
    const int Uw = act1 ? 128 - (int)__builtin_clzll(act1) : (act0 ? 64 - (int)__builtin_clzll(act0) : 0);
    // operand images of a unit: its own variant, or - skipped unit - whichever variant of the tile the workgroup loads
    auto var_of = [&](const unsigned e) -> int {
        const int kt = (int)(e >> 4);
        return (e & 3u) ? (int)((e >> 2) & 1u) : (((same_st >> kt) & 1ull) ? 0 : 1);
    };
    auto kimg_of = [&](const unsigned e) -> const char* { return smem + KRING + ((e >> 4) & 1) * SLOT_B + var_of(e) * TILE_B; };
    auto vimg_of = [&](const unsigned e) -> const char* { return smem + VRING + ((e >> 4) & 1) * SLOT_B + var_of(e) * TILE_B; };
    auto sm_phase = [&](const unsigned e, const unsigned dm) {
        const int kt = (int)(e >> 4);
        STAMP();                                                    // [0] SM start
        if (grp == 1 && dm) issue_all(dm, kt);                      // (group 0 issues the same stage between its MFMAs)
        if (e & 3u) {
#if !(LIBRA_ATTN_DBG & 32)
            if ((e & 3u) == 2u) apply_mask(kt, (int)((e >> 2) & 1u));
#endif
            softmax();
        } else {
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int j = 0; j < 4; ++j) pk[st].u[j] = 0u;
        }
        STAMP();                                                    // [1] softmax done
        if (grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stage this group requested one phase ago
        STAMP();                                                    // [2] staging wait done
        __builtin_amdgcn_sched_barrier(0);
        LOOP_BARRIER();
        __builtin_amdgcn_sched_barrier(0);
        STAMP();                                                    // [3] barrier passed = M start
    };
    auto m_end = [&]() {
        STAMP();                                                    // [4] MFMAs issued
        if (grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stage this group requested in its SM phase
        STAMP();                                                    // [5] staging wait done
        __builtin_amdgcn_sched_barrier(0);
        LOOP_BARRIER();
        __builtin_amdgcn_sched_barrier(0);
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                // K_0, V_0, K_1 landed
    unsigned e_cur = entry(0);
    if (grp == 1) __builtin_amdgcn_s_barrier();
    if (Uw > 0) m_phase(std::false_type{}, std::true_type{}, nullptr, kimg_of(e_cur), 0u, 0);
    __builtin_amdgcn_s_barrier();
    int u = 0;
    for (; u + 1 < Uw; ++u) {
        const unsigned e_nxt = entry(u + 1);
        const int kt = (int)(e_cur >> 4);
        const unsigned dm = (e_cur & 8u) ? dma_mask(kt) : 0u;
        sm_phase(e_cur, dm);
#if LIBRA_ATTN_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
        m_phase(std::true_type{}, std::true_type{}, vimg_of(e_cur), kimg_of(e_nxt), grp == 0 ? dm : 0u, kt);
#if LIBRA_ATTN_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        m_end();
        e_cur = e_nxt;
    }
    if (u < Uw) {                                                   // this wave's last unit: no next S
        const int kt = (int)(e_cur >> 4);
        const unsigned dm = (e_cur & 8u) ? dma_mask(kt) : 0u;
        sm_phase(e_cur, dm);
#if LIBRA_ATTN_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
        m_phase(std::true_type{}, std::false_type{}, vimg_of(e_cur), nullptr, grp == 0 ? dm : 0u, kt);
#if LIBRA_ATTN_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        m_end();
        ++u;
    }
    for (; u < U; ++u) {                                            // units above this wave's diagonal: staging duty only
        const unsigned e = entry(u);
        const int kt = (int)(e >> 4);
        const unsigned dm = (e & 8u) ? dma_mask(kt) : 0u;
        if (grp == 1 && dm) issue_all(dm, kt);
        if (grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LOOP_BARRIER();
        if (grp == 0 && dm) issue_all(dm, kt);
        m_end();
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();                     // re-align the two groups

#if LIBRA_ATTN_DBG & 128
    __syncthreads();
    if (dbg_blk && p.out_lo) for (int i = tid; i < 2048; i += 512) ((unsigned*)p.out_lo)[i] = ((unsigned*)(smem + STAMP_OFF))[i];
    if (dbg_blk && p.out_lo && tid == 0) { ((unsigned*)p.out_lo)[2048] = (unsigned)U; }
    if (dbg_blk && p.out_lo && lane == 0) ((unsigned*)p.out_lo)[2049 + wave] = (unsigned)Uw;
#endif
    // ---- finish ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    __syncthreads();
    constexpr int OROW = 264;                           // 128 bf16 + 8 B pad
    char* so = smem + wave * (32 * OROW);
    // two passes through the per-wave staging rows: the bf16 output, then (when asked for) its rounding residual
#pragma unroll 1
    for (int part = 0; part < ((p.out_lo && !(LIBRA_ATTN_DBG & 128)) ? 2 : 1); ++part) {
        if (part) __syncthreads();
        if (active) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = dt * 32 + 8 * g + 4 * fk;
                    float x[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        x[e] = o[dt][4 * g + e] * inv;
                        if (part) x[e] -= bf2f(f2bf(x[e]));
                    }
                    u32x2 w;
                    w[0] = pack2bf(x[0], x[1]);
                    w[1] = pack2bf(x[2], x[3]);
                    *(u32x2*)(so + l31 * OROW + d * 2) = w;
                }
            if (!part && p.lse && fk == 0 && q0w + l31 < S)
                p.lse[((long)b * p.H + h) * S + q0w + l31] =
                    l_tot > 0.f ? (m_run + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f : -INFINITY;
        }
        __syncthreads();
        if (active) {
            bf16_t* dst = part ? p.out_lo : p.out;
            // 32 rows x 256 B: lane -> (row = pass*4 + lane/16, 16-byte chunk lane%16)
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
                    *(u32x4*)(dst + (tok0 + qq) * p.ldo + h * BD + (lane & 15) * 8) = v;
                }
            }
        }
    }
}

}  // namespace libra

using namespace libra;

extern "C" int libra_bridge_attn_fwd(const void* q, int64_t ldq, const void* k_same, int64_t ldk, const void* k_cross,
                                     int64_t ldkc, const void* v_same, int64_t ldv, const void* v_cross, int64_t ldvc,
                                     const uint8_t* flag,
                                     const int32_t* kv_len, const int32_t* kv_start, void* out, int64_t ldo, float* lse,
                                     void* out_lo, int64_t B, int64_t S, int64_t H, float scale, void* stream) {
    if (B <= 0 || S <= 0) return LIBRA_OK;
    if (H <= 0 || ldq < H * BD || ldk < H * BD || ldv < H * BD || ldkc < H * BD || ldvc < H * BD || ldo < H * BD || S > 4096 ||
        ldk >= (1 << 18) || ldkc >= (1 << 18) || ldv >= (1 << 18) || ldvc >= (1 << 18))
        return LIBRA_ERR_SHAPE;
    if ((ldq % 8) || (ldk % 8) || (ldv % 8) || (ldkc % 8) || (ldvc % 8) || (ldo % 8)) return LIBRA_ERR_ALIGN;
    if (!q || !k_same || !k_cross || !v_same || !v_cross || !flag || !out) return LIBRA_ERR_ALIGN;
    if (((uintptr_t)q | (uintptr_t)k_same | (uintptr_t)k_cross | (uintptr_t)v_same | (uintptr_t)v_cross | (uintptr_t)out) & 15)
        return LIBRA_ERR_ALIGN;
    BridgeArgs a;
    a.q = (const bf16_t*)q; a.ldq = ldq; a.k_same = (const bf16_t*)k_same; a.k_cross = (const bf16_t*)k_cross; a.ldk = ldk; a.ldkc = ldkc;
    a.v_same = (const bf16_t*)v_same; a.v_cross = (const bf16_t*)v_cross; a.ldv = ldv; a.ldvc = ldvc;
    a.flag = flag; a.kv_len = kv_len; a.kv_start = kv_start; a.out = (bf16_t*)out; a.ldo = ldo; a.lse = lse; a.out_lo = (bf16_t*)out_lo;
    if (out_lo && ((uintptr_t)out_lo & 15)) return LIBRA_ERR_ALIGN;
    a.B = (int)B; a.S = (int)S; a.H = (int)H; a.n_qt = (int)((S + BQ - 1) / BQ);
    a.sl2 = scale * 1.4426950408889634f;
    const long nblk = (long)B * H * a.n_qt;
    if (nblk > 0x7fffffffL) return LIBRA_ERR_SHAPE;
    static std::atomic<bool> attr_set{false};     // (idempotent call; atomic only so that concurrent first launches do not race on the flag)
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)bridge_attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BR_LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL(bridge_attn_fwd_kernel, dim3((unsigned)nblk), dim3(512), BR_LDS, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}
