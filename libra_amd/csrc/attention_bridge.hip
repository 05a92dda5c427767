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
//   * the work list of a workgroup is a list of UNITS = (key tile, operand variant): unit (t, same) exists iff some wave has a
//     (row, key) pair of equal modality in tile t, (t, cross) iff some pair of different modality - disjoint (row, key) sets are
//     valid separate online-softmax steps.  Per wave a unit is skipped, PLAIN (no per-element test of any kind: interior tile,
//     all of the wave's pairs of this kind) or MASKED (causal diagonal / padding / pair kind selected by a per-lane 64-bit key
//     mask) - decided once per (wave, unit) in the prologue, lane-parallel, and kept in two registers (v_readlane per unit:
//     no loads, no mask code, a handful of SALU in the steady state);
//   * a unit is two phases: SM (online softmax of S_u: VALU only) and M = [O += V_u^T P_u ; S_{u+1} = K_{u+1} Q^T]
//     (32 MFMAs + their LDS fragment reads, one ring of 6 fragments 5 MFMAs ahead, addresses = lane constant + immediate);
//   * waves 0-3 and waves 4-7 (one of each per SIMD) run this sequence ONE PHASE APART (the second group passes one extra
//     s_barrier at the start): on every SIMD one wave owns the matrix pipe while its partner does its softmax on the VALU
//     (measured, experiments/probes: one wave streams 32x32x16 MFMAs at 32.4 cycles from registers / 36 with LDS fragments; the
//     partner's VALU runs at ~47 % of its stand-alone rate meanwhile);
//   * a unit's K and V tile (its ONE variant, 32 KiB) arrive by direct-to-LDS loads into a ring of 4 stages, stage = unit & 3:
//     every wave requests its 4 pieces of stage u + 2 at the top of its SM_u and never inside an M phase; group 0 has them in
//     flight for three phases (counted vmcnt(4) at the end of its next SM), group 1 for two (vmcnt(0) at the end of its M).
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
constexpr int STAGE_B = 2 * TILE_B;           // ring stage: K tile | V tile of one unit
constexpr int NSTAGE = 4;
constexpr int MASK_OFF = NSTAGE * STAGE_B;    // key-modality words of the sequence (<= 130 words; 1 KiB reserved)
constexpr int BLK_OFF = MASK_OFF + 1024;      // block-level unit sets (4 words)
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
// Unit table entry: bits 0-1 mode (0 skip, 1 plain, 2 masked), bit 2 operand variant (1 = cross), bits 3.. key tile.
typedef unsigned long long u64;
__device__ __forceinline__ u64 bits_below(int n) { return n >= 64 ? ~0ull : (n <= 0 ? 0ull : ((1ull << n) - 1ull)); }

__global__ __launch_bounds__(512, 2) void bridge_attn_fwd_kernel(const BridgeArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* kmask = (unsigned*)(smem + MASK_OFF);               // per 32 keys: bit j = key j is a vision token
    unsigned* blk = (unsigned*)(smem + BLK_OFF);                  // [0,1] tiles with a same unit, [2,3] tiles with a cross unit
    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int grp = wave >> 2;                                    // waves w and w + 4 share a SIMD: one of each group

    // ---- persistent workgroups, static schedule.  The launcher starts P workgroups (one per CU, P a multiple of n_qt); workgroup
    // w (after the XCD remap: XCD x owns w in [32x, 32x + 32)) handles the items i = w + k P, k = 0, 1, ...: sequence-head
    // bh = i / n_qt and query block (i + k) mod n_qt.  (i) at step k an XCD's 32 workgroups cover 4 (sequence, head) pairs in all
    // their 8 query blocks, which stream the same K / V tiles through that XCD's L2 at about the same time; (ii) the rotation by k
    // gives every workgroup every query block once per n_qt steps - the causal weights 4..32 tiles balance exactly, with no queue
    // and no atomics; (iii) a CU does not wait for a new workgroup to be dispatched (measured: ~7 us between two 8-wave / 134-KiB
    // workgroups on one CU, eight times per CU) and the slowest CU no longer carries 16 % more than the median.
    const int nitems = p.B * p.H * p.n_qt;
    const int P = (int)gridDim.x;
    const int w_id = xcd_remap(blockIdx.x, P);
#pragma unroll 1
    for (int step = 0, item = w_id; item < nitems; ++step, item += P) {
    int tid = tid0;                                                 // opaque per item: lane constants are re-derived inside the item instead of
    asm volatile("" : "+v"(tid));                                   // being hoisted out of the persistent loop and held in registers across it
    const int lane = tid & 63;
    const int fk = lane >> 5, l31 = lane & 31;
    const int qt = p.n_qt - 1 - ((item % p.n_qt + step) % p.n_qt);
    const int bh = item / p.n_qt;
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
#if LIBRA_ATTN_DBG & 256          // per-workgroup record over out_lo: start, loop start, loop end, end (cycles), hardware id, qt, units
    const unsigned long long wg_t0 = __builtin_readcyclecounter();
    const unsigned long long wg_r0 = __builtin_amdgcn_s_memrealtime();     // constant 100 MHz
    unsigned long long wg_t1 = 0, wg_t2 = 0, wg_pa = 0, wg_pb = 0, wg_pc = 0, wg_pd = 0;
#endif
#if LIBRA_ATTN_DBG & 128
    const bool dbg_blk = blockIdx.x == 0 && step == 0;
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
    // the 4 pieces (K j = 0,1; V j = 0,1) of this wave for the unit (key tile t, variant var) into ring stage `st`.
    // Interior tiles: loop-invariant per-lane byte offsets + a wave-uniform tile base (3 SALU per operand); the sequence's last,
    // ragged tile clamps its rows to the last token (their keys are masked).
    const unsigned ldkb_s = (unsigned)(p.ldk * 2), ldkb_c = (unsigned)(p.ldkc * 2), ldvb_s = (unsigned)(p.ldv * 2), ldvb_c = (unsigned)(p.ldvc * 2);
    const unsigned oKs0 = (unsigned)rowK * ldkb_s + colK0, oKs1 = (unsigned)(rowK + 8) * ldkb_s + colK1;
    const unsigned oKc0 = (unsigned)rowK * ldkb_c + colK0, oKc1 = (unsigned)(rowK + 8) * ldkb_c + colK1;
    const unsigned oVs0 = (unsigned)rV * ldvb_s + colV, oVs1 = (unsigned)(rV + 4) * ldvb_s + colV;
    const unsigned oVc0 = (unsigned)rV * ldvb_c + colV, oVc1 = (unsigned)(rV + 4) * ldvb_c + colV;
    auto stage_unit = [&](const int t, const int var, const int st) {
        const unsigned dst = lds0 + (unsigned)(st * STAGE_B + wave * 2048);
        if (t * BKV + BKV <= S) {
            if (var) {
                const bf16_t* kbase = kc_base + (long)t * BKV * p.ldkc;
                const bf16_t* vbase = vc_base + (long)t * BKV * p.ldvc;
                glds16_off_at(kbase, oKc0, dst); glds16_off_at(kbase, oKc1, dst + 1024);
                glds16_off_at(vbase, oVc0, dst + TILE_B); glds16_off_at(vbase, oVc1, dst + TILE_B + 1024);
            } else {
                const bf16_t* kbase = ks_base + (long)t * BKV * p.ldk;
                const bf16_t* vbase = vs_base + (long)t * BKV * p.ldv;
                glds16_off_at(kbase, oKs0, dst); glds16_off_at(kbase, oKs1, dst + 1024);
                glds16_off_at(vbase, oVs0, dst + TILE_B); glds16_off_at(vbase, oVs1, dst + TILE_B + 1024);
            }
            return;
        }
        const int lim = S - 1 - t * BKV;
        const unsigned ldkb = var ? ldkb_c : ldkb_s, ldvb = var ? ldvb_c : ldvb_s;
        const bf16_t* kbase = (var ? kc_base : ks_base) + (long)t * BKV * (var ? p.ldkc : p.ldk);
        const bf16_t* vbase = (var ? vc_base : vs_base) + (long)t * BKV * (var ? p.ldvc : p.ldv);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int row = rowK + 8 * j;
            row = row < lim ? row : lim;
            glds16_off_at(kbase, (unsigned)row * ldkb + (j ? colK1 : colK0), dst + j * 1024);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int row = rV + 4 * j;
            row = row < lim ? row : lim;
            glds16_off_at(vbase, (unsigned)row * ldvb + colV, dst + TILE_B + j * 1024);
        }
    };

    // ---- prologue: everything the first phases need is REQUESTED before the first wait (one workgroup per CU: nothing else
    // covers a prologue's serial round trips).  The first two stages are requested on a guess - (tile 0, same), (tile 0, cross):
    // right whenever tile 0 holds keys of both modalities (a BOS token in front of an image) - and re-requested below if the unit
    // list, known only after the mask pass and two barriers, starts differently.
    stage_unit(0, 0, 0);
    stage_unit(0, 1, 1);
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
#if LIBRA_ATTN_DBG & 256
    wg_pa = __builtin_readcyclecounter();
#endif
    __syncthreads();
#if LIBRA_ATTN_DBG & 256
    wg_pb = __builtin_readcyclecounter();
#endif

    // ---- per-wave classification of every key tile, lane = tile
    const bool wV = __ballot(q_vis && (q0w + l31) < S) != 0;        // this wave's query modalities
    const bool wL = __ballot(!q_vis && (q0w + l31) < S) != 0;
    unsigned m_same = 0, m_cross = 0;                               // this lane's tile: the wave's mode for its same / cross unit
    {
        const int kv0 = lane * BKV;
        const u64 mm = (u64)kmask[2 * lane] | ((u64)kmask[2 * lane + 1] << 32);
        const u64 rng = bits_below(len - kv0) & ~bits_below(start - kv0);       // valid keys of the tile
        const bool kV = (mm & rng) != 0, kL = (~mm & rng) != 0;
        const bool in = active && lane < nkt && kv0 <= q0w + 31;
        const bool wsame = in && ((wL && kL) || (wV && kV)), wcross = in && ((wL && kV) || (wV && kL));
        const bool full = kv0 + BKV - 1 <= q0w && kv0 >= start && kv0 + BKV <= len;
        const bool plain = full && !(wsame && wcross);              // every pair of the tile is of the wave's one kind
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
#if LIBRA_ATTN_DBG & 256
    wg_pc = __builtin_readcyclecounter();
#endif
    const u64 same_blk = (u64)(unsigned)__builtin_amdgcn_readfirstlane((int)blk[0]) | ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)blk[1]) << 32);
    const u64 cross_blk = (u64)(unsigned)__builtin_amdgcn_readfirstlane((int)blk[2]) | ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)blk[3]) << 32);
#if LIBRA_ATTN_DBG & 16
    const int U = 0;
#else
    const int U = __popcll(same_blk) + __popcll(cross_blk);         // units of this workgroup (<= 128)
#endif
    unsigned tab0, tab1;                                            // lane i: entry of unit i / unit 64 + i (0 past the end)
    {
        unsigned short* tab = (unsigned short*)(smem + TAB_OFF) + wave * 128;
        if (lane < nkt) {
            const u64 below = bits_below(lane);
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
    // the first two stages: keep the guesses that were right, re-request the others (after everybody's guesses have landed)
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

    // ---- fragment addressing: lane constants; stage base, k-step and key half are uniform / immediate
    int kb[4], vb[4];
    {
        const int kswz = (l31 >> 1) & 7;
#pragma unroll
        for (int j = 0; j < 4; ++j) kb[j] = l31 * 128 + (((2 * j + fk) ^ kswz) << 4);
        const int pp = lane & 15, g16 = (lane >> 4) & 1;
        const int vrow = TILE_B + (4 * fk + (pp >> 2)) * 256 + ((pp & 1) << 3);       // keys 4fk + (p>>2), 2nd read +8
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
    // ahead of its consumer and the phase holds 24 fragment registers.  k-step st of P.V consumes accumulator regs 8(st&1)..+7
    // of half st>>1 = local keys 32(st>>1) + 16(st&1) + 4fk + {0..3, 8..11}.
    constexpr int NF = 6;
    auto m_phase = [&](auto pv_c, auto qk_c, const char* vstage, const char* kstage) {
        constexpr bool PV = decltype(pv_c)::value && !(LIBRA_ATTN_DBG & 4), QK = decltype(qk_c)::value && !(LIBRA_ATTN_DBG & 8);
        if constexpr (!QK && decltype(qk_c)::value) { asm volatile("" : "+v"(sA), "+v"(sB)); }
        constexpr int N = (PV ? 16 : 0) + (QK ? 16 : 0), I0 = PV ? 0 : 16;
        bf16x8 F[NF];
        auto fread = [&](const int i) -> bf16x8 {
            if (i < 16) {
                const char* a = vstage + (i >> 2) * 4096 + vb[i & 3];
                VA t;
                t.h2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(a));
                t.h2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(a + 2048));
                return t.v;
            }
            const int ks = (i - 16) >> 1, hh = (i - 16) & 1;
            return *(const bf16x8*)(kstage + kb[ks & 3] + hh * 8192 + (ks >> 2) * 4096);
        };
        if constexpr (N > 0) {
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
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // per-lane key mask of a MASKED unit: key <= query (causal), start <= key < len (padding), and the pair's modality relation
    // == the unit's variant.  One 32-bit word per key half, bit positions compile-time after a shift by 4 fk.
    auto key_masks = [&](const int kt, const int var, unsigned& v0, unsigned& v1) {
        const int kv0 = kt * BKV;
        const unsigned km0 = (unsigned)__builtin_amdgcn_readfirstlane((int)kmask[2 * kt]);
        const unsigned km1 = (unsigned)__builtin_amdgcn_readfirstlane((int)kmask[2 * kt + 1]);
        // cross pair <=> key bit != query bit; wanted <=> cross == var  =>  valid = ~(km ^ QV ^ VAR)
        const unsigned flip = ~((q_vis ? ~0u : 0u) ^ (var ? ~0u : 0u));
        // left padding: keys before `start` are masked for real queries; a padding QUERY row keeps them (its output is never used)
        const int lo = qabs < start ? 0 : start;
        const int hi = qabs < len - 1 ? qabs : len - 1;             // last valid key of this row
        const u64 rng = bits_below(hi - kv0 + 1) & ~bits_below(lo - kv0);
        v0 = ((km0 ^ flip) & (unsigned)rng) >> (4 * fk);
        v1 = ((km1 ^ flip) & (unsigned)(rng >> 32)) >> (4 * fk);
    };
    // online softmax of (xA, xB) -> pk; the running max only advances when a tile exceeds it by 2^DEFER_THR.  The scores are
    // read-only here (a masked unit passes masked copies): an in-place mask made hipcc copy all 32 scores in every unit.
    auto softmax = [&](const float* xA, const float* xB) {
#if LIBRA_ATTN_DBG & 2
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r0 = 8 * (st & 1) + 2 * j;
                pk[st].u[j] = st < 2 ? pack2bf(xA[r0], xA[r0 + 1]) : pack2bf(xB[r0], xB[r0 + 1]);
            }
        return;
#endif
        // row max as a tree of v_max3 (independent ops: no wait states between them, depth 4 instead of 16)
        float t0 = max3f(xA[0], xA[1], xA[2]), t1 = max3f(xA[3], xA[4], xA[5]), t2 = max3f(xA[6], xA[7], xA[8]);
        float t3 = max3f(xA[9], xA[10], xA[11]), t4 = max3f(xA[12], xA[13], xA[14]), t5 = max3f(xA[15], xB[0], xB[1]);
        float t6 = max3f(xB[2], xB[3], xB[4]), t7 = max3f(xB[5], xB[6], xB[7]), t8 = max3f(xB[8], xB[9], xB[10]);
        float t9 = max3f(xB[11], xB[12], xB[13]), t10 = fmaxf(xB[14], xB[15]);
        t0 = max3f(t0, t1, t2); t3 = max3f(t3, t4, t5); t6 = max3f(t6, t7, t8); t9 = fmaxf(t9, t10);
        float tmax = fmaxf(max3f(t0, t3, t6), t9);
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
        float ps[4] = {0.f, 0.f, 0.f, 0.f};
        float pA[16], pB[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            pA[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(xA[r], p.sl2, nm));
            pB[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(xB[r], p.sl2, nm));
            ps[r & 1] += pA[r];
            ps[2 + (r & 1)] += pB[r];
        }
        l_run += (ps[0] + ps[1]) + (ps[2] + ps[3]);
        asm volatile("" : "+v"(l_run));                             // HERE: left alone, hipcc sinks the 32 adds into the tail of the M phase
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r0 = 8 * (st & 1) + 2 * j;
                pk[st].u[j] = st < 2 ? pack2bf(pA[r0], pA[r0 + 1]) : pack2bf(pB[r0], pB[r0 + 1]);
            }
    };

    // ---- main loop.  Global phase g: group 0 does SM_u at g = 2u and M_u at g = 2u + 1, group 1 one phase later.
    // A wave computes units [0, Uw) - Uw = one past its last non-skipped unit - with ONE code path (a skipped unit inside that
    // range, i.e. a unit of other waves' pair kind, multiplies P = 0 into the staged tile); the remaining units [Uw, U)
    // (tiles above this wave's diagonal) only keep the staging and barrier protocol going.
    const u64 act0 = __ballot((tab0 & 3u) != 0), act1 = __ballot((tab1 & 3u) != 0);
    const int Uw = act1 ? 128 - (int)__builtin_clzll(act1) : (act0 ? 64 - (int)__builtin_clzll(act0) : 0);
    auto stage_of = [&](const int u) -> const char* { return smem + (u & (NSTAGE - 1)) * STAGE_B; };
    // SM phase of unit u (entry e): request stage u + 2 (entry e2), softmax, wait for the stage the next M phases read
    auto sm_phase = [&](const int u, const unsigned e, const unsigned e2) {
        STAMP();                                                    // [0] SM start
        const bool issue = u + 2 < U && !(LIBRA_ATTN_DBG & 1);
        if (issue) stage_unit((int)(e2 >> 3), (int)((e2 >> 2) & 1u), (u + 2) & (NSTAGE - 1));
        // One softmax instance on 32 scalars.  A unit that is not plain first masks them IN PLACE (asm with tied operands: a
        // C select made hipcc copy all 32 scores - or all of P - in every unit, plain ones included); a skipped unit inside
        // the wave's range is a masked unit with an empty key set: P = 0, no rescale, nothing added to l.
        float a[16], c[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { a[r] = sA[r]; c[r] = sB[r]; }
        if ((e & 3u) != 1u && !(LIBRA_ATTN_DBG & 32)) {
            unsigned v0 = 0u, v1 = 0u;
            if ((e & 3u) == 2u) key_masks((int)(e >> 3), (int)((e >> 2) & 1u), v0, v1);
            const float ninf = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int bpos = (r & 3) + 8 * (r >> 2);            // local key of accumulator row r (minus 4 fk)
                const u64 k0 = __builtin_amdgcn_ballot_w64(((v0 >> bpos) & 1u) != 0), k1 = __builtin_amdgcn_ballot_w64(((v1 >> bpos) & 1u) != 0);
                asm volatile("v_cndmask_b32 %0, %2, %0, %1" : "+v"(a[r]) : "s"(k0), "v"(ninf));
                asm volatile("v_cndmask_b32 %0, %2, %0, %1" : "+v"(c[r]) : "s"(k1), "v"(ninf));
            }
        }
        softmax(a, c);
        STAMP();                                                    // [1] softmax done
        // group 0: its pieces of stage u + 1 (requested one SM ago) must have landed before the next phase reads K_{u+1}
        if (grp == 0) { if (issue) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        STAMP();                                                    // [2] staging wait done
        __builtin_amdgcn_sched_barrier(0);
        LOOP_BARRIER();
        __builtin_amdgcn_sched_barrier(0);
        STAMP();                                                    // [3] barrier passed = M start
    };
    auto m_end = [&]() {
        STAMP();                                                    // [4] MFMAs issued
        if (grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // group 1: its pieces of stage u + 2, requested in its SM_u
        STAMP();                                                    // [5] staging wait done
        __builtin_amdgcn_sched_barrier(0);
        LOOP_BARRIER();
        __builtin_amdgcn_sched_barrier(0);
    };
#if LIBRA_ATTN_DBG & 256
    wg_pd = __builtin_readcyclecounter();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                // stages 0 and 1 landed
#if LIBRA_ATTN_DBG & 256
    wg_t1 = __builtin_readcyclecounter();
#endif
    unsigned e_cur = entry(0), e_nxt = entry(1), e_dma = entry(2);
    if (grp == 1) __builtin_amdgcn_s_barrier();
    if (Uw > 0) m_phase(std::false_type{}, std::true_type{}, nullptr, stage_of(0));
    __builtin_amdgcn_s_barrier();
    int u = 0;
    for (; u + 1 < Uw; ++u) {
        sm_phase(u, e_cur, e_dma);
#if LIBRA_ATTN_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
        m_phase(std::true_type{}, std::true_type{}, stage_of(u), stage_of(u + 1));
#if LIBRA_ATTN_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        m_end();
        e_cur = e_nxt; e_nxt = e_dma; e_dma = entry(u + 3);
    }
    if (u < Uw) {                                                   // this wave's last unit: no next S
        sm_phase(u, e_cur, e_dma);
#if LIBRA_ATTN_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
        m_phase(std::true_type{}, std::false_type{}, stage_of(u), nullptr);
#if LIBRA_ATTN_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        m_end();
        e_cur = e_nxt; e_nxt = e_dma; e_dma = entry(u + 3);
        ++u;
    }
    for (; u < U; ++u) {                                            // units above this wave's diagonal: staging duty only
        const bool issue = u + 2 < U && !(LIBRA_ATTN_DBG & 1);
        if (issue) stage_unit((int)(e_dma >> 3), (int)((e_dma >> 2) & 1u), (u + 2) & (NSTAGE - 1));
        if (grp == 0) { if (issue) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        LOOP_BARRIER();
        m_end();
        e_cur = e_nxt; e_nxt = e_dma; e_dma = entry(u + 3);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();                     // re-align the two groups
#if LIBRA_ATTN_DBG & 256
    wg_t2 = __builtin_readcyclecounter();
#endif

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
    for (int part = 0; part < ((p.out_lo && !(LIBRA_ATTN_DBG & 384)) ? 2 : 1); ++part) {
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
#if LIBRA_ATTN_DBG & 256
    __syncthreads();
    if (tid == 0 && p.out_lo) {
        unsigned long long* rec = (unsigned long long*)p.out_lo + (long)item * 8;
        unsigned long long* rec2 = (unsigned long long*)p.out_lo + (long)nitems * 8 + (long)item * 4;
        rec2[0] = wg_pa - wg_t0; rec2[1] = wg_pb - wg_pa; rec2[2] = wg_pc - wg_pb; rec2[3] = wg_pd - wg_pc;
        unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        rec[0] = wg_t0; rec[1] = wg_t1; rec[2] = wg_t2; rec[3] = __builtin_readcyclecounter();
        rec[4] = hw; rec[5] = (unsigned long long)xcc | (wg_r0 << 8); rec[6] = (unsigned long long)qt | (__builtin_amdgcn_s_memrealtime() << 8); rec[7] = (unsigned long long)U;
    }
#endif
    __syncthreads();                                                // the next item's staging overwrites the output rows' LDS
    }   // persistent item loop
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
    const long nitems = (long)B * H * a.n_qt;
    if (nitems > 0x7fffffffL) return LIBRA_ERR_SHAPE;
    // persistent grid: one workgroup per (budgeted) CU, a multiple of n_qt (the kernel's rotation needs it), at most one per item
    const long nblk = persistent_grid(nitems, a.n_qt);
    static std::atomic<bool> attr_set{false};     // (idempotent call; atomic only so that concurrent first launches do not race on the flag)
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)bridge_attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BR_LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL(bridge_attn_fwd_kernel, dim3((unsigned)nblk), dim3(512), BR_LDS, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}
