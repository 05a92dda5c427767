// Fused routed-"bridge" causal flash attention (forward) for Libra's decoder layers, gfx950, head_dim 128.
//
// Reference semantics (LibraAttention.forward + attn_with_bridge, modeling_libra.py:267-414), closed form:
//     S_ij = q_i . (k_j + [m_i != m_j] kb_j) / sqrt(d) + causal/padding mask,   P = softmax_fp32(S)
//     O_i  = sum_j P_ij (v_j + [m_i != m_j] vb_j)
// where m is the per-token modality flag.  The reference evaluates this with TWO full QK^T and TWO full PV
// products and ~6 materialised [B,H,S,S] tensors (its own "TODO: make it more efficient", :288).  Here the
// caller provides the four operands K_same = rope(k), K_cross = rope(k + kb), V_same = v, V_cross = v + vb
// (libra_rope_bridge) and this kernel streams 32-key tiles; a tile pair whose queries and keys are all of one
// modality combination (the overwhelmingly common case: one contiguous 578-token image span per sequence)
// loads and multiplies only ONE variant; only modality-boundary tiles pay for both, selected per element.
//
// Structure = the ViT kernel's transposed scheme (S^T = K Q^T, O^T = V^T P^T with P^T fed straight from the
// accumulator registers), plus: V tiles are staged row-major as they lie in HBM and read with the LDS
// transpose load (ds_read_b64_tr_b16) — no V^T copy exists; keys beyond the causal diagonal or the
// sequence's valid length are masked; work-groups are ordered heaviest-first (causal imbalance).
#include <atomic>
#include <type_traits>
#include <utility>
#include "hip_common.hpp"
#include "gemm_tiles.hpp"
#include "attention_bridge_args.hpp"
#include "../../include/libra_hip.h"

namespace libra {

constexpr int BD = 128;            // head dim
constexpr int BQ = 256;            // query rows per workgroup (8 waves x 32)
constexpr int BKV = 64;            // keys per tile (two 32-key halves)
constexpr int VAR_BYTES = 2 * BKV * BD * 2;     // one variant: K tile (16 KiB) + V tile (16 KiB)
constexpr int STAGE_BYTES = 2 * VAR_BYTES;      // same + cross
constexpr int BR_LDS = 2 * STAGE_BYTES + 1024;  // double buffered + key-modality masks

// K tile image: four N-type [32 keys][64 d] sub-tiles (128-byte rows, chunk ^ ((row>>1)&7)), 4 KiB each, ordered
//               (key half, d half).  V tile image: T-type [64 keys][128 d] (256-byte rows, chunk ^ ((row&3)<<2)), 16 KiB.
__device__ __forceinline__ void stage_kv(const bf16_t* __restrict__ kp, unsigned ldk_b, const bf16_t* __restrict__ vp,
                                         unsigned ldv_b, int key0, int S, char* dst, int wave, int lane) {
    // (kp, vp: wave-uniform sequence/head bases; ld*_b: row strides in bytes; per-lane part is a 32-bit byte offset)
    // K: 16 pieces of 1 KiB (8 rows x 128 B); piece pc -> sub-tile pc>>2, rows 8*(pc&3)..; wave w takes pieces 2w, 2w+1
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pc = wave * 2 + j;
        const int st = pc >> 2, r = (pc & 3) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int key = key0 + (st >> 1) * 32 + r; key = key < S ? key : S - 1;
        glds16_off(kp, (unsigned)key * ldk_b + (unsigned)((st & 1) * 128 + c * 16), dst + pc * 1024);
    }
    // V: 16 pieces of 1 KiB (4 rows x 256 B); wave w takes pieces 2w, 2w+1
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pc = wave * 2 + j;
        const int r = pc * 4 + (lane >> 4);
        const int c = (lane & 15) ^ ((r & 3) << 2);
        int key = key0 + r; key = key < S ? key : S - 1;
        glds16_off(vp, (unsigned)key * ldv_b + (unsigned)(c * 16), dst + 16384 + pc * 1024);
    }
}

__global__ __launch_bounds__(512, 1) void bridge_attn_fwd_kernel(const BridgeArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* kmask = (unsigned*)(smem + 2 * STAGE_BYTES);        // per 32 keys: bit j = key j is a vision token
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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

    // ---- every per-lane global operand of the prologue is REQUESTED before the first wait: the query's modality byte and its Q
    // fragments (lane (q = l31, half fk) holds Q[q][16*ks + 8*fk .. +8], ks = 0..7) ride the same memory round trip as the flag
    // bytes of the mask pass (one workgroup per CU: nothing else covers a prologue's serial round trips - there were three)
    // (the first K / V tile goes out first of all, BOTH variants: which of them the tile needs is only known after the mask pass
    //  and two barriers - one more 32 KiB per workgroup buys the loop's first wait a head start of a full round trip)
    {
        const bf16_t* ks0 = p.k_same + tok0 * p.ldk + h * BD;
        const bf16_t* kc0 = p.k_cross + tok0 * p.ldkc + h * BD;
        const bf16_t* vs0 = p.v_same + tok0 * p.ldv + h * BD;
        const bf16_t* vc0 = p.v_cross + tok0 * p.ldvc + h * BD;
        stage_kv(ks0, (unsigned)p.ldk * 2u, vs0, (unsigned)p.ldv * 2u, 0, S, smem, wave, lane);
        stage_kv(kc0, (unsigned)p.ldkc * 2u, vc0, (unsigned)p.ldvc * 2u, 0, S, smem + VAR_BYTES, wave, lane);
    }
    const int q_vis_raw = p.flag[tok0 + q];
    bf16x8 qf[8];
    {
        const bf16_t* qp = p.q + (tok0 + q) * p.ldq + h * BD + fk * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
    }
    // ---- key-modality masks of this sequence into LDS (ballot over 32 flags) ----
    modality_masks(p.flag + tok0, S, kmask, tid, 512);
    const bool q_vis = q_vis_raw != 0;
    // block-level query modality presence (for staging decisions all waves must agree on)
    int* qpres = (int*)(kmask + 192);        // all LDS lives in the one dynamic array (a second __shared__ object
    if (tid < 2) qpres[tid] = 0;             // would make hipcc drain the direct-to-LDS queue before every ds_read)
    __syncthreads();
    {
        const bool valid = (q0w + l31) < S && fk == 0;
        if (__ballot(valid && q_vis)) if (lane == 0) atomicOr(&qpres[1], 1);
        if (__ballot(valid && !q_vis)) if (lane == 0) atomicOr(&qpres[0], 1);
    }
    __syncthreads();
    const bool blkL = __builtin_amdgcn_readfirstlane(qpres[0]) != 0, blkV = __builtin_amdgcn_readfirstlane(qpres[1]) != 0;
    const bool wV = __ballot(q_vis && (q0w + l31) < S) != 0;        // this wave's query modalities
    const bool wL = __ballot(!q_vis && (q0w + l31) < S) != 0;

    const bf16_t* ks_base = p.k_same + tok0 * p.ldk + h * BD;
    const bf16_t* kc_base = p.k_cross + tok0 * p.ldkc + h * BD;
    const bf16_t* vs_base = p.v_same + tok0 * p.ldv + h * BD;
    const bf16_t* vc_base = p.v_cross + tok0 * p.ldvc + h * BD;

    f32x16 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // causal: keys 0 .. min(S, (qt+1)*BQ) - 1
    int kend = (qt + 1) * BQ; kend = kend < S ? kend : S;
    const int nkt = (kend + BKV - 1) / BKV;

    // modality content of `n` keys starting at 32-key word w0 (n = 32 or 64), valid keys only
    auto key_mods = [&](int w0, int n, bool& kV, bool& kL) {
        // (readfirstlane: LDS data is wave-uniform here, and MFMAs under a branch the compiler believes divergent cost a
        //  full copy of every accumulator they touch)
        unsigned long long m = (unsigned)__builtin_amdgcn_readfirstlane((int)kmask[w0]);
        if (n == 64) m |= (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)kmask[w0 + 1]) << 32;
        int nvalid = S - w0 * 32; nvalid = nvalid > n ? n : nvalid;
        const unsigned long long full = nvalid >= 64 ? ~0ull : ((1ull << nvalid) - 1ull);
        kV = (m & full) != 0; kL = ((~m) & full) != 0;
    };
    auto stage = [&](int buf, int t) {
        bool kV, kL;
        key_mods(2 * t, 64, kV, kL);
        char* dst = smem + buf * STAGE_BYTES;
        if ((blkL && kL) || (blkV && kV))
            stage_kv(ks_base, (unsigned)p.ldk * 2u, vs_base, (unsigned)p.ldv * 2u, t * BKV, S, dst, wave, lane);
        if ((blkL && kV) || (blkV && kL))
            stage_kv(kc_base, (unsigned)p.ldkc * 2u, vc_base, (unsigned)p.ldvc * 2u, t * BKV, S, dst + VAR_BYTES, wave, lane);
    };
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) pin(qf[ks]);                     // Q has landed before the loop's LDS-DMA traffic starts

    // fragment addressing.  The lane-derived LDS offsets are recomputed per tile from an opaque copy of the lane id:
    // hoisted to kernel entry they are ten long-lived registers that hipcc spills around the tile loop, and a scratch
    // reload inside the loop is a vmcnt(0) drain of the LDS-DMA queue.
    int lane_o = lane;
    // S^T (2 x 32 keys x 32 queries) of both key halves of the K image at `kimg`.  The two accumulators alternate: eight
    // back-to-back MFMAs on ONE accumulator are a dependent chain that runs at half rate (round-1 cycle stamps: 1400 cycles
    // for the 16 QK MFMAs of a tile).
    auto qk_pair = [&](const char* kimg, f32x16& s0, f32x16& s1) {
        const int l31o = lane_o & 31, fko = lane_o >> 5;
        const int kswz = (l31o >> 1) & 7;
        const char* krow = kimg + l31o * 128;
        // Fragment reads run FOUR k-steps (8 MFMAs, > one LDS round trip) ahead of their MFMAs: left to itself hipcc keeps one pair
        // in flight and every MFMA waits out most of an LDS latency (s_waitcnt lgkmcnt(0) in front of each: the 16 MFMAs of a tile
        // took ~1400 cycles for 512 of matrix pipe).  kf[h][j]: key half h, k-step j (mod 4).
        bf16x8 kf[2][4];
        auto rd = [&](int h, int ks) -> bf16x8 {
            const int c = (2 * (ks & 3) + fko) ^ kswz;
            return *(const bf16x8*)(krow + h * 8192 + (ks >> 2) * 4096 + (c << 4));
        };
#pragma unroll
        for (int j = 0; j < 4; ++j) { kf[0][j] = rd(0, j); kf[1][j] = rd(1, j); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0][ks & 3], qf[ks], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1][ks & 3], qf[ks], s1, 0, 0, 0);
            if (ks < 4) { kf[0][ks] = rd(0, ks + 4); kf[1][ks] = rd(1, ks + 4); }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // O^T += V^T P^T for one 16-key step: `vstep` = V image + 4096 * step
    // the 8 transpose reads of a 16-key step go out before its 4 MFMAs (one LDS latency per step instead of one per MFMA)
    auto pv_step = [&](const char* vstep, const bf16x8 pk) {
        const int pp = lane_o & 15, g16 = (lane_o >> 4) & 1;
        const char* vrow = vstep + (4 * (lane_o >> 5) + (pp >> 2)) * 256 + ((pp & 1) << 3);   // keys 4fk + (p>>2), 2nd read +8
        const int tlo = (2 * g16 + ((pp & 3) >> 1)) << 4;
        union { bf16x8 v; s16x4 h2[2]; } va[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            // 32-line block dt of the 128-line (d) T-type tile
            const char* a = vrow + ((((dt ^ (pp >> 2)) & 3) << 6) | tlo);
            va[dt].h2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(a));
            va[dt].h2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(a + 2048));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[dt].v, pk, o[dt], 0, 0, 0);
    };
    auto rescale = [&](float alpha) {
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
    };

    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int cur = kt & 1;
        if (kt + 1 < nkt) stage(cur ^ 1, kt + 1);
        if (!active) continue;
        const int kv0 = kt * BKV;
        if (kv0 > q0w + 31) continue;                               // tile entirely above this wave's diagonal
        asm volatile("" : "+v"(lane_o));
        const char* sks = smem + cur * STAGE_BYTES;                 // same variant: K (4 x 4 KiB), V at +16384
        const char* skc = sks + VAR_BYTES;
        bool kV, kL;
        key_mods(2 * kt, 64, kV, kL);
        const bool wsame = (wL && kL) || (wV && kV);
        const bool wcross = (wL && kV) || (wV && kL);

        const bool mixed = wsame && wcross;                         // both variants present: select per element
        const char* img1 = wsame ? sks : skc;                       // primary variant (same unless only cross is needed)

        // ---- S^T = K Q^T, 64 keys x 32 queries ----
        f32x16 sA, sB;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sA[r] = 0.f; sB[r] = 0.f; }
        qk_pair(img1, sA, sB);
        unsigned crA = 0, crB = 0;                                  // bit r: element r takes the cross variant (mixed tiles)
        if (mixed) {
            f32x16 tA, tB;
#pragma unroll
            for (int r = 0; r < 16; ++r) { tA[r] = 0.f; tB[r] = 0.f; }
            qk_pair(skc, tA, tB);
            const unsigned km0 = (unsigned)__builtin_amdgcn_readfirstlane((int)kmask[2 * kt]);
            const unsigned km1 = (unsigned)__builtin_amdgcn_readfirstlane((int)kmask[2 * kt + 1]);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kl = (r & 3) + 8 * (r >> 2) + 4 * fk;     // local key of accumulator row r
                const bool ca = (((km0 >> kl) & 1u) != 0) != q_vis, cb = (((km1 >> kl) & 1u) != 0) != q_vis;
                sA[r] = ca ? tA[r] : sA[r];
                sB[r] = cb ? tB[r] : sB[r];
                crA |= (ca ? 1u : 0u) << r;
                crB |= (cb ? 1u : 0u) << r;
            }
        }
        if (kv0 + BKV - 1 > q0w || kv0 + BKV > len || kv0 < start) {   // causal diagonal / padded keys inside this tile
            const int qabs = q0w + l31;
            // left padding: keys before `start` are masked for real queries; a padding QUERY row keeps them (its output is
            // never used, but an all-masked row would be NaN and 0 x NaN would leak through P.V of later rows' tiles)
            const int lo = qabs < start ? 0 : start;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kv0 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                sA[r] = (key <= qabs && key < len && key >= lo) ? sA[r] : -INFINITY;
                sB[r] = (key + 32 <= qabs && key + 32 < len && key + 32 >= lo) ? sB[r] : -INFINITY;
            }
        }
        // ---- online softmax; the running max only advances when a tile exceeds it by 2^DEFER_THR ----
        float tmax = max3f(sA[0], sA[1], sB[0]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) tmax = max3f(tmax, sA[r], sA[r + 1]);
#pragma unroll
        for (int r = 1; r < 15; r += 2) tmax = max3f(tmax, sB[r], sB[r + 1]);
        tmax = fmaxf(tmax, sB[15]);
        tmax = half_swap_max(tmax * p.sl2);
        const float m_new = fmaxf(m_run, tmax);
        if (__any(m_new > m_run + DEFER_THR)) {                     // wave-uniform; the first tile always lands here
            // a row that has seen no key yet (left padding: a whole tile masked for the real rows while the pad rows of the
            // same wave keep theirs) has m_run = m_new = -inf: exp2(-inf - -inf) = NaN would poison o and l for good
            rescale(m_new == -INFINITY ? 1.f : __builtin_amdgcn_exp2f(m_run - m_new));
            m_run = m_new;
        }
        const float nm = m_run == -INFINITY ? 0.f : -m_run;         // (a row with no visible key yet stays at exactly 0)
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sA[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sA[r], p.sl2, nm));
            sB[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sB[r], p.sl2, nm));
            psum += sA[r] + sB[r];
        }
        l_run += psum;
        // ---- O^T += V^T P^T; k-step st consumes accumulator regs 8(st&1)..+7 of half st>>1 = local keys
        //      32(st>>1) + 16(st&1) + 4fk + {0..3, 8..11}
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            union { bf16x8 v; unsigned u[4]; } pk, pk2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r0 = 8 * (st & 1) + 2 * j;
                pk.u[j] = st < 2 ? pack2bf(sA[r0], sA[r0 + 1]) : pack2bf(sB[r0], sB[r0 + 1]);
            }
            if (mixed) {                                            // split P by variant (bf16 pair masks)
                const unsigned cr = (st < 2 ? crA : crB) >> (8 * (st & 1));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned keep2 = (((cr >> (2 * j)) & 1u) ? 0xffffu : 0u) | (((cr >> (2 * j + 1)) & 1u) ? 0xffff0000u : 0u);
                    pk2.u[j] = pk.u[j] & keep2;
                    pk.u[j] &= ~keep2;
                }
            }
            pv_step(img1 + 16384 + st * 4096, pk.v);
            if (mixed) pv_step(skc + 16384 + st * 4096, pk2.v);
        }
    }

    // ---- finish ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    __syncthreads();
    constexpr int OROW = 264;                           // 128 bf16 + 8 B pad
    char* so = smem + wave * (32 * OROW);
    // two passes through the per-wave staging rows: the bf16 output, then (when asked for) its rounding residual
#pragma unroll 1
    for (int part = 0; part < (p.out_lo ? 2 : 1); ++part) {
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


// ================================================================================================
// Second structure of the same contract: FOUR waves x 64 query rows, ONE wave per SIMD with the whole register file (O^T
// 128 + S^T 64 + Q 64 accumulator / operand registers; hipcc places what only the matrix pipe touches in AGPRs).  Why: the 8 x 32
// structure above spends ~615 instructions per wave and 64-key tile on 32 MFMAs (10 VALU + 6 SALU + 2 LDS per MFMA by PMC) and its
// two in-order waves per SIMD do not hide each other; with 64 rows per wave every K / V^T fragment read feeds TWO MFMAs, the
// per-tile control (tile classification, loop, barrier, DMA issue) is paid once per 64 MFMAs, and a pure interior tile runs a
// loop body without mask, padding or variant-select code (`tile<false>`); boundary / mixed-modality / padded tiles take the
// general body (`tile<true>`).  Same LDS images, same staging, same numerics (running max, deferred rescale, exp2, bf16 P).
constexpr int F_BQ = 256;            // query rows per workgroup (4 waves x 64)

// K / V tile of one variant by 4 waves: K pieces 4w..4w+3, V pieces 4w..4w+3 (piece = 1 KiB, images as in stage_kv)
__device__ __forceinline__ void stage_kv4(const bf16_t* __restrict__ kp, unsigned ldk_b, const bf16_t* __restrict__ vp,
                                          unsigned ldv_b, int key0, int S, char* dst, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pc = wave * 4 + j;
        const int st = pc >> 2, r = (pc & 3) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int key = key0 + (st >> 1) * 32 + r; key = key < S ? key : S - 1;
        glds16_off(kp, (unsigned)key * ldk_b + (unsigned)((st & 1) * 128 + c * 16), dst + pc * 1024);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pc = wave * 4 + j;
        const int r = pc * 4 + (lane >> 4);
        const int c = (lane & 15) ^ ((r & 3) << 2);
        int key = key0 + r; key = key < S ? key : S - 1;
        glds16_off(vp, (unsigned)key * ldv_b + (unsigned)(c * 16), dst + 16384 + pc * 1024);
    }
}

// ---- asm-owned accumulators.  O^T (8 blocks of 32 d x 32 queries = 128 registers) lives in AGPRs a[0:127] that ONLY the inline asm
// below names: hipcc never sees it as a value, so it cannot shuffle it between register files (left to the compiler the 4 x 64 kernel
// spent ~200 v_accvgpr moves per tile and spilled).  Every MFMA of the kernel is inline asm (with builtin MFMAs hipcc may pick the
// AGPR form and allocate the same registers); the arch VGPRs carry S^T, Q, the fragments and the softmax.  hipcc neither inserts
// wait states for these statements nor sees their latency: they are placed here (cdna_hip_programming.md 5.7).  Audit after every
// edit (experiments/tools/isa_blocks.py): no spills, no scratch, and no v_accvgpr / a[..] outside these statements.
#define LIBRA_A16(n) "a" #n
#define LIBRA_ACC_CLOBBER                                                                                                  \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19",   \
    "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37",     \
    "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55",     \
    "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73",     \
    "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91",     \
    "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108",  \
    "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", \
    "a125", "a126", "a127"
//   S^T block (arch VGPRs only)
__device__ __forceinline__ void mfma_s_first(f32x16& d, const bf16x8 k, const bf16x8 q) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(k), "v"(q));
}
__device__ __forceinline__ void mfma_s(f32x16& d, const bf16x8 k, const bf16x8 q) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(k), "v"(q));
}
//   O^T block B (= 2 dt + qb) += V^T fragment x P^T fragment; `s_nop 1` covers a P fragment the VALU has just packed
template <int B>
__device__ __forceinline__ void mfma_o(const bf16x8 v, const bf16x8 pk) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(v), "v"(pk), "i"(B * 16), "i"(B * 16 + 15)
                 : LIBRA_ACC_CLOBBER);
}
template <int N>
__device__ __forceinline__ float acc_get() {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(x) : "i"(N));
    return x;
}
template <int N>
__device__ __forceinline__ void acc_set(float x) {
    asm volatile("v_accvgpr_write_b32 a%c1, %0" ::"v"(x), "i"(N) : LIBRA_ACC_CLOBBER);
}
template <int N0, int... I>
__device__ __forceinline__ void acc_zero_seq(std::integer_sequence<int, I...>) { (acc_set<N0 + I>(0.f), ...); }
template <int N0, int... I>
__device__ __forceinline__ void acc_scale_seq(float f, std::integer_sequence<int, I...>) { (acc_set<N0 + I>(acc_get<N0 + I>() * f), ...); }
// the 16 registers of block B as floats (epilogue)
template <int B, int... I>
__device__ __forceinline__ void acc_read_block(float* x, std::integer_sequence<int, I...>) { ((x[I] = acc_get<B * 16 + I>()), ...); }
// MFMA write -> v_accvgpr_read of the same register: the wait states nobody inserts for asm
__device__ __forceinline__ void acc_settle() { asm volatile("s_nop 15" ::: "memory"); }

__global__ __launch_bounds__(256, 1) void bridge_attn_fwd64_kernel(const BridgeArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* kmask = (unsigned*)(smem + 2 * STAGE_BYTES);        // per 32 keys: bit j = key j is a vision token
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const int q0w = qt * F_BQ + wave * 64;
    const bool active = q0w < S;

    const bf16_t* ks_base = p.k_same + tok0 * p.ldk + h * BD;
    const bf16_t* kc_base = p.k_cross + tok0 * p.ldkc + h * BD;
    const bf16_t* vs_base = p.v_same + tok0 * p.ldv + h * BD;
    const bf16_t* vc_base = p.v_cross + tok0 * p.ldvc + h * BD;
    // prologue: one memory round trip (first K / V tile of BOTH variants, modality bytes, Q fragments, mask pass)
    stage_kv4(ks_base, (unsigned)p.ldk * 2u, vs_base, (unsigned)p.ldv * 2u, 0, S, smem, wave, lane);
    stage_kv4(kc_base, (unsigned)p.ldkc * 2u, vc_base, (unsigned)p.ldvc * 2u, 0, S, smem + VAR_BYTES, wave, lane);
    int qrow[2], qvis_raw[2];
    bf16x8 qf[2][8];                                               // Q[q][16 ks + 8 fk .. +8] of query block qb
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        int q = q0w + 32 * qb + l31;
        q = q < S ? q : S - 1;
        qrow[qb] = q;
        qvis_raw[qb] = p.flag[tok0 + q];
        const bf16_t* qp = p.q + (tok0 + q) * p.ldq + h * BD + fk * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[qb][ks] = *(const bf16x8*)(qp + ks * 16);
    }
    modality_masks(p.flag + tok0, S, kmask, tid, 256);
    const bool q_vis[2] = {qvis_raw[0] != 0, qvis_raw[1] != 0};
    int* qpres = (int*)(kmask + 192);
    if (tid < 2) qpres[tid] = 0;
    __syncthreads();
    bool wV = false, wL = false;                                   // this wave's query modalities (valid rows only)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const bool valid = (q0w + 32 * qb + l31) < S;
        wV = wV || __ballot(valid && q_vis[qb]) != 0;
        wL = wL || __ballot(valid && !q_vis[qb]) != 0;
    }
    if (lane == 0) { if (wV) atomicOr(&qpres[1], 1); if (wL) atomicOr(&qpres[0], 1); }
    __syncthreads();
    const bool blkL = __builtin_amdgcn_readfirstlane(qpres[0]) != 0, blkV = __builtin_amdgcn_readfirstlane(qpres[1]) != 0;

    // O^T [32 d x 32 queries] blocks (d tile dt, query block qb) = AGPRs a[(2 dt + qb) 16 .. +15], asm-owned (see above)
    acc_zero_seq<0>(std::make_integer_sequence<int, 128>{});
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};

    int kend = (qt + 1) * F_BQ; kend = kend < S ? kend : S;
    const int nkt = (kend + BKV - 1) / BKV;
    auto key_mods = [&](int w0, bool& kV, bool& kL) {               // modality content of the 64 keys at mask word w0 (valid keys only)
        unsigned long long m = (unsigned)__builtin_amdgcn_readfirstlane((int)kmask[w0]);
        m |= (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)kmask[w0 + 1]) << 32;
        int nvalid = S - w0 * 32; nvalid = nvalid > 64 ? 64 : nvalid;
        const unsigned long long full = nvalid >= 64 ? ~0ull : ((1ull << nvalid) - 1ull);
        kV = (m & full) != 0; kL = ((~m) & full) != 0;
    };
    auto stage = [&](int buf, int t) {
        bool kV, kL;
        key_mods(2 * t, kV, kL);
        char* dst = smem + buf * STAGE_BYTES;
        if ((blkL && kL) || (blkV && kV))
            stage_kv4(ks_base, (unsigned)p.ldk * 2u, vs_base, (unsigned)p.ldv * 2u, t * BKV, S, dst, wave, lane);
        if ((blkL && kV) || (blkV && kL))
            stage_kv4(kc_base, (unsigned)p.ldkc * 2u, vc_base, (unsigned)p.ldvc * 2u, t * BKV, S, dst + VAR_BYTES, wave, lane);
    };
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) pin(qf[qb][ks]);            // Q has landed before the loop's LDS-DMA traffic starts

    int lane_o = lane;                                             // (opaque per tile: see the 8-wave kernel)
    // S^T of both key halves and both query blocks against the K image at `kimg`: every K fragment feeds two MFMAs
    auto qk = [&](const char* kimg, f32x16 (&s)[2][2]) {
        const int l31o = lane_o & 31, fko = lane_o >> 5;
        const int kswz = (l31o >> 1) & 7;
        const char* krow = kimg + l31o * 128;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int c = (2 * (ks & 3) + fko) ^ kswz;
            const bf16x8 k0 = *(const bf16x8*)(krow + (ks >> 2) * 4096 + (c << 4));
            const bf16x8 k1 = *(const bf16x8*)(krow + 8192 + (ks >> 2) * 4096 + (c << 4));
            if (ks == 0) {
                mfma_s_first(s[0][0], k0, qf[0][ks]); mfma_s_first(s[1][0], k1, qf[0][ks]);
                mfma_s_first(s[0][1], k0, qf[1][ks]); mfma_s_first(s[1][1], k1, qf[1][ks]);
            } else {
                mfma_s(s[0][0], k0, qf[0][ks]); mfma_s(s[1][0], k1, qf[0][ks]);
                mfma_s(s[0][1], k0, qf[1][ks]); mfma_s(s[1][1], k1, qf[1][ks]);
            }
        }
        // the last MFMAs' results are read by the VALU next: their wait states (8-pass XDL write -> VALU read)
        asm volatile("s_nop 15" : "+v"(s[0][0]), "+v"(s[1][0]), "+v"(s[0][1]), "+v"(s[1][1]));
    };
    // O^T += V^T P^T for one 16-key step and both query blocks: every V^T fragment feeds two MFMAs
    auto pv_step = [&](const char* vstep, const bf16x8 pk0, const bf16x8 pk1) {
        const int pp = lane_o & 15, g16 = (lane_o >> 4) & 1;
        const char* vrow = vstep + (4 * (lane_o >> 5) + (pp >> 2)) * 256 + ((pp & 1) << 3);
        const int tlo = (2 * g16 + ((pp & 3) >> 1)) << 4;
        union { bf16x8 v; s16x4 h2[2]; } va[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const char* a = vrow + ((((dt ^ (pp >> 2)) & 3) << 6) | tlo);
            va[dt].h2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(a));
            va[dt].h2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(a + 2048));
        }
        mfma_o<0>(va[0].v, pk0); mfma_o<1>(va[0].v, pk1);
        mfma_o<2>(va[1].v, pk0); mfma_o<3>(va[1].v, pk1);
        mfma_o<4>(va[2].v, pk0); mfma_o<5>(va[2].v, pk1);
        mfma_o<6>(va[3].v, pk0); mfma_o<7>(va[3].v, pk1);
    };

    // One 64-key tile.  GEN = false: every row of the wave sees every key of the tile through ONE operand variant (`img1`): no mask,
    // no select.  GEN = true: the general tile (causal diagonal, padding, both variants present).
    auto tile = [&](auto gen, const int kt, const char* img1, const char* skc, const bool mixed) {
        constexpr bool GEN = decltype(gen)::value;
        const int kv0 = kt * BKV;
        f32x16 s[2][2];                                             // [key half][query block]
        qk(img1, s);
        unsigned cr[2][2] = {{0, 0}, {0, 0}};                       // [key half][qb] bit r: element r takes the cross variant
        if constexpr (GEN) {
            if (mixed) {
                // (one 32 x 32 block of the cross product at a time: a second full S^T set would push the kernel past the register
                //  file - this is the rare path, its K fragments are simply read once per query block)
                const unsigned km[2] = {(unsigned)__builtin_amdgcn_readfirstlane((int)kmask[2 * kt]),
                                        (unsigned)__builtin_amdgcn_readfirstlane((int)kmask[2 * kt + 1])};
                const int l31o = lane_o & 31, fko = lane_o >> 5;
                const int kswz = (l31o >> 1) & 7;
                const char* krow = skc + l31o * 128;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb) {
                        f32x16 t;
#pragma unroll
                        for (int ks = 0; ks < 8; ++ks) {
                            const int c = (2 * (ks & 3) + fko) ^ kswz;
                            const bf16x8 kx = *(const bf16x8*)(krow + i * 8192 + (ks >> 2) * 4096 + (c << 4));
                            if (ks == 0) mfma_s_first(t, kx, qf[qb][ks]);
                            else mfma_s(t, kx, qf[qb][ks]);
                        }
                        asm volatile("s_nop 15" : "+v"(t));
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int kl = (r & 3) + 8 * (r >> 2) + 4 * fk;
                            const bool c = (((km[i] >> kl) & 1u) != 0) != q_vis[qb];
                            s[i][qb][r] = c ? t[r] : s[i][qb][r];
                            cr[i][qb] |= (c ? 1u : 0u) << r;
                        }
                    }
            }
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const int q0b = q0w + 32 * qb;
                if (kv0 + BKV - 1 > q0b || kv0 + BKV > len || kv0 < start) {   // diagonal / padded keys for this query block
                    const int qabs = q0b + l31;
                    const int lo = qabs < start ? 0 : start;               // (a padding QUERY row keeps the padded keys: see above)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int key = kv0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fk;
                            s[i][qb][r] = (key <= qabs && key < len && key >= lo) ? s[i][qb][r] : -INFINITY;
                        }
                }
            }
        }
        // ---- online softmax per query block; the running max only advances when a tile exceeds it by 2^DEFER_THR ----
        float nm[2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float tmax = max3f(s[0][qb][0], s[0][qb][1], s[1][qb][0]);
#pragma unroll
            for (int r = 2; r < 16; r += 2) tmax = max3f(tmax, s[0][qb][r], s[0][qb][r + 1]);
#pragma unroll
            for (int r = 1; r < 15; r += 2) tmax = max3f(tmax, s[1][qb][r], s[1][qb][r + 1]);
            tmax = fmaxf(tmax, s[1][qb][15]);
            tmax = half_swap_max(tmax * p.sl2);
            const float m_new = fmaxf(m_run[qb], tmax);
            if (__any(m_new > m_run[qb] + DEFER_THR)) {             // wave-uniform; the first tile always lands here
                const float alpha = m_new == -INFINITY ? 1.f : __builtin_amdgcn_exp2f(m_run[qb] - m_new);
                l_run[qb] *= alpha;
                if (m_run[qb] != -INFINITY || true) {               // O of this query block (blocks 2 dt + qb) *= alpha: rare past the first tiles
                    acc_settle();
                    if (qb == 0) {
                        acc_scale_seq<0>(alpha, std::make_integer_sequence<int, 16>{}); acc_scale_seq<32>(alpha, std::make_integer_sequence<int, 16>{});
                        acc_scale_seq<64>(alpha, std::make_integer_sequence<int, 16>{}); acc_scale_seq<96>(alpha, std::make_integer_sequence<int, 16>{});
                    } else {
                        acc_scale_seq<16>(alpha, std::make_integer_sequence<int, 16>{}); acc_scale_seq<48>(alpha, std::make_integer_sequence<int, 16>{});
                        acc_scale_seq<80>(alpha, std::make_integer_sequence<int, 16>{}); acc_scale_seq<112>(alpha, std::make_integer_sequence<int, 16>{});
                    }
                }
                m_run[qb] = m_new;
            }
            nm[qb] = m_run[qb] == -INFINITY ? 0.f : -m_run[qb];
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float psum = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[i][qb][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[i][qb][r], p.sl2, nm[qb]));
                    psum += s[i][qb][r];
                }
            l_run[qb] += psum;
        }
        // ---- O^T += V^T P^T; k-step st consumes accumulator regs 8(st&1)..+7 of key half st>>1
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            union { bf16x8 v; unsigned u[4]; } pk[2], pk2[2];
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r0 = 8 * (st & 1) + 2 * j;
                    pk[qb].u[j] = pack2bf(s[st >> 1][qb][r0], s[st >> 1][qb][r0 + 1]);
                }
                if constexpr (GEN) {
                    if (mixed) {                                    // split P by variant (bf16 pair masks)
                        const unsigned c = cr[st >> 1][qb] >> (8 * (st & 1));
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const unsigned keep2 = (((c >> (2 * j)) & 1u) ? 0xffffu : 0u) | (((c >> (2 * j + 1)) & 1u) ? 0xffff0000u : 0u);
                            pk2[qb].u[j] = pk[qb].u[j] & keep2;
                            pk[qb].u[j] &= ~keep2;
                        }
                    }
                }
            }
            pv_step(img1 + 16384 + st * 4096, pk[0].v, pk[1].v);
            if constexpr (GEN) {
                if (mixed) pv_step(skc + 16384 + st * 4096, pk2[0].v, pk2[1].v);
            }
        }
    };

    // ---- the plain tile, scheduled by hand.  Every MFMA is an `asm volatile` statement and hipcc keeps the program order around those,
    // so the SOURCE ORDER below is the schedule: the tile's two 32-key halves are two online-softmax steps, and the softmax of one
    // half (VALU: max, exp2, sums, bf16 pack) is written between the MFMAs of the other half's product:
    //      A  S(h0) = K(h0) Q^T                     16 MFMA   | K(h0) fragments 4 k-steps ahead, then K(h1)'s first four
    //      B  S(h1) = K(h1) Q^T                     16 MFMA   | softmax(h0), V^T fragments of key step 0
    //      C  O += V(h0)^T P(h0)^T                  16 MFMA   | softmax(h1), V^T fragments one key step ahead
    //      D  O += V(h1)^T P(h1)^T                  16 MFMA   | -
    // (one wave per SIMD: nothing else hides the VALU under the matrix pipe - the 8-wave structure relied on its second wave)
    auto tile_fast = [&](const char* img) {
        const int l31o = lane_o & 31, fko = lane_o >> 5;
        const int kswz = (l31o >> 1) & 7;
        const char* krow = img + l31o * 128;
        auto kr = [&](int h, int ks) -> bf16x8 {
            const int c = (2 * (ks & 3) + fko) ^ kswz;
            return *(const bf16x8*)(krow + h * 8192 + (ks >> 2) * 4096 + (c << 4));
        };
        const int pp = lane_o & 15, g16 = (lane_o >> 4) & 1;
        const char* vrow = img + 16384 + (4 * (lane_o >> 5) + (pp >> 2)) * 256 + ((pp & 1) << 3);
        const int tlo = (2 * g16 + ((pp & 3) >> 1)) << 4;
        auto vr = [&](int st, int dt) -> bf16x8 {                    // V^T fragment: 32-d block dt of key step st
            const char* a = vrow + st * 4096 + ((((dt ^ (pp >> 2)) & 3) << 6) | tlo);
            union { bf16x8 v; s16x4 h2[2]; } va;
            va.h2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(a));
            va.h2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(a + 2048));
            return va.v;
        };
        f32x16 s[2][2];                                              // [key half][query block]
        float nm[2][2];                                              // [key half][query block]: -running max used by that half's exponentials
        union PK { bf16x8 v; unsigned u[4]; };
        PK pk[2][2][2];                                              // [key half][key step within the half][query block]
        // --- softmax pieces of key half h, query block qb (each a handful of VALU instructions between two MFMAs)
        auto sm_max = [&](const int h, const int qb) {               // running max (+ the rare rescale of O and l)
            float t = max3f(s[h][qb][0], s[h][qb][1], s[h][qb][2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) t = max3f(t, s[h][qb][r], s[h][qb][r + 1]);
            t = fmaxf(t, s[h][qb][15]);
            t = half_swap_max(t * p.sl2);
            const float m_new = fmaxf(m_run[qb], t);
            if (__any(m_new > m_run[qb] + DEFER_THR)) {             // wave-uniform; the first tile always lands here
                const float alpha = m_new == -INFINITY ? 1.f : __builtin_amdgcn_exp2f(m_run[qb] - m_new);
                l_run[qb] *= alpha;
                acc_settle();
                if (qb == 0) {
                    acc_scale_seq<0>(alpha, std::make_integer_sequence<int, 16>{}); acc_scale_seq<32>(alpha, std::make_integer_sequence<int, 16>{});
                    acc_scale_seq<64>(alpha, std::make_integer_sequence<int, 16>{}); acc_scale_seq<96>(alpha, std::make_integer_sequence<int, 16>{});
                } else {
                    acc_scale_seq<16>(alpha, std::make_integer_sequence<int, 16>{}); acc_scale_seq<48>(alpha, std::make_integer_sequence<int, 16>{});
                    acc_scale_seq<80>(alpha, std::make_integer_sequence<int, 16>{}); acc_scale_seq<112>(alpha, std::make_integer_sequence<int, 16>{});
                }
                m_run[qb] = m_new;
            }
            nm[h][qb] = m_run[qb] == -INFINITY ? 0.f : -m_run[qb];
        };
        auto sm_exp = [&](const int h, const int qb, const int r0) {  // four elements: P = exp2(S sl2 - m)
#pragma unroll
            for (int r = r0; r < r0 + 4; ++r) s[h][qb][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[h][qb][r], p.sl2, nm[h][qb]));
        };
        auto sm_sum_pack = [&](const int h, const int qb, const int st) {   // row sum + bf16 pack of key step st (8 elements)
            float ps = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r0 = 8 * st + 2 * j;
                ps += s[h][qb][r0] + s[h][qb][r0 + 1];
                pk[h][st][qb].u[j] = pack2bf(s[h][qb][r0], s[h][qb][r0 + 1]);
            }
            l_run[qb] += ps;
        };
        // softmax of half h in 8 slices (slice i goes into MFMA gap pair i of the phase it hides under)
        auto sm_slice = [&](const int h, const int i) {
#ifdef F64_NO_SM
            if (i == 0) { asm volatile("s_nop 15" : "+v"(s[h][0]), "+v"(s[h][1]));
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                        for (int j = 0; j < 4; ++j) pk[h][st][qb].u[j] = __float_as_uint(s[h][qb][8 * st + 2 * j]);
            }
            return;
#endif
            if (i == 0) { asm volatile("s_nop 15" : "+v"(s[h][0]), "+v"(s[h][1])); sm_max(h, 0); }
            else if (i == 1) { sm_max(h, 1); }
            else if (i == 2) { sm_exp(h, 0, 0); sm_exp(h, 0, 4); sm_exp(h, 1, 0); }
            else if (i == 3) { sm_exp(h, 1, 4); sm_exp(h, 0, 8); sm_exp(h, 0, 12); }
            else if (i == 4) { sm_exp(h, 1, 8); sm_exp(h, 1, 12); }
            else if (i == 5) { sm_sum_pack(h, 0, 0); sm_sum_pack(h, 1, 0); }
            else if (i == 6) { sm_sum_pack(h, 0, 1); }
            else { sm_sum_pack(h, 1, 1); }
        };
        bf16x8 kf[4];
        // ---- A: S(h0)
#pragma unroll
        for (int j = 0; j < 4; ++j) kf[j] = kr(0, j);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (ks == 0) { mfma_s_first(s[0][0], kf[0], qf[0][0]); mfma_s_first(s[0][1], kf[0], qf[1][0]); }
            else { mfma_s(s[0][0], kf[ks & 3], qf[0][ks]); mfma_s(s[0][1], kf[ks & 3], qf[1][ks]); }
            kf[ks & 3] = ks < 4 ? kr(0, ks + 4) : kr(1, ks - 4);
        }
        // ---- B: S(h1) under it softmax(h0); V^T fragments of key step 0 requested at the end
        bf16x8 vf[4];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (ks == 0) { mfma_s_first(s[1][0], kf[0], qf[0][0]); mfma_s_first(s[1][1], kf[0], qf[1][0]); }
            else { mfma_s(s[1][0], kf[ks & 3], qf[0][ks]); mfma_s(s[1][1], kf[ks & 3], qf[1][ks]); }
            if (ks < 4) kf[ks] = kr(1, ks + 4);
            else vf[ks - 4] = vr(0, ks - 4);
            sm_slice(0, ks);
        }
#ifdef F64_NO_PV
        asm volatile("" :: "v"(s[1][0]), "v"(s[1][1]), "v"(pk[0][0][0].v), "v"(pk[0][1][1].v), "v"(vf[0]), "v"(vf[3]));
        return;
#endif
        // ---- C: O += V(h0)^T P(h0)^T (key steps 0, 1) under it softmax(h1); fragments one key step ahead
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            mfma_o<0>(vf[0], pk[0][st][0].v); mfma_o<1>(vf[0], pk[0][st][1].v); vf[0] = vr(st + 1, 0); sm_slice(1, 4 * st + 0);
            mfma_o<2>(vf[1], pk[0][st][0].v); mfma_o<3>(vf[1], pk[0][st][1].v); vf[1] = vr(st + 1, 1); sm_slice(1, 4 * st + 1);
            mfma_o<4>(vf[2], pk[0][st][0].v); mfma_o<5>(vf[2], pk[0][st][1].v); vf[2] = vr(st + 1, 2); sm_slice(1, 4 * st + 2);
            mfma_o<6>(vf[3], pk[0][st][0].v); mfma_o<7>(vf[3], pk[0][st][1].v); vf[3] = vr(st + 1, 3); sm_slice(1, 4 * st + 3);
        }
        // ---- D: O += V(h1)^T P(h1)^T (key steps 2, 3)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            mfma_o<0>(vf[0], pk[1][st][0].v); mfma_o<1>(vf[0], pk[1][st][1].v); if (st == 0) vf[0] = vr(3, 0);
            mfma_o<2>(vf[1], pk[1][st][0].v); mfma_o<3>(vf[1], pk[1][st][1].v); if (st == 0) vf[1] = vr(3, 1);
            mfma_o<4>(vf[2], pk[1][st][0].v); mfma_o<5>(vf[2], pk[1][st][1].v); if (st == 0) vf[2] = vr(3, 2);
            mfma_o<6>(vf[3], pk[1][st][0].v); mfma_o<7>(vf[3], pk[1][st][1].v); if (st == 0) vf[3] = vr(3, 3);
        }
    };

#ifdef F64_NO_LOOP
    for (int kt = 0; kt < 0; ++kt) {
#else
    for (int kt = 0; kt < nkt; ++kt) {
#endif
#ifndef F64_NO_BAR
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#endif
        const int cur = kt & 1;
#ifndef F64_NO_DMA
        if (kt + 1 < nkt) stage(cur ^ 1, kt + 1);
#endif
        if (!active) continue;
        const int kv0 = kt * BKV;
        if (kv0 > q0w + 63) continue;                               // tile entirely above this wave's diagonal
        asm volatile("" : "+v"(lane_o));
        const char* sks = smem + cur * STAGE_BYTES;
        const char* skc = sks + VAR_BYTES;
        bool kV, kL;
        key_mods(2 * kt, kV, kL);
        const bool wsame = (wL && kL) || (wV && kV);
        const bool wcross = (wL && kV) || (wV && kL);
        const bool mixed = wsame && wcross;
        const char* img1 = wsame ? sks : skc;
        const bool plain = !mixed && kv0 + BKV - 1 <= q0w && kv0 + BKV <= len && kv0 >= start && q0w + 64 <= S;
#ifdef F64_FORCE_FAST
        tile_fast(img1);
#else
        if (plain) tile_fast(img1);
        else tile(std::true_type{}, kt, img1, skc, mixed);
#endif
    }

    // ---- finish: per query block through the wave's private staging rows ----
    acc_settle();
    __syncthreads();
    constexpr int OROW = 264;                           // 128 bf16 + 8 B pad
    char* so = smem + wave * (32 * OROW);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {                    // (unrolled: a run-time index into o / m_run / l_run would move them to scratch)
        const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        const int q0b = q0w + 32 * qb;
#pragma unroll 1
        for (int part = 0; part < (p.out_lo ? 2 : 1); ++part) {
            if (q0b < S) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    float ob[16];
                    if (qb == 0) {
                        if (dt == 0) acc_read_block<0>(ob, std::make_integer_sequence<int, 16>{});
                        else if (dt == 1) acc_read_block<2>(ob, std::make_integer_sequence<int, 16>{});
                        else if (dt == 2) acc_read_block<4>(ob, std::make_integer_sequence<int, 16>{});
                        else acc_read_block<6>(ob, std::make_integer_sequence<int, 16>{});
                    } else {
                        if (dt == 0) acc_read_block<1>(ob, std::make_integer_sequence<int, 16>{});
                        else if (dt == 1) acc_read_block<3>(ob, std::make_integer_sequence<int, 16>{});
                        else if (dt == 2) acc_read_block<5>(ob, std::make_integer_sequence<int, 16>{});
                        else acc_read_block<7>(ob, std::make_integer_sequence<int, 16>{});
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int d = dt * 32 + 8 * g + 4 * fk;
                        float x[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            x[e] = ob[4 * g + e] * inv;
                            if (part) x[e] -= bf2f(f2bf(x[e]));
                        }
                        u32x2 w;
                        w[0] = pack2bf(x[0], x[1]);
                        w[1] = pack2bf(x[2], x[3]);
                        *(u32x2*)(so + l31 * OROW + d * 2) = w;
                    }
                }
                if (!part && p.lse && fk == 0 && q0b + l31 < S)
                    p.lse[((long)b * p.H + h) * S + q0b + l31] =
                        l_tot > 0.f ? (m_run[qb] + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f : -INFINITY;
                // same-wave LDS write -> read: one wave's LDS operations execute in order and no other wave touches `so`
                bf16_t* dst = part ? p.out_lo : p.out;
#pragma unroll
                for (int pass = 0; pass < 8; ++pass) {
                    const int r = pass * 4 + (lane >> 4);
                    const int qq = q0b + r;
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
#ifdef LIBRA_ATTN_FWD64
    a.n_qt = (int)((S + F_BQ - 1) / F_BQ);
    {
        const long nb = (long)B * H * a.n_qt;
        if (nb > 0x7fffffffL) return LIBRA_ERR_SHAPE;
        static std::atomic<bool> attr64{false};
        if (!attr64) {
            (void)hipFuncSetAttribute((const void*)bridge_attn_fwd64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BR_LDS);
            attr64 = true;
        }
        hipLaunchKernelGGL(bridge_attn_fwd64_kernel, dim3((unsigned)nb), dim3(256), BR_LDS, (hipStream_t)stream, a);
        return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
    }
#endif
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
