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
