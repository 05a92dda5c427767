// Routed-"bridge" causal flash attention forward, second structure (round 3): 4-wave workgroups, TWO per CU.
//
// Same mathematics and the same per-wave tile body as attention_bridge.hip (S^T = K Q^T, P^T fed to O^T += V^T P^T straight from
// the accumulator registers, deferred running max, LDS transpose reads of V), restructured around what the round-2 cycle
// anatomy showed: with one 8-wave workgroup per CU the two waves of a SIMD are phase-locked by the per-tile
// barrier - both in the MFMA phases (sharing the matrix pipe) and both in the softmax (matrix pipe idle) - issuing the LDS-DMA
// pieces costs the issuing wave 650-830 cycles per tile, and every workgroup's prologue / epilogue runs with the CU otherwise idle.
//   * 128 queries per workgroup (4 waves x 32), 256 threads, <= 68 KiB LDS: two workgroups per CU whose barriers are
//     independent, so one workgroup's softmax / prologue / epilogue sits under the other's MFMAs;
//   * the iteration unit is (64-key tile, operand variant): one variant (K image + V image, 32 KiB) per LDS stage; a tile that
//     needs both variants (a modality boundary) is simply visited twice with complementary element masks - the online softmax
//     does not care - instead of the 64 KiB both-variant stage and the select / split code of the first structure;
//   * K / V tiles are staged through registers (cdna_hip_programming.md T14): 8 coalesced 16-byte global loads per thread are
//     issued right after QK^T, travel under softmax + P.V, and are written to the other stage just before the iteration's
//     single barrier - no LDS-DMA issue stalls, no m0 traffic, and hipcc counts the loads itself.
// Reference semantics: LibraAttention.forward + attn_with_bridge, modeling_libra.py:267-414 (closed form in attention_bridge.hip).
#include <atomic>
#include <cstdlib>
#include "hip_common.hpp"
#include "gemm_tiles.hpp"
#include "attention_bridge_args.hpp"
#include "../../include/libra_hip.h"

namespace libra {

constexpr int F2_BQ = 128;                         // query rows per workgroup (4 waves x 32)
constexpr int F2_BKV = 64;                         // keys per tile
constexpr int F2_STAGE = 2 * F2_BKV * BR_D * 2;    // K image (16 KiB) + V image (16 KiB) of ONE variant
constexpr int F2_OROW = 264;                       // epilogue staging row: 128 bf16 + 8 B pad
constexpr int F2_OBYTES = 4 * 32 * F2_OROW;        // one output image of the workgroup
constexpr int F2_MAIN = 2 * F2_OBYTES > 2 * F2_STAGE ? 2 * F2_OBYTES : 2 * F2_STAGE;
constexpr int F2_LDS = F2_MAIN + 1024;             // + key-modality masks (<= 129 words) and the query-presence words

// DBG != 0: timing-only builds for the cycle anatomy (results wrong): 1 = no global stores in the epilogue, 2 = no epilogue,
// 3 = no main loop, 4 = no softmax arithmetic, 5 = no mask code, 6 = no prefetch loads / LDS writes after the first unit
template <int DBG, bool DEEP>
__global__ __launch_bounds__(256, 2) void bridge_attn_fwd2_kernel(const BridgeArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* kmask = (unsigned*)(smem + F2_MAIN);                 // per 32 keys: bit j = key j is a vision token
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
    const int q0w = qt * F2_BQ + wave * 32;
    const bool active = q0w < S;
    int q = q0w + l31;
    q = q < S ? q : S - 1;

    modality_masks(p.flag + tok0, S, kmask, tid, 256);
    const bool q_vis = p.flag[tok0 + q] != 0;
    int* qpres = (int*)(kmask + 192);
    if (tid < 2) qpres[tid] = 0;
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

    // ---- Q fragments: lane (q = l31, half fk) holds Q[q][16*ks + 8*fk .. +8], ks = 0..7 ----
    bf16x8 qf[8];
    {
        const bf16_t* qp = p.q + (tok0 + q) * p.ldq + h * BR_D + fk * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
    }
    // K / V operands as buffer resources (wave-uniform descriptors, hardware bounds check: rows past the sequence end read as
    // zeros and are masked by `key < len`): no per-lane 64-bit address arithmetic and no clamping in the tile loop
    auto mk_rsrc = [&](const bf16_t* base, long ld) {
        const unsigned long pb = (unsigned long)(base + tok0 * ld + h * BR_D);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)pb), hi = __builtin_amdgcn_readfirstlane((unsigned)(pb >> 32));
        const unsigned bytes = __builtin_amdgcn_readfirstlane((unsigned)((long)(S - 1) * ld * 2 + BR_D * 2));
        return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long)hi << 32) | lo), 0, (int)bytes, 0x00020000);
    };
    const auto rs_ks = mk_rsrc(p.k_same, p.ldk), rs_kc = mk_rsrc(p.k_cross, p.ldkc);
    const auto rs_vs = mk_rsrc(p.v_same, p.ldv), rs_vc = mk_rsrc(p.v_cross, p.ldvc);

    f32x16 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    int kend = (qt + 1) * F2_BQ; kend = kend < S ? kend : S;       // causal: keys 0 .. kend - 1
    const int nkt = (kend + F2_BKV - 1) / F2_BKV;

    // modality content of the 64 keys of tile t (valid keys only); wave-uniform scalars
    auto key_mods = [&](int t, bool& kV, bool& kL) {
        unsigned long long m = (unsigned)__builtin_amdgcn_readfirstlane((int)kmask[2 * t]);
        m |= (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)kmask[2 * t + 1]) << 32;
        int nvalid = S - t * 64; nvalid = nvalid > 64 ? 64 : nvalid;
        const unsigned long long full = nvalid >= 64 ? ~0ull : ((1ull << nvalid) - 1ull);
        kV = (m & full) != 0; kL = ((~m) & full) != 0;
    };
    // which variants the WORKGROUP needs of tile t (bit 0 = same, bit 1 = cross)
    auto needs = [&](int t) -> int {
        bool kV, kL;
        key_mods(t, kV, kL);
        return (((blkL && kL) || (blkV && kV)) ? 1 : 0) | (((blkL && kV) || (blkV && kL)) ? 2 : 0);
    };
    // iteration units (tile, variant) in order; `t == nkt` = done
    auto advance = [&](int& t, int& var) {
        if (var == 0 && (needs(t) & 2)) { var = 1; return; }
        ++t;
        while (t < nkt && needs(t) == 0) ++t;                       // (cannot happen for t < nkt; keeps the walk total)
        var = (t < nkt && (needs(t) & 1)) ? 0 : 1;
    };

    // ---- staging: thread -> 4 chunks of K and 4 of V; chunk c = it*256 + tid: key row c>>4, 16-byte column c&15 ----
    // K image: four N-type [32 keys][64 d] sub-tiles (128-byte rows, chunk ^ ((row>>1)&7)), ordered (key half, d half).
    // V image: T-type [64 keys][128 d] (256-byte rows, chunk ^ ((row&3)<<2)) at +16384.
    u32x4 rk[4], rv[4];
    auto load_unit = [&](int t, int var) {
        const unsigned ldkb = (unsigned)(var ? p.ldkc : p.ldk) * 2u, ldvb = (unsigned)(var ? p.ldvc : p.ldv) * 2u;
        const unsigned vk = (unsigned)(tid >> 4) * ldkb + (unsigned)((tid & 15) * 16);
        const unsigned vv = (unsigned)(tid >> 4) * ldvb + (unsigned)((tid & 15) * 16);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const unsigned row0 = (unsigned)(t * F2_BKV + it * 16);
            rk[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(var ? rs_kc : rs_ks, vk, row0 * ldkb, 0));
            rv[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(var ? rs_vc : rs_vs, vv, row0 * ldvb, 0));
        }
    };
    // chunk it of a thread: key row it*16 + tid/16, column tid%16.  Row + 16 leaves both swizzles unchanged, so the four stores
    // of an image are one base address + compile-time offsets
    auto write_unit = [&](char* dst) {
        const int row = tid >> 4, col = tid & 15;
        char* kd = dst + (col >> 3) * 4096 + row * 128 + ((((col & 7) ^ ((row >> 1) & 7))) << 4);
        char* vd = dst + 16384 + row * 256 + ((col ^ ((row & 3) << 2)) << 4);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            *(u32x4*)(kd + (it & 1) * 2048 + (it >> 1) * 8192) = rk[it];
            *(u32x4*)(vd + it * 4096) = rv[it];
        }
    };

    int lane_o = lane;
    // S^T of both key halves.  All 16 K fragments are requested before the first MFMA (64 VGPRs: the LDS round trip is paid once
    // per tile, not once per MFMA pair - hipcc's own schedule keeps only 2-4 reads in flight and parks the wave at lgkmcnt)
    auto qk_pair = [&](const char* kimg, f32x16& s0, f32x16& s1) {
        const int l31o = lane_o & 31, fko = lane_o >> 5;
        const int kswz = (l31o >> 1) & 7;
        const char* krow = kimg + l31o * 128;
        bf16x8 kf[16];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int c = (2 * (ks & 3) + fko) ^ kswz;
            kf[2 * ks] = *(const bf16x8*)(krow + (ks >> 2) * 4096 + (c << 4));
            kf[2 * ks + 1] = *(const bf16x8*)(krow + 8192 + (ks >> 2) * 4096 + (c << 4));
        }
        if (DEEP) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2 * ks], qf[ks], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2 * ks + 1], qf[ks], s1, 0, 0, 0);
        }
    };
    // O^T += V^T P^T for two 16-key steps (8 MFMAs): all 8 V^T fragments requested first
    auto pv_step2 = [&](const char* vstep, const bf16x8 pk0, const bf16x8 pk1) {
        const int pp = lane_o & 15, g16 = (lane_o >> 4) & 1;
        const char* vrow = vstep + (4 * (lane_o >> 5) + (pp >> 2)) * 256 + ((pp & 1) << 3);
        const int tlo = (2 * g16 + ((pp & 3) >> 1)) << 4;
        union { bf16x8 v; s16x4 h2[2]; } va[8];
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const char* a = vrow + st * 4096 + ((((dt ^ (pp >> 2)) & 3) << 6) | tlo);
                va[st * 4 + dt].h2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(a));
                va[st * 4 + dt].h2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(a + 2048));
            }
        if (DEEP) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[dt].v, pk0, o[dt], 0, 0, 0);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[4 + dt].v, pk1, o[dt], 0, 0, 0);
    };
    auto rescale = [&](float alpha) {
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
    };

#pragma unroll
    for (int ks = 0; ks < 8; ++ks) pin(qf[ks]);                     // Q's wait belongs in the prologue, not at the loop's first MFMA
    // ---- first unit into stage 0 ----
    int t = 0, var = (needs(0) & 1) ? 0 : 1;
    load_unit(t, var);
    write_unit(smem);
    __syncthreads();

    if (DBG == 3) t = nkt;
    for (int u = 0; t < nkt; ++u) {
        int nt = t, nvar = var;
        advance(nt, nvar);
        const bool has_next = nt < nkt;
        const char* img = smem + (u & 1) * F2_STAGE;
        const int kv0 = t * F2_BKV;
        bool kV, kL;
        key_mods(t, kV, kL);
        const bool wsame = (wL && kL) || (wV && kV);
        const bool wcross = (wL && kV) || (wV && kL);
        // this wave takes part in the unit: has queries, the tile is not entirely above its diagonal, and it needs this variant
        const bool part = active && kv0 <= q0w + 31 && (var ? wcross : wsame);
        asm volatile("" : "+v"(lane_o));

        f32x16 sA, sB;
        if (part) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { sA[r] = 0.f; sB[r] = 0.f; }
            qk_pair(img, sA, sB);
        }
        if (has_next && DBG != 6) load_unit(nt, nvar);             // in flight under softmax + P.V
        if (part) {
            const bool mixedw = wsame && wcross;                    // both variants inside this wave x tile: select per element
            if (DBG != 5 && (mixedw || kv0 + F2_BKV - 1 > q0w || kv0 + F2_BKV > len || kv0 < start)) {
                const int qabs = q0w + l31;
                // left padding: keys before `start` are masked for real queries; a padding QUERY row keeps them (its output is
                // never used, but an all-masked row must stay at exactly 0, see the rescale guard)
                const int lo = qabs < start ? 0 : start;
                const unsigned km0 = (unsigned)__builtin_amdgcn_readfirstlane((int)kmask[2 * t]);
                const unsigned km1 = (unsigned)__builtin_amdgcn_readfirstlane((int)kmask[2 * t + 1]);
                const unsigned want = (q_vis ? 1u : 0u) ^ (unsigned)var;       // modality bit a key must have for this variant
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kl = (r & 3) + 8 * (r >> 2) + 4 * fk;
                    const int key = kv0 + kl;
                    const bool ma = !mixedw || ((km0 >> kl) & 1u) == want, mb = !mixedw || ((km1 >> kl) & 1u) == want;
                    sA[r] = (key <= qabs && key < len && key >= lo && ma) ? sA[r] : -INFINITY;
                    sB[r] = (key + 32 <= qabs && key + 32 < len && key + 32 >= lo && mb) ? sB[r] : -INFINITY;
                }
            }
            if (DBG != 4) {
            // ---- online softmax; the running max only advances when a tile exceeds it by 2^DEFER_THR ----
            float tmax = max3f(sA[0], sA[1], sB[0]);
#pragma unroll
            for (int r = 2; r < 16; r += 2) tmax = max3f(tmax, sA[r], sA[r + 1]);
#pragma unroll
            for (int r = 1; r < 15; r += 2) tmax = max3f(tmax, sB[r], sB[r + 1]);
            tmax = fmaxf(tmax, sB[15]);
            tmax = half_swap_max(tmax * p.sl2);
            const float m_new = fmaxf(m_run, tmax);
            if (__any(m_new > m_run + DEFER_THR)) {                 // wave-uniform; the first visible tile always lands here
                // a row that has seen no key yet has m_run = m_new = -inf: exp2(-inf - -inf) would be NaN
                rescale(m_new == -INFINITY ? 1.f : __builtin_amdgcn_exp2f(m_run - m_new));
                m_run = m_new;
            }
            const float nm = m_run == -INFINITY ? 0.f : -m_run;     // (a row with no visible key yet stays at exactly 0)
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sA[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sA[r], p.sl2, nm));
                sB[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sB[r], p.sl2, nm));
                psum += sA[r] + sB[r];
            }
            l_run += psum;
            }
            union { bf16x8 v; unsigned u[4]; } pk[4];
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r0 = 8 * (st & 1) + 2 * j;
                    pk[st].u[j] = st < 2 ? pack2bf(sA[r0], sA[r0 + 1]) : pack2bf(sB[r0], sB[r0 + 1]);
                }
            pv_step2(img + 16384, pk[0].v, pk[1].v);
            pv_step2(img + 16384 + 8192, pk[2].v, pk[3].v);
        }
        if (has_next && DBG != 6) write_unit(smem + ((u + 1) & 1) * F2_STAGE);
        __syncthreads();
        t = nt; var = nvar;
    }

    if (DBG == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(o[i]));
        asm volatile("" :: "v"(l_run), "v"(m_run));
        return;
    }
    // ---- finish: both output images (bf16 O and its rounding residual) through LDS, whole 256-byte rows to memory ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    char* so = smem + wave * (32 * F2_OROW);
    const bool lo_out = p.out_lo != nullptr;
    if (active) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = dt * 32 + 8 * g + 4 * fk;
                float x[4], y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    x[e] = o[dt][4 * g + e] * inv;
                    y[e] = x[e] - bf2f(f2bf(x[e]));
                }
                u32x2 w;
                w[0] = pack2bf(x[0], x[1]);
                w[1] = pack2bf(x[2], x[3]);
                *(u32x2*)(so + l31 * F2_OROW + d * 2) = w;
                if (lo_out) {
                    w[0] = pack2bf(y[0], y[1]);
                    w[1] = pack2bf(y[2], y[3]);
                    *(u32x2*)(so + F2_OBYTES + l31 * F2_OROW + d * 2) = w;
                }
            }
        if (DBG != 1 && p.lse && fk == 0 && q0w + l31 < S)
            p.lse[((long)b * p.H + h) * S + q0w + l31] =
                l_tot > 0.f ? (m_run + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f : -INFINITY;
    }
    __syncthreads();
    if (active) {
#pragma unroll 1
        for (int part = 0; part < (lo_out ? 2 : 1); ++part) {
            bf16_t* dst = part ? p.out_lo : p.out;
            const char* sp = so + part * F2_OBYTES;
#pragma unroll
            for (int pass = 0; pass < 8; ++pass) {                  // 32 rows x 256 B: lane -> (row = pass*4 + lane/16, chunk lane%16)
                const int r = pass * 4 + (lane >> 4);
                const int qq = q0w + r;
                if (qq < S && (DBG != 1 || p.B < 0)) {
                    const char* src = sp + r * F2_OROW + (lane & 15) * 16;
                    const u32x2 a = *(const u32x2*)src;
                    const u32x2 c2 = *(const u32x2*)(src + 8);
                    u32x4 v;
                    v[0] = a[0]; v[1] = a[1]; v[2] = c2[0]; v[3] = c2[1];
                    *(u32x4*)(dst + (tok0 + qq) * p.ldo + h * BR_D + (lane & 15) * 8) = v;
                }
            }
        }
    }
}

int bridge_attn_fwd2_launch(BridgeArgs a, hipStream_t stream) {
    a.n_qt = (a.S + F2_BQ - 1) / F2_BQ;
    const long nblk = (long)a.B * a.H * a.n_qt;
    if (nblk > 0x7fffffffL) return LIBRA_ERR_SHAPE;
    static const int dbg = [] { const char* e = getenv("LIBRA_ATTN_DBG"); return e ? atoi(e) : 0; }();
    static std::atomic<bool> attr_set{false};
    void (*kern)(const BridgeArgs) = bridge_attn_fwd2_kernel<0, true>;
    switch (dbg) {
        case 1: kern = bridge_attn_fwd2_kernel<1, true>; break;
        case 2: kern = bridge_attn_fwd2_kernel<2, true>; break;
        case 3: kern = bridge_attn_fwd2_kernel<3, true>; break;
        case 4: kern = bridge_attn_fwd2_kernel<4, true>; break;
        case 5: kern = bridge_attn_fwd2_kernel<5, true>; break;
        case 6: kern = bridge_attn_fwd2_kernel<6, true>; break;
        case 10: kern = bridge_attn_fwd2_kernel<0, false>; break;      // hipcc's own fragment-read schedule
        default: break;
    }
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, F2_LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), F2_LDS, stream, a);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}

}  // namespace libra
