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
#include "hip_common.hpp"
#include "gemm_tiles.hpp"
#include "../../include/libra_hip.h"

namespace libra {

constexpr int BD = 128;            // head dim
constexpr int BQ = 128;            // query rows per workgroup (4 waves x 32)
constexpr int BKV = 32;            // keys per tile
constexpr int VAR_BYTES = 2 * BKV * BD * 2;     // one variant: K tile (8 KiB) + V tile (8 KiB)
constexpr int STAGE_BYTES = 2 * VAR_BYTES;      // same + cross
constexpr int BR_LDS = 2 * STAGE_BYTES + 1024;  // double buffered + key-modality masks

struct BridgeArgs {
    const bf16_t* q; long ldq;
    const bf16_t* k_same; const bf16_t* k_cross; long ldk, ldkc;
    const bf16_t* v_same; const bf16_t* v_cross; long ldv, ldvc;
    const unsigned char* flag;     // [B*S] 1 = vision token
    const int* kv_len;             // [B] valid (non-padded) length, right padding
    bf16_t* out; long ldo;
    float* lse;                    // [B,H,S] or null
    int B, S, H, n_qt;
    float sl2;
};

// K tile image: two N-type [32 keys][64 d] sub-tiles (128-byte rows, chunk ^ ((row>>1)&7)), 4 KiB each.
// V tile image: T-type [32 keys][128 d] (256-byte rows, chunk ^ ((row&3)<<2)), 8 KiB.
__device__ __forceinline__ void stage_kv(const bf16_t* __restrict__ kp, long ldk, const bf16_t* __restrict__ vp, long ldv,
                                         int key0, int S, char* dst, int wave, int lane) {
    // K: 8 pieces of 1 KiB (8 rows x 128 B); piece pc -> sub-tile pc>>2, rows 8*(pc&3)..; wave w takes pieces 2w, 2w+1
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pc = wave * 2 + j;
        const int sub = pc >> 2, r = (pc & 3) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int key = key0 + r; key = key < S ? key : S - 1;
        glds16(kp + (long)key * ldk + sub * 64 + c * 8, dst + pc * 1024);
    }
    // V: 8 pieces of 1 KiB (4 rows x 256 B); wave w takes pieces 2w, 2w+1
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pc = wave * 2 + j;
        const int r = pc * 4 + (lane >> 4);
        const int c = (lane & 15) ^ ((r & 3) << 2);
        int key = key0 + r; key = key < S ? key : S - 1;
        glds16(vp + (long)key * ldv + c * 8, dst + 8192 + pc * 1024);
    }
}

__global__ __launch_bounds__(256, 2) void bridge_attn_fwd_kernel(const BridgeArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* kmask = (unsigned*)(smem + 2 * STAGE_BYTES);        // per 32-key tile: bit j = key j is a vision token
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
    const int q0w = qt * BQ + wave * 32;
    const bool active = q0w < S;
    int q = q0w + l31;
    q = q < S ? q : S - 1;

    // ---- key-modality masks of this sequence into LDS (ballot over 32 flags per tile) ----
    const int ntile_all = (S + BKV - 1) / BKV;
    for (int t = wave; t < ntile_all; t += 4) {
        const int key = t * BKV + l31;
        const bool vis = (key < S) && (fk == 0) && p.flag[tok0 + key] != 0;
        const unsigned long long bal = __ballot(vis);
        if (lane == 0) kmask[t] = (unsigned)bal;
    }
    const bool q_vis = p.flag[tok0 + q] != 0;
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
    const bool blkL = qpres[0] != 0, blkV = qpres[1] != 0;
    const unsigned long long wbal_v = __ballot(q_vis && (q0w + l31) < S);
    const unsigned long long wbal_l = __ballot(!q_vis && (q0w + l31) < S);
    const bool wV = wbal_v != 0, wL = wbal_l != 0;                  // this wave's query modalities

    // ---- Q fragments: lane (q = l31, half fk) holds Q[q][16*ks + 8*fk .. +8], ks = 0..7 ----
    bf16x8 qf[8];
    {
        const bf16_t* qp = p.q + (tok0 + q) * p.ldq + h * BD + fk * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
    }
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

    auto needs = [&](int t, bool& same, bool& cross) {
        const unsigned km = kmask[t];
        int nvalid = S - t * BKV; nvalid = nvalid > BKV ? BKV : nvalid;
        const unsigned full = nvalid >= 32 ? 0xffffffffu : ((1u << nvalid) - 1u);
        const bool kV = (km & full) != 0, kL = ((~km) & full) != 0;
        same = (blkL && kL) || (blkV && kV);
        cross = (blkL && kV) || (blkV && kL);
    };
    auto stage = [&](int buf, int t) {
        bool same, cross;
        needs(t, same, cross);
        char* dst = smem + buf * STAGE_BYTES;
        if (same) stage_kv(ks_base, p.ldk, vs_base, p.ldv, t * BKV, S, dst, wave, lane);
        if (cross) stage_kv(kc_base, p.ldkc, vc_base, p.ldvc, t * BKV, S, dst + VAR_BYTES, wave, lane);
    };
    stage(0, 0);

    // fragment addressing
    const int pp = lane & 15, g16 = (lane >> 4) & 1;
    const int vrow = (4 * fk + (pp >> 2)) * 256;                    // T-type V: keys 4*fk + (p>>2) (+8 for the 2nd read)

    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int cur = kt & 1;
        if (kt + 1 < nkt) stage(cur ^ 1, kt + 1);
        if (!active) continue;
        const int kv0 = kt * BKV;
        if (kv0 > q0w + 31) continue;                               // tile entirely above this wave's diagonal
        bool bsame, bcross;
        needs(kt, bsame, bcross);
        const unsigned km = kmask[kt];
        // wave-level needs (subset of the block-level ones)
        int nvalid = S - kv0; nvalid = nvalid > BKV ? BKV : nvalid;
        const unsigned full = nvalid >= 32 ? 0xffffffffu : ((1u << nvalid) - 1u);
        const bool kV = (km & full) != 0, kL = ((~km) & full) != 0;
        const bool wsame = (wL && kL) || (wV && kV);
        const bool wcross = (wL && kV) || (wV && kL);
        const char* sks = smem + cur * STAGE_BYTES;                 // K same (2 x 4 KiB), V same at +8192
        const char* skc = sks + VAR_BYTES;

        // ---- S^T = K Q^T (32 keys x 32 queries), per needed variant ----
        f32x16 s_s, s_c;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s_s[r] = 0.f; s_c[r] = 0.f; }
        if (wsame) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int sub = ks >> 2, c = (2 * (ks & 3) + fk) ^ ((l31 >> 1) & 7);
                const bf16x8 kf = *(const bf16x8*)(sks + sub * 4096 + l31 * 128 + (c << 4));
                s_s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s_s, 0, 0, 0);
            }
        }
        if (wcross) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int sub = ks >> 2, c = (2 * (ks & 3) + fk) ^ ((l31 >> 1) & 7);
                const bf16x8 kf = *(const bf16x8*)(skc + sub * 4096 + l31 * 128 + (c << 4));
                s_c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s_c, 0, 0, 0);
            }
        }
        // ---- select per element, scale, mask, online softmax ----
        const int qabs = q0w + l31;
        float tmax = -INFINITY;
        unsigned crossbits = 0;                                     // bit r: element r uses the cross variant
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kl = (r & 3) + 8 * (r >> 2) + 4 * fk;         // local key of accumulator row r
            const int key = kv0 + kl;
            const bool kvis = (km >> kl) & 1u;
            const bool cr = kvis != q_vis;
            float v = (cr ? s_c[r] : s_s[r]) * p.sl2;
            v = (key <= qabs && key < len) ? v : -INFINITY;
            s_s[r] = v;
            crossbits |= (cr ? 1u : 0u) << r;
            tmax = fmaxf(tmax, v);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float m_use = m_new == -INFINITY ? 0.f : m_new;       // fully masked so far: keep everything at 0
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(s_s[r] - m_use);
            s_s[r] = e;
            psum += e;
        }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] *= alpha;

        // ---- O^T += V^T P^T per variant; k-step sx consumes accumulator regs 8sx..8sx+7 = local keys
        //      16sx + 4fk + {0..3, 8..11}
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
            union { bf16x8 v; unsigned u[4]; } ps, pc;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r0 = 8 * sx + 2 * j, r1 = r0 + 1;
                const float a0 = s_s[r0], a1 = s_s[r1];
                const bool c0 = (crossbits >> r0) & 1u, c1 = (crossbits >> r1) & 1u;
                ps.u[j] = pack2bf(c0 ? 0.f : a0, c1 ? 0.f : a1);
                pc.u[j] = pack2bf(c0 ? a0 : 0.f, c1 ? a1 : 0.f);
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                // 32-line block dt of the 128-line (d) T-type tile; key rows 16sx + 4fk + (p>>2), second read +8
                const int toff = (((((dt ^ (pp >> 2)) & 3) << 2) | (2 * g16 + ((pp & 3) >> 1))) << 4) + ((pp & 1) << 3);
                if (wsame) {
                    const char* a = sks + 8192 + sx * 4096 + vrow + toff;
                    union { bf16x8 v; s16x4 h2[2]; } va;
                    va.h2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(a));
                    va.h2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(a + 2048));
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va.v, ps.v, o[dt], 0, 0, 0);
                }
                if (wcross) {
                    const char* a = skc + 8192 + sx * 4096 + vrow + toff;
                    union { bf16x8 v; s16x4 h2[2]; } va;
                    va.h2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(a));
                    va.h2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(a + 2048));
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va.v, pc.v, o[dt], 0, 0, 0);
                }
            }
        }
    }

    // ---- finish ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    __syncthreads();
    constexpr int OROW = 264;                           // 128 bf16 + 8 B pad
    char* so = smem + wave * (32 * OROW);
    if (active) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = dt * 32 + 8 * g + 4 * fk;
                u32x2 w;
                w[0] = pack2bf(o[dt][4 * g + 0] * inv, o[dt][4 * g + 1] * inv);
                w[1] = pack2bf(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
                *(u32x2*)(so + l31 * OROW + d * 2) = w;
            }
        if (p.lse && fk == 0 && q0w + l31 < S)
            p.lse[((long)b * p.H + h) * S + q0w + l31] =
                l_tot > 0.f ? (m_run + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f : -INFINITY;
    }
    __syncthreads();
    if (active) {
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
                *(u32x4*)(p.out + (tok0 + qq) * p.ldo + h * BD + (lane & 15) * 8) = v;
            }
        }
    }
}

}  // namespace libra

using namespace libra;

extern "C" int libra_bridge_attn_fwd(const void* q, int64_t ldq, const void* k_same, int64_t ldk, const void* k_cross,
                                     int64_t ldkc, const void* v_same, int64_t ldv, const void* v_cross, int64_t ldvc,
                                     const uint8_t* flag,
                                     const int32_t* kv_len, void* out, int64_t ldo, float* lse, int64_t B, int64_t S,
                                     int64_t H, float scale, void* stream) {
    if (B <= 0 || S <= 0) return LIBRA_OK;
    if (H <= 0 || ldq < H * BD || ldk < H * BD || ldv < H * BD || ldkc < H * BD || ldvc < H * BD || ldo < H * BD || S > 4096)
        return LIBRA_ERR_SHAPE;
    if ((ldq % 8) || (ldk % 8) || (ldv % 8) || (ldkc % 8) || (ldvc % 8) || (ldo % 8)) return LIBRA_ERR_ALIGN;
    if (!q || !k_same || !k_cross || !v_same || !v_cross || !flag || !out) return LIBRA_ERR_ALIGN;
    if (((uintptr_t)q | (uintptr_t)k_same | (uintptr_t)k_cross | (uintptr_t)v_same | (uintptr_t)v_cross | (uintptr_t)out) & 15)
        return LIBRA_ERR_ALIGN;
    BridgeArgs a;
    a.q = (const bf16_t*)q; a.ldq = ldq; a.k_same = (const bf16_t*)k_same; a.k_cross = (const bf16_t*)k_cross; a.ldk = ldk; a.ldkc = ldkc;
    a.v_same = (const bf16_t*)v_same; a.v_cross = (const bf16_t*)v_cross; a.ldv = ldv; a.ldvc = ldvc;
    a.flag = flag; a.kv_len = kv_len; a.out = (bf16_t*)out; a.ldo = ldo; a.lse = lse;
    a.B = (int)B; a.S = (int)S; a.H = (int)H; a.n_qt = (int)((S + BQ - 1) / BQ);
    a.sl2 = scale * 1.4426950408889634f;
    const long nblk = (long)B * H * a.n_qt;
    if (nblk > 0x7fffffffL) return LIBRA_ERR_SHAPE;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)bridge_attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BR_LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL(bridge_attn_fwd_kernel, dim3((unsigned)nblk), dim3(256), BR_LDS, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}
