// ViT flash attention backward for gfx950 (head_dim 64, non-causal).
//
//   P_ij  = exp(scale q_i.k_j - L_i)            (L = log-sum-exp saved by the forward)
//   dV_j  = sum_i P_ij dO_i        dP_ij = dO_i . v_j        D_i = dO_i . O_i
//   dS_ij = P_ij (dP_ij - D_i)     dQ_i = scale sum_j dS_ij k_j     dK_j = scale sum_i dS_ij q_i
//
// Two deterministic passes (no atomics), both built on the forward's "transposed" trick — the
// reduction-side matrix of every second-stage product is taken straight from the first-stage MFMA
// accumulator registers, so P / dS never leave their lane:
//   * dq kernel : workgroup = 128 query rows (lane <-> query), loops over 64-key tiles.
//                 S^T = K Q^T, dP^T = V dO^T (A from LDS, B = Q / dO fragments in registers),
//                 dQ^T += K^T dS^T          (A = K^T fragments: LDS transpose reads of the same K tile)
//   * dkv kernel: workgroup = 128 keys (lane <-> key), loops over 64-query tiles.
//                 S = Q K^T, dP = dO V^T    (A from LDS, B = K / V fragments in registers),
//                 dV^T += dO^T P, dK^T += Q^T dS   (A = dO^T / Q^T fragments: LDS transpose reads of the dO / Q tiles)
// Every operand tile is staged once, as it lies in HBM, in the dual-use image of attn_tiles64.hpp; D = rowsum(dO*O)
// comes from libra_vit_attn_delta.
#include <atomic>
#include "hip_common.hpp"
#include "attn_tiles64.hpp"
#include "../../include/libra_hip.h"

namespace libra {

constexpr int HD = 64, KB = 64, TILE = KB * HD * 2;    // 8 KiB tiles of 64 rows x 128 B

struct AttnBwdArgs {
    const bf16_t* qkv; long ld_qkv;
    const bf16_t* dout; long ld_out;
    const float* lse; const float* delta;
    bf16_t* dqkv; long ld_dqkv;
    int B, T, H, n_t;
    float sl2, scale;
};

__device__ __forceinline__ bf16x8 pack_regs(const f32x16& a, int sx) {
    union { bf16x8 v; unsigned u[4]; } pb;
#pragma unroll
    for (int j = 0; j < 4; ++j) pb.u[j] = pack2bf(a[8 * sx + 2 * j], a[8 * sx + 2 * j + 1]);
    return pb.v;
}

// write a wave's transposed accumulator pair X^T[64 d][32 tokens] (times mul) to rows tok0.. of a
// [*, ld] bf16 matrix at column col0, via a 32 x 136-byte LDS staging area.
__device__ __forceinline__ void store_wave_tile(const f32x16* acc, float mul, char* so, bf16_t* __restrict__ dst,
                                                long ld, long row_base, int rows_valid, int col0, int lane) {
    constexpr int OROW = 136;
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = dt * 32 + 8 * g + 4 * half;
            u32x2 w;
            w[0] = pack2bf(acc[dt][4 * g + 0] * mul, acc[dt][4 * g + 1] * mul);
            w[1] = pack2bf(acc[dt][4 * g + 2] * mul, acc[dt][4 * g + 3] * mul);
            *(u32x2*)(so + l31 * OROW + d * 2) = w;
        }
    // same-wave LDS write -> read: LDS ops of one wave execute in order and no other wave touches `so`
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int r = pass * 8 + (lane >> 3);
        if (r < rows_valid) {
            const char* src = so + r * OROW + (lane & 7) * 16;
            const u32x2 a = *(const u32x2*)src;
            const u32x2 c2 = *(const u32x2*)(src + 8);
            u32x4 v;
            v[0] = a[0]; v[1] = a[1]; v[2] = c2[0]; v[3] = c2[1];
            *(u32x4*)(dst + (row_base + r) * ld + col0 + (lane & 7) * 8) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dQ pass
constexpr int DQ_STAGE = 2 * TILE;               // K, V
constexpr int DQ_LDS = 2 * DQ_STAGE;             // 32 KiB

__global__ __launch_bounds__(256, 2) void vit_attn_bwd_dq_kernel(const AttnBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int nblk = p.B * p.H * p.n_t;
    const int L = xcd_remap(blockIdx.x, nblk);
    const int qt = L % p.n_t, bh = L / p.n_t;
    const int h = bh % p.H, b = bh / p.H;
    const int T = p.T;
    const long tok0 = (long)b * T;
    const int q0 = qt * 128 + wave * 32;
    const bool active = q0 < T;

    int q = q0 + l31;
    const bool qvalid = q < T;
    q = qvalid ? q : T - 1;
    bf16x8 qf[4], dof[4];
    {
        const bf16_t* qp = p.qkv + (tok0 + q) * p.ld_qkv + h * HD + half * 8;
        const bf16_t* dp = p.dout + (tok0 + q) * p.ld_out + h * HD + half * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { qf[ks] = *(const bf16x8*)(qp + ks * 16); dof[ks] = *(const bf16x8*)(dp + ks * 16); }
    }
    const long sidx = ((long)b * p.H + h) * T + q;
    float Lq2 = p.lse[sidx] * 1.4426950408889634f;
    float Dq = p.delta[sidx];

    const bf16_t* kbase = p.qkv + tok0 * p.ld_qkv + (long)p.H * HD + h * HD;
    const bf16_t* vbase = kbase + (long)p.H * HD;

    f32x16 dq[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;

    const int nkt = (T + KB - 1) / KB;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { pin(qf[ks]); pin(dof[ks]); }   // prologue loads have landed before any LDS-DMA is in flight
    pin(Lq2); pin(Dq);
    stage_tile64(kbase, p.ld_qkv, 0, T, smem, wave, lane);
    stage_tile64(vbase, p.ld_qkv, 0, T, smem + TILE, wave, lane);

    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int cur = kt & 1;
        if (kt + 1 < nkt) {
            char* nb = smem + (cur ^ 1) * DQ_STAGE;
            stage_tile64(kbase, p.ld_qkv, (kt + 1) * KB, T, nb, wave, lane);
            stage_tile64(vbase, p.ld_qkv, (kt + 1) * KB, T, nb + TILE, wave, lane);
        }
        if (!active) continue;
        const char* sk = smem + cur * DQ_STAGE;
        const char* sv = sk + TILE;
        const int kv0 = kt * KB;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(sk + off64(c * 32 + l31, ks * 2 + half));
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 vf = *(const bf16x8*)(sv + off64(c * 32 + l31, ks * 2 + half));
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[ks], dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kv0 + c * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float pr = key < T ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], p.sl2, -Lq2)) : 0.f;
                s[r] = pr * (dp[r] - Dq);                      // dS^T (clamped tail keys: exactly 0)
            }
#pragma unroll
            for (int sx = 0; sx < 2; ++sx) {
                const bf16x8 dsb = pack_regs(s, sx);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tread64(sk, lane, dt, c * 2 + sx), dsb, dq[dt], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    if (active) {
        int rows_valid = T - q0; rows_valid = rows_valid > 32 ? 32 : rows_valid;
        store_wave_tile(dq, p.scale, smem + wave * (32 * 136), p.dqkv, p.ld_dqkv, tok0 + q0, rows_valid, h * HD, lane);
    }
}

// ------------------------------------------------------------------------------------------------
// dK / dV pass
constexpr int DKV_STAGE = 2 * TILE + 512;        // Q, dO, L[64], D[64]
constexpr int DKV_LDS = 2 * DKV_STAGE;           // 33 KiB

__global__ __launch_bounds__(256, 2) void vit_attn_bwd_dkv_kernel(const AttnBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int nblk = p.B * p.H * p.n_t;
    const int L = xcd_remap(blockIdx.x, nblk);
    const int ktile = L % p.n_t, bh = L / p.n_t;
    const int h = bh % p.H, b = bh / p.H;
    const int T = p.T;
    const long tok0 = (long)b * T;
    const int k0 = ktile * 128 + wave * 32;
    const bool active = k0 < T;

    int key = k0 + l31;
    key = key < T ? key : T - 1;
    bf16x8 kf[4], vf[4];
    {
        const bf16_t* kp = p.qkv + (tok0 + key) * p.ld_qkv + (long)p.H * HD + h * HD + half * 8;
        const bf16_t* vp = kp + (long)p.H * HD;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { kf[ks] = *(const bf16x8*)(kp + ks * 16); vf[ks] = *(const bf16x8*)(vp + ks * 16); }
    }
    const bf16_t* qbase = p.qkv + tok0 * p.ld_qkv + h * HD;
    const bf16_t* dobase = p.dout + tok0 * p.ld_out + h * HD;
    const float* lbase = p.lse + ((long)b * p.H + h) * T;
    const float* dbase = p.delta + ((long)b * p.H + h) * T;

    f32x16 dk[2], dv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[i][r] = 0.f; dv[i][r] = 0.f; }

    const int nqt = (T + KB - 1) / KB;
    auto stage_all = [&](char* st, int qtile) {
        const int r0 = qtile * KB;
        stage_tile64(qbase, p.ld_qkv, r0, T, st, wave, lane);
        stage_tile64(dobase, p.ld_out, r0, T, st + TILE, wave, lane);
        if (wave < 2) {                                         // 64 fp32 each: one 4-byte direct-to-LDS op
            int qi = r0 + lane; qi = qi < T ? qi : T - 1;
            const float* src = (wave == 0 ? lbase : dbase) + qi;
            glds4(src, st + 2 * TILE + wave * 256);
        }
    };
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { pin(kf[ks]); pin(vf[ks]); }   // K/V fragments have landed before any LDS-DMA is in flight
    stage_all(smem, 0);

    for (int it = 0; it < nqt; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int cur = it & 1;
        if (it + 1 < nqt) stage_all(smem + (cur ^ 1) * DKV_STAGE, it + 1);
        if (!active) continue;
        const char* sq = smem + cur * DKV_STAGE;
        const char* sdo = sq + TILE;
        const float* sL = (const float*)(sq + 2 * TILE);
        const float* sD = sL + 64;
        const int q0 = it * KB;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 a = *(const bf16x8*)(sq + off64(c * 32 + l31, ks * 2 + half));
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, kf[ks], s, 0, 0, 0);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 a = *(const bf16x8*)(sdo + off64(c * 32 + l31, ks * 2 + half));
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, vf[ks], dp, 0, 0, 0);
            }
            // accumulator row (r) <-> query  q0 + c*32 + (r&3) + 8*(r>>2) + 4*half ; column <-> this lane's key
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ql = c * 32 + 8 * g + 4 * half;
                const f32x4 Lv = *(const f32x4*)(sL + ql);
                const f32x4 Dv = *(const f32x4*)(sD + ql);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    const float pr = (q0 + ql + e) < T
                                         ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], p.sl2, -Lv[e] * 1.4426950408889634f)) : 0.f;
                    s[r] = pr;                                   // P
                    dp[r] = pr * (dp[r] - Dv[e]);                // dS
                }
            }
#pragma unroll
            for (int sx = 0; sx < 2; ++sx) {
                const bf16x8 pb = pack_regs(s, sx);
                const bf16x8 dsb = pack_regs(dp, sx);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tread64(sdo, lane, dt, c * 2 + sx), pb, dv[dt], 0, 0, 0);
                    dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tread64(sq, lane, dt, c * 2 + sx), dsb, dk[dt], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();
    if (active) {
        int rows_valid = T - k0; rows_valid = rows_valid > 32 ? 32 : rows_valid;
        char* so = smem + wave * (32 * 136);
        store_wave_tile(dk, p.scale, so, p.dqkv, p.ld_dqkv, tok0 + k0, rows_valid, p.H * HD + h * HD, lane);
        store_wave_tile(dv, 1.0f, so, p.dqkv, p.ld_dqkv, tok0 + k0, rows_valid, 2 * p.H * HD + h * HD, lane);
    }
}

// delta[b,h,q] = sum_d dO[b*T+q, h*64+d] * O[...]   (8 lanes per (token, head))
__global__ __launch_bounds__(256) void vit_attn_delta_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ o_lo,
                                                             const bf16_t* __restrict__ dout, long ld, float* __restrict__ delta,
                                                             int T, int H, long total_chunks) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = i < total_chunks;
    const int cpr = H * 8;                                   // 8-element chunks per token row
    const long row = ok ? i / cpr : 0;
    const int ch = ok ? (int)(i - row * cpr) : 0;
    float s = 0.f;
    if (ok) {
        float a[8], g[8];
        unpack8(*(const u32x4*)(o + row * ld + ch * 8), a);
        unpack8(*(const u32x4*)(dout + row * ld + ch * 8), g);
        if (o_lo) {                                           // O to ~16 mantissa bits: see libra_vit_attn_fwd
            float l[8];
            unpack8(*(const u32x4*)(o_lo + row * ld + ch * 8), l);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += l[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) s += a[e] * g[e];
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (ok && (ch & 7) == 0) {
        const int h = ch >> 3;
        const long b = row / T;
        const int t = (int)(row - b * T);
        delta[(b * H + h) * T + t] = s;
    }
}

}  // namespace libra

using namespace libra;

extern "C" int libra_vit_attn_delta(const void* out, const void* out_lo, const void* dout, int64_t ld, float* delta, int64_t B,
                                    int64_t T, int64_t H, void* stream) {
    const long rows = B * T;
    if (rows <= 0) return LIBRA_OK;
    if (H <= 0 || ld < H * HD || (ld % 8)) return LIBRA_ERR_SHAPE;
    if (!out || !dout || !delta || (((uintptr_t)out | (uintptr_t)dout | (uintptr_t)out_lo) & 15)) return LIBRA_ERR_ALIGN;
    const long total = rows * H * 8;
    hipLaunchKernelGGL(vit_attn_delta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)out, (const bf16_t*)out_lo, (const bf16_t*)dout, (long)ld, delta, (int)T, (int)H, total);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}

extern "C" int libra_vit_attn_bwd(const void* qkv, int64_t ld_qkv, const void* dout, int64_t ld_out, const float* lse,
                                  const float* delta, void* dqkv, int64_t ld_dqkv, int64_t B, int64_t T, int64_t H,
                                  float scale, void* stream) {
    if (B <= 0 || T <= 0) return LIBRA_OK;
    if (H <= 0 || ld_qkv < 3 * H * HD || ld_out < H * HD || ld_dqkv < 3 * H * HD) return LIBRA_ERR_SHAPE;
    if ((ld_qkv % 8) || (ld_out % 8) || (ld_dqkv % 8)) return LIBRA_ERR_ALIGN;
    if (!qkv || !dout || !lse || !delta || !dqkv) return LIBRA_ERR_ALIGN;
    if (((uintptr_t)qkv | (uintptr_t)dout | (uintptr_t)dqkv) & 15) return LIBRA_ERR_ALIGN;
    AttnBwdArgs a;
    a.qkv = (const bf16_t*)qkv; a.ld_qkv = ld_qkv;
    a.dout = (const bf16_t*)dout; a.ld_out = ld_out; a.lse = lse; a.delta = delta;
    a.dqkv = (bf16_t*)dqkv; a.ld_dqkv = ld_dqkv;
    a.B = (int)B; a.T = (int)T; a.H = (int)H; a.n_t = (int)((T + 127) / 128);
    a.scale = scale; a.sl2 = scale * 1.4426950408889634f;
    const long nblk = (long)B * H * a.n_t;
    if (nblk > 0x7fffffffL) return LIBRA_ERR_SHAPE;
    static std::atomic<bool> attr_set{false};     // (idempotent call; atomic only so that concurrent first launches do not race on the flag)
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)vit_attn_bwd_dkv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DKV_LDS);
        (void)hipFuncSetAttribute((const void*)vit_attn_bwd_dq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DQ_LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL(vit_attn_bwd_dq_kernel, dim3((unsigned)nblk), dim3(256), DQ_LDS, (hipStream_t)stream, a);
    if (hipGetLastError() != hipSuccess) return LIBRA_ERR_LAUNCH;
    hipLaunchKernelGGL(vit_attn_bwd_dkv_kernel, dim3((unsigned)nblk), dim3(256), DKV_LDS, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}
