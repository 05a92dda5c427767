// Routed-bridge attention for ONE new query token per sequence against the KV cache (generation):
// LibraAttention.forward with past_key_value (modeling_libra.py:344-361, :364-391) + attn_with_bridge (:267-296) at
// q_len = 1.  In this build's operands the reference's cache ([K_for_vision, K_for_language], V, V_bridge, flag) is the four
// row buffers the training kernels already produce - K_same = rope(k), K_cross = rope(k + kb), V_same = v,
// V_cross = v + vb - plus the modality flag of every cached token:
//     s_j = q . (m_q != m_j ? K_cross[j] : K_same[j]) / sqrt(128),   o = sum_j softmax(s)_j (m_q != m_j ? V_cross[j] : V_same[j]).
// HBM-bound (one pass over one K row and one V row per cached token and head): a workgroup = one (sequence, head); a 16-lane
// group owns a key at a time (16 B = 8 channels per lane: a wave reads four 256-byte rows per instruction), the dot product is
// finished with four DPP row rotations, each group keeps its own online-softmax state and the 16 groups are merged through
// LDS at the end (fp32 probabilities throughout: at q_len = 1 there is no MFMA that would want them in bf16).
#include "hip_common.hpp"
#include "../../include/libra_hip.h"

namespace libra {

struct DecodeArgs {
    const bf16_t* q; long ldq;                       // [B, H*128] roped query of the new token
    const bf16_t* ks; const bf16_t* kc; const bf16_t* vs; const bf16_t* vc;   // caches [B, Lmax, H*128]
    long ldc, bstride;                               // row stride, batch stride (elements)
    const unsigned char* kflag; long fstride;        // [B, Lmax] modality of every cached token
    const unsigned char* qflag;                      // [B] modality of the query token
    const int* lens;                                 // [B] end of the valid cached tokens (the new one included)
    const int* starts;                               // [B] first valid cached token (left-padded prompts), or null
    bf16_t* out; long ldo;                           // [B, H*128]
    int H; float sl2;                                // scale * log2(e)
};

__global__ __launch_bounds__(256) void bridge_attn_decode_kernel(const DecodeArgs p) {
    __shared__ float red_m[16], red_l[16];
    __shared__ __attribute__((aligned(16))) float red_o[16][128];
    const int h = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & 15, slot = lane >> 4;     // 8 channels [8 sub, 8 sub + 8) of key (wave * 4 + slot) + 16 i
    const int grp = wave * 4 + slot;
    const int len = p.lens[b];
    const int mq = p.qflag[b] != 0;
    float qf[8];
    unpack8(*(const u32x4*)(p.q + (long)b * p.ldq + h * 128 + sub * 8), qf);
    const long base = (long)b * p.bstride + h * 128 + sub * 8;
    const unsigned char* fl = p.kflag + (long)b * p.fstride;
    float m = -INFINITY, l = 0.f, o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
    for (int j = (p.starts ? p.starts[b] : 0) + grp; j < len; j += 16) {
        const bool cross = (fl[j] != 0) != (mq != 0);
        const long off = base + (long)j * p.ldc;
        float kf[8], vf[8];
        unpack8(*(const u32x4*)((cross ? p.kc : p.ks) + off), kf);
        unpack8(*(const u32x4*)((cross ? p.vc : p.vs) + off), vf);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s = fmaf(qf[e], kf[e], s);
        s = row_ror_add<1>(row_ror_add<2>(row_ror_add<4>(row_ror_add<8>(s)))) * p.sl2;      // all 16 lanes: q . k (log2 units)
        const float mn = fmaxf(m, s);
        const float alpha = __builtin_amdgcn_exp2f(m - mn);               // first key: exp2(-inf) = 0
        const float pr = __builtin_amdgcn_exp2f(s - mn);
        l = l * alpha + pr;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = fmaf(pr, vf[e], o[e] * alpha);
        m = mn;
    }
    if (sub == 0) { red_m[grp] = m; red_l[grp] = l; }
    *(f32x4*)(&red_o[grp][sub * 8]) = f32x4{o[0], o[1], o[2], o[3]};
    *(f32x4*)(&red_o[grp][sub * 8 + 4]) = f32x4{o[4], o[5], o[6], o[7]};
    __syncthreads();
    if (threadIdx.x < 128) {
        float M = -INFINITY;
#pragma unroll
        for (int g = 0; g < 16; ++g) M = fmaxf(M, red_m[g]);
        float L = 0.f, acc = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const float w = red_m[g] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(red_m[g] - M);   // groups that saw no key
            L = fmaf(red_l[g], w, L);
            acc = fmaf(red_o[g][threadIdx.x], w, acc);
        }
        p.out[(long)b * p.ldo + h * 128 + threadIdx.x] = f2bf(L > 0.f ? acc / L : 0.f);
    }
}

static inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace libra

using namespace libra;

extern "C" int libra_bridge_attn_decode(const void* q, int64_t ldq, const void* k_same, const void* k_cross, const void* v_same,
                                        const void* v_cross, int64_t ldc, int64_t batch_stride, const uint8_t* key_flag,
                                        int64_t flag_stride, const uint8_t* query_flag, const int* kv_len,
                                        const int* kv_start, void* out, int64_t ldo, int64_t B, int64_t H, float scale,
                                        void* stream) {
    if (B <= 0) return LIBRA_OK;
    if (H <= 0 || H > 65535 || B > 65535 || ldq < H * 128 || ldc < H * 128 || ldo < H * 128 || batch_stride < ldc) return LIBRA_ERR_SHAPE;
    if ((ldq % 8) || (ldc % 8) || (batch_stride % 8)) return LIBRA_ERR_ALIGN;
    if (!q || !k_same || !k_cross || !v_same || !v_cross || !key_flag || !query_flag || !kv_len || !out) return LIBRA_ERR_ALIGN;
    if (!al16(q) || !al16(k_same) || !al16(k_cross) || !al16(v_same) || !al16(v_cross)) return LIBRA_ERR_ALIGN;
    DecodeArgs a;
    a.q = (const bf16_t*)q; a.ldq = ldq; a.ks = (const bf16_t*)k_same; a.kc = (const bf16_t*)k_cross;
    a.vs = (const bf16_t*)v_same; a.vc = (const bf16_t*)v_cross; a.ldc = ldc; a.bstride = batch_stride;
    a.kflag = key_flag; a.fstride = flag_stride; a.qflag = query_flag; a.lens = kv_len; a.starts = kv_start; a.out = (bf16_t*)out; a.ldo = ldo;
    a.H = (int)H; a.sl2 = scale * 1.4426950408889634f;
    hipLaunchKernelGGL(bridge_attn_decode_kernel, dim3((unsigned)H, (unsigned)B), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}
