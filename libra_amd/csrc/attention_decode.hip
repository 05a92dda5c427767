// Routed-bridge attention for ONE new query token per sequence against the KV cache (generation):
// LibraAttention.forward with past_key_value (modeling_libra.py:344-361, :364-391) + attn_with_bridge (:267-296) at
// q_len = 1.  In this build's operands the reference's cache ([K_for_vision, K_for_language], V, V_bridge, flag) is the four
// row buffers the training kernels already produce - K_same = rope(k), K_cross = rope(k + kb), V_same = v,
// V_cross = v + vb - plus the modality flag of every cached token:
//     s_j = q . (m_q != m_j ? K_cross[j] : K_same[j]) / sqrt(128),   o = sum_j softmax(s)_j (m_q != m_j ? V_cross[j] : V_same[j]).
// HBM-bound (one pass over one K row and one V row per cached token and head): a workgroup = one (sequence, head, key split); a
// 16-lane group owns a key at a time (16 B = 8 channels per lane: a wave reads four 256-byte rows per instruction), the dot
// product is finished with four DPP row rotations, each group keeps its own online-softmax state and the 16 groups are merged
// through LDS at the end (fp32 probabilities throughout: at q_len = 1 there is no MFMA that would want them in bf16).
// Round 3: (i) the cached keys of a (sequence, head) are split over DEC_SPLIT workgroups (B x H = 256 workgroups of 4 waves left
// three quarters of the chip's wave slots - and of its outstanding-load capacity - empty: 73 us per layer for 134 MB, 1.8 TB/s),
// their partial (max, sum, unnormalised output) states are folded by a second small kernel; (ii) a group handles FOUR keys per
// loop trip: the four modality flags first, then all eight row loads, then the arithmetic - one memory round trip per four keys
// instead of two dependent ones (flag, then the selected row) per key.
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
    float* part; int nsplit;                         // nsplit > 1: partial states [B][H][nsplit][DEC_PART] instead of `out`
};

constexpr int DEC_SPLIT = 4;                         // key splits per (sequence, head) when a workspace is given
constexpr int DEC_PART = 132;                        // floats per partial state: 128 outputs, max, sum, 2 pad

__global__ __launch_bounds__(256) void bridge_attn_decode_kernel(const DecodeArgs p) {
    __shared__ float red_m[16], red_l[16];
    __shared__ __attribute__((aligned(16))) float red_o[16][128];
    const int h = blockIdx.x, b = blockIdx.y, z = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & 15, slot = lane >> 4;     // 8 channels [8 sub, 8 sub + 8) of key (wave * 4 + slot) + 16 i
    const int grp = wave * 4 + slot;
    const int len_all = p.lens[b];
    const int start_all = p.starts ? p.starts[b] : 0;
    // this workgroup's share of the keys: whole multiples of 64 (4 keys x 16 groups) per split
    int per = (len_all - start_all + p.nsplit - 1) / p.nsplit;
    per = (per + 63) & ~63;
    const int j0 = start_all + z * per;
    int len = j0 + per; len = len < len_all ? len : len_all;
    const int mq = p.qflag[b] != 0;
    float qf[8];
    unpack8(*(const u32x4*)(p.q + (long)b * p.ldq + h * 128 + sub * 8), qf);
    const long base = (long)b * p.bstride + h * 128 + sub * 8;
    const unsigned char* fl = p.kflag + (long)b * p.fstride;
    float m = -INFINITY, l = 0.f, o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
    for (int j = j0 + grp; j < len; j += 64) {
        bool cross[4], ok[4];
        u32x4 kq[4], vq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int jj = j + 16 * u;
            ok[u] = jj < len;
            cross[u] = (fl[ok[u] ? jj : j] != 0) != (mq != 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long off = base + (long)(ok[u] ? j + 16 * u : j) * p.ldc;
            kq[u] = *(const u32x4*)((cross[u] ? p.kc : p.ks) + off);
            vq[u] = *(const u32x4*)((cross[u] ? p.vc : p.vs) + off);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!ok[u]) continue;                                            // (uniform within the 16-lane group)
            float kf[8], vf[8];
            unpack8(kq[u], kf);
            unpack8(vq[u], vf);
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s = fmaf(qf[e], kf[e], s);
            s = row_ror_add<1>(row_ror_add<2>(row_ror_add<4>(row_ror_add<8>(s)))) * p.sl2;  // all 16 lanes: q . k (log2 units)
            const float mn = fmaxf(m, s);
            const float alpha = __builtin_amdgcn_exp2f(m - mn);           // first key: exp2(-inf) = 0
            const float pr = __builtin_amdgcn_exp2f(s - mn);
            l = l * alpha + pr;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fmaf(pr, vf[e], o[e] * alpha);
            m = mn;
        }
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
        if (p.nsplit > 1) {
            float* pt = p.part + (((long)b * p.H + h) * p.nsplit + z) * DEC_PART;
            pt[threadIdx.x] = acc;
            if (threadIdx.x == 0) { pt[128] = M; pt[129] = L; }
        } else {
            p.out[(long)b * p.ldo + h * 128 + threadIdx.x] = f2bf(L > 0.f ? acc / L : 0.f);
        }
    }
}

// fold the key splits of one (sequence, head): out = sum_z o_z 2^(m_z - M) / sum_z l_z 2^(m_z - M), in split order
__global__ __launch_bounds__(128) void bridge_attn_decode_merge_kernel(const float* __restrict__ part, int nsplit, bf16_t* __restrict__ out,
                                                                       long ldo, int H) {
    const int h = blockIdx.x, b = blockIdx.y;
    const float* pt = part + ((long)b * H + h) * nsplit * DEC_PART;
    float M = -INFINITY;
    for (int z = 0; z < nsplit; ++z) M = fmaxf(M, pt[z * DEC_PART + 128]);
    float L = 0.f, acc = 0.f;
    for (int z = 0; z < nsplit; ++z) {
        const float mz = pt[z * DEC_PART + 128];
        const float w = mz == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mz - M);
        L = fmaf(pt[z * DEC_PART + 129], w, L);
        acc = fmaf(pt[z * DEC_PART + threadIdx.x], w, acc);
    }
    out[(long)b * ldo + h * 128 + threadIdx.x] = f2bf(L > 0.f ? acc / L : 0.f);
}

static inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace libra

using namespace libra;

extern "C" size_t libra_bridge_attn_decode_workspace_bytes(int64_t B, int64_t H) {
    return (B <= 0 || H <= 0) ? 0 : (size_t)B * (size_t)H * DEC_SPLIT * DEC_PART * sizeof(float);
}

extern "C" int libra_bridge_attn_decode(const void* q, int64_t ldq, const void* k_same, const void* k_cross, const void* v_same,
                                        const void* v_cross, int64_t ldc, int64_t batch_stride, const uint8_t* key_flag,
                                        int64_t flag_stride, const uint8_t* query_flag, const int* kv_len,
                                        const int* kv_start, void* out, int64_t ldo, int64_t B, int64_t H, float scale,
                                        float* workspace, size_t workspace_bytes, void* stream) {
    if (B <= 0) return LIBRA_OK;
    if (H <= 0 || H > 65535 || B > 65535 || ldq < H * 128 || ldc < H * 128 || ldo < H * 128 || batch_stride < ldc) return LIBRA_ERR_SHAPE;
    if ((ldq % 8) || (ldc % 8) || (batch_stride % 8)) return LIBRA_ERR_ALIGN;
    if (!q || !k_same || !k_cross || !v_same || !v_cross || !key_flag || !query_flag || !kv_len || !out) return LIBRA_ERR_ALIGN;
    if (!al16(q) || !al16(k_same) || !al16(k_cross) || !al16(v_same) || !al16(v_cross)) return LIBRA_ERR_ALIGN;
    if (workspace && (!al16(workspace) || workspace_bytes < libra_bridge_attn_decode_workspace_bytes(B, H))) return LIBRA_ERR_ALIGN;
    DecodeArgs a;
    a.q = (const bf16_t*)q; a.ldq = ldq; a.ks = (const bf16_t*)k_same; a.kc = (const bf16_t*)k_cross;
    a.vs = (const bf16_t*)v_same; a.vc = (const bf16_t*)v_cross; a.ldc = ldc; a.bstride = batch_stride;
    a.kflag = key_flag; a.fstride = flag_stride; a.qflag = query_flag; a.lens = kv_len; a.starts = kv_start; a.out = (bf16_t*)out; a.ldo = ldo;
    a.H = (int)H; a.sl2 = scale * 1.4426950408889634f;
    a.part = workspace; a.nsplit = workspace ? DEC_SPLIT : 1;          // no workspace: one workgroup per (sequence, head) writes `out`
    hipLaunchKernelGGL(bridge_attn_decode_kernel, dim3((unsigned)H, (unsigned)B, (unsigned)a.nsplit), dim3(256), 0, (hipStream_t)stream, a);
    if (hipGetLastError() != hipSuccess) return LIBRA_ERR_LAUNCH;
    if (a.nsplit > 1) {
        hipLaunchKernelGGL(bridge_attn_decode_merge_kernel, dim3((unsigned)H, (unsigned)B), dim3(128), 0, (hipStream_t)stream,
                           (const float*)workspace, a.nsplit, (bf16_t*)out, (long)ldo, (int)H);
        if (hipGetLastError() != hipSuccess) return LIBRA_ERR_LAUNCH;
    }
    return LIBRA_OK;
}
