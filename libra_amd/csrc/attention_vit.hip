// ViT (non-causal, head_dim 64) flash attention for gfx950 — forward.
//
// One workgroup = 128 query rows of one (image, head): 4 wave64, 32 query rows per wave.
// Everything is computed TRANSPOSED so that a softmax row lives in ONE lane pair:
//     S^T[key, q] = K · Q^T        (MFMA A = K fragment from LDS,  B = Q fragment held in registers)
//     O^T[d,  q] = V^T · P^T       (MFMA A = V^T fragment from LDS, B = P^T taken *directly* from the
//                                   S^T accumulator registers — the 32x32 C/D layout of lane (q, half)
//                                   already is a valid B-operand slot assignment, because the k-index
//                                   order of an MFMA contraction is free as long as A and B agree)
// so the row max / row sum need one cross-half exchange instead of a 32-lane butterfly and P never
// moves between lanes or through LDS.
//
// K and V tiles [64 keys][64 d] stream HBM -> LDS with 16-byte direct-to-LDS loads (source-side XOR swizzle,
// mirrored on the fragment reads; attn_tiles64.hpp), double buffered, one barrier per tile.  V is staged as it lies in
// the fused qkv activation and read with the LDS transpose load: no V^T copy exists.
// Work-group ids are XCD-remapped so the 5 query tiles that share one (image, head)'s K/V run on the
// same XCD/L2.
//
// Scores are kept in fp32 (never rounded to bf16), P is rounded to bf16 only as the MFMA operand and
// the row sum is taken from the un-rounded fp32 P; the output is normalised once at the end.
#include "hip_common.hpp"
#include "attn_tiles64.hpp"
#include "../../include/libra_hip.h"

namespace libra {

constexpr int HD = 64;            // head dim
constexpr int QB = 128;           // query rows per workgroup
constexpr int KB = 64;            // keys per tile
constexpr int KV_TILE_BYTES = KB * HD * 2;   // 8 KiB
constexpr int ATT_LDS = 4 * KV_TILE_BYTES;   // K0 V0 K1 V1 = 32 KiB

struct AttnFwdArgs {
    const bf16_t* qkv; long ld_qkv;
    bf16_t* out; long ld_out;
    float* lse;
    bf16_t* out_lo;                // optional: bf16(O_fp32 - float(bf16(O_fp32))), same layout as out (see libra_vit_attn_fwd)
    int B, T, H, n_qt;
    float sl2;      // scale * log2(e)
};

__global__ __launch_bounds__(256, 2) void vit_attn_fwd_kernel(const AttnFwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int l31 = lane & 31;

    const int nblk = p.B * p.H * p.n_qt;
    const int L = xcd_remap(blockIdx.x, nblk);
    const int qt = L % p.n_qt;
    const int bh = L / p.n_qt;
    const int h = bh % p.H;
    const int b = bh / p.H;

    const int T = p.T;
    const long tok0 = (long)b * T;
    const int q0 = qt * QB + wave * 32;
    const bool active = q0 < T;                         // wave-uniform

    // ---- Q fragments (B operand of S^T = K Q^T): lane (q = l31, half) holds Q[q][ks*16 + half*8 .. +8]
    bf16x8 qf[4];
    {
        int q = q0 + l31;
        q = q < T ? q : T - 1;
        const bf16_t* qp = p.qkv + (tok0 + q) * p.ld_qkv + h * HD + half * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
    }

    const bf16_t* kbase = p.qkv + tok0 * p.ld_qkv + (long)p.H * HD + h * HD;            // K rows of this image/head
    const bf16_t* vbase = kbase + (long)p.H * HD;                                          // V rows of this image/head

    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int nkt = (T + KB - 1) / KB;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) pin(qf[ks]);         // Q has landed before any LDS-DMA is in flight
    stage_tile64(kbase, p.ld_qkv, 0, T, smem, wave, lane);
    stage_tile64(vbase, p.ld_qkv, 0, T, smem + KV_TILE_BYTES, wave, lane);

    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int cur = kt & 1;
        if (kt + 1 < nkt) {
            char* nb = smem + (cur ^ 1) * 2 * KV_TILE_BYTES;
            stage_tile64(kbase, p.ld_qkv, (kt + 1) * KB, T, nb, wave, lane);
            stage_tile64(vbase, p.ld_qkv, (kt + 1) * KB, T, nb + KV_TILE_BYTES, wave, lane);
        }
        if (!active) continue;
        const char* sk = smem + cur * 2 * KV_TILE_BYTES;
        const char* sv = sk + KV_TILE_BYTES;
        const int kv0 = kt * KB;

        // ---- S^T = K Q^T : two 32-key sub-tiles
        f32x16 s[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[c][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(sk + off64(c * 32 + l31, ks * 2 + half));
                s[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[c], 0, 0, 0);
            }
        }
        // ---- mask the key tail (last tile only), running max (advanced only when a tile exceeds it by 2^DEFER_THR:
        //      P <= 256 then, which bf16 carries at the same relative precision), exponentiate
        if (kv0 + KB > T) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + c * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    s[c][r] = key < T ? s[c][r] : -INFINITY;
                }
        }
        float tmax = max3f(s[0][0], s[0][1], s[1][0]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) tmax = max3f(tmax, s[0][r], s[0][r + 1]);
#pragma unroll
        for (int r = 1; r < 15; r += 2) tmax = max3f(tmax, s[1][r], s[1][r + 1]);
        tmax = fmaxf(tmax, s[1][15]);
        tmax = half_swap_max(tmax * p.sl2);
        const float m_new = fmaxf(m_run, tmax);
        if (__any(m_new > m_run + DEFER_THR)) {          // wave-uniform; the first tile always lands here
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        }
        const float nm = -m_run;
        float psum = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[c][r], p.sl2, nm));
                s[c][r] = e;
                psum += e;
            }
        l_run += psum;

        // ---- O^T += V^T P^T : k-step (c, sx) consumes S^T accumulator registers 8sx..8sx+7, i.e. for this
        // lane half the keys  c*32 + 16sx + 4half + {0,1,2,3, 8,9,10,11}; V^T fragments come from the V tile by
        // LDS transpose reads (clamped tail rows meet P = 0)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int sx = 0; sx < 2; ++sx) {
                union { bf16x8 v; unsigned u[4]; } pb;
#pragma unroll
                for (int j = 0; j < 4; ++j) pb.u[j] = pack2bf(s[c][8 * sx + 2 * j], s[c][8 * sx + 2 * j + 1]);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tread64(sv, lane, dt, c * 2 + sx), pb.v, o[dt], 0, 0, 0);
            }
    }

    // ---- finish: combine the two halves' partial row sums, normalise, transpose through LDS, store
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    __syncthreads();                                   // everyone is done with the K/V buffers
    constexpr int OROW = 136;                          // bytes per staged output row (64 bf16 + 8 B pad)
    char* so = smem + wave * (32 * OROW);
    // two passes through the per-wave staging rows: the bf16 output, then (when asked for) its rounding residual
#pragma unroll 1
    for (int part = 0; part < (p.out_lo ? 2 : 1); ++part) {
        if (part) __syncthreads();
        if (active) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = dt * 32 + 8 * g + 4 * half;
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
            if (!part && p.lse && half == 0 && q0 + l31 < T)
                p.lse[((long)b * p.H + h) * T + q0 + l31] = (m_run + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f;
        }
        __syncthreads();
        if (active) {
            bf16_t* dst = part ? p.out_lo : p.out;
            // 32 rows x 128 B: lane -> (row = pass*8 + lane/8, 16-byte chunk lane%8)
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int r = pass * 8 + (lane >> 3);
                const int q = q0 + r;
                if (q < T) {
                    const char* src = so + r * OROW + (lane & 7) * 16;
                    u32x4 v;
                    const u32x2 a = *(const u32x2*)src;
                    const u32x2 c2 = *(const u32x2*)(src + 8);
                    v[0] = a[0]; v[1] = a[1]; v[2] = c2[0]; v[3] = c2[1];
                    *(u32x4*)(dst + (tok0 + q) * p.ld_out + h * HD + (lane & 7) * 8) = v;
                }
            }
        }
    }
}

}  // namespace libra

using namespace libra;

extern "C" int libra_vit_attn_fwd(const void* qkv, int64_t ld_qkv, void* out, int64_t ld_out, float* lse, void* out_lo, int64_t B,
                                  int64_t T, int64_t H, float scale, void* stream) {
    if (B <= 0 || T <= 0) return LIBRA_OK;
    if (H <= 0 || ld_qkv < 3 * H * HD || ld_out < H * HD) return LIBRA_ERR_SHAPE;
    if ((ld_qkv % 8) || (ld_out % 8)) return LIBRA_ERR_ALIGN;
    if (!qkv || !out || (((uintptr_t)qkv | (uintptr_t)out) & 15)) return LIBRA_ERR_ALIGN;
    AttnFwdArgs a;
    a.qkv = (const bf16_t*)qkv; a.ld_qkv = ld_qkv;
    if (out_lo && ((uintptr_t)out_lo & 15)) return LIBRA_ERR_ALIGN;
    a.out = (bf16_t*)out; a.ld_out = ld_out; a.lse = lse; a.out_lo = (bf16_t*)out_lo;
    a.B = (int)B; a.T = (int)T; a.H = (int)H; a.n_qt = (int)((T + QB - 1) / QB);
    a.sl2 = scale * 1.4426950408889634f;
    const long nblk = (long)B * H * a.n_qt;
    if (nblk > 0x7fffffffL) return LIBRA_ERR_SHAPE;
    hipLaunchKernelGGL(vit_attn_fwd_kernel, dim3((unsigned)nblk), dim3(256), ATT_LDS, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}
