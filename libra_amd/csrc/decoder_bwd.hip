// Backward row kernels of the routed decoder (HBM-bound): CE gradient, routed RMSNorm backward (dx and the
// per-modality weight gradients), SwiGLU backward, RoPE/bridge backward.
#include "hip_common.hpp"
#include "../../include/libra_hip.h"

namespace libra {

// ---------------------------------------------------------------------------------------------------
// dz[r, v] = sum_q [t_q[r] >= 0] * coef_q * (softmax(z_r)[v] - [v == t_q[r] - sub])     (CrossEntropyLoss, mean reduction,
// averaged over the codebooks: coef_q = 1 / (count_q * Q)); one wave per row, three passes over the row.
__global__ __launch_bounds__(256) void ce_rows_bwd_kernel(const bf16_t* __restrict__ z, long ldz, int V,
                                                          const long long* __restrict__ t0, const long long* __restrict__ t1,
                                                          long long sub, float c0, float c1, const float* __restrict__ scale_dev,
                                                          bf16_t* __restrict__ dz, long lddz, long rows) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long long a0 = t0 ? t0[row] : -1, a1 = t1 ? t1[row] : -1;
    const float sc = scale_dev ? *scale_dev : 1.0f;
    const float w0 = a0 >= 0 ? c0 * sc : 0.f, w1 = a1 >= 0 ? c1 * sc : 0.f;
    const float wsum = w0 + w1;
    const bf16_t* zr = z + row * ldz;
    bf16_t* dr = dz + row * lddz;
    float mx = -INFINITY, se = 0.f;
    if (wsum != 0.f) {
        for (int c = lane; c < V; c += 64) mx = fmaxf(mx, bf2f(zr[c]));
        mx = wave_max(mx);
        for (int c = lane; c < V; c += 64) se += __expf(bf2f(zr[c]) - mx);
        se = wave_sum(se);
    }
    const float inv = wsum != 0.f ? 1.0f / se : 0.f;
    const long long i0 = a0 - sub, i1 = a1 - sub;
    for (int c = lane; c < V; c += 64) {
        float g = 0.f;
        if (wsum != 0.f) {
            g = wsum * __expf(bf2f(zr[c]) - mx) * inv;
            if (a0 >= 0 && c == i0) g -= w0;
            if (a1 >= 0 && c == i1) g -= w1;
        }
        dr[c] = f2bf(g);
    }
}

// ---------------------------------------------------------------------------------------------------
// Routed RMSNorm backward, dx part:  y = w_m * (x * rstd)  =>  dx = rstd * (g - xh * mean(g * xh)),  g = dy * w_m, xh = x * rstd
template <int NC>
__global__ __launch_bounds__(256, NC <= 8 ? 3 : 1) void rmsnorm_routed_bwd_kernel(const bf16_t* __restrict__ dy, long lddy, const bf16_t* __restrict__ x,
                                                                 long ldx, const bf16_t* __restrict__ w_lang,
                                                                 const bf16_t* __restrict__ w_vis,
                                                                 const unsigned char* __restrict__ flag,
                                                                 const float* __restrict__ rstd_i, const bf16_t* __restrict__ dres,
                                                                 long lddr, bf16_t* __restrict__ dx, long lddx, long rows, int D) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = D >> 3;
    const bf16_t* w = (flag && flag[row]) ? w_vis : w_lang;
    // Every load of the row - dy, x, the weight and the residual gradient - is issued before the first use, and the row
    // is kept as it arrived (packed bf16: 4 registers per 8 elements and operand) and unpacked again for the second
    // sweep: one exposed memory latency per row instead of two, at the register footprint of three waves per SIMD.
    u32x4 ra[NC], rb[NC], rw[NC], rr[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
        const bool on = c < nch;
        ra[i] = on ? *(const u32x4*)(dy + row * lddy + c * 8) : u32x4{0, 0, 0, 0};
        rb[i] = on ? *(const u32x4*)(x + row * ldx + c * 8) : u32x4{0, 0, 0, 0};
        rw[i] = on ? *(const u32x4*)(w + c * 8) : u32x4{0, 0, 0, 0};
        rr[i] = (on && dres) ? *(const u32x4*)(dres + row * lddr + c * 8) : u32x4{0, 0, 0, 0};
    }
    const float rstd = rstd_i[row];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        float a[8], b[8], ww[8];
        unpack8(ra[i], a); unpack8(rb[i], b); unpack8(rw[i], ww);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (a[e] * ww[e]) * (b[e] * rstd);
    }
    s = wave_sum(s) / (float)D;
#pragma unroll
    for (int i = 0; i < NC; ++i) { pin(ra[i]); pin(rb[i]); pin(rw[i]); }      // (opaque: unpack again, do not keep 3 fp32 copies)
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            float a[8], b[8], ww[8], r[8], o[8];
            unpack8(ra[i], a); unpack8(rb[i], b); unpack8(rw[i], ww); unpack8(rr[i], r);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = rstd * (a[e] * ww[e] - (b[e] * rstd) * s) + r[e];
            *(u32x4*)(dx + row * lddx + c * 8) = pack8(o);
        }
    }
}

// weight-gradient part: dw_m[c] = sum_{rows of modality m} dy[r,c] * x[r,c] * rstd[r]; thread = 8 columns, strips of rows,
// partial [strip][2][D] fp32 then a deterministic second stage.
constexpr int RW_STRIPS = 256;
__global__ __launch_bounds__(256) void rmsnorm_wgrad_partial_kernel(const bf16_t* __restrict__ dy, long lddy, const bf16_t* __restrict__ x,
                                                                    long ldx, const float* __restrict__ rstd,
                                                                    const unsigned char* __restrict__ flag, long rows, int D,
                                                                    float* __restrict__ part, const int* __restrict__ sel) {
    // sel (optional): the rows to visit, `rows` of them - when only one modality's weight is trainable (frozen-language
    // pretraining) the other modality's rows are not read at all
    const int c8 = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (c8 >= D) return;
    const long per = (rows + gridDim.y - 1) / gridDim.y;
    const long r0 = (long)blockIdx.y * per, r1 = min(rows, r0 + per);
    auto row_of = [&](long i) -> long { return sel ? (long)sel[i] : i; };
    float sl[8], sv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sl[e] = 0.f; sv[e] = 0.f; }
    auto add_row = [&](const u32x4 ra, const u32x4 rb, const float rs, const bool vis) {
        float a[8], b[8];
        unpack8(ra, a); unpack8(rb, b);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float t = a[e] * b[e] * rs;
            sl[e] += vis ? 0.f : t;
            sv[e] += vis ? t : 0.f;
        }
    };
    // four rows (eight 16-byte loads) in flight per thread; rows are accumulated in order, so the sums do not depend on it
    long r = r0;
    for (; r + 4 <= r1; r += 4) {
        u32x4 ra[4], rb[4]; float rs[4]; bool vis[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long rr = row_of(r + u);
            ra[u] = *(const u32x4*)(dy + rr * lddy + c8);
            rb[u] = *(const u32x4*)(x + rr * ldx + c8);
            rs[u] = rstd[rr];
            vis[u] = flag && flag[rr];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) add_row(ra[u], rb[u], rs[u], vis[u]);
    }
    for (; r < r1; ++r) {
        const long rr = row_of(r);
        add_row(*(const u32x4*)(dy + rr * lddy + c8), *(const u32x4*)(x + rr * ldx + c8), rstd[rr], flag && flag[rr]);
    }
    float* d0 = part + ((long)blockIdx.y * 2) * D + c8;
#pragma unroll
    for (int e = 0; e < 8; ++e) { d0[e] = sl[e]; d0[D + e] = sv[e]; }
}
__global__ __launch_bounds__(1024) void rmsnorm_wgrad_final_kernel(const float* __restrict__ part, int strips, int D,
                                                                   float* __restrict__ out_l, float* __restrict__ out_v) {
    __shared__ float red[32][33];
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int which = blockIdx.y;
    const int col = blockIdx.x * 32 + c;
    float s = 0.f;
    if (col < D)
        for (int p = g; p < strips; p += 32) s += part[((long)p * 2 + which) * D + col];
    red[g][c] = s;
    __syncthreads();
    if (g == 0 && col < D) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) t += red[i][c];
        float* o = which ? out_v : out_l;
        if (o) o[col] += t;
    }
}

// ---------------------------------------------------------------------------------------------------
// SwiGLU backward: y = silu(g) * u  =>  dg = dy * u * s * (1 + g (1 - s)),  du = dy * silu(g),  s = sigmoid(g)
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* __restrict__ dy, long lddy, const bf16_t* __restrict__ g,
                                                         const bf16_t* __restrict__ u, long ldgu, bf16_t* __restrict__ dg,
                                                         bf16_t* __restrict__ du, long ldd, long rows, int I) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int i8 = I >> 3;
    if (i >= rows * i8) return;
    const long r = i / i8;
    const int c = (int)(i - r * i8) * 8;
    float a[8], b[8], d[8], og[8], ou[8];
    unpack8(*(const u32x4*)(g + r * ldgu + c), a);
    unpack8(*(const u32x4*)(u + r * ldgu + c), b);
    unpack8(*(const u32x4*)(dy + r * lddy + c), d);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float s = 1.0f / (1.0f + __expf(-a[e]));
        og[e] = d[e] * b[e] * s * (1.0f + a[e] * (1.0f - s));
        ou[e] = d[e] * a[e] * s;
    }
    *(u32x4*)(dg + r * ldd + c) = pack8(og);
    *(u32x4*)(du + r * ldd + c) = pack8(ou);
}

// ---------------------------------------------------------------------------------------------------
// RoPE / bridge backward.  Forward:  q' = R q,  K_same = R k,  K_cross = R (k + kb),  V_cross = v + vb  (R = rotation by
// position), kb = B_k[m] t_k, vb = B_v[m] t_v.  So  dq = R^T dq',  dk = R^T (dK_same + dK_cross),  dkb = R^T dK_cross,
// dv = dV_same + dV_cross,  dvb = dV_cross, and the gradient of the rank-8 bridge activations dt_k = B_k[m]^T dkb,
// dt_v = B_v[m]^T dvb - taken here, while dkb / dvb are in registers, instead of by four skinny GEMMs that re-read them.
// R^T: (y1, y2) at (d, d+64) with c, s:  x1 = y1 c + y2 s,  x2 = y2 c - y1 s.
// One thread = 4 channels d in [4c, 4c+4) and their partners d + 64; a workgroup = all H*16 threads of one token at a time,
// ROPE_BWD_TOK consecutive tokens; the thread's rows of B_k / B_v stay in registers and are reloaded when the modality flips.
struct RopeBwdArgs {
    const bf16_t* dq; const bf16_t* dks; const bf16_t* dkc; const bf16_t* dvs; const bf16_t* dvc; long ld;   // [N, H*128]
    const bf16_t* cos; const bf16_t* sin;
    bf16_t* dqkv; long ldo;      // [N, 3*H*128]
    bf16_t* dkb; long ldb;       // [N, H*128]
    const bf16_t* bk_l; const bf16_t* bk_v; const bf16_t* bv_l; const bf16_t* bv_v;   // weight_B [H*128, 8]
    const unsigned char* flag;
    bf16_t* dtb; long ldt;       // [N, >= 16]: cols 0..7 = dt_k, 8..15 = dt_v
    long N; int S, H;
    const int* positions; int pos_stride, max_pos;   // optional explicit positions [N, pos_stride] (see libra_rope_bridge_pos)
};
constexpr int ROPE_BWD_TOK = 16;

typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
// c + a.lo * b.lo + a.hi * b.hi on packed bf16 pairs (v_dot2_f32_bf16, fp32 accumulate)
__device__ __forceinline__ float dot2bf(unsigned a, unsigned b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_hw, a), __builtin_bit_cast(bf16x2_hw, b), c, false);
}
__device__ __forceinline__ void unpack4b(const u32x2 v, float* f) {
    f[0] = __uint_as_float(v[0] << 16); f[1] = __uint_as_float(v[0] & 0xffff0000u);
    f[2] = __uint_as_float(v[1] << 16); f[3] = __uint_as_float(v[1] & 0xffff0000u);
}
__device__ __forceinline__ u32x2 pack4b(const float* f) {
    u32x2 v; v[0] = pack2bf(f[0], f[1]); v[1] = pack2bf(f[2], f[3]); return v;
}

__global__ __launch_bounds__(512) void rope_bridge_bwd_kernel(const RopeBwdArgs p) {
    __shared__ float part[2][8][16];                          // [token parity][wave][16 partial sums]
    const int LT = p.H * 16;                                  // threads per token (blockDim.x = LT rounded up to 64)
    const int lt = threadIdx.x;
    const bool on = lt < LT;
    const int c = lt & 15, h = on ? lt >> 4 : 0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nwaves = (blockDim.x + 63) >> 6;
    const int HD = p.H * 128;
    const long col0 = (long)h * 128 + c * 4, col1 = col0 + 64;
    // bridge weights of this thread's 4 + 4 channels for all 8 ranks, as bf16 channel PAIRS (the operands are B^T [8, D]):
    // one v_dot2_f32_bf16 consumes two channels of a packed gradient dword against them
    u32x2 wk[8][2], wv[8][2];
    int cur_mod = -1;
    const long n0 = (long)blockIdx.x * ROPE_BWD_TOK;
    int s = (int)(n0 % p.S);                                  // one division per thread, then stepped
    for (int j = 0; j < ROPE_BWD_TOK; ++j, ++s) {
        const long n = n0 + j;
        if (n >= p.N) break;                                  // uniform over the workgroup
        if (s >= p.S) s -= p.S;
        float acc[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        if (on) {
            const int vis = p.dtb ? p.flag[n] != 0 : 0;             // (flag may be null without the bridge operands)
            if (p.dtb && vis != cur_mod) {
                const bf16_t* bk = vis ? p.bk_v : p.bk_l;
                const bf16_t* bv = vis ? p.bv_v : p.bv_l;
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        wk[r][hf] = *(const u32x2*)(bk + (long)r * HD + (hf ? col1 : col0));
                        wv[r][hf] = *(const u32x2*)(bv + (long)r * HD + (hf ? col1 : col0));
                    }
                cur_mod = vis;
            }
            float cs[4], sn[4];
            const int pos = p.positions ? min(max(p.positions[n * p.pos_stride + (p.pos_stride == 2 ? (h & 1) : 0)], 0), p.max_pos - 1) : s;
            unpack4b(*(const u32x2*)(p.cos + (long)pos * 128 + c * 4), cs);
            unpack4b(*(const u32x2*)(p.sin + (long)pos * 128 + c * 4), sn);
            auto ld2 = [&](const bf16_t* t, float* a, float* b) {
                unpack4b(*(const u32x2*)(t + n * p.ld + col0), a);
                unpack4b(*(const u32x2*)(t + n * p.ld + col1), b);
            };
            float q1[4], q2[4], ks1[4], ks2[4], kc1[4], kc2[4], vs1[4], vs2[4], vc1[4], vc2[4];
            const u32x2 rvc1 = *(const u32x2*)(p.dvc + n * p.ld + col0), rvc2 = *(const u32x2*)(p.dvc + n * p.ld + col1);
            ld2(p.dq, q1, q2); ld2(p.dks, ks1, ks2); ld2(p.dkc, kc1, kc2); ld2(p.dvs, vs1, vs2);
            unpack4b(rvc1, vc1); unpack4b(rvc2, vc2);
            float oq1[4], oq2[4], ok1[4], ok2[4], ob1[4], ob2[4], ov1[4], ov2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                oq1[e] = q1[e] * cs[e] + q2[e] * sn[e];
                oq2[e] = q2[e] * cs[e] - q1[e] * sn[e];
                ob1[e] = kc1[e] * cs[e] + kc2[e] * sn[e];
                ob2[e] = kc2[e] * cs[e] - kc1[e] * sn[e];
                const float t1 = ks1[e] + kc1[e], t2 = ks2[e] + kc2[e];
                ok1[e] = t1 * cs[e] + t2 * sn[e];
                ok2[e] = t2 * cs[e] - t1 * sn[e];
                ov1[e] = vs1[e] + vc1[e];
                ov2[e] = vs2[e] + vc2[e];
            }
            const u32x2 b1 = pack4b(ob1), b2 = pack4b(ob2);
            *(u32x2*)(p.dqkv + n * p.ldo + col0) = pack4b(oq1);
            *(u32x2*)(p.dqkv + n * p.ldo + col1) = pack4b(oq2);
            *(u32x2*)(p.dqkv + n * p.ldo + HD + col0) = pack4b(ok1);
            *(u32x2*)(p.dqkv + n * p.ldo + HD + col1) = pack4b(ok2);
            *(u32x2*)(p.dqkv + n * p.ldo + 2 * HD + col0) = pack4b(ov1);
            *(u32x2*)(p.dqkv + n * p.ldo + 2 * HD + col1) = pack4b(ov2);
            *(u32x2*)(p.dkb + n * p.ldb + col0) = b1;
            *(u32x2*)(p.dkb + n * p.ldb + col1) = b2;
            if (p.dtb) {
                // dt_k[r] += dkb[c] B_k[c][r] with dkb as stored (bf16), dt_v[r] += dvb[c] B_v[c][r] with dvb = dV_cross:
                // both gradients are already packed channel pairs
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    acc[r] = dot2bf(b1[0], wk[r][0][0], dot2bf(b1[1], wk[r][0][1], dot2bf(b2[0], wk[r][1][0], dot2bf(b2[1], wk[r][1][1], acc[r]))));
                    acc[8 + r] = dot2bf(rvc1[0], wv[r][0][0], dot2bf(rvc1[1], wv[r][0][1], dot2bf(rvc2[0], wv[r][1][0], dot2bf(rvc2[1], wv[r][1][1], acc[8 + r]))));
                }
            }
        }
        if (p.dtb) {
            // workgroup reduction of the 16 sums of this token: wave sums (registers only), then 16 threads add the waves
            float rs[4];
            wave_sum16(acc, rs);                              // rs[e] in row rho = total of acc[e + 4 rho]
            if ((lane & 15) == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) part[j & 1][wave][e + 4 * (lane >> 4)] = rs[e];
            }
            __syncthreads();                                  // (double-buffered by token parity: one barrier per token)
            if (threadIdx.x < 16) {
                float t = 0.f;
                for (int w = 0; w < nwaves; ++w) t += part[j & 1][w][threadIdx.x];
                p.dtb[n * p.ldt + threadIdx.x] = f2bf(t);
            }
        }
    }
}

static inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
static inline int launched() { return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH; }

}  // namespace libra

using namespace libra;

extern "C" int libra_ce_rows_bwd(const void* logits, int64_t ldz, int64_t V, const int64_t* target0, const int64_t* target1,
                                 int64_t target_sub, float coef0, float coef1, const float* scale_dev, void* dlogits,
                                 int64_t lddz, int64_t rows, void* stream) {
    if (rows <= 0) return LIBRA_OK;
    if (V <= 0 || ldz < V || lddz < V) return LIBRA_ERR_SHAPE;
    if (!logits || !dlogits || (!target0 && !target1)) return LIBRA_ERR_ALIGN;
    hipLaunchKernelGGL(ce_rows_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)logits, (long)ldz, (int)V, (const long long*)target0, (const long long*)target1,
                       (long long)target_sub, coef0, coef1, scale_dev, (bf16_t*)dlogits, (long)lddz, (long)rows);
    return launched();
}

extern "C" int libra_rmsnorm_routed_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* w_lang,
                                        const void* w_vis, const uint8_t* flag, const float* rstd, const void* dres,
                                        int64_t lddr, void* dx, int64_t lddx, int64_t rows, int64_t D, void* stream) {
    if (rows <= 0) return LIBRA_OK;
    if (D <= 0 || (D % 8) || D > 8192 || lddy < D || ldx < D || lddx < D || (lddy % 8) || (ldx % 8) || (lddx % 8)) return LIBRA_ERR_SHAPE;
    if (!dy || !x || !w_lang || !rstd || !dx || (flag && !w_vis) || (dres && (lddr % 8))) return LIBRA_ERR_ALIGN;
    if (!al16(dy) || !al16(x) || !al16(w_lang) || !al16(dx) || (w_vis && !al16(w_vis)) || (dres && !al16(dres))) return LIBRA_ERR_ALIGN;
    const unsigned grid = (unsigned)((rows + 3) / 4);
    const int nc = (int)((D / 8 + 63) / 64);
#define LAUNCH_RB(NC)                                                                                                \
    hipLaunchKernelGGL((rmsnorm_routed_bwd_kernel<NC>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, \
                       (long)lddy, (const bf16_t*)x, (long)ldx, (const bf16_t*)w_lang, (const bf16_t*)w_vis, flag, rstd,  \
                       (const bf16_t*)dres, (long)lddr, (bf16_t*)dx, (long)lddx, (long)rows, (int)D)
    if (nc <= 1) LAUNCH_RB(1); else if (nc <= 2) LAUNCH_RB(2); else if (nc <= 4) LAUNCH_RB(4);
    else if (nc <= 8) LAUNCH_RB(8); else LAUNCH_RB(16);
#undef LAUNCH_RB
    return launched();
}

extern "C" size_t libra_rmsnorm_wgrad_workspace_bytes(int64_t rows, int64_t D) {
    return (rows > 0 && D > 0) ? (size_t)RW_STRIPS * 2 * D * sizeof(float) : 0;
}

extern "C" int libra_rmsnorm_routed_wgrad(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* rstd,
                                          const uint8_t* flag, float* dw_lang, float* dw_vis, void* workspace,
                                          size_t workspace_bytes, int64_t rows, int64_t D, const int32_t* rows_sel,
                                          int64_t n_sel, void* stream) {
    if (rows_sel) {                                                   // visit only these rows (all inside [0, rows): caller's contract)
        if (n_sel < 0 || n_sel > rows) return LIBRA_ERR_SHAPE;
        rows = n_sel;
    }
    if (rows <= 0) return LIBRA_OK;                                   // nothing to add
    if (D <= 0 || (D % 8) || lddy < D || ldx < D || (lddy % 8) || (ldx % 8)) return LIBRA_ERR_SHAPE;
    if (!dy || !x || !rstd || !workspace || !al16(dy) || !al16(x) || !al16(workspace)) return LIBRA_ERR_ALIGN;
    if (workspace_bytes < libra_rmsnorm_wgrad_workspace_bytes(rows, D)) return LIBRA_ERR_ALIGN;
    const int strips = (int)(rows < RW_STRIPS ? rows : RW_STRIPS);
    dim3 grid((unsigned)((D + 2047) / 2048), (unsigned)strips);
    hipLaunchKernelGGL(rmsnorm_wgrad_partial_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (long)lddy,
                       (const bf16_t*)x, (long)ldx, rstd, flag, (long)rows, (int)D, (float*)workspace, (const int*)rows_sel);
    if (hipGetLastError() != hipSuccess) return LIBRA_ERR_LAUNCH;
    hipLaunchKernelGGL(rmsnorm_wgrad_final_kernel, dim3((unsigned)((D + 31) / 32), 2), dim3(1024), 0, (hipStream_t)stream,
                       (const float*)workspace, strips, (int)D, dw_lang, dw_vis);
    return launched();
}

extern "C" int libra_swiglu_bwd(const void* dy, int64_t lddy, const void* gate, const void* up, int64_t ldgu, void* dgate,
                                void* dup, int64_t ldd, int64_t rows, int64_t I, void* stream) {
    if (rows <= 0) return LIBRA_OK;
    if (I <= 0 || (I % 8) || lddy < I || ldgu < I || ldd < I || (lddy % 8) || (ldgu % 8) || (ldd % 8)) return LIBRA_ERR_SHAPE;
    if (!dy || !gate || !up || !dgate || !dup || !al16(dy) || !al16(gate) || !al16(up) || !al16(dgate) || !al16(dup)) return LIBRA_ERR_ALIGN;
    const long total = rows * (I / 8);
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)dy, (long)lddy, (const bf16_t*)gate, (const bf16_t*)up, (long)ldgu, (bf16_t*)dgate,
                       (bf16_t*)dup, (long)ldd, (long)rows, (int)I);
    return launched();
}

extern "C" int libra_rope_bridge_bwd(const void* dq, const void* dk_same, const void* dk_cross, const void* dv_same,
                                     const void* dv_cross, int64_t ld, const void* cos, const void* sin, int64_t max_pos,
                                     void* dqkv, int64_t ldo, void* dkb, int64_t ldb, const void* bk_l, const void* bk_v,
                                     const void* bv_l, const void* bv_v, const uint8_t* flag, void* dtb, int64_t ldt,
                                     int64_t N, int64_t S, int64_t H, const int* positions, int64_t pos_stride, void* stream) {
    if (N <= 0) return LIBRA_OK;
    if (positions && pos_stride != 1 && pos_stride != 2) return LIBRA_ERR_SHAPE;
    if (positions) S = S > max_pos ? max_pos : S;                      // (S only paces the implicit positions)
    if (H <= 0 || H > 32 || S <= 0 || S > max_pos || ld < H * 128 || ldo < 3 * H * 128 || ldb < H * 128) return LIBRA_ERR_SHAPE;
    if ((ld % 8) || (ldo % 8) || (ldb % 8)) return LIBRA_ERR_ALIGN;
    if (!dq || !dk_same || !dk_cross || !dv_same || !dv_cross || !cos || !sin || !dqkv || !dkb) return LIBRA_ERR_ALIGN;
    if (!al16(dq) || !al16(dk_same) || !al16(dk_cross) || !al16(dv_same) || !al16(dv_cross) || !al16(dqkv) || !al16(dkb)) return LIBRA_ERR_ALIGN;
    if (dtb && (!bk_l || !bk_v || !bv_l || !bv_v || !flag || ldt < 16 || !al16(bk_l) || !al16(bk_v) || !al16(bv_l) || !al16(bv_v)))
        return LIBRA_ERR_ALIGN;
    RopeBwdArgs a;
    a.dq = (const bf16_t*)dq; a.dks = (const bf16_t*)dk_same; a.dkc = (const bf16_t*)dk_cross; a.dvs = (const bf16_t*)dv_same;
    a.dvc = (const bf16_t*)dv_cross; a.ld = ld; a.cos = (const bf16_t*)cos; a.sin = (const bf16_t*)sin;
    a.dqkv = (bf16_t*)dqkv; a.ldo = ldo; a.dkb = (bf16_t*)dkb; a.ldb = ldb;
    a.bk_l = (const bf16_t*)bk_l; a.bk_v = (const bf16_t*)bk_v; a.bv_l = (const bf16_t*)bv_l; a.bv_v = (const bf16_t*)bv_v;
    a.flag = flag; a.dtb = (bf16_t*)dtb; a.ldt = ldt; a.N = N; a.S = (int)S; a.H = (int)H;
    a.positions = positions; a.pos_stride = (int)pos_stride; a.max_pos = (int)max_pos;
    const int threads = (int)((H * 16 + 63) / 64 * 64);
    const long grid = (N + ROPE_BWD_TOK - 1) / ROPE_BWD_TOK;
    if (grid > 0x7fffffffL) return LIBRA_ERR_SHAPE;
    hipLaunchKernelGGL(rope_bridge_bwd_kernel, dim3((unsigned)grid), dim3(threads), 0, (hipStream_t)stream, a);
    return launched();
}
