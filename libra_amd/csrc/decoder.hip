// Row kernels of the routed decoder (all HBM-bound): modality-routed RMSNorm, RoPE + rank-8 bridge
// expansion, SwiGLU, multi-codebook embedding assembly, fused cross-entropy statistics.
#include "hip_common.hpp"
#include "../../include/libra_hip.h"

namespace libra {

// ---------------------------------------------------------------------------------------------------
// Routed RMSNorm (LlamaRMSNorm, modeling_llama.py:127-132, routed by cal_language_vision, modeling_libra.py:463,
// :481,:817): y = w_m * bf16(x * rsqrt(mean(x^2) + eps)), w_m chosen per row by the modality flag.  The x*rstd
// product is rounded to bf16 BEFORE the weight multiply, as `self.weight * hidden_states.to(input_dtype)` does.
template <int NC>
__global__ __launch_bounds__(256) void rmsnorm_routed_kernel(const bf16_t* __restrict__ x, long ldx,
                                                             const bf16_t* __restrict__ w_lang,
                                                             const bf16_t* __restrict__ w_vis,
                                                             const unsigned char* __restrict__ flag,
                                                             bf16_t* __restrict__ y, long ldy, float* __restrict__ rstd_o,
                                                             long rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = D >> 3;
    // the weight row is requested together with x (its modality flag first): one exposed memory latency per row, not two
    const bf16_t* w = (flag && flag[row]) ? w_vis : w_lang;
    u32x4 rx[NC], rw[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
        rx[i] = c < nch ? *(const u32x4*)(x + row * ldx + c * 8) : u32x4{0, 0, 0, 0};
        rw[i] = c < nch ? *(const u32x4*)(w + c * 8) : u32x4{0, 0, 0, 0};
    }
    float v[NC][8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        unpack8(rx[i], v[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) ss += v[i][e] * v[i][e];
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)D + eps);
    if (rstd_o && lane == 0) rstd_o[row] = rstd;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            float g[8], o[8];
            unpack8(rw[i], g);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = g[e] * bf2f(f2bf(v[i][e] * rstd));
            *(u32x4*)(y + row * ldy + c * 8) = pack8(o);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// RoPE + bridge (modeling_libra.py:318-340, :282-286, apply_rotary_pos_emb :39-61).  Per token n (position
// s = n % S) and head, head_dim 128:
//   kb = B_k[m] t_k,  vb = B_v[m] t_v          (rank-8 expansion of the bridge low-rank activations, m = modality)
//   q' = rope(q);  k_same = rope(k);  k_cross = rope(bf16(k + kb));  v_cross = bf16(v + vb)
// q and k_same overwrite qkv in place; every product / sum is rounded to bf16 where the reference's bf16 ops do.
// One thread = 8 channels d in [8c, 8c+8) and their RoPE partners d + 64.
struct RopeArgs {
    bf16_t* qkv; long ld; int H;                  // [N, 3*H*128]: q | k | v
    const bf16_t* tb; long ldt;                   // [N, >=16]: cols 0..7 = k-bridge t, 8..15 = v-bridge t
    const bf16_t* bk_l; const bf16_t* bk_v; const bf16_t* bv_l; const bf16_t* bv_v;   // weight_B [H*128, 8]
    const unsigned char* flag;
    const bf16_t* cos; const bf16_t* sin;         // [max_pos, 128] bf16 (the reference casts its fp32 cache to x.dtype)
    bf16_t* k_cross; bf16_t* v_cross; long ldc;   // [N, H*128]
    long N; int S;
    const int* positions;                         // optional [N, pos_stride]: RoPE position of token n (NULL: n % S)
    int pos_stride;                               // 1, or 2 = use_2d_rope: even heads take column 0 (row position), odd heads column 1
    int max_pos;                                  // rows of cos / sin: explicit positions are clamped into the table
    int tok;                                      // consecutive tokens per thread (ROPE_TOK; 1 for the few rows of a generation step)
    // generation step (one token per sequence, token n = sequence n): also store the token's four K / V rows at cache slot *slot
    bf16_t* cache_ks; bf16_t* cache_kc; bf16_t* cache_vs; bf16_t* cache_vc;      // [B, Lmax, H*128] each, or all null
    long c_row, c_batch; const long long* slot;
};

__device__ __forceinline__ float rbf(float x) { return bf2f(f2bf(x)); }
typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
// c + a.lo * b.lo + a.hi * b.hi on packed bf16 pairs (v_dot2_f32_bf16, fp32 accumulate)
__device__ __forceinline__ float dot2bf(unsigned a, unsigned b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_hw, a), __builtin_bit_cast(bf16x2_hw, b), c, false);
}

// One thread = 4 channels d in [4c, 4c+4) and their RoPE partners d + 64.  It keeps those 8 channels' rows of B_k / B_v
// (64 dwords) in registers across ROPE_TOK consecutive tokens and reloads them only when the modality flips: read per
// token from L2 they were 128 KB of weight traffic per token (2.1 GB per call at N = 16384, more than the activations).
constexpr int ROPE_TOK = 16;

__device__ __forceinline__ void unpack4(const u32x2 v, float* f) {
    f[0] = __uint_as_float(v[0] << 16); f[1] = __uint_as_float(v[0] & 0xffff0000u);
    f[2] = __uint_as_float(v[1] << 16); f[3] = __uint_as_float(v[1] & 0xffff0000u);
}
__device__ __forceinline__ u32x2 pack4(const float* f) {
    u32x2 v; v[0] = pack2bf(f[0], f[1]); v[1] = pack2bf(f[2], f[3]); return v;
}

__global__ __launch_bounds__(256) void rope_bridge_kernel(const RopeArgs p) {
    const int LT = p.H * 16;                                  // threads per token
    const int tpb = LT >= 256 ? 1 : 256 / LT;                 // token slots per block
    const int slot = LT >= 256 ? 0 : (int)threadIdx.x / LT;
    const int lt = LT >= 256 ? (int)(blockIdx.y * 256 + threadIdx.x) : (int)threadIdx.x % LT;
    if (slot >= tpb || lt >= LT) return;
    const int c = lt & 15, h = lt >> 4;
    const int HD = p.H * 128;
    const long n_first = (long)blockIdx.x * p.tok * tpb + slot;
    u32x4 wk[2][4], wv[2][4];
    int cur_mod = -1;
    int s = (int)(n_first % p.S);                             // position of the token in its sequence: one division per
    for (int j = 0; j < p.tok; ++j, s += tpb) {               // thread, then stepped (a 64-bit modulo per token was ~100 VALU ops)
        const long n = n_first + (long)j * tpb;
        if (n >= p.N) break;
        while (s >= p.S) s -= p.S;
        const int pos = p.positions ? min(max(p.positions[n * p.pos_stride + (p.pos_stride == 2 ? (h & 1) : 0)], 0), p.max_pos - 1) : s;
        const int vis = p.flag[n] != 0;
        if (vis != cur_mod) {
            const bf16_t* bk = vis ? p.bk_v : p.bk_l;
            const bf16_t* bv = vis ? p.bv_v : p.bv_l;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const long col = (long)h * 128 + hf * 64 + c * 4 + e;
                    wk[hf][e] = *(const u32x4*)(bk + col * 8);
                    wv[hf][e] = *(const u32x4*)(bv + col * 8);
                }
            cur_mod = vis;
        }
        // t_k, t_v stay packed: (rank 2i, 2i+1) pairs are exactly the operand shape of v_dot2_f32_bf16 against weight_B rows
        const u32x4 tk = *(const u32x4*)(p.tb + n * p.ldt), tv = *(const u32x4*)(p.tb + n * p.ldt + 8);
        float cs[2][4], sn[2][4];
        float q[2][4], k[2][4], v[2][4], kb[2][4], vb[2][4];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int d = hf * 64 + c * 4;
            const long col = (long)h * 128 + d;
            unpack4(*(const u32x2*)(p.cos + (long)pos * 128 + d), cs[hf]);
            unpack4(*(const u32x2*)(p.sin + (long)pos * 128 + d), sn[hf]);
            unpack4(*(const u32x2*)(p.qkv + n * p.ld + col), q[hf]);
            unpack4(*(const u32x2*)(p.qkv + n * p.ld + HD + col), k[hf]);
            unpack4(*(const u32x2*)(p.qkv + n * p.ld + 2 * HD + col), v[hf]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = 0.f, b2 = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) { a = dot2bf(tk[r], wk[hf][e][r], a); b2 = dot2bf(tv[r], wv[hf][e][r], b2); }
                kb[hf][e] = rbf(a);
                vb[hf][e] = rbf(b2);
            }
        }
        float qo[2][4], ko[2][4], kc[2][4], vc[2][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float kx0 = rbf(k[0][e] + kb[0][e]), kx1 = rbf(k[1][e] + kb[1][e]);
            // x*cos + rotate_half(x)*sin ; rotate_half = cat(-x2, x1)
            qo[0][e] = rbf(rbf(q[0][e] * cs[0][e]) + rbf(-q[1][e] * sn[0][e]));
            qo[1][e] = rbf(rbf(q[1][e] * cs[1][e]) + rbf(q[0][e] * sn[1][e]));
            ko[0][e] = rbf(rbf(k[0][e] * cs[0][e]) + rbf(-k[1][e] * sn[0][e]));
            ko[1][e] = rbf(rbf(k[1][e] * cs[1][e]) + rbf(k[0][e] * sn[1][e]));
            kc[0][e] = rbf(rbf(kx0 * cs[0][e]) + rbf(-kx1 * sn[0][e]));
            kc[1][e] = rbf(rbf(kx1 * cs[1][e]) + rbf(kx0 * sn[1][e]));
            vc[0][e] = v[0][e] + vb[0][e];
            vc[1][e] = v[1][e] + vb[1][e];
        }
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const long col = (long)h * 128 + hf * 64 + c * 4;
            *(u32x2*)(p.qkv + n * p.ld + col) = pack4(qo[hf]);
            *(u32x2*)(p.qkv + n * p.ld + HD + col) = pack4(ko[hf]);
            *(u32x2*)(p.k_cross + n * p.ldc + col) = pack4(kc[hf]);
            *(u32x2*)(p.v_cross + n * p.ldc + col) = pack4(vc[hf]);
            if (p.cache_ks) {
                // (a slot outside the sequence's own [0, c_batch / c_row) rows stores nothing: the slot is a device value the host
                //  cannot check when a captured graph is replayed - `index_copy_`, which this replaces, would have asserted)
                const long slot = (long)p.slot[0];
                if (slot >= 0 && (slot + 1) * p.c_row <= p.c_batch) {
                    const long at = n * p.c_batch + slot * p.c_row + col;
                    *(u32x2*)(p.cache_ks + at) = pack4(ko[hf]);
                    *(u32x2*)(p.cache_kc + at) = pack4(kc[hf]);
                    *(u32x2*)(p.cache_vs + at) = pack4(v[hf]);
                    *(u32x2*)(p.cache_vc + at) = pack4(vc[hf]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// SwiGLU: y = silu(g) * u  with g = gu[:, :I], u = gu[:, I:2I]  (LlamaMLP act_fn(gate) * up, modeling_llama.py:199-201;
// both bf16 roundings of the reference kept: bf16(silu(g)) * u -> bf16).
__global__ __launch_bounds__(256) void swiglu_kernel(const bf16_t* __restrict__ g, const bf16_t* __restrict__ u, long ldgu,
                                                     bf16_t* __restrict__ y, long ldy, long rows, int I) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int i8 = I >> 3;
    if (i >= rows * i8) return;
    const long r = i / i8;
    const int c = (int)(i - r * i8) * 8;
    float a[8], b[8], o[8];
    unpack8(*(const u32x4*)(g + r * ldgu + c), a);
    unpack8(*(const u32x4*)(u + r * ldgu + c), b);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = rbf(a[e] / (1.0f + __expf(-a[e]))) * b[e];
    *(u32x4*)(y + r * ldy + c) = pack8(o);
}

// ---------------------------------------------------------------------------------------------------
// Row gather:  out[r, col0 + :] = table[idx[r] - sub]   (embedding lookups; idx int64)
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* __restrict__ table, int D, const long long* __restrict__ idx,
                                                          long long sub, const int* __restrict__ rows_sel, long n,
                                                          bf16_t* __restrict__ out, long ldo, int col0) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int d8 = D >> 3;
    if (i >= n * d8) return;
    const long r = i / d8;
    const int c = (int)(i - r * d8) * 8;
    const long src_row = rows_sel ? rows_sel[r] : r;           // which token this output row stands for
    const long long id = idx[src_row] - sub;
    *(u32x4*)(out + r * ldo + col0 + c) = *(const u32x4*)(table + id * D + c);
}

// copy rows: out[r, col0:col0+D] = in[rows_sel[r], :D]
__global__ __launch_bounds__(256) void copy_rows_kernel(const bf16_t* __restrict__ in, long ldi, int D, const int* __restrict__ rows_sel,
                                                        long n, bf16_t* __restrict__ out, long ldo, int col0) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int d8 = D >> 3;
    if (i >= n * d8) return;
    const long r = i / d8;
    const int c = (int)(i - r * d8) * 8;
    const long sr = rows_sel ? rows_sel[r] : r;
    *(u32x4*)(out + r * ldo + col0 + c) = *(const u32x4*)(in + sr * ldi + c);
}

// ---------------------------------------------------------------------------------------------------
// Cross-entropy row statistics over logits [rows, V] (bf16): loss_r = logsumexp(z_r) - z_r[target_r], fp32,
// 0 and not counted when target_r == ignore (-100).  One wave per row (torch's CrossEntropyLoss on the bf16 logits
// upcasts to fp32 internally the same way).  Accumulates sum(loss) and count with one atomic per block pair —
// kept deterministic by writing per-row losses and reducing on the host side tensor instead.
__global__ __launch_bounds__(256) void ce_rows_kernel(const bf16_t* __restrict__ z, long ldz, int V, const long long* __restrict__ target,
                                                      long long tsub, float* __restrict__ loss_rows, long rows) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long long t = target[row];
    if (t < 0) { if (lane == 0) loss_rows[row] = 0.f; return; }
    const bf16_t* zr = z + row * ldz;
    float mx = -INFINITY;
    for (int c = lane * 8; c < V; c += 512) {
        if (c + 8 <= V) {
            float v[8];
            unpack8(*(const u32x4*)(zr + c), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) mx = fmaxf(mx, v[e]);
        } else {
            for (int e = 0; c + e < V; ++e) mx = fmaxf(mx, bf2f(zr[c + e]));
        }
    }
    mx = wave_max(mx);
    float se = 0.f;
    for (int c = lane * 8; c < V; c += 512) {
        if (c + 8 <= V) {
            float v[8];
            unpack8(*(const u32x4*)(zr + c), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) se += __expf(v[e] - mx);
        } else {
            for (int e = 0; c + e < V; ++e) se += __expf(bf2f(zr[c + e]) - mx);
        }
    }
    se = wave_sum(se);
    if (lane == 0) {
        const long long tl = t - tsub;
        const float zt = (tl >= 0 && tl < V) ? bf2f(zr[tl]) : -INFINITY;     // label outside this head's range -> +inf loss, as in the reference
        loss_rows[row] = (mx + __logf(se)) - zt;
    }
}

static inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
static inline int launched() { return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH; }

// Append one new token per sequence to the four K/V caches of a layer in ONE launch (the decode step made four index_copy
// launches per layer for it).  src_x [B, W] (row strides ld_x), cache_x [B, Lmax, W] (row / batch strides), slot = the cache
// position, read from device memory so that the step stays free of host-side indices (hipGraph capture).
struct KvAppendArgs {
    const bf16_t* src[4]; long ld[4];
    bf16_t* dst[4];
    long row_stride, batch_stride;
    const long long* slot;
    int B, W;
};
__global__ __launch_bounds__(256) void kv_cache_append_kernel(const KvAppendArgs p) {
    const int b = blockIdx.x, which = blockIdx.y;
    const long long s = p.slot[0];
    const bf16_t* src = p.src[0]; long ld = p.ld[0]; bf16_t* dst = p.dst[0];
    if (which == 1) { src = p.src[1]; ld = p.ld[1]; dst = p.dst[1]; }
    else if (which == 2) { src = p.src[2]; ld = p.ld[2]; dst = p.dst[2]; }
    else if (which == 3) { src = p.src[3]; ld = p.ld[3]; dst = p.dst[3]; }
    if (s < 0 || (s + 1) * p.row_stride > p.batch_stride) return;      // outside the sequence's rows: nothing is stored (see above)
    const bf16_t* sr = src + (long)b * ld;
    bf16_t* dr = dst + (long)b * p.batch_stride + s * p.row_stride;
    for (int c = threadIdx.x * 8; c < p.W; c += 256 * 8) *(u32x4*)(dr + c) = *(const u32x4*)(sr + c);
}

}  // namespace libra

using namespace libra;

extern "C" int libra_rmsnorm_routed_fwd(const void* x, int64_t ldx, const void* w_lang, const void* w_vis,
                                        const uint8_t* flag, void* y, int64_t ldy, float* rstd, int64_t rows, int64_t D,
                                        float eps, void* stream) {
    if (rows <= 0) return LIBRA_OK;
    if (D <= 0 || (D % 8) || D > 8192 || ldx < D || ldy < D || (ldx % 8) || (ldy % 8)) return LIBRA_ERR_SHAPE;
    if (!x || !w_lang || !y || (flag && !w_vis) || !al16(x) || !al16(y) || !al16(w_lang) || (w_vis && !al16(w_vis)))
        return LIBRA_ERR_ALIGN;
    const unsigned grid = (unsigned)((rows + 3) / 4);
    const int nc = (int)((D / 8 + 63) / 64);
#define LAUNCH_RMS(NC)                                                                                              \
    hipLaunchKernelGGL((rmsnorm_routed_kernel<NC>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, \
                       (long)ldx, (const bf16_t*)w_lang, (const bf16_t*)w_vis, flag, (bf16_t*)y, (long)ldy, rstd,    \
                       (long)rows, (int)D, eps)
    if (nc <= 1) LAUNCH_RMS(1); else if (nc <= 2) LAUNCH_RMS(2); else if (nc <= 4) LAUNCH_RMS(4);
    else if (nc <= 8) LAUNCH_RMS(8); else LAUNCH_RMS(16);
#undef LAUNCH_RMS
    return launched();
}

static int rope_bridge_run(void* qkv, int64_t ld, const void* tb, int64_t ldt, const void* bk_l, const void* bk_v,
                           const void* bv_l, const void* bv_v, const uint8_t* flag, const void* cos, const void* sin,
                           int64_t max_pos, void* k_cross, void* v_cross, int64_t ldc, int64_t N, int64_t S,
                           int64_t H, const int* positions, int64_t pos_stride, void* stream, void* const* cache = nullptr,
                           int64_t c_row = 0, int64_t c_batch = 0, const int64_t* slot = nullptr) {
    if (N <= 0) return LIBRA_OK;
    if (cache) {
        if (!slot || c_row < H * 128 || c_batch < c_row || (c_row % 8) || (c_batch % 8)) return LIBRA_ERR_SHAPE;
        for (int i = 0; i < 4; ++i) if (!cache[i] || !al16(cache[i])) return LIBRA_ERR_ALIGN;
    }
    if (positions && pos_stride != 1 && pos_stride != 2) return LIBRA_ERR_SHAPE;
    if (H <= 0 || S <= 0 || S > max_pos || ld < 3 * H * 128 || ldt < 16 || ldc < H * 128) return LIBRA_ERR_SHAPE;
    if ((ld % 8) || (ldt % 8) || (ldc % 8)) return LIBRA_ERR_ALIGN;
    if (!qkv || !tb || !bk_l || !bk_v || !bv_l || !bv_v || !flag || !cos || !sin || !k_cross || !v_cross) return LIBRA_ERR_ALIGN;
    if (!al16(qkv) || !al16(tb) || !al16(bk_l) || !al16(bk_v) || !al16(bv_l) || !al16(bv_v) || !al16(cos) || !al16(sin) ||
        !al16(k_cross) || !al16(v_cross)) return LIBRA_ERR_ALIGN;
    RopeArgs a;
    a.qkv = (bf16_t*)qkv; a.ld = ld; a.H = (int)H; a.tb = (const bf16_t*)tb; a.ldt = ldt;
    a.bk_l = (const bf16_t*)bk_l; a.bk_v = (const bf16_t*)bk_v; a.bv_l = (const bf16_t*)bv_l; a.bv_v = (const bf16_t*)bv_v;
    a.flag = flag; a.cos = (const bf16_t*)cos; a.sin = (const bf16_t*)sin;
    a.k_cross = (bf16_t*)k_cross; a.v_cross = (bf16_t*)v_cross; a.ldc = ldc; a.N = N; a.S = (int)S;
    a.positions = positions; a.pos_stride = (int)pos_stride; a.max_pos = (int)max_pos;
    a.cache_ks = cache ? (bf16_t*)cache[0] : nullptr; a.cache_kc = cache ? (bf16_t*)cache[1] : nullptr;
    a.cache_vs = cache ? (bf16_t*)cache[2] : nullptr; a.cache_vc = cache ? (bf16_t*)cache[3] : nullptr;
    a.c_row = c_row; a.c_batch = c_batch; a.slot = (const long long*)slot;
    const long LT = H * 16, tpb = LT >= 256 ? 1 : 256 / LT;
    // tokens per thread: ROPE_TOK amortises the bridge weights' registers over a run of tokens; with few rows (a generation step:
    // one token per sequence) that serialised them - 8 dependent round trips in 2 workgroups, 14 us for 8 rows - so small problems
    // take one token per thread and spread over the chip instead
    const int tok = N >= 4096 ? ROPE_TOK : 1;
    a.tok = tok;
    const long gx = (N + tok * tpb - 1) / (tok * tpb), gy = LT >= 256 ? (LT + 255) / 256 : 1;
    if (gx > 0x7fffffffL || gy > 65535) return LIBRA_ERR_SHAPE;
    hipLaunchKernelGGL(rope_bridge_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, a);
    return launched();
}

extern "C" int libra_rope_bridge(void* qkv, int64_t ld, const void* tb, int64_t ldt, const void* bk_l, const void* bk_v,
                                 const void* bv_l, const void* bv_v, const uint8_t* flag, const void* cos, const void* sin,
                                 int64_t max_pos, void* k_cross, void* v_cross, int64_t ldc, int64_t N, int64_t S,
                                 int64_t H, void* stream) {
    return rope_bridge_run(qkv, ld, tb, ldt, bk_l, bk_v, bv_l, bv_v, flag, cos, sin, max_pos, k_cross, v_cross, ldc, N, S, H,
                           nullptr, 1, stream);
}

extern "C" int libra_rope_bridge_pos(void* qkv, int64_t ld, const void* tb, int64_t ldt, const void* bk_l, const void* bk_v,
                                     const void* bv_l, const void* bv_v, const uint8_t* flag, const void* cos, const void* sin,
                                     int64_t max_pos, void* k_cross, void* v_cross, int64_t ldc, int64_t N,
                                     const int* positions, int64_t pos_stride, int64_t H, void* stream) {
    if (!positions) return LIBRA_ERR_ALIGN;
    return rope_bridge_run(qkv, ld, tb, ldt, bk_l, bk_v, bv_l, bv_v, flag, cos, sin, max_pos, k_cross, v_cross, ldc, N, max_pos,
                           H, positions, pos_stride, stream);
}

extern "C" int libra_rope_bridge_pos_append(void* qkv, int64_t ld, const void* tb, int64_t ldt, const void* bk_l, const void* bk_v,
                                            const void* bv_l, const void* bv_v, const uint8_t* flag, const void* cos, const void* sin,
                                            int64_t max_pos, void* k_cross, void* v_cross, int64_t ldc, int64_t B,
                                            const int* positions, int64_t pos_stride, int64_t H, void* cache_k_same,
                                            void* cache_k_cross, void* cache_v_same, void* cache_v_cross, int64_t row_stride,
                                            int64_t batch_stride, const int64_t* slot, void* stream) {
    if (!positions) return LIBRA_ERR_ALIGN;
    void* const cache[4] = {cache_k_same, cache_k_cross, cache_v_same, cache_v_cross};
    return rope_bridge_run(qkv, ld, tb, ldt, bk_l, bk_v, bv_l, bv_v, flag, cos, sin, max_pos, k_cross, v_cross, ldc, B, max_pos, H,
                           positions, pos_stride, stream, cache, row_stride, batch_stride, slot);
}

extern "C" int libra_swiglu(const void* gate, const void* up, int64_t ldgu, void* y, int64_t ldy, int64_t rows,
                            int64_t I, void* stream) {
    if (rows <= 0) return LIBRA_OK;
    if (I <= 0 || (I % 8) || ldgu < I || ldy < I || (ldgu % 8) || (ldy % 8)) return LIBRA_ERR_SHAPE;
    if (!gate || !up || !y || !al16(gate) || !al16(up) || !al16(y)) return LIBRA_ERR_ALIGN;
    const long total = rows * (I / 8);
    hipLaunchKernelGGL(swiglu_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)gate, (const bf16_t*)up, (long)ldgu, (bf16_t*)y, (long)ldy, (long)rows, (int)I);
    return launched();
}

extern "C" int libra_gather_rows(const void* table, int64_t D, const int64_t* idx, int64_t sub, const int32_t* rows_sel,
                                 int64_t n, void* out, int64_t ldo, int64_t col0, void* stream) {
    if (n <= 0) return LIBRA_OK;
    if (D <= 0 || (D % 8) || (ldo % 8) || (col0 % 8) || ldo < col0 + D) return LIBRA_ERR_SHAPE;
    if (!table || !idx || !out || !al16(table) || !al16(out)) return LIBRA_ERR_ALIGN;
    const long total = n * (D / 8);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)table, (int)D, (const long long*)idx, (long long)sub, rows_sel, (long)n,
                       (bf16_t*)out, (long)ldo, (int)col0);
    return launched();
}

extern "C" int libra_copy_rows(const void* in, int64_t ldi, int64_t D, const int32_t* rows_sel, int64_t n, void* out,
                               int64_t ldo, int64_t col0, void* stream) {
    if (n <= 0) return LIBRA_OK;
    if (D <= 0 || (D % 8) || (ldi % 8) || (ldo % 8) || (col0 % 8) || ldo < col0 + D || ldi < D) return LIBRA_ERR_SHAPE;
    if (!in || !out || !al16(in) || !al16(out)) return LIBRA_ERR_ALIGN;
    const long total = n * (D / 8);
    hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)in, (long)ldi, (int)D, rows_sel, (long)n, (bf16_t*)out, (long)ldo, (int)col0);
    return launched();
}

extern "C" int libra_ce_rows(const void* logits, int64_t ldz, int64_t V, const int64_t* target, int64_t target_sub,
                             float* loss_rows, int64_t rows, void* stream) {
    if (rows <= 0) return LIBRA_OK;
    if (V <= 0 || ldz < V || (ldz % 8)) return LIBRA_ERR_SHAPE;
    if (!logits || !target || !loss_rows || !al16(logits)) return LIBRA_ERR_ALIGN;
    hipLaunchKernelGGL(ce_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)logits, (long)ldz, (int)V, (const long long*)target, (long long)target_sub, loss_rows,
                       (long)rows);
    return launched();
}

extern "C" int libra_kv_cache_append(const void* k_same, int64_t ld_ks, const void* k_cross, int64_t ld_kc, const void* v_same,
                                     int64_t ld_vs, const void* v_cross, int64_t ld_vc, void* cache_k_same, void* cache_k_cross,
                                     void* cache_v_same, void* cache_v_cross, int64_t row_stride, int64_t batch_stride,
                                     const int64_t* slot, int64_t B, int64_t W, void* stream) {
    if (B <= 0) return LIBRA_OK;
    if (W <= 0 || (W % 8) || ld_ks < W || ld_kc < W || ld_vs < W || ld_vc < W || row_stride < W || batch_stride < row_stride ||
        (ld_ks % 8) || (ld_kc % 8) || (ld_vs % 8) || (ld_vc % 8) || (row_stride % 8) || (batch_stride % 8) || B > 65535)
        return LIBRA_ERR_SHAPE;
    if (!k_same || !k_cross || !v_same || !v_cross || !cache_k_same || !cache_k_cross || !cache_v_same || !cache_v_cross || !slot)
        return LIBRA_ERR_ALIGN;
    if (!al16(k_same) || !al16(k_cross) || !al16(v_same) || !al16(v_cross) || !al16(cache_k_same) || !al16(cache_k_cross) ||
        !al16(cache_v_same) || !al16(cache_v_cross))
        return LIBRA_ERR_ALIGN;
    KvAppendArgs a;
    a.src[0] = (const bf16_t*)k_same; a.src[1] = (const bf16_t*)k_cross; a.src[2] = (const bf16_t*)v_same; a.src[3] = (const bf16_t*)v_cross;
    a.ld[0] = ld_ks; a.ld[1] = ld_kc; a.ld[2] = ld_vs; a.ld[3] = ld_vc;
    a.dst[0] = (bf16_t*)cache_k_same; a.dst[1] = (bf16_t*)cache_k_cross; a.dst[2] = (bf16_t*)cache_v_same; a.dst[3] = (bf16_t*)cache_v_cross;
    a.row_stride = row_stride; a.batch_stride = batch_stride; a.slot = (const long long*)slot; a.B = (int)B; a.W = (int)W;
    hipLaunchKernelGGL(kv_cache_append_kernel, dim3((unsigned)B, 4), dim3(256), 0, (hipStream_t)stream, a);
    return launched();
}
