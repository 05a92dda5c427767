// LayerNorm forward/backward and the ViT embedding-assemble + pre-LN kernel (gfx950).
//
// All of these are HBM-bound row kernels: one wave64 per row, the whole row lives in registers
// (16-byte bf16x8 loads, D <= 8192), statistics in fp32 with wave shuffles — no LDS, no re-reads.
// Algorithmic bytes per row: fwd 2*D*2 (+8 stats); bwd 4*D*2 (dy, x, [dres], dx).
#include "hip_common.hpp"
#include "../../include/libra_hip.h"

namespace libra {

constexpr int LN_MAXC = 16;   // chunks of 8 per lane: D <= 64*8*16 = 8192
constexpr int ROWS_PER_BLOCK = 4;

template <int NC>
__device__ __forceinline__ void load_row(const bf16_t* __restrict__ p, int nchunks, int lane, float (*v)[8]) {
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunks) unpack8(*(const u32x4*)(p + c * 8), v[i]);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
        }
    }
}

template <int NC>
__device__ __forceinline__ void row_stats(float (*v)[8], int nchunks, int lane, int D, float eps, float& mean,
                                          float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[i][e];
    mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        if (lane + 64 * i < nchunks) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
        }
    }
    rstd = rsqrtf(wave_sum(q) / (float)D + eps);
}

template <int NC>
__global__ __launch_bounds__(64 * ROWS_PER_BLOCK) void layernorm_fwd_kernel(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta,
    bf16_t* __restrict__ y, float* __restrict__ mean_o, float* __restrict__ rstd_o, long rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunks = D >> 3;
    float v[NC][8];
    load_row<NC>(x + row * D, nchunks, lane, v);
    float mean, rstd;
    row_stats<NC>(v, nchunks, lane, D, eps, mean, rstd);
    if (lane == 0) {
        if (mean_o) mean_o[row] = mean;
        if (rstd_o) rstd_o[row] = rstd;
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunks) {
            float g[8], b[8], o[8];
            unpack8(*(const u32x4*)(gamma + c * 8), g);
            unpack8(*(const u32x4*)(beta + c * 8), b);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
            *(u32x4*)(y + row * D + c * 8) = pack8(o);
        }
    }
}

// dx = rstd * (g*dy - mean_D(g*dy) - xhat * mean_D(g*dy*xhat)) [+ dres];  dgamma += dy*xhat; dbeta += dy; optionally
// dxsum += dx (the column sum of the OUTPUT = the bias gradient of the Linear whose dY this dx is: it saves the separate
// column-sum pass over dx).  Each block walks a strip of rows so the parameter-gradient partials stay in registers; the
// next row's operands are fetched while the current row is reduced (one row at a time was latency bound: 2.9 TB/s).
template <int NC, bool DXSUM>
__global__ __launch_bounds__(64 * ROWS_PER_BLOCK) void layernorm_bwd_kernel(
    const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const bf16_t* __restrict__ gamma,
    const float* __restrict__ mean_i, const float* __restrict__ rstd_i, const bf16_t* __restrict__ dres,
    bf16_t* __restrict__ dx, float* __restrict__ part, long rows, int D, int rows_per_block) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nchunks = D >> 3;
    constexpr int NP = DXSUM ? 3 : 2;
    float g[NC][8];
    load_row<NC>(gamma, nchunks, lane, g);
    float ag[NC][8], ab[NC][8], ad[DXSUM ? NC : 1][8];
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) { ag[i][e] = 0.f; ab[i][e] = 0.f; if (DXSUM) ad[i][e] = 0.f; }

    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(rows, r0 + rows_per_block);
    struct RowIn { u32x4 y[NC], x[NC], r[NC]; float mean, rstd; };
    auto fetch = [&](long row, RowIn& in) {
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            const bool ok = c < nchunks;
            in.y[i] = ok ? *(const u32x4*)(dy + row * D + c * 8) : u32x4{0u, 0u, 0u, 0u};
            in.x[i] = ok ? *(const u32x4*)(x + row * D + c * 8) : u32x4{0u, 0u, 0u, 0u};
            in.r[i] = (ok && dres) ? *(const u32x4*)(dres + row * D + c * 8) : u32x4{0u, 0u, 0u, 0u};
        }
        in.mean = mean_i[row]; in.rstd = rstd_i[row];
    };
    RowIn cur, nxt;
    long row = r0 + wave;
    if (row < r1) fetch(row, cur);
    for (; row < r1; row += ROWS_PER_BLOCK) {
        const bool more = row + ROWS_PER_BLOCK < r1;
        if (more) fetch(row + ROWS_PER_BLOCK, nxt);
        float vy[NC][8], vx[NC][8];
#pragma unroll
        for (int i = 0; i < NC; ++i) { unpack8(cur.y[i], vy[i]); unpack8(cur.x[i], vx[i]); }
        const float mean = cur.mean, rstd = cur.rstd;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = (vx[i][e] - mean) * rstd;     // padded chunks: dy = 0 so they add nothing
                const float gd = g[i][e] * vy[i][e];
                s1 += gd; s2 += gd * xh;
                ag[i][e] += vy[i][e] * xh; ab[i][e] += vy[i][e];
                vx[i][e] = xh;
            }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunks) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rstd * (g[i][e] * vy[i][e] - s1 - vx[i][e] * s2);
                if (dres) {
                    float r[8];
                    unpack8(cur.r[i], r);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += r[e];
                }
                const u32x4 ob = pack8(o);
                *(u32x4*)(dx + row * D + c * 8) = ob;
                if (DXSUM) {                                   // sum of the values as stored (bf16), like a column sum of dx
                    float q[8];
                    unpack8(ob, q);
#pragma unroll
                    for (int e = 0; e < 8; ++e) ad[i][e] += q[e];
                }
            }
        }
        if (more) cur = nxt;
    }
    if (part) {
        // Deterministic parameter-gradient partials, no atomics.  For D <= 2048 the four waves of the workgroup are first
        // summed through LDS in wave order (one wave's rows at a time: NP * D floats), so ONE partial row set per workgroup
        // reaches HBM - a quarter of the bytes written here and read again by ln_param_reduce_kernel; wider rows keep one
        // set per wave (the LDS slab would not fit).
        constexpr bool WG_SUM = NC <= 4;
        if constexpr (WG_SUM) {
            __shared__ __attribute__((aligned(16))) float comb[NP * NC * 512];
            for (int w = 1; w < ROWS_PER_BLOCK; ++w) {
                if (wave == w) {
#pragma unroll
                    for (int i = 0; i < NC; ++i) {
                        float* q = comb + (lane + 64 * i) * 8;
                        *(f32x4*)(q) = f32x4{ag[i][0], ag[i][1], ag[i][2], ag[i][3]};
                        *(f32x4*)(q + 4) = f32x4{ag[i][4], ag[i][5], ag[i][6], ag[i][7]};
                        *(f32x4*)(q + NC * 512) = f32x4{ab[i][0], ab[i][1], ab[i][2], ab[i][3]};
                        *(f32x4*)(q + NC * 512 + 4) = f32x4{ab[i][4], ab[i][5], ab[i][6], ab[i][7]};
                        if (DXSUM) {
                            *(f32x4*)(q + 2 * NC * 512) = f32x4{ad[i][0], ad[i][1], ad[i][2], ad[i][3]};
                            *(f32x4*)(q + 2 * NC * 512 + 4) = f32x4{ad[i][4], ad[i][5], ad[i][6], ad[i][7]};
                        }
                    }
                }
                __syncthreads();
                if (wave == 0) {
#pragma unroll
                    for (int i = 0; i < NC; ++i) {
                        const float* q = comb + (lane + 64 * i) * 8;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            ag[i][e] += q[e]; ab[i][e] += q[NC * 512 + e];
                            if (DXSUM) ad[i][e] += q[2 * NC * 512 + e];
                        }
                    }
                }
                __syncthreads();
            }
            if (wave != 0) return;
        }
        float* pg = part + ((long)(WG_SUM ? blockIdx.x : blockIdx.x * ROWS_PER_BLOCK + wave) * NP) * D;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunks) {
                *(f32x4*)(pg + c * 8) = f32x4{ag[i][0], ag[i][1], ag[i][2], ag[i][3]};
                *(f32x4*)(pg + c * 8 + 4) = f32x4{ag[i][4], ag[i][5], ag[i][6], ag[i][7]};
                *(f32x4*)(pg + D + c * 8) = f32x4{ab[i][0], ab[i][1], ab[i][2], ab[i][3]};
                *(f32x4*)(pg + D + c * 8 + 4) = f32x4{ab[i][4], ab[i][5], ab[i][6], ab[i][7]};
                if (DXSUM) {
                    *(f32x4*)(pg + 2 * D + c * 8) = f32x4{ad[i][0], ad[i][1], ad[i][2], ad[i][3]};
                    *(f32x4*)(pg + 2 * D + c * 8 + 4) = f32x4{ad[i][4], ad[i][5], ad[i][6], ad[i][7]};
                }
            }
        }
    }
}

// out_k[d] += sum_p part[p][k][d] for the np (2 or 3) accumulators of the backward.  32 columns x 32 row groups per block.
__global__ __launch_bounds__(1024) void ln_param_reduce_kernel(const float* __restrict__ part, int nparts, int np, int D,
                                                               float* __restrict__ out_gamma, float* __restrict__ out_beta,
                                                               float* __restrict__ out_dxsum) {
    __shared__ float red[32][33];
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int which = blockIdx.y;
    const int col = blockIdx.x * 32 + c;
    float s = 0.f;
    if (col < D) {
        for (int p = rg; p < nparts; p += 32) s += part[((long)p * np + which) * D + col];
    }
    red[rg][c] = s;
    __syncthreads();
    if (rg == 0 && col < D) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) t += red[i][c];
        float* o = which == 0 ? out_gamma : which == 1 ? out_beta : out_dxsum;
        o[col] += t;
    }
}

// emb[b,t] = (t == 0 ? cls : patches[b*(T-1) + t-1]) + pos[t]  (rounded to bf16, as the reference's
// bf16 add does), hs0 = LN(emb).
template <int NC>
__global__ __launch_bounds__(64 * ROWS_PER_BLOCK) void vit_embed_ln_kernel(
    const bf16_t* __restrict__ patches, const bf16_t* __restrict__ cls, const bf16_t* __restrict__ pos,
    const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta, bf16_t* __restrict__ emb,
    bf16_t* __restrict__ hs0, float* __restrict__ mean_o, float* __restrict__ rstd_o, long rows, int T, int D,
    float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunks = D >> 3;
    const long b = row / T;
    const int t = (int)(row - b * T);
    const bf16_t* src = (t == 0) ? cls : patches + (b * (T - 1) + (t - 1)) * (long)D;
    float v[NC][8], pv[NC][8];
    load_row<NC>(src, nchunks, lane, v);
    load_row<NC>(pos + (long)t * D, nchunks, lane, pv);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] = bf2f(f2bf(v[i][e] + pv[i][e]));
        if (c < nchunks && emb) *(u32x4*)(emb + row * D + c * 8) = pack8(v[i]);
    }
    float mean, rstd;
    row_stats<NC>(v, nchunks, lane, D, eps, mean, rstd);
    if (lane == 0) {
        if (mean_o) mean_o[row] = mean;
        if (rstd_o) rstd_o[row] = rstd;
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunks) {
            float g[8], bb[8], o[8];
            unpack8(*(const u32x4*)(gamma + c * 8), g);
            unpack8(*(const u32x4*)(beta + c * 8), bb);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + bb[e];
            *(u32x4*)(hs0 + row * D + c * 8) = pack8(o);
        }
    }
}

template <typename F>
static int dispatch_nc(int D, F&& f) {
    const int nc = (D / 8 + 63) / 64;
    if (nc <= 1) return f(std::integral_constant<int, 1>());
    if (nc <= 2) return f(std::integral_constant<int, 2>());
    if (nc <= 4) return f(std::integral_constant<int, 4>());
    if (nc <= 8) return f(std::integral_constant<int, 8>());
    return f(std::integral_constant<int, 16>());
}

static inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace libra

using namespace libra;

extern "C" int libra_layernorm_fwd(const void* x, const void* gamma, const void* beta, void* y, float* mean,
                                   float* rstd, int64_t rows, int64_t D, float eps, void* stream) {
    if (rows <= 0) return LIBRA_OK;
    if (D <= 0 || (D % 8) || D > 64 * 8 * LN_MAXC) return LIBRA_ERR_SHAPE;
    if (!x || !gamma || !beta || !y || !al16(x) || !al16(gamma) || !al16(beta) || !al16(y)) return LIBRA_ERR_ALIGN;
    const unsigned grid = (unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
    return dispatch_nc((int)D, [&](auto nc) {
        hipLaunchKernelGGL((layernorm_fwd_kernel<decltype(nc)::value>), dim3(grid), dim3(64 * ROWS_PER_BLOCK), 0,
                           (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)gamma, (const bf16_t*)beta,
                           (bf16_t*)y, mean, rstd, (long)rows, (int)D, eps);
        return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
    });
}

static long ln_bwd_rows_per_block(long rows) {
    // ~2 workgroups per CU worth of row strips
    long rpb = (rows + 511) / 512;
    return ((rpb + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK) * ROWS_PER_BLOCK;
}

extern "C" size_t libra_layernorm_bwd_workspace_bytes(int64_t rows, int64_t D) {
    if (rows <= 0 || D <= 0) return 0;
    const long rpb = ln_bwd_rows_per_block(rows);
    const long grid = (rows + rpb - 1) / rpb;
    return (size_t)grid * ROWS_PER_BLOCK * 3 * D * sizeof(float);                // room for the optional dx column sum
}

extern "C" int libra_layernorm_bwd(const void* dy, const void* x, const void* gamma, const float* mean,
                                   const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta, float* dxsum,
                                   void* workspace, size_t workspace_bytes, int64_t rows, int64_t D, void* stream) {
    if (rows <= 0) return LIBRA_OK;
    if (D <= 0 || (D % 8) || D > 4096) return LIBRA_ERR_SHAPE;     // the whole row + partials stay in registers
    if (!dy || !x || !gamma || !mean || !rstd || !dx) return LIBRA_ERR_ALIGN;
    if (!al16(dy) || !al16(x) || !al16(gamma) || !al16(dx) || (dres && !al16(dres))) return LIBRA_ERR_ALIGN;
    if ((dgamma == nullptr) != (dbeta == nullptr) || (dxsum && !dgamma)) return LIBRA_ERR_ALIGN;
    const long rpb = ln_bwd_rows_per_block(rows);
    const unsigned grid = (unsigned)((rows + rpb - 1) / rpb);
    float* part = nullptr;
    if (dgamma) {
        if (!workspace || (((uintptr_t)workspace) & 15) || workspace_bytes < libra_layernorm_bwd_workspace_bytes(rows, D))
            return LIBRA_ERR_ALIGN;
        part = (float*)workspace;
    }
    const int rc = dispatch_nc((int)D, [&](auto nc) {
        constexpr int NCV = decltype(nc)::value;
        if (dxsum)
            hipLaunchKernelGGL((layernorm_bwd_kernel<NCV, true>), dim3(grid), dim3(64 * ROWS_PER_BLOCK), 0, (hipStream_t)stream,
                               (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)gamma, mean, rstd, (const bf16_t*)dres,
                               (bf16_t*)dx, part, (long)rows, (int)D, (int)rpb);
        else
            hipLaunchKernelGGL((layernorm_bwd_kernel<NCV, false>), dim3(grid), dim3(64 * ROWS_PER_BLOCK), 0, (hipStream_t)stream,
                               (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)gamma, mean, rstd, (const bf16_t*)dres,
                               (bf16_t*)dx, part, (long)rows, (int)D, (int)rpb);
        return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
    });
    if (rc != LIBRA_OK || !dgamma) return rc;
    const int np = dxsum ? 3 : 2;
    const int nparts = (int)(D <= 2048 ? grid : grid * ROWS_PER_BLOCK);       // (rows of D <= 2048 are summed per workgroup first)
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((unsigned)((D + 31) / 32), (unsigned)np), dim3(1024), 0, (hipStream_t)stream,
                       part, nparts, np, (int)D, dgamma, dbeta, dxsum);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}

extern "C" int libra_vit_embed_ln(const void* patches, const void* cls, const void* pos, const void* gamma,
                                  const void* beta, void* emb, void* hs0, float* mean, float* rstd, int64_t B,
                                  int64_t T, int64_t D, float eps, void* stream) {
    const long rows = B * T;
    if (rows <= 0) return LIBRA_OK;
    if (T < 2 || D <= 0 || (D % 8) || D > 64 * 8 * LN_MAXC) return LIBRA_ERR_SHAPE;
    if (!patches || !cls || !pos || !gamma || !beta || !hs0) return LIBRA_ERR_ALIGN;
    if (!al16(patches) || !al16(cls) || !al16(pos) || !al16(gamma) || !al16(beta) || !al16(hs0) || (emb && !al16(emb)))
        return LIBRA_ERR_ALIGN;
    const unsigned grid = (unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
    return dispatch_nc((int)D, [&](auto nc) {
        hipLaunchKernelGGL((vit_embed_ln_kernel<decltype(nc)::value>), dim3(grid), dim3(64 * ROWS_PER_BLOCK), 0,
                           (hipStream_t)stream, (const bf16_t*)patches, (const bf16_t*)cls, (const bf16_t*)pos,
                           (const bf16_t*)gamma, (const bf16_t*)beta, (bf16_t*)emb, (bf16_t*)hs0, mean, rstd, rows,
                           (int)T, (int)D, eps);
        return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
    });
}
