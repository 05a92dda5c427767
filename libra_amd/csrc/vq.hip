// LFQ sign-quantise + bit-pack + token-id framing (gfx950).  HBM-bound: reads h [rows,E] once.
//
// One wave64 per row (grid-strided), the project_in weight [Q*9, E] staged once per workgroup in LDS.
//   x_j   = bf16( sum_e h[e] * w_in[j,e] + b_in[j] )      fp32 accumulate, one rounding — the rounding
//                                                         points of the reference's bf16 F.linear
//   bit_j = x_j > 0 ;  index_q = sum_{j<9} bit_{9q+j} << (8-j)         (MSB first)
//   ids[q,b,1+p] = offset + index_q ; ids[q,b,0] = BOI ; ids[q,b,hw+1] = EOI
// Integer outputs are exact functions of the bits; the bits are exact functions of x's sign.
#include <atomic>
#include "hip_common.hpp"
#include "../../include/libra_hip.h"

namespace libra {

constexpr int LFQ_MAX_CD = 36;     // Q*9, Q <= 4

struct LfqArgs {
    const bf16_t* h; const bf16_t* w_in; const bf16_t* b_in; const bf16_t* w_out; const bf16_t* b_out;
    long long* indices; long long* ids; bf16_t* xpre; bf16_t* quant;
    long rows, ld_h; int B, hw, E, Q, CD;
    long long offset, boi, eoi;
    int has_proj;
};

__global__ __launch_bounds__(256) void lfq_encode_kernel(const LfqArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* sw = (bf16_t*)smem;                       // [CD][E] project_in weight
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int E = p.E, CD = p.CD;
    if (p.has_proj) {
        const int n8 = CD * E / 8;
        for (int i = tid; i < n8; i += 256) *(u32x4*)(sw + i * 8) = *(const u32x4*)(p.w_in + i * 8);
    }
    __syncthreads();
    const int nch = E >> 3;
    for (long row = (long)blockIdx.x * 4 + wave; row < p.rows; row += (long)gridDim.x * 4) {
        float x[LFQ_MAX_CD];
        if (p.has_proj) {
#pragma unroll
            for (int j = 0; j < LFQ_MAX_CD; ++j) x[j] = 0.f;
            for (int c = lane; c < nch; c += 64) {
                float hv[8];
                unpack8(*(const u32x4*)(p.h + row * p.ld_h + c * 8), hv);
#pragma unroll
                for (int j = 0; j < LFQ_MAX_CD; ++j) {
                    if (j < CD) {
                        float wv[8];
                        unpack8(*(const u32x4*)(sw + j * E + c * 8), wv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) x[j] = fmaf(hv[e], wv[e], x[j]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < LFQ_MAX_CD; ++j) {
                if (j < CD) {
                    const float s = wave_sum(x[j]) + bf2f(p.b_in[j]);
                    x[j] = bf2f(f2bf(s));
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < LFQ_MAX_CD; ++j) x[j] = (j < CD) ? bf2f(p.h[row * p.ld_h + j]) : 0.f;
        }
        // every lane now holds all CD values
        unsigned long long bits = 0ull;
#pragma unroll
        for (int j = 0; j < LFQ_MAX_CD; ++j)
            if (j < CD && x[j] > 0.f) bits |= (1ull << j);
        const long b = row / p.hw;
        const int pp = (int)(row - b * p.hw);
        if (lane < p.Q) {
            const int q = lane;
            long long idx = 0;
#pragma unroll
            for (int j = 0; j < 9; ++j) idx |= (long long)((bits >> (9 * q + j)) & 1ull) << (8 - j);
            if (p.indices) p.indices[row * p.Q + q] = idx;
            if (p.ids) {
                long long* dst = p.ids + ((long)q * p.B + b) * (p.hw + 2);
                dst[1 + pp] = p.offset + idx;
                if (pp == 0) dst[0] = p.boi;
                if (pp == p.hw - 1) dst[p.hw + 1] = p.eoi;
            }
        }
        if (p.xpre && lane < CD) {
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < LFQ_MAX_CD; ++j) v = (j == lane) ? x[j] : v;
            p.xpre[row * CD + lane] = f2bf(v);
        }
        if (p.quant) {
            if (p.has_proj) {
                for (int e = lane; e < E; e += 64) {
                    float acc = 0.f;
                    for (int j = 0; j < CD; ++j) {
                        const float w = bf2f(p.w_out[(long)e * CD + j]);
                        acc += ((bits >> j) & 1ull) ? w : -w;
                    }
                    p.quant[row * E + e] = f2bf(acc + bf2f(p.b_out[e]));
                }
            } else {
                if (lane < CD) p.quant[row * E + lane] = ((bits >> lane) & 1ull) ? (bf16_t)0x3f80 : (bf16_t)0xbf80;
            }
        }
    }
}

}  // namespace libra

using namespace libra;

extern "C" int libra_lfq_encode(const void* h, int64_t ld_h, const void* w_in, const void* b_in, const void* w_out,
                                const void* b_out, int64_t* indices, int64_t* ids, void* xpre, void* quant,
                                int64_t B, int64_t hw, int64_t E, int64_t Q, int64_t offset, int64_t boi, int64_t eoi,
                                void* stream) {
    const long rows = B * hw;
    if (rows <= 0) return LIBRA_OK;
    const int64_t CD = Q * 9;
    if (Q < 1 || CD > LFQ_MAX_CD || E < CD || ld_h < E) return LIBRA_ERR_SHAPE;
    const int has_proj = (E != CD);
    if (has_proj && ((E % 8) || (ld_h % 8) || CD * E * 2 > 64 * 1024)) return LIBRA_ERR_SHAPE;
    if (!h || (((uintptr_t)h) & 15)) return LIBRA_ERR_ALIGN;
    if (has_proj && (!w_in || !b_in || (((uintptr_t)w_in) & 15))) return LIBRA_ERR_ALIGN;
    if (quant && has_proj && (!w_out || !b_out)) return LIBRA_ERR_ALIGN;
    LfqArgs a;
    a.h = (const bf16_t*)h; a.w_in = (const bf16_t*)w_in; a.b_in = (const bf16_t*)b_in;
    a.w_out = (const bf16_t*)w_out; a.b_out = (const bf16_t*)b_out;
    a.indices = (long long*)indices; a.ids = (long long*)ids; a.xpre = (bf16_t*)xpre; a.quant = (bf16_t*)quant;
    a.rows = rows; a.ld_h = ld_h; a.B = (int)B; a.hw = (int)hw; a.E = (int)E; a.Q = (int)Q; a.CD = (int)CD;
    a.offset = offset; a.boi = boi; a.eoi = eoi; a.has_proj = has_proj;
    const size_t lds = has_proj ? (size_t)(CD * E * 2) : 16;
    static std::atomic<bool> attr_set{false};     // (idempotent call; atomic only so that concurrent first launches do not race on the flag)
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)lfq_encode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        attr_set = true;
    }
    long grid = (rows + 3) / 4;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(lfq_encode_kernel, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}
