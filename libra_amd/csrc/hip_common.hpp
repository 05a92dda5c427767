// Shared device helpers for the Libra gfx950 (CDNA4 / MI355X) kernels.
// wave = 64 lanes everywhere; MFMA shapes used: v_mfma_f32_32x32x16_bf16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace libra {

typedef unsigned short bf16_t;                                   // raw bf16 bits in memory
typedef __attribute__((ext_vector_type(8))) short bf16x8;        // one MFMA A/B fragment (8 bf16 = 4 VGPR)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;       // 32x32 accumulator fragment
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define LIBRA_LDS __attribute__((address_space(3)))
#define LIBRA_GLB __attribute__((address_space(1)))

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }

// round-to-nearest-even, NaN kept quiet (matches torch's float->bfloat16)
__device__ __forceinline__ bf16_t f2bf(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
}
__device__ __forceinline__ void unpack8(const u32x4 v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(v[i] << 16);
        f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack2bf(f[2 * i], f[2 * i + 1]);
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// 16-byte async global -> LDS copy. LDS destination is wave-uniform base + lane*16.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const LIBRA_GLB void*)gsrc, (LIBRA_LDS void*)lds_wave_base, 16, 0, 0);
}

// XCD-aware, bijective remap of a linear block id so that each of the 8 XCDs (block b runs on
// XCD b % 8) receives a contiguous range of the logical tile space (L2 locality; speed only).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, j = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

}  // namespace libra
