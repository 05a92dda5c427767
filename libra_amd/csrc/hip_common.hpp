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
typedef __attribute__((ext_vector_type(2))) float f32x2;         // operand of the packed fp32 VALU ops (v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

typedef __attribute__((ext_vector_type(4))) short s16x4;     // operand type of the LDS transpose read builtin
#define LIBRA_LDS __attribute__((address_space(3)))
#define LIBRA_GLB __attribute__((address_space(1)))

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }

// round-to-nearest-even, NaN kept quiet (matches torch's float->bfloat16): gfx950's v_cvt_pk_bf16_f32
typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float hw_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    const hw_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hw_bf16x2));
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
// Issued as inline asm on purpose: for the builtin, hipcc's waitcnt pass cannot tell which LDS bytes a pending
// LDS-DMA will write and drains the whole VMEM queue (s_waitcnt vmcnt(0)) before the next ds_read of ANY address,
// which serialises every prefetch with the compute it was meant to hide under.  Hidden from that pass, the copies
// are ordered only by the kernels' own explicit `s_waitcnt vmcnt(n)` + barrier (every kernel has one per tile).
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    const unsigned lds_off = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(LIBRA_LDS char*)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_off) : "memory");
}

// the same with the LDS destination given as a byte address (wave-uniform): no generic -> LDS pointer conversion per call
__device__ __forceinline__ void glds16_at(const void* gsrc, unsigned lds_byte_addr) {
    const unsigned lds_off = __builtin_amdgcn_readfirstlane(lds_byte_addr);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_off) : "memory");
}

// 4-byte-per-lane variant (wave-uniform base + lane*4), same reasoning as glds16.
__device__ __forceinline__ void glds4(const void* gsrc, void* lds_wave_base) {
    const unsigned lds_off = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(LIBRA_LDS char*)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gsrc), "s"(lds_off) : "memory");
}
// 16-byte copy addressed as wave-uniform 64-bit base (SGPR pair) + per-lane unsigned 32-bit byte offset: no 64-bit
// per-lane address arithmetic and no VGPR pair per source pointer.
__device__ __forceinline__ void glds16_off(const void* sbase, unsigned voff, void* lds_wave_base) {
    const unsigned lds_off = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(LIBRA_LDS char*)lds_wave_base);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_off) : "memory");
}
// the same with the LDS destination given as a byte address (wave-uniform)
__device__ __forceinline__ void glds16_off_at(const void* sbase, unsigned voff, unsigned lds_byte_addr) {
    const unsigned lds_off = __builtin_amdgcn_readfirstlane(lds_byte_addr);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_off) : "memory");
}
// Make the compiler finish (wait for) whatever produces `v` HERE: values loaded in a prologue and first used inside a
// tile loop would otherwise get their s_waitcnt vmcnt inside the loop, where it also drains the hidden LDS-DMA queue.
template <typename T>
__device__ __forceinline__ void pin(T& v) { asm volatile("" : "+v"(v)); }

// Modality bit masks of one sequence into LDS: word t, bit i = token 32 t + i is a vision token, words [0, ceil(S/32)] (the
// spare last word is zero so a ragged 64-token tile reads zeros).  Fast path (flags 4-byte aligned, S a multiple of 4): every
// thread loads ONE dword = 4 flags, coalesced, and 8 lanes OR their nibbles into a word (3 cross-lane steps): one memory round
// trip and ~40 VALU.  [Round 5 cycle stamps: the previous form - one thread per word, 32 byte loads each, 65 threads busy - was
// 8.4k of the attention forward's 10.5k prologue cycles.]  Generic path: one thread per word, its 32 flag bytes as 32
// independent loads (still ONE round trip; the per-wave ballot loop before that made 9 dependent ones at S = 2048).
__device__ __forceinline__ void modality_masks(const unsigned char* __restrict__ flag_seq, int S, unsigned* masks, int tid, int nthreads) {
    const int n32 = (S + 31) / 32;
    if ((((unsigned long)flag_seq | (unsigned long)S) & 3ul) == 0) {
        const int n_dw = S >> 2;
        for (int i0 = 0; i0 < n32 * 8; i0 += nthreads) {            // (n32 * 8 dwords cover whole words; lanes past n_dw hold 0)
            const int i = i0 + tid;
            unsigned d = 0;
            if (i < n_dw) d = ((const unsigned*)flag_seq)[i];
            unsigned nib = ((d & 0xffu) ? 1u : 0u) | ((d & 0xff00u) ? 2u : 0u) | ((d & 0xff0000u) ? 4u : 0u) | ((d & 0xff000000u) ? 8u : 0u);
            unsigned v = nib << (4 * (tid & 7));
            v |= (unsigned)__shfl_xor((int)v, 1, 64);
            v |= (unsigned)__shfl_xor((int)v, 2, 64);
            v |= (unsigned)__shfl_xor((int)v, 4, 64);
            if ((tid & 7) == 0 && (i >> 3) < n32) masks[i >> 3] = v;
        }
        if (tid == 0) masks[n32] = 0;
        return;
    }
    for (int t = tid; t < n32 + 1; t += nthreads) {
        unsigned char f[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {                              // clamped, unconditional loads: all 32 in flight together
            const int tok = t * 32 + i;
            f[i] = flag_seq[tok < S ? tok : S - 1];
        }
        unsigned m = 0;
#pragma unroll
        for (int i = 0; i < 32; ++i) m |= ((f[i] != 0 && t * 32 + i < S) ? 1u : 0u) << i;
        masks[t] = m;
    }
}

// max of three without fmaxf's NaN-quieting canonicalisation moves (scores are finite or -inf here)
__device__ __forceinline__ float max3f(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// max(v, v of lane ^ 32): one v_permlane32_swap instead of an LDS round trip
__device__ __forceinline__ float half_swap_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_swap_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// Sixteen wave-wide sums at once, without the LDS crossbar: the two cross-half / cross-row steps exchange HALF of the values
// (v_permlane32_swap, v_permlane16_swap: 12 swaps + 12 adds leave 4 registers whose 16-lane rows each carry a different
// value), then four DPP row rotations finish every row.  On return out[e], read in any lane of row rho = lane >> 4, is the
// total of v[e + 4 * rho]  (40 VALU ops instead of 96 ds_bpermute round trips).
template <int N>
__device__ __forceinline__ float row_ror_add(float x) {
    const unsigned r = (unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), 0x120 + N, 0xf, 0xf, false);
    return x + __uint_as_float(r);
}
__device__ __forceinline__ void wave_sum16(const float* v, float* out) {
    float h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {       // lanes 0-31 keep value e, lanes 32-63 value e + 8
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[e]), __float_as_uint(v[e + 8]), false, false);
        h[e] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {       // rows: [e, e + 4, e + 8, e + 12]
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(h[e]), __float_as_uint(h[e + 4]), false, false);
        out[e] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) out[e] = row_ror_add<1>(row_ror_add<2>(row_ror_add<4>(row_ror_add<8>(out[e]))));
}
constexpr float DEFER_THR = 8.f;   // online-softmax running max is only advanced when a tile exceeds it by 2^8

// runtime.hip: physical CU count and the grid size of a persistent kernel under the caller's CU budget (libra_set_cu_budget)
int cu_count();
long persistent_grid(long nitems, int period);

// XCD-aware, bijective remap of a linear block id so that each of the 8 XCDs (block b runs on
// XCD b % 8) receives a contiguous range of the logical tile space (L2 locality; speed only).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, j = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}


// Tile (row, column) of workgroup `bid` of a tiles_m x tiles_n GEMM grid.  The hardware deals consecutive workgroup ids round-robin to
// the 8 XCDs, so the WT = 8*XR*XC workgroups resident at one time ("a wave of workgroups") are given one compact patch of the output:
// XCD x takes an XR x XC sub-patch (its private L2 sees XR + XC operand panels for XR*XC tiles) and the 8 sub-patches tile a
// (XR*WR) x (XC*WC) patch, so the whole chip reads (XR*WR + XC*WC) panels per wave from the memory side and the 8 XCDs fetch the SAME
// B panels at the same time (one HBM read + Infinity-Cache hits) instead of at 8 different moments of the launch.  Waves sweep the
// columns of one XR*WR-row group before moving to the next group: the group's A panels stay cache-resident over the sweep.
// Ragged edges shrink the group / block they touch; the map stays a bijection.  The last partial wave is dealt in XCD-contiguous chunks.
struct TileRC { int tm, tn; };
template <int XR, int XC, int WR, int WC>
__device__ __forceinline__ TileRC tile_order(int bid, int tiles_m, int tiles_n) {
    static_assert(WR * WC == 8, "8 XCDs");
    constexpr int XT = XR * XC, WT = XT * 8, GR = XR * WR;
    const int ntiles = tiles_m * tiles_n;
    const int full = ntiles / WT * WT;
    int u;
    if (bid < full) u = (bid / WT) * WT + (bid & 7) * XT + ((bid % WT) >> 3);
    else u = full + xcd_remap(bid - full, ntiles - full);
    const int gw = GR * tiles_n;                    // level 1: groups of GR rows, each sweeping every column
    const int g = u / gw;
    int v = u - g * gw;
    const int r0 = g * GR, gsz = min(GR, tiles_m - r0);
    const int cb = v / (gsz * XC);                  // level 2: blocks of XC columns inside the group
    v -= cb * gsz * XC;
    const int c0 = cb * XC, csz = min(XC, tiles_n - c0);
    const int rb = v / (XR * csz);                  // level 3: XR-row sub-blocks inside the column block (one XCD's patch)
    v -= rb * XR * csz;
    const int rr0 = rb * XR, rsz = min(XR, gsz - rr0);
    TileRC t;
    t.tm = r0 + rr0 + v % rsz;
    t.tn = c0 + v / rsz;
    return t;
}

}  // namespace libra
