// LDS tile images and MFMA fragment loads shared by the two bf16 GEMM structures (gfx950).
//
// An operand tile always covers 128 "output-index" lines (rows of A = m, rows of B = n) x 64 reduction
// steps (k) = 16 KiB.  Two global layouts are supported per operand, chosen at compile time:
//
//   N-type  the operand is stored [line][k]   (k contiguous: a Linear's x[M,K] / weight[N,K]).
//           LDS image [128 lines][64 k] = 128-byte rows, 16-byte chunk c of row r stored at chunk
//           c ^ ((r >> 1) & 7).  Fragment = one ds_read_b128 (8 consecutive k of one line), conflict free.
//   T-type  the operand is stored [k][line]   (line contiguous: dY[M_tokens, N_out] used as A^T for
//           wgrad, W[N_out, K_in] used as B for dgrad) — no transposed copy is ever materialised.
//           LDS image [64 k][128 lines] = 256-byte rows, chunk c of row r stored at c ^ ((r & 3) << 2).
//           Fragment = two ds_read_b64_tr_b16 (hardware transpose read: 4 k x 16 lines per 16-lane group;
//           semantics measured with experiments/tools/probe_tr.hip), conflict free: the 32 lanes of a service group
//           read 4 rows x 64 B that the swizzle spreads over all 64 banks.
//
// Both images are filled by 16-byte direct-to-LDS loads (lane-linear destination), so the swizzle is
// applied to the per-lane SOURCE address and mirrored on the fragment read.
#pragma once
#include "hip_common.hpp"

namespace libra {


// ---- fragment addressing (per lane, computed once) ----
struct FragAddr {
    int koffN[4];   // N-type: byte offset of (line = l31, chunk = 2*ks + fk); add 4096 * (32-line block index)
    int rowT;       // T-type: byte offset of k-row (8*fk + ((lane&15)>>2)); add 4096*ks, +1024 for the 2nd read
};

__device__ __forceinline__ FragAddr make_frag_addr(int lane) {
    FragAddr f;
    const int l31 = lane & 31, fk = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) f.koffN[ks] = l31 * 128 + ((((2 * ks + fk) ^ ((l31 >> 1) & 7))) << 4);
    const int p = lane & 15, g16 = (lane >> 4) & 1;
    f.rowT = (8 * fk + (p >> 2)) * 256;
    (void)g16;
    return f;
}

// byte offset contributed by the 32-line block index t (0..3) of a 128-line tile; computed arithmetically so
// that a wave-dependent t never turns into a runtime-indexed register array (scratch)
template <bool T>
__device__ __forceinline__ int frag_toff(int lane, int t) {
    if constexpr (!T) return t * 4096;
    const int p = lane & 15, g16 = (lane >> 4) & 1;
    return (((((t ^ (p >> 2)) & 3) << 2) | (2 * g16 + ((p & 3) >> 1))) << 4) + ((p & 1) << 3);
}

// fragment of the 32-line block whose frag_toff() is `toff`, k-step ks (0..3)
template <bool T>
__device__ __forceinline__ bf16x8 load_frag(const char* tile, const FragAddr& f, int toff, int ks) {
    if constexpr (!T) {
        return *(const bf16x8*)(tile + toff + f.koffN[ks]);
    } else {
        const char* a = tile + ks * 4096 + f.rowT + toff;
        union { bf16x8 v; s16x4 h[2]; } u;
        u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(a));
        u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(a + 1024));
        return u.v;
    }
}

// ---- staging: per-lane source element offset of one 1-KiB piece (relative to the tile's K origin) ----
// piece index `pc` counts 1-KiB pieces of the 16-KiB tile (0..15).
//   N-type piece = 8 lines x 128 B: line = 8*pc + lane/8, LDS position lane%8
//   T-type piece = 4 k-rows x 256 B: row = 4*pc + lane/16, LDS position lane%16
template <bool T, bool SPLIT = false>
__device__ __forceinline__ unsigned stage_src(int pc, int lane, int line0, int nlines, long ld,
                                              const int* __restrict__ rows = nullptr) {
    // SPLIT: the tile's 128 lines are two 64-line runs 128 operand lines apart (lines 0-63 <-> line0 + 0..63, lines 64-127 <->
    // line0 + 128..191): gemm_bf16_w.hip's A units, which hold the FIRST (or second) 64 rows of both 128-row wave blocks
    if constexpr (!T) {
        const int r = pc * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int g = line0 + (SPLIT ? (r & 63) + ((r >> 6) << 7) : r);
        g = g < nlines ? g : nlines - 1;                 // clamp the tail (masked at the store)
        if (rows) g = rows[g];                           // routed gather: logical line -> physical row
        return (unsigned)g * (unsigned)ld + c * 8;
    } else {
        const int r = pc * 4 + (lane >> 4);
        const int c = (lane & 15) ^ ((r & 3) << 2);
        int col = line0 + (SPLIT ? (c & 7) * 8 + ((c >> 3) << 7) : c * 8);
        col = col + 8 <= nlines ? col : nlines - 8;       // clamp (nlines % 8 == 0, >= 8)
        return (unsigned)r * (unsigned)ld + col;
    }
}
// element advance of the tile origin per K tile (64 reduction steps)
template <bool T>
__device__ __forceinline__ long ktile_stride(long ld) { return T ? 64 * ld : 64; }

}  // namespace libra
