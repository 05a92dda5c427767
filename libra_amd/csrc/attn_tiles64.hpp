// [64 rows][64 d] bf16 operand tiles of the ViT attention kernels (head_dim 64): 128-byte rows, 16-byte chunk c of row r
// stored at position c ^ g64(r).  One image serves both uses of an operand:
//   * row fragments (ds_read_b128: lane <-> row, 8 consecutive d) - the 16 rows of a lane group land on 16 distinct
//     (row parity, chunk position) pairs = all 64 banks;
//   * transposed fragments (ds_read_b64_tr_b16: A operand X^T[d][rows]) - the four rows of a lane group differ in
//     (row parity, bit 2 of the chunk position), again all 64 banks.
// So the backward needs no Q^T / K^T / dO^T copies and the forward no V^T copy (round 1 spent 57 us per layer on them).
#pragma once
#include "hip_common.hpp"

namespace libra {

__device__ __forceinline__ int g64(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }

// rows [row_lo, row_lo+64) of a row-major matrix (clamped to row_hi_excl-1); 4 waves, two 1-KiB pieces each
__device__ __forceinline__ void stage_tile64(const bf16_t* __restrict__ base, long ld, int row_lo, int row_hi_excl,
                                             char* lds_tile, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = wave * 16 + j * 8 + (lane >> 3);
        int gr = row_lo + r;
        gr = gr < row_hi_excl ? gr : row_hi_excl - 1;
        const int c = (lane & 7) ^ g64(r);
        glds16_off(base, (unsigned)(((long)gr * ld + c * 8) * 2), lds_tile + (wave * 16 + j * 8) * 128);   // (tile base in SGPRs + 32-bit lane offset)
    }
}
// byte offset of (row, 16-byte chunk)
__device__ __forceinline__ int off64(int row, int chunk) { return row * 128 + ((chunk ^ g64(row)) << 4); }

// A-operand fragment X^T[d = 32*dt + (lane&31)][rows 16*step + 4*(lane>>5) + {0..3, 8..11}]  (step = 0..3)
__device__ __forceinline__ bf16x8 tread64(const char* tile, int lane, int dt, int step) {
    const int pp = lane & 15, g16 = (lane >> 4) & 1, fk = lane >> 5;
    const int r1 = 16 * step + 4 * fk + (pp >> 2);
    const int chunk = dt * 4 + 2 * g16 + ((pp & 3) >> 1);
    const char* a = tile + r1 * 128 + ((pp & 1) << 3);
    union { bf16x8 v; s16x4 h2[2]; } u;
    u.h2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(a + ((chunk ^ g64(r1)) << 4)));
    u.h2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LIBRA_LDS s16x4*)(a + 1024 + ((chunk ^ g64(r1 + 8)) << 4)));
    return u.v;
}

}  // namespace libra
