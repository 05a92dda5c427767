// Argument block of the bridge-attention forward (attention_bridge.hip).
#pragma once
#include "hip_common.hpp"

namespace libra {

constexpr int BR_D = 128;          // head dim of Libra's decoder (LLaMA-2-7B: 32 heads x 128)

struct BridgeArgs {
    const bf16_t* q; long ldq;
    const bf16_t* k_same; const bf16_t* k_cross; long ldk, ldkc;
    const bf16_t* v_same; const bf16_t* v_cross; long ldv, ldvc;
    const unsigned char* flag;     // [B*S] 1 = vision token
    const int* kv_len;             // [B] end of the valid keys (right padding), or null
    const int* kv_start;           // [B] first valid key (LEFT padding: generation prompts, demo/libra_demo.ipynb), or null
    bf16_t* out; long ldo;
    float* lse;                    // [B,H,S] or null
    bf16_t* out_lo;                // optional rounding residual of `out` (same layout), see libra_bridge_attn_fwd
    int B, S, H, n_qt;
    float sl2;
};

}  // namespace libra
