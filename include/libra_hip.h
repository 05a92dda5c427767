/*
 * libra_hip.h — C ABI of the MI355X (gfx950) kernels for Libra's vision-to-LLM hot path.
 *
 * The reference (YifanXu74/Libra) has no FFI: its hot path is plain PyTorch ops inside
 * libra/models/{clip,libra,llama}.  This library is what sits *under* the Python module surface
 * (SURVEY.md §8b): every entry point replaces the torch-op sequence of the cited reference lines.
 *
 * Conventions
 *   - all pointers are DEVICE pointers borrowed for the duration of the enqueue; the library never
 *     allocates, frees, synchronises or keeps global mutable state;
 *   - bf16 tensors are raw uint16 storage, row-major, leading dimensions in ELEMENTS;
 *   - `stream` is a hipStream_t (NULL = default stream); work is only enqueued;
 *   - return value: LIBRA_OK (0) or a negative LIBRA_ERR_* code; nothing is launched on error.
 *     (The Python host raises ValueError / RuntimeError from these, mirroring the reference's
 *     shape-check exceptions, e.g. libra/models/libra/modeling_libra.py:374-403.)
 */
#ifndef LIBRA_HIP_H
#define LIBRA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LIBRA_OK 0
#define LIBRA_ERR_SHAPE (-1)   /* unsupported / inconsistent shape */
#define LIBRA_ERR_ALIGN (-2)   /* pointer or leading dimension not 16-byte aligned / null */
#define LIBRA_ERR_LAUNCH (-3)  /* HIP refused the launch */

/* ABI version, bumped on any signature change. */
int libra_hip_abi_version(void);

/* ---- bf16 GEMM, C[M,N] = epi(A[M,K] . B[N,K]^T), fp32 accumulate ---------------------------------
 * Replaces every nn.Linear / F.linear / 1x1-conv on the path: CLIP q/k/v/out_proj, fc1, fc2
 * (libra/models/clip/modeling_clip.py:299-301,:361,:375-377), the patch-embedding conv as an im2col
 * GEMM (:195), quant_conv (libra/models/libra/taming/models/vqgan.py:108), and their autograd
 * dgrad/wgrad products.  K % 64 == 0; lda/ldb/ldc/ldr/ldaux % 8 == 0; 16-byte aligned pointers.
 * epilogue, in order:  v = (acc + bias[n]) * (n < alpha_cols ? alpha : 1);  [store preact];
 *                      quick_gelu(bf16(v));  * qgelu'(aux);  + resid;  round to bf16.                                                     */
#define LIBRA_GEMM_BIAS 1            /* + bias[N] (bf16) */
#define LIBRA_GEMM_QUICK_GELU 2      /* x*sigmoid(1.702x)   (HF ACT2FN["quick_gelu"], modeling_clip.py:376) */
#define LIBRA_GEMM_RESIDUAL 4        /* + resid[M,N] (bf16, ldr) — residual add of modeling_clip.py:413,:418 */
#define LIBRA_GEMM_MUL_QGELU_GRAD 8  /* * d/dx quick_gelu(aux[M,N])  (backward of the fc1 activation) */
#define LIBRA_GEMM_STORE_PREACT 16   /* also store the pre-activation (bf16) to preact[M,N] (saved for backward) */
#define LIBRA_GEMM_A_T 32            /* A is given reduction-major: A^T [K, M] row-major, lda >= M, M % 8 == 0 (wgrad: dY) */
#define LIBRA_GEMM_B_T 64            /* B is given reduction-major: [K, N] row-major, ldb >= N, N % 8 == 0 (dgrad: the weight itself; wgrad: X) */
int libra_gemm_bf16_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                       int64_t M, int64_t N, int64_t K, const void* bias, const void* resid, int64_t ldr,
                       const void* aux, int64_t ldaux, void* preact, int64_t ldpre, float alpha,
                       int64_t alpha_cols, int flags, void* stream);

/* Routed variant (Libra's per-token modality routing, cal_language_vision: modeling_libra.py:111-147, without the
 * boolean-mask gather/scatter passes): logical row m of A is physical row a_rows[m] (K-contiguous A only;
 * a_phys_rows = number of physical rows of A), and logical output row m is written to row c_rows[m] of C (and
 * read from that row of resid / aux, written to that row of preact).  Either map may be NULL.       */
int libra_gemm_bf16_nt_routed(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                              int64_t M, int64_t N, int64_t K, const void* bias, const void* resid, int64_t ldr,
                              const void* aux, int64_t ldaux, void* preact, int64_t ldpre, float alpha,
                              int64_t alpha_cols, int flags, const int32_t* a_rows, int64_t a_phys_rows,
                              const int32_t* c_rows, void* stream);

/* The same with the tile structure chosen by the caller instead of the library's cost model (speed only - every structure
 * computes the same contract, bit for bit per output element up to the fp32 summation order inside a K tile, which is equal too):
 * AUTO = as libra_gemm_bf16_nt_routed; 128 = 128x128 tiles, two workgroups per CU; 256 = 256x256 tiles, one workgroup per CU
 * (falls back to 128 when M or N < 256); W = 256x128 tiles, 4 waves, two workgroups per CU (gemm_bf16_w.hip).  For tests (every
 * structure against the oracle on every shape) and for tools/gemm_sweep.py, which fits the cost model.                          */
#define LIBRA_GEMM_TILE_AUTO 0
#define LIBRA_GEMM_TILE_128 1
#define LIBRA_GEMM_TILE_256 2
#define LIBRA_GEMM_TILE_W 3
int libra_gemm_bf16_nt_tile(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                            int64_t M, int64_t N, int64_t K, const void* bias, const void* resid, int64_t ldr,
                            const void* aux, int64_t ldaux, void* preact, int64_t ldpre, float alpha,
                            int64_t alpha_cols, int flags, const int32_t* a_rows, int64_t a_phys_rows,
                            const int32_t* c_rows, int tile, void* stream);

/* Grouped launch: `groups` (<= 4) GEMMs of identical shape, strides, flags (A_T / B_T only) and row maps but different
 * operand pointers run as ONE launch (blockIdx.z = group), so that several mid-sized problems fill whole waves of
 * workgroups together - e.g. the three low-rank vision projections LibraLinear.weight_B of q, k, v (modeling_libra.py:64-90),
 * each 1.19 waves of 256^2 tiles on their own.  No fused epilogue operands.  tile: LIBRA_GEMM_TILE_* as above (0 = cost model).                                          */
int libra_gemm_bf16_nt_grouped(const void* const* A, int64_t lda, const void* const* B, int64_t ldb, void* const* C,
                               int64_t ldc, int64_t groups, int64_t M, int64_t N, int64_t K, float alpha,
                               int64_t alpha_cols, int flags, const int32_t* a_rows, int64_t a_phys_rows,
                               const int32_t* c_rows, int tile, void* stream);

/* Multi-problem launch (round 6): up to LIBRA_GEMM_MULTI_MAX INDEPENDENT GEMMs - each with its own shapes, strides, operand
 * layouts (A_T / B_T), row maps and fused epilogue, i.e. each one a complete libra_gemm_bf16_nt_routed call - as ONE persistent
 * launch over a common list of 256x256 tiles, longest K first, handed out dynamically.  What it replaces on the reference side is
 * still the per-module F.linear calls of cal_language_vision (modeling_libra.py:111-147): the language module's product on the
 * text rows and the vision module's products on the vision rows are independent and share a launch here (e.g. q|k|v of the text
 * rows + the three LibraLinear.weight_B expansions of the vision rows, modeling_libra.py:192-199), and so do the weight
 * gradients of a layer, which nothing inside the layer waits for.  Per output element the arithmetic is that of
 * libra_gemm_bf16_nt_tile(..., LIBRA_GEMM_TILE_256), bit for bit.  No problem may read what another problem of the same call writes.
 * splitk > 1: the problem's K range is cut into that many slices, each slice a tile-list entry of its own (a weight gradient with a
 * few very long tiles becomes filler for the others' last wave); slab = fp32 [splitk][M][N] workspace (16-byte aligned), reduced
 * to bf16 C by a second kernel in slice order (deterministic) - the arithmetic of libra_gemm_bf16_nt_splitk(_routed); only
 * LIBRA_GEMM_RESIDUAL may be fused, N % 8 == 0.  splitk <= 1: slab is ignored.
 * wait_on >= 0: this problem READS (as A, B, residual ...) the C of problems[wait_on] - the one exception to "independent": its tiles
 * are listed behind the producer's and wait, on the device, until all of the producer's tiles are stored (release / acquire at
 * agent scope).  The producer must be an unsplit problem that waits for nothing itself.  (The second low-rank stage of a LibraLinear
 * pair, modeling_libra.py:192-199, behind its first stage, in the launch of the text GEMM that runs beside both.)  -1: none.
 * queue_ws: 128 bytes of device memory, 16-byte aligned, ALL ZERO at the first use and owned by one stream - the kernel leaves it
 * zero again (its counters clear themselves), so it is allocated and cleared once, not per call; word 10 turns non-zero, and stays
 * so, if a device-side wait ever ran out (the results of that call are then invalid).
 */
#define LIBRA_GEMM_MULTI_MAX 12
typedef struct libra_gemm_problem {
    const void* A; int64_t lda; const void* B; int64_t ldb; void* C; int64_t ldc;
    int64_t M, N, K;
    const void* bias; const void* resid; int64_t ldr; const void* aux; int64_t ldaux; void* preact; int64_t ldpre;
    float alpha; int32_t flags; int64_t alpha_cols;
    const int32_t* a_rows; int64_t a_phys_rows; const int32_t* c_rows;
    int64_t splitk; void* slab;
    int64_t wait_on;
} libra_gemm_problem;
int libra_gemm_bf16_multi(const libra_gemm_problem* problems, int64_t n_problems, void* queue_ws, void* stream);

/* The first half of LlamaMLP for a generation step (M <= 16 rows) in one launch: Y[M, I] = silu(A W_gate^T) * (A W_up^T),
 * modeling_llama.py:199-201 (act_fn(gate_proj(x)) * up_proj(x)), W_gate_up = [2I, K] = gate rows then up rows (the packed operand
 * the training GEMM uses).  Arithmetic = libra_gemm_bf16_nt followed by libra_swiglu, bit for bit (both products rounded to bf16,
 * bf16(silu(g)) * u): HBM-bound weight streaming like the other M <= 16 GEMMs.  a_rows: optional row gather on A.                 */
int libra_gemm_swiglu_skinny(const void* A, int64_t lda, const void* W_gate_up, int64_t ldw, void* Y, int64_t ldy, int64_t M,
                             int64_t I, int64_t K, const int32_t* a_rows, int64_t a_phys_rows, void* stream);

/* split-K variant for wgrad-shaped problems (small M,N, very long K; no epilogue): K slices on the 256^2
 * kernel, fp32 partial slabs in `workspace`, deterministic reduction to bf16 C.  plan() suggests the number
 * of slices for a shape (1 = use libra_gemm_bf16_nt).  N % 8 == 0; flags: LIBRA_GEMM_A_T / _B_T only.                                  */
int libra_gemm_splitk_plan(int64_t M, int64_t N, int64_t K);
size_t libra_gemm_splitk_workspace_bytes(int64_t M, int64_t N, int64_t splits);
int libra_gemm_bf16_nt_splitk(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                              int64_t M, int64_t N, int64_t K, int64_t splits, int flags, void* workspace,
                              size_t workspace_bytes, void* stream);
/* The routed form of the same (row maps as libra_gemm_bf16_nt_routed; K-contiguous A; flags: LIBRA_GEMM_B_T, LIBRA_GEMM_RESIDUAL -
 * the residual is read at the scattered row by the reduction stage): the decoder's text-stream projections of a step with few
 * text rows (libra_pretrain.yaml:19 - 700-token sequences leave 976 text rows per 8-sequence step).                          */
int libra_gemm_bf16_nt_splitk_routed(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                                     int64_t M, int64_t N, int64_t K, int64_t splits, int flags, const void* resid, int64_t ldr,
                                     const int32_t* a_rows, int64_t a_phys_rows, const int32_t* c_rows, void* workspace,
                                     size_t workspace_bytes, void* stream);

/* ---- LayerNorm over the last dim (nn.LayerNorm, eps 1e-5: modeling_clip.py:386-388,:866) ---------
 * x,y [rows,D] bf16 contiguous, gamma/beta [D] bf16, mean/rstd [rows] fp32 (saved for backward, may be
 * NULL).  D % 8 == 0, D <= 8192.                                                                     */
int libra_layernorm_fwd(const void* x, const void* gamma, const void* beta, void* y, float* mean,
                        float* rstd, int64_t rows, int64_t D, float eps, void* stream);
/* dx [rows,D] bf16 (optionally dx += dres, the residual-branch gradient, bf16 [rows,D]);
 * dgamma/dbeta: fp32 [D], ADDED to (deterministic two-stage reduction through `workspace`, no atomics),
 * may both be NULL (then no workspace is needed).  dxsum (fp32 [D], may be NULL; needs dgamma/dbeta): ADDED the
 * column sum of the dx this call writes = the bias gradient of the nn.Linear whose output gradient dx is
 * (modeling_clip.py:404-414: dx of layer_norm2 feeds out_proj, dx of layer_norm1 feeds the previous layer's fc2),
 * which saves a separate column-sum pass over dx.  D <= 4096.                                         */
size_t libra_layernorm_bwd_workspace_bytes(int64_t rows, int64_t D);
int libra_layernorm_bwd(const void* dy, const void* x, const void* gamma, const float* mean,
                        const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta, float* dxsum,
                        void* workspace, size_t workspace_bytes, int64_t rows, int64_t D, void* stream);

/* ---- patch embedding front end (CLIPVisionEmbeddings.forward, modeling_clip.py:193-228) -----------
 * im2col of non-overlapping PxP patches: pixel [B,C,H,W] bf16 -> cols [B*(H/P)*(W/P), Kpad] bf16 with
 * column (c,ky,kx) = c*P*P + ky*P + kx, zero-padded to Kpad (% 64 == 0).                            */
int libra_patch_im2col(const void* pixel, void* cols, int64_t B, int64_t C, int64_t H, int64_t W,
                       int64_t P, int64_t Kpad, void* stream);
/* col2im-free backward helper is not needed: d(pixel) = dcols scattered back by the same index map. */
int libra_patch_col2im(const void* dcols, void* dpixel, int64_t B, int64_t C, int64_t H, int64_t W,
                       int64_t P, int64_t Kpad, void* stream);
/* emb[b,0] = cls + pos[0]; emb[b,1+p] = patches[b*Np+p] + pos[1+p]  (bf16 [B,T,D], T = Np+1), then
 * hs0 = LayerNorm(emb) (the `pre_layrnorm`, modeling_clip.py:893).  mean/rstd may be NULL.          */
int libra_vit_embed_ln(const void* patches, const void* cls, const void* pos, const void* gamma,
                       const void* beta, void* emb, void* hs0, float* mean, float* rstd, int64_t B,
                       int64_t T, int64_t D, float eps, void* stream);

/* ---- column sums (bias gradient db[n] = sum_m dY[m,n]); out fp32 [cols] is ADDED to; deterministic two-stage */
size_t libra_colsum_workspace_bytes(int64_t rows, int64_t cols);
int libra_colsum_bf16(const void* x, int64_t ld, int64_t rows, int64_t cols, float* out, void* workspace,
                      size_t workspace_bytes, void* stream);

/* ---- ViT self-attention, flash style (CLIPAttention.forward, modeling_clip.py:287-363) -------------
 * qkv [B*T, 3*H*64] bf16 (q | k | v, head h at columns h*64 of each third; q UNscaled).
 * out [B*T, H*64] bf16, lse [B,H,T] fp32 (natural-log-sum-exp of the scaled scores; may be NULL).
 * head_dim is fixed at 64 (CLIP ViT-L/14).  K and V tiles are read where they lie (no transposed copies).
 * out_lo (optional, layout of out): the rounding residual bf16(O - bf16(O)) of the fp32 output.  The backward's
 * D_i = sum_d dO_id O_id is a difference-sensitive term (dS = P (dP - D)): taken from the bf16 O alone, q / k weight gradients
 * of ViT-L came out 3-9 % off where the reference's bf16 autograd is 1-2 % off (profiles/r02_parity.txt); with O + out_lo
 * (~16 mantissa bits) they are within 1 %. */
int libra_vit_attn_fwd(const void* qkv, int64_t ld_qkv, void* out, int64_t ld_out, float* lse, void* out_lo, int64_t B,
                       int64_t T, int64_t H, float scale, void* stream);
/* D = rowsum(dO * O) per (image, head, token): delta [B,H,T] fp32.                                   */
int libra_vit_attn_delta(const void* out, const void* out_lo, const void* dout, int64_t ld, float* delta, int64_t B, int64_t T,
                         int64_t H, void* stream);
/* Backward (autograd of modeling_clip.py:308-348): from qkv, dO [B*T,H*64], lse and delta produce
 * dqkv [B*T, 3*H*64] (dq/dk already multiplied by `scale`, i.e. gradients w.r.t. the UNscaled q, k).
 * Deterministic: two passes (dQ; dK/dV), no atomics, no transposed operand copies.                    */
int libra_vit_attn_bwd(const void* qkv, int64_t ld_qkv, const void* dout, int64_t ld_out, const float* lse,
                       const float* delta, void* dqkv, int64_t ld_dqkv, int64_t B, int64_t T, int64_t H,
                       float scale, void* stream);

/* ---- vision-tower feature select (CLIPVisionTower.feature_select, clip_encoder.py:31-45) -----------
 * feat[b*(T-1)+p, j*D + d] = hs_j[b, 1+p, d] for j < n_sel (<= 4): channel concat, CLS dropped.      */
int libra_feature_select(const void* const* hs, int64_t n_sel, void* feat, int64_t B, int64_t T,
                         int64_t D, void* stream);
/* backward scatter: dhs_j[b,1+p,:] (+)= dfeat[..., j*D:(j+1)*D]; CLS rows get 0 unless accumulate.  */
int libra_feature_select_bwd(const void* dfeat, void* const* dhs, const int* accumulate, int64_t n_sel,
                             int64_t B, int64_t T, int64_t D, void* stream);

/* ---- LFQ sign-quantise + pack (LFQ.forward eval branch, lookup_free_quantization.py:185-208, and
 *      ImageTokenizer.encode, image_tokenizer.py:77-86) -------------------------------------------
 * h [rows, E] bf16, row stride ld_h (rows = B*hw, the quant_conv output); w_in [Q*9, E], b_in [Q*9] bf16 or NULL when
 * E == Q*9 (no projection).  x = bf16(h . w_in^T + b_in); bit = x > 0; index_q = sum bit * 2^(8-j).
 * Outputs (any may be NULL): indices int64 [rows, Q];  ids int64 [Q, B, hw+2] = offset+index framed by
 * BOI/EOI;  xpre bf16 [rows, Q*9] (the pre-sign value, for margin reports);
 * quant bf16 [rows, E] = project_out(+-1) (w_out [E, Q*9], b_out [E]; or +-1 itself when no projection). */
int libra_lfq_encode(const void* h, int64_t ld_h, const void* w_in, const void* b_in, const void* w_out,
                     const void* b_out, int64_t* indices, int64_t* ids, void* xpre, void* quant,
                     int64_t B, int64_t hw, int64_t E, int64_t Q, int64_t offset, int64_t boi,
                     int64_t eoi, void* stream);

/* ==== routed ("bridge") decoder rows (libra/models/libra/modeling_libra.py) ===========================*/
/* Routed RMSNorm: y = w_m * bf16(x * rsqrt(mean(x^2)+eps)), w_m = flag[row] ? w_vis : w_lang
 * (LlamaRMSNorm, models/llama/modeling_llama.py:127-132, routed at modeling_libra.py:463,:481,:817).
 * flag NULL = unrouted (w_lang for every row: vision_signal_norm, :640).  rstd [rows] fp32 optional.     */
int libra_rmsnorm_routed_fwd(const void* x, int64_t ldx, const void* w_lang, const void* w_vis,
                             const uint8_t* flag, void* y, int64_t ldy, float* rstd, int64_t rows, int64_t D,
                             float eps, void* stream);
/* RoPE + rank-8 bridge expansion (modeling_libra.py:318-340, apply_rotary_pos_emb :39-61), head_dim 128.
 * qkv [N, 3*H*128] (q|k|v): q and k are rotated IN PLACE (k -> K_same); tb [N, ldt>=16] holds the bridge
 * low-rank activations (cols 0..7 key bridge, 8..15 value bridge); bk_x / bv_x are the bridges' weight_B
 * [H*128, 8] for language / vision tokens; outputs K_cross = rope(bf16(k + kb)), V_cross = bf16(v + vb)
 * [N, H*128].  Token n has position n % S; cos/sin are bf16 [max_pos, 128] tables.                    */
int libra_rope_bridge(void* qkv, int64_t ld, const void* tb, int64_t ldt, const void* bk_l, const void* bk_v,
                      const void* bv_l, const void* bv_v, const uint8_t* flag, const void* cos, const void* sin,
                      int64_t max_pos, void* k_cross, void* v_cross, int64_t ldc, int64_t N, int64_t S, int64_t H,
                      void* stream);
/* The same with explicit RoPE positions (int32 [N, pos_stride], clamped into [0, max_pos)) instead of n % S: the cached decode
 * step and left-padded prompts (position_ids = attention_mask.cumsum - 1, prepare_inputs_for_generation, modeling_libra.py:
 * 1196-1209), and - pos_stride = 2 - `use_2d_rope` (:43-49, :663-678): even heads rotate by column 0 (the row position), odd heads
 * by column 1 (the column position). */
int libra_rope_bridge_pos(void* qkv, int64_t ld, const void* tb, int64_t ldt, const void* bk_l, const void* bk_v,
                          const void* bv_l, const void* bv_v, const uint8_t* flag, const void* cos, const void* sin,
                          int64_t max_pos, void* k_cross, void* v_cross, int64_t ldc, int64_t N, const int* positions,
                          int64_t pos_stride, int64_t H, void* stream);
/* The same for a generation step (one new token per sequence, row n = sequence n), which also stores the token's four rows
 * K_same = rope(k), K_cross = rope(k + kb), V_same = v, V_cross = v + vb at slot *slot of the layer's caches [B, Lmax, H*128]
 * (row / batch strides in elements; the reference's cache update, modeling_libra.py:344-361) - libra_rope_bridge_pos followed by
 * libra_kv_cache_append in one launch.  `slot` is read on the device (capturable); a slot outside the sequence's own rows
 * [0, batch_stride / row_stride) stores NOTHING in the caches (no out-of-bounds write when a replayed graph outruns the cache).      */
int libra_rope_bridge_pos_append(void* qkv, int64_t ld, const void* tb, int64_t ldt, const void* bk_l, const void* bk_v,
                                 const void* bv_l, const void* bv_v, const uint8_t* flag, const void* cos, const void* sin,
                                 int64_t max_pos, void* k_cross, void* v_cross, int64_t ldc, int64_t B, const int* positions,
                                 int64_t pos_stride, int64_t H, void* cache_k_same, void* cache_k_cross, void* cache_v_same,
                                 void* cache_v_cross, int64_t row_stride, int64_t batch_stride, const int64_t* slot, void* stream);
/* Append the new token of every sequence to a layer's four caches (the reference's `torch.cat` of past and present key / value
 * states, modeling_libra.py:344-361) in one launch: cache_x[b][*slot][0..W) = x[b][0..W) for x in (K_same, K_cross, V_same,
 * V_cross); x [B, W] with row strides ld_x, caches [B, Lmax, W] with row / batch strides in elements, slot = the position,
 * read on the device (one int64): the decode step stays capturable in a hipGraph.  A slot outside [0, batch_stride / row_stride)
 * stores nothing. */
int libra_kv_cache_append(const void* k_same, int64_t ld_ks, const void* k_cross, int64_t ld_kc, const void* v_same, int64_t ld_vs,
                          const void* v_cross, int64_t ld_vc, void* cache_k_same, void* cache_k_cross, void* cache_v_same,
                          void* cache_v_cross, int64_t row_stride, int64_t batch_stride, const int64_t* slot, int64_t B, int64_t W,
                          void* stream);
/* Routed-bridge attention of ONE new query token per sequence against the KV cache (LibraAttention.forward with
 * past_key_value, modeling_libra.py:344-391, at q_len = 1).  The reference's cache ([K_for_vision, K_for_language], V,
 * V_bridge, flag) is held as the four row buffers the training path produces - K_same, K_cross, V_same, V_cross
 * [B, Lmax, H*128] bf16 (row stride ldc, batch stride batch_stride elements) - plus key_flag [B, Lmax] (row stride
 * flag_stride): key j uses the *_cross buffers iff key_flag[b][j] != query_flag[b].  q, out [B, H*128]; kv_len[b] = number
 * of valid cached tokens including the new one; scale = 1/sqrt(128).
 * workspace (fp32, 16-byte aligned, >= libra_bridge_attn_decode_workspace_bytes(B, H)) or NULL: with it the cached keys of
 * every (sequence, head) are split over 4 workgroups whose partial softmax states a second kernel folds (4x the loads in
 * flight: the B x H workgroups of the unsplit launch fill a quarter of the chip); without it one workgroup per (sequence, head). */
size_t libra_bridge_attn_decode_workspace_bytes(int64_t B, int64_t H);
int libra_bridge_attn_decode(const void* q, int64_t ldq, const void* k_same, const void* k_cross, const void* v_same,
                             const void* v_cross, int64_t ldc, int64_t batch_stride, const uint8_t* key_flag,
                             int64_t flag_stride, const uint8_t* query_flag, const int* kv_len, const int* kv_start,
                             void* out, int64_t ldo, int64_t B, int64_t H, float scale, float* workspace,
                             size_t workspace_bytes, void* stream);
/* Fused routed-bridge causal flash attention, forward (LibraAttention.forward + attn_with_bridge,
 * modeling_libra.py:267-414):  S_ij = q_i.(k_j + [m_i!=m_j] kb_j)/sqrt(d) (+causal, +right padding via
 * kv_len[b] = end of the valid keys, +left padding via kv_start[b] = first valid key; either may be NULL), O_i = sum_j softmax(S)_ij (v_j + [m_i!=m_j] vb_j).  Operands are [B*S, H*128] views with row
 * strides ldq/ldk/ldkc/ldv/ldvc; flag [B*S] (1 = vision token); out [B*S, H*128]; lse [B,H,S] fp32 optional;
 * out_lo optional (layout of out): rounding residual of the output for the backward's D term, see libra_vit_attn_fwd.   */
int libra_bridge_attn_fwd(const void* q, int64_t ldq, const void* k_same, int64_t ldk, const void* k_cross,
                          int64_t ldkc, const void* v_same, int64_t ldv, const void* v_cross, int64_t ldvc,
                          const uint8_t* flag,
                          const int32_t* kv_len, const int32_t* kv_start, void* out, int64_t ldo, float* lse, void* out_lo,
                          int64_t B, int64_t S, int64_t H, float scale, void* stream);
/* Backward of libra_bridge_attn_fwd (deterministic, two passes): from dO and the forward's operands / lse produce
 * dq [B*S,H*128] (w.r.t. the rotated q) and the four operand gradients dK_same, dK_cross, dV_same, dV_cross
 * ([B*S, H*128], row stride ldg).  `out` (+ optional `out_lo`) is the forward output (D = rowsum(dO*O) is formed by the dQ pass
 * from the dO fragments it already holds and handed to the dK / dV pass through `delta` [B,H,S], scratch).
 * err_word (device int32, 4-byte aligned, or NULL): sticky error word.  Rounds 3-4 raised bit 0 when the dK / dV pass's bounded
 * in-workgroup wait ran out; since round 5 the pass synchronises with workgroup barriers only and never writes the word (the
 * parameter stays for ABI stability and for future device-side checks).                                                      */
int libra_bridge_attn_bwd(const void* q, int64_t ldq, const void* k_same, int64_t ldk, const void* k_cross,
                          int64_t ldkc, const void* v_same, int64_t ldv, const void* v_cross, int64_t ldvc,
                          const void* out, const void* out_lo, int64_t ldout, const void* dout, int64_t lddo, const uint8_t* flag,
                          const int32_t* kv_len, const float* lse, float* delta, void* dq, int64_t lddq,
                          void* dk_same, void* dk_cross, void* dv_same, void* dv_cross, int64_t ldg, int64_t B,
                          int64_t S, int64_t H, float scale, int32_t* err_word, void* stream);
/* y = bf16(silu(gate)) * up  (LlamaMLP, models/llama/modeling_llama.py:199-201)                          */
int libra_swiglu(const void* gate, const void* up, int64_t ldgu, void* y, int64_t ldy, int64_t rows, int64_t I,
                 void* stream);
/* out[r, col0:col0+D] = table[idx[rows_sel ? rows_sel[r] : r] - sub]   (nn.Embedding lookups of
 * get_inputs_embeds_from_multicodebook, modeling_libra.py:625-636); idx int64                          */
int libra_gather_rows(const void* table, int64_t D, const int64_t* idx, int64_t sub, const int32_t* rows_sel,
                      int64_t n, void* out, int64_t ldo, int64_t col0, void* stream);
/* out[r, col0:col0+D] = in[rows_sel ? rows_sel[r] : r, :D]                                              */
int libra_copy_rows(const void* in, int64_t ldi, int64_t D, const int32_t* rows_sel, int64_t n, void* out,
                    int64_t ldo, int64_t col0, void* stream);
/* per-row cross-entropy terms of CrossEntropyLoss over bf16 logits [rows, V] (modeling_libra.py:1160-1174):
 * loss_rows[r] = logsumexp(z_r) - z_r[target[r] - target_sub], 0 when target[r] < 0 (ignore_index -100),
 * +inf when the target lies outside this head's column range (the reference's -inf padded logit).       */
int libra_ce_rows(const void* logits, int64_t ldz, int64_t V, const int64_t* target, int64_t target_sub,
                  float* loss_rows, int64_t rows, void* stream);

/* ---- decoder backward row kernels ------------------------------------------------------------------*/
/* dlogits[r,v] = sum_q [t_q[r]>=0] coef_q (softmax(z_r)[v] - [v == t_q[r]-target_sub]); target1 may be NULL.
 * scale_dev (device fp32 scalar, may be NULL): both coefficients are multiplied by *scale_dev in the kernel - the upstream
 * gradient of the loss (autograd's incoming scalar, a loss scale) stays on the device, no host read at the start of backward. */
int libra_ce_rows_bwd(const void* logits, int64_t ldz, int64_t V, const int64_t* target0, const int64_t* target1,
                      int64_t target_sub, float coef0, float coef1, const float* scale_dev, void* dlogits, int64_t lddz,
                      int64_t rows, void* stream);
/* routed RMSNorm backward: dx = rstd (g - xh mean(g xh)) [+ dres], g = dy w_m, xh = x rstd (rstd from the forward) */
int libra_rmsnorm_routed_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* w_lang,
                             const void* w_vis, const uint8_t* flag, const float* rstd, const void* dres, int64_t lddr,
                             void* dx, int64_t lddx, int64_t rows, int64_t D, void* stream);
/* dw_m[c] += sum_{rows of modality m} dy x rstd  (fp32 [D], deterministic two-stage; dw_vis NULL when unrouted).
 * rows_sel (int32 [n_sel], optional): visit only these rows of the [rows, D] operands - e.g. the vision rows when only the
 * vision norm weight is trainable (frozen-language pretraining): the other modality's rows are then not read.          */
size_t libra_rmsnorm_wgrad_workspace_bytes(int64_t rows, int64_t D);
int libra_rmsnorm_routed_wgrad(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* rstd,
                               const uint8_t* flag, float* dw_lang, float* dw_vis, void* workspace,
                               size_t workspace_bytes, int64_t rows, int64_t D, const int32_t* rows_sel, int64_t n_sel,
                               void* stream);
/* SwiGLU backward: dgate = dy up s (1 + gate (1 - s)), dup = dy silu(gate)                                   */
int libra_swiglu_bwd(const void* dy, int64_t lddy, const void* gate, const void* up, int64_t ldgu, void* dgate,
                     void* dup, int64_t ldd, int64_t rows, int64_t I, void* stream);
/* backward of libra_rope_bridge: dqkv [N,3*H*128] = (R^T dq', R^T (dK_same + dK_cross), dV_same + dV_cross),
 * dkb [N,H*128] = R^T dK_cross  (dvb = dV_cross itself), and - when dtb is not NULL - the gradient of the rank-8 bridge
 * activations dtb[n, 0:8] = B_k[m_n]^T dkb[n], dtb[n, 8:16] = B_v[m_n]^T dvb[n] (LibraAttention.forward,
 * modeling_libra.py:318-340), taken while dkb / dvb are in registers.  The four bridge operands are given TRANSPOSED here,
 * bk_* / bv_* = weight_B^T, [8, H*128] row-major bf16: a dword then holds the weights of two adjacent channels for one rank,
 * the operand shape of v_dot2_f32_bf16 against the packed gradient.  H <= 32. */
int libra_rope_bridge_bwd(const void* dq, const void* dk_same, const void* dk_cross, const void* dv_same,
                          const void* dv_cross, int64_t ld, const void* cos, const void* sin, int64_t max_pos,
                          void* dqkv, int64_t ldo, void* dkb, int64_t ldb, const void* bk_l, const void* bk_v,
                          const void* bv_l, const void* bv_v, const uint8_t* flag, void* dtb, int64_t ldt, int64_t N,
                          int64_t S, int64_t H, const int* positions, int64_t pos_stride, void* stream);

/* ---- small elementwise helpers -------------------------------------------------------------------*/
/* out_bf16[i] = bf16(in_f32[i])  (parameter-gradient accumulators -> bf16 .grad) */
int libra_f32_to_bf16(const float* in, void* out, int64_t n, void* stream);
/* y = a + b (bf16, n % 8 == 0 not required) */
int libra_add_bf16(const void* a, const void* b, void* y, int64_t n, void* stream);

/* ---- VQ image decoder (image generation: LFQ.indices_to_codes -> post_quant_conv -> taming Decoder) -----------------------
 * Activations are NHWC bf16 ([pixels, channels] row-major); 1x1 convs and the gathered 3x3 convs run on libra_gemm_bf16_nt. */
/* codes[m, q*nbits + j] = +-1 from bit (nbits-1-j) of indices[m,q] (MSB first; lookup_free_quantization.py:111,:129-158);
 * columns >= Q*nbits up to ldc are zero (GEMM K granule).  ldc % 8 == 0. */
int libra_lfq_codes(const int64_t* indices, void* codes, int64_t M, int64_t Q, int64_t nbits, int64_t ldc, void* stream);
/* GroupNorm(G groups, eps) statistics of x [B, HW, C] folded with gamma / beta into the per-(image, channel) affine
 * y = x * scale + shift (fp32 [B, C] each; Normalize = GroupNorm(32, eps 1e-6), diffusionmodules/model.py:34-35).
 * Deterministic two-stage reduction; C % 8 == 0, C <= 1024. */
size_t libra_groupnorm_workspace_bytes(int64_t B, int64_t HW, int64_t C);
int libra_groupnorm_affine(const void* x, const void* gamma, const void* beta, float* scale, float* shift, int64_t B, int64_t HW,
                           int64_t C, int64_t G, float eps, void* workspace, size_t workspace_bytes, void* stream);
/* Gathered conv operand: out[(b,y,x), tap*C + c] = f(x[b, sy, sx, c]) over the ksize x ksize window (ksize 1 or 3, zero padding)
 * of the nearest-neighbour upsampled image [H, W] of x [B, Hs, Ws, C] (source index = min(int(floorf(dst * inv_scale)), n-1),
 * PyTorch's rule; Upsample, model.py:38-56); f = identity, or GroupNorm affine (scale/shift [B,C], result rounded to bf16) then
 * optionally swish (x * sigmoid(x), model.py:28-31) with the reference's bf16 rounding points.  Columns up to ldo are zero. */
int libra_conv_gather(const void* x, void* out, int64_t ldo, const float* scale, const float* shift, int swish, int64_t B,
                      int64_t Hs, int64_t Ws, int64_t C, int64_t H, int64_t W, int64_t ksize, float inv_scale_h,
                      float inv_scale_w, void* stream);
/* x[r, :cols] <- softmax(bf16(x[r, :cols] * scale)), x[r, cols:ld] <- 0  (AttnBlock, model.py:170-196: bmm * c^-0.5, softmax) */
int libra_softmax_rows(void* x, int64_t rows, int64_t cols, int64_t ld, float scale, void* stream);

/* ---- image input pipeline (decoded uint8 image -> CLIP pixel_values / patch-embed operand) ---------------------------------
 * The reference's CPU chain: [expand2square] -> PIL BICUBIC resize of the shortest edge -> center crop -> /255 -> normalise -> CHW
 * (libra/models/clip/image_processing_clip.py:219-337; libra/data/datasets/caption_datasets.py:45-56), then `.to(bfloat16)`.
 * The resize is Pillow's two-pass 8-bit resampler; bounds [out][2] (first tap, tap count) and coeffs [out][ksize] (fixed point,
 * 22 fractional bits) are Pillow's `precompute_coeffs` + `normalize_coeffs_8bpc`, computed by the host. */
/* horizontal pass: canvas rows [row0, row0 + rows) -> out uint8 [rows, out_w, 3].  The canvas is the image [in_h, in_w, 3] at
 * offset (pad_y, pad_x) on a background of colour (bg_r, bg_g, bg_b) (expand2square); pad 0 = the image itself. */
int libra_resample_h_u8(const uint8_t* in, int64_t in_h, int64_t in_w, int64_t pad_y, int64_t pad_x, int bg_r, int bg_g, int bg_b,
                        const int32_t* bounds, const int32_t* coeffs, int64_t ksize, uint8_t* out, int64_t rows, int64_t out_w,
                        int64_t row0, void* stream);
/* vertical pass + center crop (window top/left/crop of the resized image) + normalisation through lut bf16 [3][256] (level ->
 * bf16((level/255 - mean) / std) with the reference's float32 rounding points) + layout: patch == 0 -> out bf16 [3, crop, crop];
 * patch > 0 -> the patch-embedding GEMM's im2col rows [(crop/patch)^2, kpad], column c*patch^2 + py*patch + px (pad columns
 * untouched: zero them once). */
int libra_resample_v_u8_norm(const uint8_t* tmp, int64_t tmp_w, int64_t row0, const int32_t* bounds, const int32_t* coeffs,
                             int64_t ksize, int64_t top, int64_t left, int64_t crop, const void* lut, void* out, int64_t patch,
                             int64_t kpad, void* stream);

/* ---- rank-8 bridge weight gradients -------------------------------------------------------------*/
/* out_m[j][c] = bf16( sum_{t < N, modality(t) = m} coef[t][j] * x[t][c] ), j < ncoef (8 or 16), c < C: the weight gradients of
 * Libra's rank-8 bridge LibraLinears (modeling_libra.py:150-189, :310-340) as one HBM pass over x instead of N = 8 GEMMs on
 * row-compacted copies.  dB (weight_B [H, 8]): x = dkb, coef = t_k, transpose_out = 1 (out[c][j], row stride ldo >= ncoef);
 * dA (weight_A [8, H]): x = h, coef = dt (16 columns: k rows 0-7, v rows 8-15), transpose_out = 0 (out[j][c], ldo >= C).
 * flag [N] (1 = vision token; NULL: every token counts for out_l); out_l / out_v may be NULL.  fp32 accumulation, deterministic.
 * workspace >= libra_rank_outer_wgrad_workspace_bytes(N, C, ncoef), 16-byte aligned. */
size_t libra_rank_outer_wgrad_workspace_bytes(int64_t N, int64_t C, int64_t ncoef);
int libra_rank_outer_wgrad(const void* x, int64_t ldx, const void* coef, int64_t ldcoef, int64_t ncoef, const uint8_t* flag,
                           void* out_l, void* out_v, int64_t ldo, int transpose_out, int64_t N, int64_t C, float* workspace,
                           size_t workspace_bytes, void* stream);

/* ---- optimizer -----------------------------------------------------------------------------------*/
/* Fused AdamW on a flat range of n elements (the data-parallel optimizer step of the reference's recipes: AdamW via HF
 * Trainer / DeepSpeed fused Adam with bf16 + fp32 master weights, libra/configs/libra_pretrain.yaml:83-91,
 * libra/configs/deepspeed_configs/ZeRO-2.json).  g = grad_scale * grad;  master -= lr*wd*master;
 * m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  master -= lr/bias_corr1 * m / (sqrt(v)/sqrt(bias_corr2) + eps);
 * param = bf16(master).  master / m / v fp32, grad / param bf16; all pointers 16-byte aligned.  One HBM pass (28 B/element).
 * grad_norm_sq (device scalar or NULL) + max_grad_norm: global-norm gradient clipping as torch.nn.utils.clip_grad_norm_
 * (HF Trainer `max_grad_norm: 1.0`, libra_pretrain.yaml / deepspeed_configs/ZeRO-2.json "gradient_clipping": "auto"):
 * *grad_norm_sq = squared norm of the UNSCALED gradient range(s); the applied gradients are grad_scale * grad, so grad_scale is
 * further multiplied by min(1, max_grad_norm / (|grad_scale| * sqrt(*grad_norm_sq) + 1e-6)), read on the device. */
int libra_adamw_step(float* master, float* m, float* v, const void* grad, void* param, int64_t n, float lr, float beta1,
                     float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2, float grad_scale,
                     const float* grad_norm_sq, float max_grad_norm, void* stream);

/* out[0] (+)= sum of squares of the bf16 range x[0..n) - the gradient-norm pass of the clipping above.  Deterministic
 * (fixed partition, fixed fold order); workspace >= libra_sumsq_workspace_bytes(n).  HBM-bound: 2 B / element. */
size_t libra_sumsq_workspace_bytes(int64_t n);
int libra_sumsq_bf16(const void* x, int64_t n, float* out, int accumulate, float* workspace, size_t workspace_bytes,
                     void* stream);

/* ---- compute-unit budget (data-parallel overlap) ------------------------------------------------------------
 * The reference overlaps DeepSpeed's reduce-scatter with backward (libra/configs/deepspeed_configs/ZeRO-2.json:15-21,
 * "overlap_comm": true); on MI355X RCCL's reduction kernels need compute units of their own while the GEMM / attention kernels
 * (one 128-KiB-LDS workgroup per CU) occupy all 256.  Two knobs, both off by default:
 *  - libra_stream_create_cu_reserved: a HIP stream whose kernels may use all CUs EXCEPT `reserve_cus` of them (a CU mask with the
 *    highest `reserve_cus` bits cleared: mask bit b is a CU of XCC b % 8, shader engine (b / 8) % 4 - measured,
 *    profiles/r05_cu_mask_layout.txt - so 8 = one CU per XCD).  The hardware deals workgroups to the shader engines of an XCC
 *    round-robin whatever their CU counts: a reserve that is not a multiple of 32 (one CU per SE per XCC) leaves SE 3 of every
 *    XCC short and a one-workgroup-per-CU grid waits for it (measured: GEMM +4 %, persistent attention forward +90 % at 8 reserved,
 *    +20 % at 32; profiles/r05_cu_budget_probe*.txt).  The caller runs the step on the stream (torch.cuda.ExternalStream) and
 *    leaves RCCL on its own unmasked stream.  *cus_out = CUs left.
 *  - libra_set_cu_budget: the PERSISTENT kernels (bridge attention forward / dQ pass) size their grid to `cus` workgroups
 *    instead of one per physical CU (0 = all).  Returns the previous value.  Thread-safe, takes effect at the next launch. */
int libra_stream_create_cu_reserved(int32_t reserve_cus, void** stream_out, int32_t* cus_out);
int libra_stream_destroy(void* stream);
int libra_set_cu_budget(int32_t cus);
int libra_get_cu_count(void);
/* Diagnostics of the above: a stream with a caller-given raw CU mask (nwords x 32 bits), and a kernel that records, per workgroup
 * (one per CU: 128 KiB of LDS each, resident for ~spin x 64 x 64 clocks), HW_REG_HW_ID and HW_REG_XCC_ID into out[2 b], out[2 b + 1]:
 * which physical CUs a mask enables (tools/cu_mask_probe.py -> profiles/r05_cu_mask_layout.txt). */
int libra_stream_create_cu_mask(const uint32_t* mask, int32_t nwords, void** stream_out);
int libra_debug_cu_map(uint32_t* out, int32_t nblocks, int32_t spin, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LIBRA_HIP_H */
