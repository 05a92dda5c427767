// Multi-problem bf16 "NT" GEMM launch (gfx950): ONE persistent launch walks the 256x256 tiles of up to 12 INDEPENDENT problems
// (own shapes, strides, operand layouts, row maps and fused epilogue each) - round 6.
//
// Why: the routed decoder issues ~25 GEMMs per layer and direction, and 17 of them are the low-rank vision branch
// (LibraLinear pairs, modeling_libra.py:167-199: M = 4 624 vision rows, K or N = 1 024 / 2 752): 0.3 - 1.9 waves of 256^2 tiles on
// 256 CUs, each launch paying its own partly filled last wave, its own ramp and its own drain (BENCH r05: 126 ms per step at
// 0.385 of the MFMA peak against 0.535 for the text shapes).  A text GEMM and the vision GEMMs that do not depend on it are
// independent work (cal_language_vision routes disjoint rows, modeling_libra.py:111-147), and a weight gradient has no consumer
// inside its layer at all.  Here they share one tile list:
//   * problems are listed longest K first, so the list ends with the cheap tiles: the unbalanced tail of the launch costs a
//     fraction of the SHORTEST tile, not a wave of the longest (dynamic longest-processing-time-first);
//   * tiles are handed out from eight queues - queue x holds the entries t = x (mod 8), its consumers are the workgroups
//     b = x (mod 8), i.e. the workgroups of XCD x under the hardware's round-robin placement - so consecutive entries of a queue
//     go to one XCD at the same time exactly as hardware dispatch would place them, and tile_order's compact per-XCD patches
//     (hip_common.hpp) keep their L2 sharing.  Every problem starts at a multiple of 8 entries for that reason;
//   * the first entry of a workgroup is static (t = blockIdx.x); each later one is fetched by ONE returning atomic that is
//     issued before the tile's prologue loads and read after its epilogue: no fetch latency on the critical path.  (The atomic
//     is the oldest VMEM operation of the tile, so the prologue's own counted wait covers it: it is issued from inline asm like
//     the LDS-DMA, DESIGN rule 1, and the compiler's waitcnt pass never sees it.)
//   * the queue words live in a 128-byte caller-owned workspace that is all zero between launches: the last workgroup to leave
//     clears it (no memset node per launch, no library-owned global state).  One workspace per stream.
//   * a problem may READ the output of one other problem of the same launch (wait_on): its tiles are listed after the producer's,
//     the producer's tiles count themselves done behind an agent-scope release, and a consumer tile that comes up early waits for
//     the count behind an acquire (bounded: a wait that runs out sets a sticky word of the workspace instead of hanging the
//     chip).  That folds the first low-rank stage of a LibraLinear pair (x A^T, 76 tiles = 0.3 waves as a launch of its own)
//     into the launch that holds its second stage and the text GEMM beside it.  Every workgroup is resident (grid <= compute
//     units, one workgroup each) and producers precede consumers in every queue, so a waiting tile always waits for work that is
//     running or already done.
// The tile itself is gemm256_body.hpp - the same code, bit for bit, as gemm_bf16_nt_256_kernel; the operand layout (A_T / B_T)
// is a per-problem, wave-uniform branch around it.  Results do not depend on which workgroup computes a tile.
#include <algorithm>
#include <atomic>
#include <cstddef>
#include "gemm256_body.hpp"

namespace libra {

constexpr int MULTI_MAXP = LIBRA_GEMM_MULTI_MAX;

struct GemmProb {                  // device view of one problem (kernel-argument resident, read with scalar loads)
    const bf16_t* A; const bf16_t* B; bf16_t* C;
    const bf16_t* bias; const bf16_t* resid; const bf16_t* aux; bf16_t* preact;
    const int* a_rows; const int* c_rows;
    int lda, ldb, ldc, ldr, ldaux, ldpre;
    int M, N, K;
    int tiles_m, tiles_n;
    int tile0;                     // first entry of this problem in the launch's tile list (a multiple of 8)
    float alpha; int alpha_cols;
    int flags;
    int splitk;                    // K slices (1 = none): slice ky of tile b is entry ky * tiles + b and writes slab[ky]
    float* slab;                   // fp32 [splitk][M][N] partial results of a split problem (reduced by a second kernel)
    int wait_on;                   // launch-order index of the problem whose C this problem reads (-1: none) ...
    int wait_need;                 // ... and how many finished tiles of it make C complete
    int signal;                    // some problem of the launch waits on this one: count finished tiles in queue[16 + own index]
    int pad_;
};
static_assert(sizeof(GemmProb) == 160, "GemmProb layout");
static_assert(sizeof(libra_gemm_problem) == 192, "libra_gemm_problem layout (tests/test_cabi_cpu.py checks the host side against it)");
struct GemmMultiArgs {
    GemmProb prob[MULTI_MAXP];
    int tile0[MULTI_MAXP];         // = prob[i].tile0 (unused slots: INT_MAX): the owner scan reads these with constant indices
    unsigned* queue;               // [0..7] entries taken from queue x beyond the static first ones, [8] workgroups done,
                                   // [10] sticky: a bounded wait ran out, [16 + i] finished tiles of problem i (if signalled)
    int nprob; int nentries;
};
static_assert(offsetof(GemmMultiArgs, prob) == 0, "the kernel reads prob[] at the start of the kernel-argument segment");

typedef const __attribute__((address_space(4))) GemmProb* ProbPtr;
constexpr unsigned MULTI_SPIN_MAX = 1u << 21;   // x ~1 us of s_sleep: seconds, then give up loudly (queue[10]) instead of hanging

// VARIANTS: bit (2 at + bt) set = that operand layout may occur in the launch (the launcher picks the smallest superset: a body
// that cannot occur is not compiled in)
template <int VARIANTS>
__global__ __launch_bounds__(G256_THREADS, 2) void gemm_bf16_multi_kernel(const GemmMultiArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* slot = (unsigned*)(smem + G256_LDS);              // next entry, thread 0 -> everyone
    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int xcd = blockIdx.x & 7;
    const int static_x = ((int)gridDim.x - xcd + 7) >> 3;       // queue x: its first static_x entries are the static first tiles
    const int nentries = p.nentries;
    unsigned* const qx = p.queue + xcd;
    const ProbPtr probs = (ProbPtr)__builtin_amdgcn_kernarg_segment_ptr();

#pragma unroll 1
    for (int t = blockIdx.x; t < nentries;) {
        // ---- fetch the NEXT entry now (returns the queue's old count), read after the tile
        unsigned got = 0;
        if (tid0 == 0) {
            const unsigned one = 1;
            asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(got) : "v"(qx), "v"(one) : "memory");
        }
        // ---- which problem owns entry t (problems are few: a scalar scan)
        int g = 0;
#pragma unroll
        for (int i = 1; i < MULTI_MAXP; ++i) g = (t >= p.tile0[i]) ? i : g;
        g = __builtin_amdgcn_readfirstlane(g);
        const ProbPtr q = probs + g;
        const int bid = t - q->tile0;
        const int tiles_m = q->tiles_m, tiles_n = q->tiles_n;
        const int ntile = tiles_m * tiles_n, sk = q->splitk;
        if (bid < ntile * sk) {                                 // (else: one of the <= 7 pad entries behind a problem)
            const int ky = sk > 1 ? bid / ntile : 0;            // slice-major: consecutive entries = neighbouring tiles of one slice
            const int bt_ = bid - ky * ntile;
            const int wait_on = q->wait_on;
            if (wait_on >= 0) {                                 // this problem reads another one's C: all of the producer's tiles done?
                if (tid0 == 0) {
                    const unsigned need = (unsigned)q->wait_need;
                    unsigned spins = 0;
                    while (__hip_atomic_load(p.queue + 16 + wait_on, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                        if (++spins > MULTI_SPIN_MAX) { __hip_atomic_store(p.queue + 10, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                        __builtin_amdgcn_s_sleep(32);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // (cache-level: the whole workgroup reads behind the barrier)
                }
                __syncthreads();
            }
            Gemm256Args a;
            a.A = q->A; a.B = q->B; a.C = q->C;
            a.bias = q->bias; a.resid = q->resid; a.aux = q->aux; a.preact = q->preact;
            a.a_rows = q->a_rows; a.c_rows = q->c_rows;
            a.lda = q->lda; a.ldb = q->ldb; a.ldc = q->ldc; a.ldr = q->ldr; a.ldaux = q->ldaux; a.ldpre = q->ldpre;
            a.M = q->M; a.N = q->N; a.K = q->K;
            a.tiles_m = tiles_m; a.tiles_n = tiles_n;
            a.alpha = q->alpha; a.alpha_cols = q->alpha_cols;
            a.flags = q->flags;
            a.slab = q->slab; a.splitk = sk;
            const int v = ((a.flags & LIBRA_GEMM_A_T) ? 2 : 0) | ((a.flags & LIBRA_GEMM_B_T) ? 1 : 0);
            if ((VARIANTS & 1) && v == 0) gemm256_tile<false, false>(a, a.A, a.B, a.C, bt_, ky, smem, tid0, wave, wr, wc);
            if ((VARIANTS & 2) && v == 1) gemm256_tile<false, true>(a, a.A, a.B, a.C, bt_, ky, smem, tid0, wave, wr, wc);
            if ((VARIANTS & 4) && v == 2) gemm256_tile<true, false>(a, a.A, a.B, a.C, bt_, ky, smem, tid0, wave, wr, wc);
            if ((VARIANTS & 8) && v == 3) gemm256_tile<true, true>(a, a.A, a.B, a.C, bt_, ky, smem, tid0, wave, wr, wc);
            if (q->signal) {                                    // someone waits for this problem: publish the tile, then count it
                __syncthreads();                                // every wave's stores are issued and complete (workgroup release)
                if (tid0 == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    __hip_atomic_fetch_add(p.queue + 16 + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        if (tid0 == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (after a tile: long complete; a pad entry: the fetch's round trip)
            slot[0] = got;
        }
        __syncthreads();                                        // slot visible; the epilogue's LDS slabs are free again
        t = (static_x + (int)slot[0]) * 8 + xcd;
        t = __builtin_amdgcn_readfirstlane(t);
        __syncthreads();                                        // everyone has read the slot before thread 0 rewrites it
    }
    // ---- leave the workspace as it was found: the last workgroup out clears the queues
    if (tid0 == 0) {
        const unsigned done = atomicAdd(p.queue + 8, 1u);
        if (done == gridDim.x - 1) {
#pragma unroll
            for (int i = 0; i < 9; ++i) __hip_atomic_store(p.queue + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int i = 0; i < MULTI_MAXP; ++i) __hip_atomic_store(p.queue + 16 + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace libra

using namespace libra;

extern "C" int libra_splitk_reduce_launch_(const float* slab, int S, int64_t M, int64_t N, void* C, int64_t ldc, const int* c_rows,
                                           const void* resid, int64_t ldr, void* stream);

extern "C" int libra_gemm_bf16_multi(const libra_gemm_problem* probs, int64_t nprob, void* queue_ws, void* stream) {
    if (nprob <= 0) return LIBRA_OK;
    if (!probs || nprob > MULTI_MAXP) return LIBRA_ERR_SHAPE;
    if (!queue_ws || ((uintptr_t)queue_ws & 15)) return LIBRA_ERR_ALIGN;
    int order[MULTI_MAXP], n = 0;
    int variants = 0;
    for (int i = 0; i < (int)nprob; ++i) {
        const libra_gemm_problem& q = probs[i];
        const int64_t M = q.M, N = q.N, K = q.K;
        if (M <= 0 || N <= 0) continue;                              // empty problem: nothing to do
        const int flags = q.flags;
        const int at = (flags & LIBRA_GEMM_A_T) ? 1 : 0, bt = (flags & LIBRA_GEMM_B_T) ? 1 : 0;
        // the checks of libra_gemm_bf16_nt_routed (gemm_bf16.hip gemm_run), per problem
        if (!q.A || !q.B || !q.C || K <= 0 || (K % 64) != 0) return LIBRA_ERR_SHAPE;
        if ((q.lda % 8) || (q.ldb % 8) || q.ldc < N) return LIBRA_ERR_SHAPE;
        if (q.a_rows && (at || q.a_phys_rows <= 0)) return LIBRA_ERR_SHAPE;
        const int64_t arows = q.a_rows ? q.a_phys_rows : M;
        if (at ? (q.lda < M || (M % 8) || K * q.lda >= (1LL << 31)) : (q.lda < K || arows * q.lda >= (1LL << 31))) return LIBRA_ERR_SHAPE;
        if (bt ? (q.ldb < N || (N % 8) || K * q.ldb >= (1LL << 31)) : (q.ldb < K || N * q.ldb >= (1LL << 31))) return LIBRA_ERR_SHAPE;
        if (((uintptr_t)q.A | (uintptr_t)q.B | (uintptr_t)q.C) & 15) return LIBRA_ERR_ALIGN;
        if (q.ldc % 8) return LIBRA_ERR_ALIGN;
        if ((flags & LIBRA_GEMM_BIAS) && (!q.bias || ((uintptr_t)q.bias & 15))) return LIBRA_ERR_ALIGN;
        if ((flags & LIBRA_GEMM_RESIDUAL) && (!q.resid || (q.ldr % 8) || ((uintptr_t)q.resid & 15))) return LIBRA_ERR_ALIGN;
        if ((flags & LIBRA_GEMM_MUL_QGELU_GRAD) && (!q.aux || (q.ldaux % 8) || ((uintptr_t)q.aux & 15))) return LIBRA_ERR_ALIGN;
        if ((flags & LIBRA_GEMM_STORE_PREACT) && (!q.preact || (q.ldpre % 8) || q.ldpre < N || ((uintptr_t)q.preact & 15))) return LIBRA_ERR_ALIGN;
        if (M > (1 << 30) || N > (1 << 30) || K > (1 << 30)) return LIBRA_ERR_SHAPE;
        if (q.ldc >= (1LL << 31) || q.ldr >= (1LL << 31) || q.ldaux >= (1LL << 31) || q.ldpre >= (1LL << 31)) return LIBRA_ERR_SHAPE;
        if (q.splitk > 1) {            // K slices: raw fp32 slabs + the deterministic reduction; only a residual may be fused (as libra_gemm_bf16_nt_splitk_routed)
            if (q.splitk > K / 64 || q.splitk > 64 || (N % 8)) return LIBRA_ERR_SHAPE;
            if (flags & ~(LIBRA_GEMM_A_T | LIBRA_GEMM_B_T | LIBRA_GEMM_RESIDUAL)) return LIBRA_ERR_SHAPE;
            if (!q.slab || ((uintptr_t)q.slab & 15)) return LIBRA_ERR_ALIGN;
        } else if (q.splitk < 0) return LIBRA_ERR_SHAPE;
        if (q.wait_on >= 0) {          // reads the C of problem wait_on: an unsplit, non-empty problem of this call that waits for nothing itself
            if (q.wait_on >= nprob || q.wait_on == i) return LIBRA_ERR_SHAPE;
            const libra_gemm_problem& w = probs[q.wait_on];
            if (w.wait_on >= 0 || w.splitk > 1 || w.M <= 0 || w.N <= 0) return LIBRA_ERR_SHAPE;
        }
        order[n++] = i;
        variants |= 1 << (2 * at + bt);
    }
    if (n == 0) return LIBRA_OK;
    // longest tiles first (stable: equal K keeps the caller's order)
    auto slice_k = [&](int x) { return probs[x].K / (probs[x].splitk > 1 ? probs[x].splitk : 1); };
    std::stable_sort(order, order + n, [&](int x, int y) { return slice_k(x) > slice_k(y); });
    // a consumer goes behind its producer (producers wait for nothing: one pass; it keeps its place among the later problems)
    for (int j = 0; j < n; ++j) {
        const int c = order[j], w = (int)probs[c].wait_on;
        if (w < 0) continue;
        int pw = -1;
        for (int k = 0; k < n; ++k) if (order[k] == w) pw = k;
        if (pw > j) {                                   // producer currently later: rotate the consumer to just behind it
            for (int k = j; k < pw; ++k) order[k] = order[k + 1];
            order[pw] = c;
            --j;                                        // the element that moved into slot j has not been looked at
        }
    }
    int pos[MULTI_MAXP];
    for (int i = 0; i < MULTI_MAXP; ++i) pos[i] = -1;
    for (int j = 0; j < n; ++j) pos[order[j]] = j;
    GemmMultiArgs a;
    long entries = 0;
    for (int j = 0; j < n; ++j) {
        const libra_gemm_problem& q = probs[order[j]];
        GemmProb& d = a.prob[j];
        d.A = (const bf16_t*)q.A; d.B = (const bf16_t*)q.B; d.C = (bf16_t*)q.C;
        d.bias = (const bf16_t*)q.bias; d.resid = (const bf16_t*)q.resid; d.aux = (const bf16_t*)q.aux; d.preact = (bf16_t*)q.preact;
        d.a_rows = q.a_rows; d.c_rows = q.c_rows;
        d.lda = (int)q.lda; d.ldb = (int)q.ldb; d.ldc = (int)q.ldc; d.ldr = (int)q.ldr; d.ldaux = (int)q.ldaux; d.ldpre = (int)q.ldpre;
        d.M = (int)q.M; d.N = (int)q.N; d.K = (int)q.K;
        d.tiles_m = (int)((q.M + 255) / 256); d.tiles_n = (int)((q.N + 255) / 256);
        d.tile0 = (int)entries;
        d.alpha = q.alpha; d.alpha_cols = (int)q.alpha_cols;
        d.flags = q.flags;
        d.splitk = q.splitk > 1 ? (int)q.splitk : 1;
        d.slab = d.splitk > 1 ? (float*)q.slab : nullptr;
        d.wait_on = q.wait_on >= 0 ? pos[q.wait_on] : -1;
        d.wait_need = 0; d.signal = 0; d.pad_ = 0;
        entries += ((long)d.tiles_m * d.tiles_n * d.splitk + 7) / 8 * 8;
        if (entries > 0x3fffffffL) return LIBRA_ERR_SHAPE;
    }
    for (int j = 0; j < n; ++j)
        if (a.prob[j].wait_on >= 0) {
            GemmProb& w = a.prob[a.prob[j].wait_on];
            w.signal = 1;
            a.prob[j].wait_need = w.tiles_m * w.tiles_n;
        }
    for (int j = n; j < MULTI_MAXP; ++j) { a.prob[j] = GemmProb{}; a.prob[j].tile0 = 0x7fffffff; a.prob[j].wait_on = -1; }
    for (int j = 0; j < MULTI_MAXP; ++j) a.tile0[j] = a.prob[j].tile0;
    a.queue = (unsigned*)queue_ws; a.nprob = n; a.nentries = (int)entries;

    // the smallest compiled superset of the operand layouts that occur
    static const int sets[] = {1, 2, 8, 10, 15};
    int set = 15;
    for (int s : sets) if ((variants & ~s) == 0) { set = s; break; }
    void (*kern)(const GemmMultiArgs) = set == 1 ? gemm_bf16_multi_kernel<1> : set == 2 ? gemm_bf16_multi_kernel<2>
                                      : set == 8 ? gemm_bf16_multi_kernel<8> : set == 10 ? gemm_bf16_multi_kernel<10>
                                                                                         : gemm_bf16_multi_kernel<15>;
    constexpr int LDS = G256_LDS + 64;
    static std::atomic<bool> attr_set[16];          // zero-initialised; idempotent call, atomic so concurrent first launches do not race
    if (!attr_set[set]) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set[set] = true;
    }
    long nblk = persistent_grid(entries, 8);        // one workgroup per (budgeted) CU, a multiple of 8; entries is one too
    if (nblk < 8) nblk = 8;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(G256_THREADS), LDS, (hipStream_t)stream, a);
    if (hipGetLastError() != hipSuccess) return LIBRA_ERR_LAUNCH;
    for (int j = 0; j < n; ++j) {                   // second stage of the split problems
        const libra_gemm_problem& q = probs[order[j]];
        if (q.splitk > 1) {
            const int rc = libra_splitk_reduce_launch_((const float*)q.slab, (int)q.splitk, q.M, q.N, q.C, q.ldc, q.c_rows,
                                                       (q.flags & LIBRA_GEMM_RESIDUAL) ? q.resid : nullptr, q.ldr, stream);
            if (rc != LIBRA_OK) return rc;
        }
    }
    return LIBRA_OK;
}
