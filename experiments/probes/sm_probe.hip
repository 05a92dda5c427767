// Cycle probe for the softmax (VALU) phase of the bridge-attention forward on gfx950: cycles per 32-score online-softmax step of one
// wave, alone on its SIMD and beside a partner wave that streams MFMAs (at s_setprio 0 / 1).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float hw_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) { const hw_f32x2 v = {lo, hi}; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hw_bf16x2)); }
__device__ __forceinline__ float max3f(float a, float b, float c) { float d; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ float half_swap_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// VAR 0: the kernel's softmax as shipped (serial max chain, 2 sum chains); 1: tree max + 4 sum chains; 2: exp only (32 v_exp + 32 fma);
// 3: everything but the exps (exp replaced by a multiply)
template <int VAR, int MF, int PRIO>
__global__ __launch_bounds__(512, 2) void probe(unsigned long long* out, float* sink, int reps) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned long long t0 = 0, t1 = 0;
    float acc_sink = 0.f;
    if (wave >= 4) {
        f32x16 sA, sB;
        for (int r = 0; r < 16; ++r) { sA[r] = 0.01f * (lane + r); sB[r] = 0.02f * (lane - r); }
        float m_run = -1e30f, l_run = 0.f; const float sl2 = 0.1275f;
        f32x16 o[4];
        for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) o[a][r] = 1.f;
        unsigned pk[16];
        t0 = __builtin_readcyclecounter();
        for (int rep = 0; rep < reps; ++rep) {
            asm volatile("" : "+v"(sA), "+v"(sB));
            float tmax;
            if (VAR == 1) {
                float a0 = max3f(sA[0], sA[1], sA[2]), a1 = max3f(sA[3], sA[4], sA[5]), a2 = max3f(sA[6], sA[7], sA[8]), a3 = max3f(sA[9], sA[10], sA[11]);
                float a4 = max3f(sA[12], sA[13], sA[14]), a5 = max3f(sA[15], sB[0], sB[1]), a6 = max3f(sB[2], sB[3], sB[4]), a7 = max3f(sB[5], sB[6], sB[7]);
                float a8 = max3f(sB[8], sB[9], sB[10]), a9 = max3f(sB[11], sB[12], sB[13]), a10 = fmaxf(sB[14], sB[15]);
                a0 = max3f(a0, a1, a2); a3 = max3f(a3, a4, a5); a6 = max3f(a6, a7, a8); a9 = fmaxf(a9, a10);
                tmax = fmaxf(max3f(a0, a3, a6), a9);
            } else {
                tmax = max3f(sA[0], sA[1], sB[0]);
#pragma unroll
                for (int r = 2; r < 16; r += 2) tmax = max3f(tmax, sA[r], sA[r + 1]);
#pragma unroll
                for (int r = 1; r < 15; r += 2) tmax = max3f(tmax, sB[r], sB[r + 1]);
                tmax = fmaxf(tmax, sB[15]);
            }
            tmax = half_swap_max(tmax * sl2);
            const float m_new = fmaxf(m_run, tmax);
            if (__any(m_new > m_run + 8.f)) {
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
                m_run = m_new;
            }
            const float nm = -m_run;
            float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (VAR == 3) { sA[r] = __builtin_fmaf(sA[r], sl2, nm) * 0.5f; sB[r] = __builtin_fmaf(sB[r], sl2, nm) * 0.5f; }
                else { sA[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sA[r], sl2, nm)); sB[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sB[r], sl2, nm)); }
                if (VAR != 2) {
                    if (VAR == 1) { ps[r & 1] += sA[r]; ps[2 + (r & 1)] += sB[r]; } else { ps[0] += sA[r]; ps[1] += sB[r]; }
                }
            }
            l_run += (ps[0] + ps[1]) + (ps[2] + ps[3]);
            if (VAR != 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { pk[j] = pack2bf(sA[2 * j], sA[2 * j + 1]); pk[8 + j] = pack2bf(sB[2 * j], sB[2 * j + 1]); }
#pragma unroll
                for (int j = 0; j < 16; ++j) asm volatile("" :: "v"(pk[j]));
            } else {
                asm volatile("" :: "v"(sA), "v"(sB));
            }
        }
        t1 = __builtin_readcyclecounter();
        for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc_sink += o[a][r];
        acc_sink += l_run + m_run;
    } else if (MF) {
        f32x16 acc[4];
        for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
        bf16x8 q[8];
        for (int k = 0; k < 8; ++k) for (int e = 0; e < 8; ++e) q[k][e] = (short)(0x3c00 + lane + k);
        if (PRIO) __builtin_amdgcn_s_setprio(1);
        t0 = __builtin_readcyclecounter();
        for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
            for (int n = 0; n < 32; ++n) {
                acc[n & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q[n & 7], q[(n + 1) & 7], acc[n & 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        t1 = __builtin_readcyclecounter();
        for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc_sink += acc[a][r];
    }
    if (acc_sink == 123.456f) sink[tid] = acc_sink;
    if (lane == 0) out[blockIdx.x * 8 + wave] = (t1 - t0);
}

template <int VAR, int MF, int PRIO>
void run(const char* name) {
    const int nblk = 256, reps = 200;
    unsigned long long* d; float* sink;
    (void)hipMalloc(&d, nblk * 8 * 8); (void)hipMalloc(&sink, 512 * 4);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((probe<VAR, MF, PRIO>), dim3(nblk), dim3(512), 0, 0, d, sink, reps);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(nblk * 8);
    (void)hipMemcpy(h.data(), d, nblk * 8 * 8, hipMemcpyDeviceToHost);
    std::vector<double> sm, mf;
    for (int b = 0; b < nblk; ++b) for (int w = 0; w < 8; ++w) (w >= 4 ? sm : mf).push_back(h[b * 8 + w] / (double)reps);
    std::sort(sm.begin(), sm.end()); std::sort(mf.begin(), mf.end());
    printf("%-62s softmax step: med %7.0f cyc   | partner 32 MFMAs: med %7.0f cyc\n", name, sm[sm.size() / 2], mf[mf.size() / 2]);
    (void)hipFree(d); (void)hipFree(sink);
}

int main() {
    run<0, 0, 0>("as shipped, alone");
    run<1, 0, 0>("tree max + 4 sum chains, alone");
    run<2, 0, 0>("32 fma + 32 exp only, alone");
    run<3, 0, 0>("no exps (multiply instead), alone");
    run<0, 1, 0>("as shipped, MFMA partner prio 0");
    run<0, 1, 1>("as shipped, MFMA partner prio 1");
    run<1, 1, 0>("tree max + 4 sum chains, MFMA partner prio 0");
    run<1, 1, 1>("tree max + 4 sum chains, MFMA partner prio 1");
    run<2, 1, 0>("32 fma + 32 exp only, MFMA partner prio 0");
    return 0;
}
