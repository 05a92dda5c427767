// Cycle probe for one wave's MFMA stream on gfx950: how many cycles per v_mfma_f32_32x32x16_bf16 does a single wave per SIMD sustain
// with (a) register operands, (b) LDS operand fragments read NF-1 MFMAs ahead (ds_read_b128 / ds_read_b64_tr_b16), with and without a
// VALU-only partner wave on the same SIMD.   hipcc -O3 --offload-arch=gfx950 mfma_probe.hip -o mfma_probe && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
#define LDSP __attribute__((address_space(3)))

template <int MODE, int NACC, int NF, int PARTNER, int PRIO = 0>
__global__ __launch_bounds__(512, 2) void probe(unsigned long long* out, float* sink, int reps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 65536 / 4; i += blockDim.x) ((unsigned*)smem)[i] = 0x3c003c00u + (i & 7);
    __syncthreads();
    const int l31 = lane & 31, fk = lane >> 5;
    int kb[4], vb[4];
    {
        const int kswz = (l31 >> 1) & 7;
        for (int j = 0; j < 4; ++j) kb[j] = l31 * 128 + (((2 * j + fk) ^ kswz) << 4);
        const int pp = lane & 15, g16 = (lane >> 4) & 1;
        const int vrow = (4 * fk + (pp >> 2)) * 256 + ((pp & 1) << 3);
        const int tlo = (2 * g16 + ((pp & 3) >> 1)) << 4;
        for (int dt = 0; dt < 4; ++dt) vb[dt] = vrow + ((((dt ^ (pp >> 2)) & 3) << 6) | tlo);
    }
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 q[8];
    for (int k = 0; k < 8; ++k) for (int e = 0; e < 8; ++e) q[k][e] = (short)(0x3c00 + lane + k);
    unsigned long long t0 = 0, t1 = 0;
    if (wave < 4) {
        if (PRIO) __builtin_amdgcn_s_setprio(1);
        t0 = __builtin_readcyclecounter();
        for (int rep = 0; rep < reps; ++rep) {
            const char* kimg = smem + (rep & 1) * 16384;
            const char* vimg = smem + 32768 + (rep & 1) * 16384;
            if constexpr (MODE == 0) {                         // register operands only
#pragma unroll
                for (int n = 0; n < 32; ++n) {
                    acc[n % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q[n & 7], q[(n + 1) & 7], acc[n % NACC], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                // MODE 1: 16 P.V-like (tr reads, 4 accs) then 16 Q.K-like (b128 reads, 2 accs)   [the kernel's M phase]
                // MODE 2: the same 32 MFMAs interleaved P.V / Q.K alternately
                // MODE 3: only the 16 Q.K-like, twice (2 accs, distance 2)
                // MODE 4: only the 16 P.V-like, twice
                bf16x8 F[NF];
                auto idx = [&](int n) -> int {                 // n-th MFMA of the phase -> logical op: 0-15 P.V (st, dt), 16-31 Q.K (ks, h)
                    if (MODE == 1) return n;
                    if (MODE == 2) return (n & 1) ? 16 + (n >> 1) : (n >> 1);
                    if (MODE == 3) return 16 + (n & 15);
                    return n & 15;
                };
                auto fread = [&](const int i) -> bf16x8 {
                    if (i < 16) {
                        const char* a = vimg + (i >> 2) * 4096 + vb[i & 3];
                        union { bf16x8 v; s16x4 h2[2]; } t;
                        t.h2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDSP s16x4*)(a));
                        t.h2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDSP s16x4*)(a + 2048));
                        return t.v;
                    }
                    const int ks = (i - 16) >> 1, hh = (i - 16) & 1;
                    return *(const bf16x8*)(kimg + kb[ks & 3] + hh * 8192 + (ks >> 2) * 4096);
                };
#pragma unroll
                for (int n = 0; n < NF; ++n) F[n] = fread(idx(n));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = 0; n < 32; ++n) {
                    const int i = idx(n);
                    if (i < 16) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[n % NF], q[i >> 2], acc[i & 3], 0, 0, 0);
                    else acc[NACC == 2 ? (i & 1) : ((i & 1) + 2 * ((i >> 1) & 1))] =
                        __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[n % NF], q[(i - 16) >> 1], acc[NACC == 2 ? (i & 1) : ((i & 1) + 2 * ((i >> 1) & 1))], 0, 0, 0);
                    if (n + NF < 32) F[n % NF] = fread(idx(n + NF));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        t1 = __builtin_readcyclecounter();
    } else if (PARTNER == 2) {                                 // the kernel's softmax step: 32 independent fma + exp, sums, 16 cvt_pk
        float sA[16], sB[16];
        for (int r = 0; r < 16; ++r) { sA[r] = 0.01f * (lane + r); sB[r] = 0.02f * (lane - r); }
        float l_run = 0.f;
        t0 = __builtin_readcyclecounter();
        for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(sA[r]), "+v"(sB[r]));
            float t = sA[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) t = fmaxf(t, fmaxf(sA[r], sB[r]));
            float ps[4] = {0.f, 0.f, 0.f, 0.f};
            float pA[16], pB[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pA[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sA[r], 0.12f, -t));
                pB[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sB[r], 0.12f, -t));
                ps[r & 1] += pA[r]; ps[2 + (r & 1)] += pB[r];
            }
            l_run += (ps[0] + ps[1]) + (ps[2] + ps[3]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                typedef __bf16 hb2 __attribute__((ext_vector_type(2))); typedef float hf2 __attribute__((ext_vector_type(2)));
                const hf2 v0 = {pA[2 * j], pA[2 * j + 1]}, v1 = {pB[2 * j], pB[2 * j + 1]};
                const unsigned u0 = __builtin_bit_cast(unsigned, __builtin_convertvector(v0, hb2)), u1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v1, hb2));
                asm volatile("" :: "v"(u0), "v"(u1));
            }
        }
        t1 = __builtin_readcyclecounter();
        if (l_run == 123.456f) sink[tid] = l_run;
    } else if (PARTNER) {                                      // VALU-only partner: exp / fma / add stream like a softmax
        float x[16];
        for (int r = 0; r < 16; ++r) x[r] = 0.001f * (lane + r);
        for (int rep = 0; rep < reps * 4; ++rep) {
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[r], 0.5f, -1.0f)) + x[(r + 1) & 15];
        }
        float s = 0; for (int r = 0; r < 16; ++r) s += x[r];
        if (s == 123.456f) sink[tid] = s;
    }
    float s = 0;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 123.456f) sink[tid] = s;
    if (lane == 0) out[blockIdx.x * 8 + wave] = (t1 - t0);
}

template <int MODE, int NACC, int NF, int PARTNER, int PRIO = 0>
void run(const char* name, int threads) {
    const int nblk = 256, reps = 200;
    unsigned long long* d; float* sink;
    hipMalloc(&d, nblk * 8 * 8); hipMalloc(&sink, 512 * 4);
    hipFuncSetAttribute((const void*)probe<MODE, NACC, NF, PARTNER, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((probe<MODE, NACC, NF, PARTNER, PRIO>), dim3(nblk), dim3(threads), 65536, 0, d, sink, reps);
    hipDeviceSynchronize();
    // tick-rate calibration: a long run of the same kernel timed with events
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int lreps = 8000;
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<MODE, NACC, NF, PARTNER, PRIO>), dim3(nblk), dim3(threads), 65536, 0, d, sink, lreps);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    {
        std::vector<unsigned long long> t(nblk * 8);
        hipMemcpy(t.data(), d, nblk * 8 * 8, hipMemcpyDeviceToHost);
        unsigned long long mx = 0; for (int b = 0; b < nblk; ++b) for (int w = 0; w < 4; ++w) mx = std::max(mx, t[b * 8 + w]);
        printf("    [long run: %.3f ms wall, %llu ticks in the slowest MFMA wave -> %.2f GHz tick rate if the kernel is that wave]\n", ms, mx, mx / (ms * 1e6));
    }
    hipLaunchKernelGGL((probe<MODE, NACC, NF, PARTNER, PRIO>), dim3(nblk), dim3(threads), 65536, 0, d, sink, reps);
    hipDeviceSynchronize();
    std::vector<unsigned long long> hh(nblk * 8), h, hp;
    hipMemcpy(hh.data(), d, nblk * 8 * 8, hipMemcpyDeviceToHost);
    for (int b = 0; b < nblk; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? h : hp).push_back(hh[b * 8 + w]);
    std::sort(h.begin(), h.end()); std::sort(hp.begin(), hp.end());
    // s_memtime / readcyclecounter ticks at a constant 100 MHz on some parts: report raw ticks per MFMA and let the reader compare modes
    printf("%-58s ticks/MFMA  min %.2f  med %.2f  max %.2f\n", name, h[0] / (reps * 32.0), h[h.size() / 2] / (reps * 32.0), h.back() / (reps * 32.0));
    if (PARTNER == 2) printf("%-58s   partner softmax step: med %.0f cyc\n", "", hp[hp.size() / 2] / (double)reps);
    hipFree(d); hipFree(sink);
}

int main() {
    run<0, 4, 6, 0>("regs only, 4 accumulators, 1 wave/SIMD", 256);
    run<0, 2, 6, 0>("regs only, 2 accumulators, 1 wave/SIMD", 256);
    run<0, 1, 6, 0>("regs only, 1 accumulator, 1 wave/SIMD", 256);
    run<1, 2, 6, 0>("M phase as shipped (PV then QK, NF=6), 1 wave/SIMD", 256);
    run<1, 2, 8, 0>("M phase, NF=8", 256);
    run<1, 4, 6, 0>("M phase, QK on 4 accumulators", 256);
    run<2, 2, 6, 0>("PV/QK interleaved, NF=6", 256);
    run<2, 2, 8, 0>("PV/QK interleaved, NF=8", 256);
    run<3, 2, 6, 0>("QK only (b128 reads, 2 acc)", 256);
    run<3, 4, 6, 0>("QK only (b128 reads, 4 acc)", 256);
    run<4, 4, 6, 0>("PV only (tr reads, 4 acc)", 256);
    run<1, 2, 6, 1>("M phase as shipped + VALU partner wave", 512);
    run<2, 2, 8, 1>("PV/QK interleaved NF=8 + VALU partner wave", 512);
    run<0, 4, 6, 1>("regs only 4 acc + VALU partner wave", 512);
    run<0, 4, 6, 2>("regs only 4 acc + softmax partner", 512);
    run<1, 2, 6, 2>("M phase NF=6 + softmax partner", 512);
    run<1, 2, 6, 2, 1>("M phase NF=6 prio 1 + softmax partner", 512);
    run<1, 2, 8, 2, 1>("M phase NF=8 prio 1 + softmax partner", 512);
    run<1, 2, 12, 2, 1>("M phase NF=12 prio 1 + softmax partner", 512);
    run<2, 2, 8, 2, 1>("interleaved NF=8 prio 1 + softmax partner", 512);
    return 0;
}
