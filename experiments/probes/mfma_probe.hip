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

template <int MODE, int NACC, int NF, bool PARTNER>
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
    if (lane == 0 && wave < 4) out[blockIdx.x * 4 + wave] = (t1 - t0);
}

template <int MODE, int NACC, int NF, bool PARTNER>
void run(const char* name, int threads) {
    const int nblk = 256, reps = 200;
    unsigned long long* d; float* sink;
    hipMalloc(&d, nblk * 4 * 8); hipMalloc(&sink, 512 * 4);
    hipFuncSetAttribute((const void*)probe<MODE, NACC, NF, PARTNER>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((probe<MODE, NACC, NF, PARTNER>), dim3(nblk), dim3(threads), 65536, 0, d, sink, reps);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nblk * 4);
    hipMemcpy(h.data(), d, nblk * 4 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    // s_memtime / readcyclecounter ticks at a constant 100 MHz on some parts: report raw ticks per MFMA and let the reader compare modes
    printf("%-58s ticks/MFMA  min %.2f  med %.2f  max %.2f\n", name, h[0] / (reps * 32.0), h[h.size() / 2] / (reps * 32.0), h.back() / (reps * 32.0));
    hipFree(d); hipFree(sink);
}

int main() {
    run<0, 4, 6, false>("regs only, 4 accumulators, 1 wave/SIMD", 256);
    run<0, 2, 6, false>("regs only, 2 accumulators, 1 wave/SIMD", 256);
    run<0, 1, 6, false>("regs only, 1 accumulator, 1 wave/SIMD", 256);
    run<1, 2, 6, false>("M phase as shipped (PV then QK, NF=6), 1 wave/SIMD", 256);
    run<1, 2, 8, false>("M phase, NF=8", 256);
    run<1, 4, 6, false>("M phase, QK on 4 accumulators", 256);
    run<2, 2, 6, false>("PV/QK interleaved, NF=6", 256);
    run<2, 2, 8, false>("PV/QK interleaved, NF=8", 256);
    run<3, 2, 6, false>("QK only (b128 reads, 2 acc)", 256);
    run<3, 4, 6, false>("QK only (b128 reads, 4 acc)", 256);
    run<4, 4, 6, false>("PV only (tr reads, 4 acc)", 256);
    run<1, 2, 6, true>("M phase as shipped + VALU partner wave", 512);
    run<2, 2, 8, true>("PV/QK interleaved NF=8 + VALU partner wave", 512);
    run<0, 4, 6, true>("regs only 4 acc + VALU partner wave", 512);
    return 0;
}
