// Hardware-semantics probe (run on the GPU box): prints the lane/element mapping of ds_read_b64_tr_b16
// so the transposing-LDS-read GEMM variants can be written against measured, not assumed, behaviour.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned short* out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[2048];
    int l = threadIdx.x;
    for (int i = l; i < 2048; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned short*)lds;   // escape the array
    unsigned addr;
    if (mode == 0) addr = l * 8;                                   // lane i -> elements 4i..4i+3
    else if (mode == 1) addr = (l & 15) * 128 + (l >> 4) * 8;      // 16 rows of 64 elements; group g -> cols 4g..
    else addr = ((l & 3) * 4 + ((l >> 2) & 3) * 64 + (l >> 4) * 256) * 2;
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr + base) : "memory");
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)(v >> (16 * j));
}
int main() {
    unsigned short* d; hipError_t e = hipMalloc(&d, 64 * 4 * 2); printf("malloc: %s\n", hipGetErrorString(e));
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
        std::vector<unsigned short> h(256);
        e = hipDeviceSynchronize(); printf("sync: %s\n", hipGetErrorString(e));
        e = hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost); printf("copy: %s\n", hipGetErrorString(e));
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    }
    return 0;
}
