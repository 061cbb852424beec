// Does LDS read-return traffic into VGPRs slow a VALU-bound stream on the same SIMD?
// Per iteration: 64 VALU ops (32 v_xor + 32 v_bcnt, 4 independent chains) and M broadcast ds_read_b128.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
template<int M, int WIDTH>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters) {
    __shared__ uint4 lds[512];
    lds[threadIdx.x] = make_uint4(threadIdx.x, 1, 2, 3); lds[threadIdx.x + 256] = make_uint4(5, 6, 7, threadIdx.x);
    __syncthreads();
    uint32_t a[4], q[8], y = blockIdx.x * 2654435761u + threadIdx.x;
    for (int i = 0; i < 4; i++) a[i] = i;
    for (int i = 0; i < 8; i++) q[i] = y * (i + 3);
    uint32_t addr = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)lds;
    asm volatile("" : "+v"(addr));
    uint32_t sink = 0;
    for (int it = 0; it < iters; it++) {
        uint4 r0, r1, r2, r3;
        if (M >= 1) { if (WIDTH == 16) asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(r0) : "v"(addr)); else asm volatile("ds_read_b32 %0, %1 offset:0" : "=v"(r0.x) : "v"(addr)); }
        if (M >= 2) { if (WIDTH == 16) asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(r1) : "v"(addr)); else asm volatile("ds_read_b32 %0, %1 offset:16" : "=v"(r1.x) : "v"(addr)); }
        if (M >= 3) asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(r2) : "v"(addr));
        if (M >= 4) asm volatile("ds_read_b128 %0, %1 offset:48" : "=v"(r3) : "v"(addr));
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int c = 0; c < 4; c++) { uint32_t x; asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "v"(q[u]), "v"(a[c])); asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[c]) : "v"(x)); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (M >= 1) asm volatile("" :: "v"(r0.x));
        if (M >= 2) asm volatile("" :: "v"(r1.x));
        if (M >= 3) asm volatile("" :: "v"(r2.x));
        if (M >= 4) asm volatile("" :: "v"(r3.x));
    }
    out[blockIdx.x * 256 + threadIdx.x] = a[0] + a[1] + a[2] + a[3] + sink;
}
template<int M, int WIDTH> int run(uint32_t* out) {
    const int iters = 4000, blocks = 256 * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<M, WIDTH><<<blocks, 256>>>(out, 100);
    CK(hipEventRecord(e0)); k<M, WIDTH><<<blocks, 256>>>(out, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("ds_read_%s x%d per 64 VALU: %.3f ms  (%.2f ns per iteration per SIMD-wave-slot)\n", WIDTH == 16 ? "b128" : "b32 ", M, ms, ms * 1e6 / iters / 8);
    return 0;
}
int main() { uint32_t* out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    run<0, 16>(out); run<1, 16>(out); run<2, 16>(out); run<3, 16>(out); run<4, 16>(out); run<1, 4>(out); run<2, 4>(out); run<0, 16>(out); return 0; }
