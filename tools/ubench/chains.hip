// VALU throughput of the xor+bcnt stream vs number of independent accumulator chains per wave, 8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
template<int C, int W>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters) {
    uint32_t a[C], q[8], y = blockIdx.x * 2654435761u + threadIdx.x;
    for (int i = 0; i < C; i++) a[i] = i;
    for (int i = 0; i < 8; i++) q[i] = y * (i + 3);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 32 / C; u++)
#pragma unroll
            for (int c = 0; c < C; c++) { uint32_t x; asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "v"(q[(u + c) & 7]), "v"(a[(c + 1) % C])); asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[c]) : "v"(x)); }
    }
    uint32_t s = 0; for (int i = 0; i < C; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template<int C, int W> int run(uint32_t* out) {
    const int iters = 4000, blocks = 256 * W;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) k<C, W><<<blocks, 256>>>(out, iters);
    CK(hipEventRecord(e0)); k<C, W><<<blocks, 256>>>(out, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("chains=%d waves/SIMD=%d: %.3f ms  -> %.1f ns per 64 VALU per wave-slot = %.0f cyc @2.38GHz (ideal 192)\n", C, W, ms, ms * 1e6 / iters / W, ms * 1e6 / iters / W * 2.38);
    return 0;
}
int main() { uint32_t* out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    run<4, 8>(out); run<1, 8>(out); run<2, 8>(out); run<4, 8>(out); run<8, 8>(out); run<1, 4>(out); run<2, 4>(out); run<4, 4>(out); run<1, 2>(out); run<4, 2>(out); return 0; }
