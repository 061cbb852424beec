// Does VALU throughput depend on operand data (switching activity)?  Same instruction stream, different data.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
// 8 "query" regs x 8 "train" regs per lane; inner body = the Hamming step without any memory op or filter
__global__ __launch_bounds__(256) void k(const uint32_t* __restrict__ in, uint32_t* out, int iters, unsigned long long* stamps) {
    uint32_t q[4][8], t[8], acc[4] = {0, 0, 0, 0};
    const int tid = blockIdx.x * 256 + threadIdx.x;
    for (int r = 0; r < 4; r++) for (int w = 0; w < 8; w++) q[r][w] = in[(tid * 32 + r * 8 + w) & 0xFFFFF];
    for (int w = 0; w < 8; w++) t[w] = in[(tid * 8 + w + 77777) & 0xFFFFF];
    const unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int w = 0; w < 8; w++)
#pragma unroll
                for (int r = 0; r < 4; r++) { uint32_t x = q[r][w] ^ t[(w + u) & 7]; asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[r]) : "v"(x)); }
        }
    }
    const unsigned long long st1 = __builtin_amdgcn_s_memtime(), sr1 = __builtin_amdgcn_s_memrealtime();
    out[tid] = acc[0] + acc[1] + acc[2] + acc[3];
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = st1 - st0; stamps[2 * blockIdx.x + 1] = sr1 - sr0; }
}
int main() {
    const int n = 1 << 20; std::vector<uint32_t> h(n);
    uint32_t *din, *dout; unsigned long long* ds; CK(hipMalloc(&din, n * 4)); CK(hipMalloc(&dout, 256 * 8 * 256 * 4)); CK(hipMalloc(&ds, 256 * 8 * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 4; mode++) {
        srand(1);
        for (auto& v : h) v = mode == 0 ? 0u : mode == 1 ? 0xFFFFFFFFu : mode == 2 ? (uint32_t)(rand() & 1 ? 0xFFFFFFFFu : 0u) : ((uint32_t)rand() << 16) ^ (uint32_t)rand();
        CK(hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice));
        for (int bpc : {2, 8}) {
            const int iters = 4000, blocks = 256 * bpc;
            k<<<blocks, 256>>>(din, dout, 10, ds);
            CK(hipEventRecord(e0)); k<<<blocks, 256>>>(din, dout, iters, ds); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> hs(2 * blocks); CK(hipMemcpy(hs.data(), ds, hs.size() * 8, hipMemcpyDeviceToHost));
            double ghz = 0; for (int i = 0; i < blocks; i++) ghz += (double)hs[2 * i] / (hs[2 * i + 1] * 10.0); ghz /= blocks;
            const double pairs = (double)iters * 4 * 4 * 64 * 4 * blocks;  // 4 u x 4 r per iter per lane
            printf("data=%s waves/SIMD=%d  %.3f ms  %.0f Gpairs/s-equivalent  clk %.2f GHz\n", mode == 0 ? "zeros " : mode == 1 ? "ones  " : mode == 2 ? "0/~0  " : "random", bpc, ms, pairs / (ms * 1e-3) / 1e9, ghz);
        }
    }
    return 0;
}
