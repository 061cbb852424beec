// VALU throughput of  x = q ^ t ; acc = bcnt(x) + acc  with independent xor operands, for different
// numbers of accumulator chains and different xor->bcnt distances (8 waves/SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
#define XOR(x, a, b) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "v"(a), "v"(b))
#define BCNT(acc, x) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(x))
template<int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters) {
    uint32_t a[4] = {0, 1, 2, 3}, q[8], t[8], y = blockIdx.x * 2654435761u + threadIdx.x;
    for (int i = 0; i < 8; i++) { q[i] = y * (i + 3); t[i] = y ^ (0x9E3779B9u * (i + 1)); }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 4; rep++) {
            if (MODE == 0) {          // one chain, xor immediately before its bcnt (the R=1 kernel today)
#pragma unroll
                for (int w = 0; w < 8; w++) { uint32_t x; XOR(x, q[w], t[(w + rep) & 7]); BCNT(a[0], x); }
            } else if (MODE == 1) {   // two chains alternating
#pragma unroll
                for (int w = 0; w < 8; w++) { uint32_t x; XOR(x, q[w], t[(w + rep) & 7]); BCNT(a[w & 1], x); }
            } else if (MODE == 2) {   // four chains
#pragma unroll
                for (int w = 0; w < 8; w++) { uint32_t x; XOR(x, q[w], t[(w + rep) & 7]); BCNT(a[w & 3], x); }
            } else if (MODE == 3) {   // one chain, all 8 xors first, then 8 bcnts
                uint32_t x[8];
#pragma unroll
                for (int w = 0; w < 8; w++) XOR(x[w], q[w], t[(w + rep) & 7]);
#pragma unroll
                for (int w = 0; w < 8; w++) BCNT(a[0], x[w]);
            } else if (MODE == 4) {   // one chain, xor kept two ahead of its bcnt
                uint32_t x[8];
                XOR(x[0], q[0], t[rep & 7]); XOR(x[1], q[1], t[(1 + rep) & 7]);
#pragma unroll
                for (int w = 0; w < 8; w++) { if (w + 2 < 8) XOR(x[w + 2], q[w + 2], t[(w + 2 + rep) & 7]); BCNT(a[0], x[w]); }
            } else if (MODE == 5) {   // two chains, xors first then bcnts
                uint32_t x[8];
#pragma unroll
                for (int w = 0; w < 8; w++) XOR(x[w], q[w], t[(w + rep) & 7]);
#pragma unroll
                for (int w = 0; w < 8; w++) BCNT(a[w & 1], x[w]);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a[0] + a[1] + a[2] + a[3];
}
template<int MODE> int run(uint32_t* out, const char* name) {
    const int iters = 4000, blocks = 256 * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) k<MODE><<<blocks, 256>>>(out, iters);
    CK(hipEventRecord(e0)); k<MODE><<<blocks, 256>>>(out, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-46s %.3f ms -> %.0f cyc per 64 VALU @2.38GHz (ideal 192)\n", name, ms, ms * 1e6 / iters / 8 * 2.38);
    return 0;
}
int main() { uint32_t* out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    run<2>(out, "warm"); run<0>(out, "1 chain, xor right before bcnt"); run<1>(out, "2 chains alternating"); run<2>(out, "4 chains");
    run<3>(out, "1 chain, 8 xors then 8 bcnts"); run<4>(out, "1 chain, xor two ahead"); run<5>(out, "2 chains, xors then bcnts"); return 0; }
