// Micro-benchmark: does the physical VGPR number (bank = index mod 4?) of the two source operands change the issue
// rate of v_xor_b32 (2-cycle class) and v_bcnt_u32_b32 (4-cycle class), alone and as the xor -> bcnt pair of the
// Hamming row loop?  Every variant is one asm block over explicitly numbered registers v16..v47.
// Build: hipcc --offload-arch=gfx950 -O3 vgpr_bank.hip -o vgpr_bank
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

#define CLOB "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
             "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47"

// 16 instructions per block; sources (A+i%4 ...) chosen per mode
#define X8(op, d0, a0, b0, sa, sb) \
    op " v" #d0 ", v" #a0 ", v" #b0 "\n"

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters) {
    uint32_t seed = threadIdx.x * 2654435761u + blockIdx.x;
    asm volatile(
        "v_mov_b32 v16, %0\n v_mov_b32 v17, %0\n v_mov_b32 v18, %0\n v_mov_b32 v19, %0\n"
        "v_mov_b32 v20, %0\n v_mov_b32 v21, %0\n v_mov_b32 v22, %0\n v_mov_b32 v23, %0\n"
        "v_mov_b32 v24, %0\n v_mov_b32 v25, %0\n v_mov_b32 v26, %0\n v_mov_b32 v27, %0\n"
        "v_mov_b32 v28, %0\n v_mov_b32 v29, %0\n v_mov_b32 v30, %0\n v_mov_b32 v31, %0\n"
        "v_mov_b32 v32, %0\n v_mov_b32 v33, %0\n v_mov_b32 v34, %0\n v_mov_b32 v35, %0\n"
        "v_mov_b32 v36, %0\n v_mov_b32 v37, %0\n v_mov_b32 v38, %0\n v_mov_b32 v39, %0\n"
        "v_mov_b32 v40, %0\n v_mov_b32 v41, %0\n v_mov_b32 v42, %0\n v_mov_b32 v43, %0\n"
        "v_mov_b32 v44, %0\n v_mov_b32 v45, %0\n v_mov_b32 v46, %0\n v_mov_b32 v47, %0\n" ::"v"(seed) : CLOB);
    for (int it = 0; it < iters; it++) {
        if (MODE == 0)  // xor, sources in the SAME bank (16 & 20, 17 & 21 ...), destinations spread
            asm volatile(
                "v_xor_b32 v32, v16, v20\n v_xor_b32 v33, v17, v21\n v_xor_b32 v34, v18, v22\n v_xor_b32 v35, v19, v23\n"
                "v_xor_b32 v36, v24, v28\n v_xor_b32 v37, v25, v29\n v_xor_b32 v38, v26, v30\n v_xor_b32 v39, v27, v31\n"
                "v_xor_b32 v40, v16, v24\n v_xor_b32 v41, v17, v25\n v_xor_b32 v42, v18, v26\n v_xor_b32 v43, v19, v27\n"
                "v_xor_b32 v44, v20, v28\n v_xor_b32 v45, v21, v29\n v_xor_b32 v46, v22, v30\n v_xor_b32 v47, v23, v31\n" ::: CLOB);
        if (MODE == 1)  // xor, sources in DIFFERENT banks (16 & 21, 17 & 22 ...)
            asm volatile(
                "v_xor_b32 v32, v16, v21\n v_xor_b32 v33, v17, v22\n v_xor_b32 v34, v18, v23\n v_xor_b32 v35, v19, v20\n"
                "v_xor_b32 v36, v24, v29\n v_xor_b32 v37, v25, v30\n v_xor_b32 v38, v26, v31\n v_xor_b32 v39, v27, v28\n"
                "v_xor_b32 v40, v16, v25\n v_xor_b32 v41, v17, v26\n v_xor_b32 v42, v18, v27\n v_xor_b32 v43, v19, v24\n"
                "v_xor_b32 v44, v20, v29\n v_xor_b32 v45, v21, v30\n v_xor_b32 v46, v22, v31\n v_xor_b32 v47, v23, v28\n" ::: CLOB);
        if (MODE == 2)  // bcnt (accumulating), x and acc in the SAME bank
            asm volatile(
                "v_bcnt_u32_b32 v32, v16, v32\n v_bcnt_u32_b32 v33, v17, v33\n v_bcnt_u32_b32 v34, v18, v34\n v_bcnt_u32_b32 v35, v19, v35\n"
                "v_bcnt_u32_b32 v36, v20, v36\n v_bcnt_u32_b32 v37, v21, v37\n v_bcnt_u32_b32 v38, v22, v38\n v_bcnt_u32_b32 v39, v23, v39\n"
                "v_bcnt_u32_b32 v40, v24, v40\n v_bcnt_u32_b32 v41, v25, v41\n v_bcnt_u32_b32 v42, v26, v42\n v_bcnt_u32_b32 v43, v27, v43\n"
                "v_bcnt_u32_b32 v44, v28, v44\n v_bcnt_u32_b32 v45, v29, v45\n v_bcnt_u32_b32 v46, v30, v46\n v_bcnt_u32_b32 v47, v31, v47\n" ::: CLOB);
        if (MODE == 3)  // bcnt, x and acc in DIFFERENT banks
            asm volatile(
                "v_bcnt_u32_b32 v32, v17, v32\n v_bcnt_u32_b32 v33, v18, v33\n v_bcnt_u32_b32 v34, v19, v34\n v_bcnt_u32_b32 v35, v16, v35\n"
                "v_bcnt_u32_b32 v36, v21, v36\n v_bcnt_u32_b32 v37, v22, v37\n v_bcnt_u32_b32 v38, v23, v38\n v_bcnt_u32_b32 v39, v20, v39\n"
                "v_bcnt_u32_b32 v40, v25, v40\n v_bcnt_u32_b32 v41, v26, v41\n v_bcnt_u32_b32 v42, v27, v42\n v_bcnt_u32_b32 v43, v24, v43\n"
                "v_bcnt_u32_b32 v44, v29, v44\n v_bcnt_u32_b32 v45, v30, v45\n v_bcnt_u32_b32 v46, v31, v46\n v_bcnt_u32_b32 v47, v28, v47\n" ::: CLOB);
        if (MODE == 4)  // the row: 8 x (xor tmp, q, t ; bcnt acc, tmp, acc) with q / t in the SAME bank, tmp / acc in the same bank
            asm volatile(
                "v_xor_b32 v32, v16, v24\n v_bcnt_u32_b32 v40, v32, v40\n v_xor_b32 v33, v17, v25\n v_bcnt_u32_b32 v40, v33, v40\n"
                "v_xor_b32 v34, v18, v26\n v_bcnt_u32_b32 v40, v34, v40\n v_xor_b32 v35, v19, v27\n v_bcnt_u32_b32 v40, v35, v40\n"
                "v_xor_b32 v32, v20, v28\n v_bcnt_u32_b32 v40, v32, v40\n v_xor_b32 v33, v21, v29\n v_bcnt_u32_b32 v40, v33, v40\n"
                "v_xor_b32 v34, v22, v30\n v_bcnt_u32_b32 v40, v34, v40\n v_xor_b32 v35, v23, v31\n v_bcnt_u32_b32 v40, v35, v40\n" ::: CLOB);
        if (MODE == 5)  // the row with q / t in DIFFERENT banks and tmp / acc in different banks
            asm volatile(
                "v_xor_b32 v33, v16, v26\n v_bcnt_u32_b32 v40, v33, v40\n v_xor_b32 v34, v17, v27\n v_bcnt_u32_b32 v40, v34, v40\n"
                "v_xor_b32 v35, v18, v24\n v_bcnt_u32_b32 v40, v35, v40\n v_xor_b32 v33, v19, v25\n v_bcnt_u32_b32 v40, v33, v40\n"
                "v_xor_b32 v34, v20, v30\n v_bcnt_u32_b32 v40, v34, v40\n v_xor_b32 v35, v21, v31\n v_bcnt_u32_b32 v40, v35, v40\n"
                "v_xor_b32 v33, v22, v28\n v_bcnt_u32_b32 v40, v33, v40\n v_xor_b32 v34, v23, v29\n v_bcnt_u32_b32 v40, v34, v40\n" ::: CLOB);
        if (MODE == 6)  // the row, q / t different banks, two independent accumulators (even / odd words)
            asm volatile(
                "v_xor_b32 v33, v16, v26\n v_bcnt_u32_b32 v40, v33, v40\n v_xor_b32 v34, v17, v27\n v_bcnt_u32_b32 v41, v34, v41\n"
                "v_xor_b32 v35, v18, v24\n v_bcnt_u32_b32 v40, v35, v40\n v_xor_b32 v33, v19, v25\n v_bcnt_u32_b32 v41, v33, v41\n"
                "v_xor_b32 v34, v20, v30\n v_bcnt_u32_b32 v40, v34, v40\n v_xor_b32 v35, v21, v31\n v_bcnt_u32_b32 v41, v35, v41\n"
                "v_xor_b32 v33, v22, v28\n v_bcnt_u32_b32 v40, v33, v40\n v_xor_b32 v34, v23, v29\n v_bcnt_u32_b32 v41, v34, v41\n" ::: CLOB);
        if (MODE == 7)  // 8 xors first, then 8 bcnts (grouped by class), different banks
            asm volatile(
                "v_xor_b32 v32, v16, v26\n v_xor_b32 v33, v17, v27\n v_xor_b32 v34, v18, v24\n v_xor_b32 v35, v19, v25\n"
                "v_xor_b32 v36, v20, v30\n v_xor_b32 v37, v21, v31\n v_xor_b32 v38, v22, v28\n v_xor_b32 v39, v23, v29\n"
                "v_bcnt_u32_b32 v41, v32, v41\n v_bcnt_u32_b32 v41, v33, v41\n v_bcnt_u32_b32 v41, v34, v41\n v_bcnt_u32_b32 v41, v35, v41\n"
                "v_bcnt_u32_b32 v41, v36, v41\n v_bcnt_u32_b32 v41, v37, v41\n v_bcnt_u32_b32 v41, v38, v41\n v_bcnt_u32_b32 v41, v39, v41\n" ::: CLOB);
    }
    uint32_t s;
    asm volatile("v_add_u32 %0, v40, v41\n v_add_u32 %0, %0, v32\n v_add_u32 %0, %0, v47" : "=v"(s)::CLOB);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> int run(const char* name, int blocks_per_cu, uint32_t* out) {
    const int iters = 4000, ncu = 256;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = ncu * blocks_per_cu;
    k<MODE><<<blocks, 256>>>(out, 50);
    CK(hipEventRecord(e0));
    k<MODE><<<blocks, 256>>>(out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // per SIMD: blocks_per_cu waves, each iters * 16 instructions
    const double inst_per_simd = (double)blocks_per_cu * iters * 16;
    printf("%-68s waves/SIMD %d: %7.3f ms  %.2f cycles per instruction per SIMD @2.4GHz\n", name, blocks_per_cu, ms,
           ms * 1e-3 * 2.4e9 / inst_per_simd);
    return 0;
}

int main() {
    uint32_t* out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_xor   sources in the same bank (idx mod 4)", w, out);
        run<1>("v_xor   sources in different banks", w, out);
        run<2>("v_bcnt  x / acc in the same bank", w, out);
        run<3>("v_bcnt  x / acc in different banks", w, out);
        run<4>("row (xor,bcnt)x8: q/t same bank", w, out);
        run<5>("row (xor,bcnt)x8: q/t and tmp/acc in different banks", w, out);
        run<6>("row, different banks, two accumulators", w, out);
        run<7>("row, 8 xor then 8 bcnt, different banks", w, out);
    }
    return 0;
}
