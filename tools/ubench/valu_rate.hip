// Micro-benchmark: issue rate of the integer VALU ops the Hamming kernel is made of.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

template<int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, unsigned long long* cyc) {
    uint32_t a[8], x = threadIdx.x * 2654435761u, y = blockIdx.x + 1;
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = x + i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(y));
                if (MODE == 1) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(y));
                if (MODE == 2) { uint32_t tmp; asm volatile("v_xor_b32 %0, %1, %2" : "=v"(tmp) : "v"(y), "v"(a[i])); asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(tmp)); }
                if (MODE == 3) asm volatile("v_min3_u32 %0, %1, %0, %1" : "+v"(a[i]) : "v"(y));
                if (MODE == 4) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(y));
                if (MODE == 5) asm volatile("v_fma_f32 %0, %1, %0, %1" : "+v"(a[i]) : "v"(y));
                if (MODE == 6) asm volatile("v_xor_b32 %0, s4, %0" : "+v"(a[i]) :: "s4");
                if (MODE == 7) asm volatile("v_xor_b32_dpp %0, %1, %0 row_ror:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(y));
                if (MODE == 8) asm volatile("v_mov_b32 %0, s4" : "=v"(a[i]) :: "s4");
                if (MODE == 9) asm volatile("v_mov_b32_dpp %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(y));
                if (MODE == 10) asm volatile("v_mov_b32_dpp %0, %1 wave_rol:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(y));
                if (MODE == 11) asm volatile("v_xor_b32_dpp %0, %1, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(y));
                if (MODE == 12) asm volatile("v_readlane_b32 s4, %0, 3" :: "v"(a[i]) : "s4");
                if (MODE == 13) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(y));
                if (MODE == 14) asm volatile("v_xor_b32_dpp %0, %1, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(a[i]) : "v"(y));
                if (MODE == 15) asm volatile("v_or_b32 %0, %1, %0" : "+v"(a[i]) : "v"(y));
                if (MODE == 16) asm volatile("v_xor_b32_dpp %0, %1, %0 row_shr:5 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(y));
                if (MODE == 17) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(y));
                if (MODE == 18) asm volatile("v_xor_b32_dpp %0, %1, %0 wave_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(y));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template<int MODE> int run(const char* name, int blocks_per_cu, uint32_t* out, unsigned long long* cyc) {
    const int iters = 2000, ncu = 256;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int blocks = ncu * blocks_per_cu;
    k<MODE><<<blocks, 256>>>(out, 10, cyc);
    CK(hipEventRecord(e0));
    k<MODE><<<blocks, 256>>>(out, iters, cyc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks); CK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    const double inst_per_wave = (double)iters * 64 * (MODE == 2 ? 2 : 1);
    // waves per SIMD = blocks_per_cu (each block = 4 waves = 1 per SIMD)
    const double cyc_per_inst_per_simd = avg / (inst_per_wave * blocks_per_cu);
    const double laneops = inst_per_wave * 64 * 4.0 * blocks;
    printf("%-18s waves/SIMD=%d  %.3f ms  %.1f Tlane-ops/s  avg memtime ticks/block %.0f  ticks per wave-inst per SIMD %.3f\n",
           name, blocks_per_cu, ms, laneops / (ms * 1e-3) / 1e12, avg, cyc_per_inst_per_simd);
    return 0;
}

int main() {
    uint32_t* out; unsigned long long* cyc;
    CK(hipMalloc(&out, 256 * 8 * 256 * 4)); CK(hipMalloc(&cyc, 256 * 8 * 8));
    for (int w : {2, 8}) {
        run<0>("xor", w, out, cyc); run<1>("bcnt", w, out, cyc); run<2>("xor+bcnt", w, out, cyc);
        run<7>("xor_dpp_row_ror", w, out, cyc); run<16>("xor_dpp_row_shr", w, out, cyc); run<11>("xor_dpp_quad", w, out, cyc); run<14>("xor_dpp_bcast15", w, out, cyc); run<18>("xor_dpp_wave_ror", w, out, cyc);
        run<8>("mov_sgpr", w, out, cyc); run<9>("mov_dpp_row_ror", w, out, cyc); run<10>("mov_dpp_wave_rol", w, out, cyc);
        run<12>("readlane", w, out, cyc); run<13>("permlane32_swap", w, out, cyc); run<17>("permlane16_swap", w, out, cyc); run<15>("or", w, out, cyc);
    }
    return 0;
}
