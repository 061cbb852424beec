// Probe: why is a lean [wait, 16 VALU, 2 ds_read] row loop slower in the real kernel than hipcc's heavier form?
// MODE 0: lean - one VGPR base, immediate offsets.   MODE 1: one v_mov from SGPR per read + scalar address math.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
typedef uint32_t u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define XOR(x, a, b) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "v"(a), "v"(b))
#define BCNT(acc, x) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(x))
#define ROW(acc, ra, rc) { u32 x; XOR(x, q[0], ra.x); BCNT(acc, x); XOR(x, q[1], ra.y); BCNT(acc, x); XOR(x, q[2], ra.z); BCNT(acc, x); XOR(x, q[3], ra.w); BCNT(acc, x); \
                           XOR(x, q[4], rc.x); BCNT(acc, x); XOR(x, q[5], rc.y); BCNT(acc, x); XOR(x, q[6], rc.z); BCNT(acc, x); XOR(x, q[7], rc.w); BCNT(acc, x); }
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ src, u32* out, int iters) {
    __shared__ uint4 tile[512 + 8];
    tile[threadIdx.x] = src[threadIdx.x]; tile[threadIdx.x + 256] = src[threadIdx.x + 256];
    if (threadIdx.x < 8) tile[512 + threadIdx.x] = src[threadIdx.x];
    __syncthreads();
    u32 q[8]; for (int i = 0; i < 8; i++) q[i] = src[600 + threadIdx.x].x * (i + 3);
    u32 acc0 = 0, acc1 = 0;
    u32 base = (u32)(uintptr_t)(const __attribute__((address_space(3))) void*)tile;
    u32 addr = base; asm volatile("" : "+v"(addr));
    for (int it = 0; it < iters; it++) {
        if (MODE == 2) {
            u32x4 r0a, r0c, r1a, r1c;
            u32 a[16];
            asm volatile("ds_read_b128 %0, %2 offset:0\n ds_read_b128 %1, %2 offset:16" : "=&v"(r0a), "=&v"(r0c) : "v"(addr));
            asm volatile("ds_read_b128 %0, %2 offset:32\n ds_read_b128 %1, %2 offset:48" : "=&v"(r1a), "=&v"(r1c) : "v"(addr));
#pragma unroll
            for (int u = 0; u < 16; u += 2) {
                a[u] = 0x80000000u - 60u; a[u + 1] = 0x80000000u - 60u;
                asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                ROW(a[u], r0a, r0c);
                asm volatile("ds_read_b128 %0, %2 offset:%3\n ds_read_b128 %1, %2 offset:%4" : "=&v"(r0a), "=&v"(r0c) : "v"(addr), "n"(32 * (u + 2)), "n"(32 * (u + 2) + 16));
                asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                ROW(a[u + 1], r1a, r1c);
                asm volatile("ds_read_b128 %0, %2 offset:%3\n ds_read_b128 %1, %2 offset:%4" : "=&v"(r1a), "=&v"(r1c) : "v"(addr), "n"(32 * (u + 3)), "n"(32 * (u + 3) + 16));
            }
            u32 m = a[0];
#pragma unroll
            for (int u = 1; u < 16; u++) m &= a[u];
            if (__builtin_expect(__ballot((int)m >= 0) != 0ull, 0)) { acc0 += m; acc1 ^= a[3]; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("" :: "v"(r0a), "v"(r0c), "v"(r1a), "v"(r1c));
        } else if (MODE == 0) {
            u32x4 r0a, r0c, r1a, r1c;
            asm volatile("ds_read_b128 %0, %2 offset:0\n ds_read_b128 %1, %2 offset:16" : "=&v"(r0a), "=&v"(r0c) : "v"(addr));
            asm volatile("ds_read_b128 %0, %2 offset:32\n ds_read_b128 %1, %2 offset:48" : "=&v"(r1a), "=&v"(r1c) : "v"(addr));
#pragma unroll
            for (int u = 0; u < 16; u += 2) {
                asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                ROW(acc0, r0a, r0c);
                asm volatile("ds_read_b128 %0, %2 offset:%3\n ds_read_b128 %1, %2 offset:%4" : "=&v"(r0a), "=&v"(r0c) : "v"(addr), "n"(32 * (u + 2)), "n"(32 * (u + 2) + 16));
                asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                ROW(acc1, r1a, r1c);
                asm volatile("ds_read_b128 %0, %2 offset:%3\n ds_read_b128 %1, %2 offset:%4" : "=&v"(r1a), "=&v"(r1c) : "v"(addr), "n"(32 * (u + 3)), "n"(32 * (u + 3) + 16));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("" :: "v"(r0a), "v"(r0c), "v"(r1a), "v"(r1c));
        } else {
            // hipcc-like: every read gets its own address register moved from an SGPR, plus scalar address arithmetic
            u32 sb = __builtin_amdgcn_readfirstlane(base) + ((it & 7) << 9);
            u32x4 r0a, r0c, r1a, r1c; u32 a0, a1, a2, a3;
            asm volatile("v_mov_b32 %0, %1" : "=v"(a0) : "s"(sb));
            asm volatile("v_mov_b32 %0, %1" : "=v"(a1) : "s"(sb + 16));
            asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %3" : "=&v"(r0a), "=&v"(r0c) : "v"(a0), "v"(a1));
            asm volatile("v_mov_b32 %0, %1" : "=v"(a2) : "s"(sb + 32));
            asm volatile("v_mov_b32 %0, %1" : "=v"(a3) : "s"(sb + 48));
            asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %3" : "=&v"(r1a), "=&v"(r1c) : "v"(a2), "v"(a3));
#pragma unroll
            for (int u = 0; u < 16; u += 2) {
                asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                ROW(acc0, r0a, r0c);
                { u32 s0 = ((sb + 32 * (u + 2)) & 0xffff), s1 = s0 + 16;
                  asm volatile("v_mov_b32 %0, %1" : "=v"(a0) : "s"(s0)); asm volatile("v_mov_b32 %0, %1" : "=v"(a1) : "s"(s1)); }
                asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %3" : "=&v"(r0a), "=&v"(r0c) : "v"(a0), "v"(a1));
                asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                ROW(acc1, r1a, r1c);
                { u32 s0 = ((sb + 32 * (u + 3)) & 0xffff), s1 = s0 + 16;
                  asm volatile("v_mov_b32 %0, %1" : "=v"(a2) : "s"(s0)); asm volatile("v_mov_b32 %0, %1" : "=v"(a3) : "s"(s1)); }
                asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %3" : "=&v"(r1a), "=&v"(r1c) : "v"(a2), "v"(a3));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("" :: "v"(r0a), "v"(r0c), "v"(r1a), "v"(r1c));
        }
        if (MODE != 1) { addr = base + (((it + 1) & 7) << 9); asm volatile("" : "+v"(addr)); }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc0 + acc1;
}
template <int MODE> int run(const uint4* src, u32* out, const char* name) {
    const int iters = 2000, blocks = 256 * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) k<MODE><<<blocks, 256>>>(src, out, iters);
    CK(hipEventRecord(e0)); k<MODE><<<blocks, 256>>>(src, out, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %.3f ms -> %.1f cycles per row per SIMD @2.38GHz\n", name, ms, ms * 1e-3 * 2.38e9 / (iters * 16.0 * 8));
    return 0;
}
int main() { uint4* src; u32* out; CK(hipMalloc(&src, 1024 * 16)); CK(hipMemset(src, 0x5A, 1024 * 16)); CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    run<0>(src, out, "warm"); run<0>(src, out, "lean: VGPR base + immediate offsets"); run<1>(src, out, "v_mov from SGPR per read + SALU"); run<2>(src, out, "lean + 16-row AND filter, branch not taken"); run<0>(src, out, "lean again"); return 0; }
