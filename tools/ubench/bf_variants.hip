// Development harness: variants of the Hamming top-2 inner loop, timed on 64k x 64k random data.
// Build: hipcc --offload-arch=gfx950 -O3 bf_variants.hip -o bf_variants
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <string.h>
typedef uint32_t u32;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);}}while(0)
#define TILE 256
#define BIAS 0x80000000u
#define NONE 0xFFFFFFFFu

__device__ __forceinline__ u32 bcnt_acc(u32 x, u32 acc) { u32 d; asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(acc)); return d; }
__device__ __forceinline__ u32 umin3(u32 a, u32 b, u32 c) { return min(min(a, b), c); }
__device__ __forceinline__ u32 umed3(u32 a, u32 b, u32 c) { return max(min(a, b), min(max(a, b), c)); }

template <int R> __device__ __forceinline__ u32 min_all(const u32 (&acc)[R]) {
    u32 m = acc[0];
    int r = 1;
#pragma unroll
    for (; r + 1 < R; r += 2) m = umin3(m, acc[r], acc[r + 1]);
    if (r < R) m = min(m, acc[r]);
    return m;
}
// AND-tree variant of the filter: sign bit clear in any acc <=> sign bit of AND clear
template <int R> __device__ __forceinline__ u32 and_all(const u32 (&acc)[R]) {
    u32 m = acc[0];
#pragma unroll
    for (int r = 1; r < R; r++) m &= acc[r];
    return m;
}

template <int R, int FILT>
__device__ __forceinline__ void step(const u32 (&q)[R][8], const uint4 a, const uint4 b, u32 ti, u32 (&b1)[R], u32 (&b2)[R], u32 (&init)[R]) {
    u32 acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = bcnt_acc(q[r][0] ^ a.x, init[r]);
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = bcnt_acc(q[r][1] ^ a.y, acc[r]);
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = bcnt_acc(q[r][2] ^ a.z, acc[r]);
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = bcnt_acc(q[r][3] ^ a.w, acc[r]);
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = bcnt_acc(q[r][4] ^ b.x, acc[r]);
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = bcnt_acc(q[r][5] ^ b.y, acc[r]);
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = bcnt_acc(q[r][6] ^ b.z, acc[r]);
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = bcnt_acc(q[r][7] ^ b.w, acc[r]);
    if (FILT == 2) {
#pragma unroll
        for (int r = 0; r < R; r++) b1[r] ^= acc[r];   // keeps the distance live at 1 fast op per pair; results are garbage
        return;
    }
    bool hit;
    if (FILT == 0 || FILT == 3) hit = min_all<R>(acc) < BIAS;
    else hit = (int)and_all<R>(acc) >= 0;
    // FILT 3: wave-uniform branch on the ballot, update block laid out as the unlikely path
    if (FILT == 3 ? __builtin_expect(__ballot(hit) != 0ull, 0) : hit) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            u32 key = ((acc[r] - init[r]) << 23) | ti;
            b2[r] = umed3(b1[r], b2[r], key);
            b1[r] = min(b1[r], key);
            init[r] = max(init[r], BIAS - (b2[r] >> 23));   // never loosen a bound learnt from other chunks
        }
    }
}

// exchange the 2nd-best distance of every query with the other chunk blocks working on the same queries
template <int R>
__device__ __forceinline__ void share_bound(u32* __restrict__ bound, int qbase, int N, const u32 (&b2)[R], u32 (&init)[R]) {
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int qi = qbase + r * 64;
        if (qi < N) {
            const u32 g = __hip_atomic_load(&bound[qi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // may be stale: only looser
            const u32 own = b2[r] >> 23;
            if (own < g) atomicMin(&bound[qi], own);
            const u32 th = min(own, g + 1);      // pass iff d < own (later index cannot win a tie) and d <= g
            init[r] = BIAS - th;
        }
    }
}

// PF: 0 = reads at the top of each 2-step group (baseline), 1 = rows for the next group prefetched into a second register set
template <int R, int PF, int FILT, int MINW, int SH>
__global__ __launch_bounds__(256, MINW) void bf(const uint4* __restrict__ q, int N, const uint4* __restrict__ t, int M, int chunk, uint2* __restrict__ partial, unsigned long long* __restrict__ stamps, u32* __restrict__ bound) {
    __shared__ uint4 tile[2][TILE * 2];
    const unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qbase = blockIdx.x * (256 * R) + wave * (64 * R) + lane;
    u32 qr[R][8];
#pragma unroll
    for (int r = 0; r < R; r++) {
        int qi = qbase + r * 64; qi = qi < N ? qi : N - 1;
        const uint4 a = q[2 * (size_t)qi], b = q[2 * (size_t)qi + 1];
        qr[r][0] = a.x; qr[r][1] = a.y; qr[r][2] = a.z; qr[r][3] = a.w; qr[r][4] = b.x; qr[r][5] = b.y; qr[r][6] = b.z; qr[r][7] = b.w;
    }
    u32 b1[R], b2[R], init[R];
#pragma unroll
    for (int r = 0; r < R; r++) { b1[r] = NONE; b2[r] = NONE; init[r] = BIAS - 511u; }
    const int t0 = blockIdx.y * chunk, t1 = min(M, t0 + chunk);
#pragma unroll
    for (int i = 0; i < 2; i++) { const int g = 2 * t0 + tid + i * 256; tile[0][tid + i * 256] = g < 2 * t1 ? t[(size_t)g] : make_uint4(0, 0, 0, 0); }
    __syncthreads();
    int buf = 0;
    for (int tb = t0; tb < t1; tb += TILE) {
        const int nb = tb + TILE;
        uint4 nxt[2];
        if (nb < t1) {
#pragma unroll
            for (int i = 0; i < 2; i++) { const int g = 2 * nb + tid + i * 256; nxt[i] = g < 2 * t1 ? t[(size_t)g] : make_uint4(0, 0, 0, 0); }
        }
        if (SH) share_bound<R>(bound, qbase, N, b2, init);
        const int cnt = __builtin_amdgcn_readfirstlane(min(TILE, t1 - tb));
        const uint4* tp = tile[buf];
        int j = 0;
        if (PF == 0) {
            for (; j + 2 <= cnt; j += 2) {
                const uint4 a0 = tp[2 * j], c0 = tp[2 * j + 1], a1 = tp[2 * j + 2], c1 = tp[2 * j + 3];
                step<R, FILT>(qr, a0, c0, (u32)(tb + j), b1, b2, init);
                step<R, FILT>(qr, a1, c1, (u32)(tb + j + 1), b1, b2, init);
            }
        } else if (PF == 1) {
            // software pipeline: row j+1 is in flight while row j is computed
            uint4 a0 = tp[0], c0 = tp[1];
            for (; j + 2 <= cnt; j += 2) {
                const uint4 a1 = tp[2 * j + 2], c1 = tp[2 * j + 3];
                step<R, FILT>(qr, a0, c0, (u32)(tb + j), b1, b2, init);
                a0 = tp[(2 * j + 4) & (2 * TILE - 1)]; c0 = tp[(2 * j + 5) & (2 * TILE - 1)];
                step<R, FILT>(qr, a1, c1, (u32)(tb + j + 1), b1, b2, init);
            }
        } else if (PF == 3) {
            uint4 a0 = tp[0], c0 = tp[1];
            for (; j + 2 <= cnt; j += 2) {
                step<R, FILT>(qr, a0, c0, (u32)(tb + j), b1, b2, init);
                a0.x += 0x9E3779B9u; a0.y += 0x7F4A7C15u; a0.z += 0x85EBCA6Bu; a0.w += 0xC2B2AE35u; c0.x += 0x27D4EB2Fu; c0.y += 0x165667B1u; c0.z += 0xD3A2646Du; c0.w += 0xFD7046C5u;
                step<R, FILT>(qr, c0, a0, (u32)(tb + j + 1), b1, b2, init);
            }
        } else {
            for (; j + 4 <= cnt; j += 4) {
                const uint4 a0 = tp[2 * j], c0 = tp[2 * j + 1], a1 = tp[2 * j + 2], c1 = tp[2 * j + 3];
                const uint4 a2 = tp[2 * j + 4], c2 = tp[2 * j + 5], a3 = tp[2 * j + 6], c3 = tp[2 * j + 7];
                step<R, FILT>(qr, a0, c0, (u32)(tb + j), b1, b2, init);
                step<R, FILT>(qr, a1, c1, (u32)(tb + j + 1), b1, b2, init);
                step<R, FILT>(qr, a2, c2, (u32)(tb + j + 2), b1, b2, init);
                step<R, FILT>(qr, a3, c3, (u32)(tb + j + 3), b1, b2, init);
            }
        }
        for (; j < cnt; j++) { const uint4 a0 = tp[2 * j], c0 = tp[2 * j + 1]; step<R, FILT>(qr, a0, c0, (u32)(tb + j), b1, b2, init); }
        if (nb < t1) {
#pragma unroll
            for (int i = 0; i < 2; i++) tile[buf ^ 1][tid + i * 256] = nxt[i];
        }
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int r = 0; r < R; r++) { const int qi = qbase + r * 64; if (qi < N) partial[(size_t)blockIdx.y * N + qi] = make_uint2(b1[r], b2[r]); }
    if (stamps && threadIdx.x == 0) {
        const unsigned long long st1 = __builtin_amdgcn_s_memtime(), sr1 = __builtin_amdgcn_s_memrealtime();
        const size_t b = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        stamps[4 * b] = st1 - st0; stamps[4 * b + 1] = sr1 - sr0; stamps[4 * b + 2] = sr0; stamps[4 * b + 3] = sr1;
    }
}

// per-wave staging: every wave streams the train chunk through its own 2 x 2 KiB LDS ring, no block barrier
template <int R, int FILT, int SH, int WT>
__global__ __launch_bounds__(256) void bfw(const uint4* __restrict__ q, int N, const uint4* __restrict__ t, int M, int chunk, uint2* __restrict__ partial, unsigned long long* __restrict__ stamps, u32* __restrict__ bound) {
    __shared__ uint4 wtile[4][2][WT * 2];
    const unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qbase = blockIdx.x * (256 * R) + wave * (64 * R) + lane;
    u32 qr[R][8];
#pragma unroll
    for (int r = 0; r < R; r++) {
        int qi = qbase + r * 64; qi = qi < N ? qi : N - 1;
        const uint4 a = q[2 * (size_t)qi], b = q[2 * (size_t)qi + 1];
        qr[r][0] = a.x; qr[r][1] = a.y; qr[r][2] = a.z; qr[r][3] = a.w; qr[r][4] = b.x; qr[r][5] = b.y; qr[r][6] = b.z; qr[r][7] = b.w;
    }
    u32 b1[R], b2[R], init[R];
#pragma unroll
    for (int r = 0; r < R; r++) { b1[r] = NONE; b2[r] = NONE; init[r] = BIAS - 511u; }
    const int t0 = blockIdx.y * chunk, t1 = min(M, t0 + chunk);
    constexpr int NL = WT * 2 / 64;   // uint4 loads per lane per tile
    uint4* my = &wtile[wave][0][0];
#pragma unroll
    for (int i = 0; i < NL; i++) { const int g = 2 * t0 + lane + i * 64; my[lane + i * 64] = g < 2 * t1 ? t[(size_t)g] : make_uint4(0, 0, 0, 0); }
    int buf = 0;
    for (int tb = t0; tb < t1; tb += WT) {
        const int nb = tb + WT;
        uint4 nxt[NL];
        if (nb < t1) {
#pragma unroll
            for (int i = 0; i < NL; i++) { const int g = 2 * nb + lane + i * 64; nxt[i] = g < 2 * t1 ? t[(size_t)g] : make_uint4(0, 0, 0, 0); }
        }
        if (SH && ((tb - t0) & 255) == 0) share_bound<R>(bound, qbase, N, b2, init);
        const int cnt = __builtin_amdgcn_readfirstlane(min(WT, t1 - tb));
        const uint4* tp = &wtile[wave][buf][0];
        __builtin_amdgcn_wave_barrier();
        int j = 0;
        uint4 a0 = tp[0], c0 = tp[1];
        for (; j + 2 <= cnt; j += 2) {
            const uint4 a1 = tp[2 * j + 2], c1 = tp[2 * j + 3];
            step<R, FILT>(qr, a0, c0, (u32)(tb + j), b1, b2, init);
            a0 = tp[(2 * j + 4) & (2 * WT - 1)]; c0 = tp[(2 * j + 5) & (2 * WT - 1)];
            step<R, FILT>(qr, a1, c1, (u32)(tb + j + 1), b1, b2, init);
        }
        for (; j < cnt; j++) { const uint4 x0 = tp[2 * j], y0 = tp[2 * j + 1]; step<R, FILT>(qr, x0, y0, (u32)(tb + j), b1, b2, init); }
        __builtin_amdgcn_wave_barrier();
        if (nb < t1) {
            uint4* wp = &wtile[wave][buf ^ 1][0];
#pragma unroll
            for (int i = 0; i < NL; i++) wp[lane + i * 64] = nxt[i];
        }
        buf ^= 1;
    }
#pragma unroll
    for (int r = 0; r < R; r++) { const int qi = qbase + r * 64; if (qi < N) partial[(size_t)blockIdx.y * N + qi] = make_uint2(b1[r], b2[r]); }
    if (stamps && threadIdx.x == 0) {
        const unsigned long long st1 = __builtin_amdgcn_s_memtime(), sr1 = __builtin_amdgcn_s_memrealtime();
        const size_t b = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        stamps[4 * b] = st1 - st0; stamps[4 * b + 1] = sr1 - sr0; stamps[4 * b + 2] = sr0; stamps[4 * b + 3] = sr1;
    }
}

__global__ void merge(const uint2* partial, int S, int N, uint2* out) {
    int n = blockIdx.x * 256 + threadIdx.x; if (n >= N) return;
    u32 b1 = NONE, b2 = NONE;
    for (int s = 0; s < S; s++) { uint2 p = partial[(size_t)s * N + n]; b2 = umed3(b1, b2, p.x); b1 = min(b1, p.x); b2 = umed3(b1, b2, p.y); b1 = min(b1, p.y); }
    out[n] = make_uint2(b1, b2);
}

static int N = 65536, M = 65536;
static u32* dbound; static unsigned long long* dstamps; static uint4 *dq, *dt; static uint2 *dpart, *dout; static std::vector<uint2> ref;

template <int R, int PF, int FILT, int MINW, int SH> void run(const char* name, int bpc) {
    { const char* sel = getenv("SEL"); char tag[64]; snprintf(tag, sizeof tag, "%s bpc%d", name, bpc); if (sel && !strstr(tag, sel)) return; }
    int qblocks = (N + 256 * R - 1) / (256 * R);
    long target = 256L * bpc; int S = (int)(target / qblocks); int maxc = (M + TILE - 1) / TILE; if (S > maxc) S = maxc; if (S < 1) S = 1;
    int chunk = (M + S - 1) / S; chunk = (chunk + TILE - 1) / TILE * TILE; S = (M + chunk - 1) / chunk;
    dim3 grid(qblocks, S);
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, (const void*)bf<R, PF, FILT, MINW, SH>));
    for (int i = 0; i < 2; i++) { CK(hipMemsetAsync(dbound, 0x7F, (size_t)N * 4)); bf<R, PF, FILT, MINW, SH><<<grid, 256>>>(dq, N, dt, M, chunk, dpart, dstamps, dbound); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int i = 0; i < 7; i++) { CK(hipMemsetAsync(dbound, 0x7F, (size_t)N * 4)); CK(hipEventRecord(e0)); bf<R, PF, FILT, MINW, SH><<<grid, 256>>>(dq, N, dt, M, chunk, dpart, dstamps, dbound); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms); }
    std::sort(ts.begin(), ts.end());
    merge<<<(N + 255) / 256, 256>>>(dpart, S, N, dout);
    std::vector<uint2> h(N); CK(hipMemcpy(h.data(), dout, N * 8, hipMemcpyDeviceToHost));
    const int nb = qblocks * S; std::vector<unsigned long long> hs(4 * nb); CK(hipMemcpy(hs.data(), dstamps, hs.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ghz(nb); double blk_us = 0; unsigned long long s_min = ~0ull, e_max = 0, s_max = 0, e_min = ~0ull; std::vector<double> dur(nb);
    for (int i = 0; i < nb; i++) { ghz[i] = (double)hs[4 * i] / ((double)hs[4 * i + 1] * 10.0); dur[i] = hs[4 * i + 1] / 100.0; blk_us += dur[i];
        s_min = std::min(s_min, hs[4 * i + 2]); s_max = std::max(s_max, hs[4 * i + 2]); e_max = std::max(e_max, hs[4 * i + 3]); e_min = std::min(e_min, hs[4 * i + 3]); }
    std::sort(dur.begin(), dur.end());
    printf("    blocks: dur min %.0f p50 %.0f p90 %.0f max %.0f us | last start +%.0f us, first end +%.0f us, last end +%.0f us\n", dur[0], dur[nb / 2], dur[nb * 9 / 10], dur[nb - 1],
           (s_max - s_min) / 100.0, (e_min - s_min) / 100.0, (e_max - s_min) / 100.0);
    std::sort(ghz.begin(), ghz.end()); blk_us /= nb;
    bool same = true;
    if (ref.empty()) ref = h; else for (int i = 0; i < N; i++) if (h[i].x != ref[i].x || h[i].y != ref[i].y) { same = false; break; }
    double pairs = (double)N * M;
    printf("%-22s R=%d bpc=%2d grid=%dx%d vgpr=%3d  min %.4f med %.4f ms  %.0f Gpairs/s  clk %.2f GHz blk %.0f us same=%d\n", name, R, bpc, qblocks, S, fa.numRegs, ts[0], ts[3], pairs / (ts[3] * 1e-3) / 1e9, ghz[nb / 2], blk_us, (int)same);
    fflush(stdout);
}

template <int R, int FILT, int SH, int WT> void runw(const char* name, int bpc) {
    { const char* sel = getenv("SEL"); char tag[64]; snprintf(tag, sizeof tag, "%s bpc%d", name, bpc); if (sel && !strstr(tag, sel)) return; }
    int qblocks = (N + 256 * R - 1) / (256 * R);
    long target = 256L * bpc; int S = (int)(target / qblocks); int maxc = (M + TILE - 1) / TILE; if (S > maxc) S = maxc; if (S < 1) S = 1;
    int chunk = (M + S - 1) / S; chunk = (chunk + 255) / 256 * 256; S = (M + chunk - 1) / chunk;
    dim3 grid(qblocks, S);
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, (const void*)bfw<R, FILT, SH, WT>));
    for (int i = 0; i < 2; i++) { CK(hipMemsetAsync(dbound, 0x7F, (size_t)N * 4)); bfw<R, FILT, SH, WT><<<grid, 256>>>(dq, N, dt, M, chunk, dpart, dstamps, dbound); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int i = 0; i < 7; i++) { CK(hipMemsetAsync(dbound, 0x7F, (size_t)N * 4)); CK(hipEventRecord(e0)); bfw<R, FILT, SH, WT><<<grid, 256>>>(dq, N, dt, M, chunk, dpart, dstamps, dbound); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms); }
    std::sort(ts.begin(), ts.end());
    merge<<<(N + 255) / 256, 256>>>(dpart, S, N, dout);
    std::vector<uint2> h(N); CK(hipMemcpy(h.data(), dout, N * 8, hipMemcpyDeviceToHost));
    const int nb = qblocks * S; std::vector<unsigned long long> hs(4 * nb); CK(hipMemcpy(hs.data(), dstamps, hs.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ghz(nb); double blk_us = 0; unsigned long long s_min = ~0ull, e_max = 0, s_max = 0, e_min = ~0ull; std::vector<double> dur(nb);
    for (int i = 0; i < nb; i++) { ghz[i] = (double)hs[4 * i] / ((double)hs[4 * i + 1] * 10.0); dur[i] = hs[4 * i + 1] / 100.0; blk_us += dur[i];
        s_min = std::min(s_min, hs[4 * i + 2]); s_max = std::max(s_max, hs[4 * i + 2]); e_max = std::max(e_max, hs[4 * i + 3]); e_min = std::min(e_min, hs[4 * i + 3]); }
    std::sort(dur.begin(), dur.end());
    printf("    blocks: dur min %.0f p50 %.0f p90 %.0f max %.0f us | last start +%.0f us, first end +%.0f us, last end +%.0f us\n", dur[0], dur[nb / 2], dur[nb * 9 / 10], dur[nb - 1],
           (s_max - s_min) / 100.0, (e_min - s_min) / 100.0, (e_max - s_min) / 100.0);
    std::sort(ghz.begin(), ghz.end()); blk_us /= nb;
    bool same = true;
    if (ref.empty()) ref = h; else for (int i = 0; i < N; i++) if (h[i].x != ref[i].x || h[i].y != ref[i].y) { same = false; break; }
    double pairs = (double)N * M;
    printf("%-22s R=%d bpc=%2d grid=%dx%d vgpr=%3d  min %.4f med %.4f ms  %.0f Gpairs/s  clk %.2f GHz blk %.0f us same=%d\n", name, R, bpc, qblocks, S, fa.numRegs, ts[0], ts[3], pairs / (ts[3] * 1e-3) / 1e9, ghz[nb / 2], blk_us, (int)same);
    fflush(stdout);
}

int main(int argc, char** argv) {
    if (argc > 1) N = atoi(argv[1]); if (argc > 2) M = atoi(argv[2]);
    std::vector<uint8_t> hq((size_t)N * 32), ht((size_t)M * 32);
    srand(228); for (auto& b : hq) b = rand() & 255; for (auto& b : ht) b = rand() & 255;
    CK(hipMalloc(&dq, hq.size())); CK(hipMalloc(&dt, ht.size())); CK(hipMalloc(&dpart, (size_t)N * 8 * 1024)); CK(hipMalloc(&dout, (size_t)N * 8)); CK(hipMalloc(&dstamps, 32 * 65536 * 8)); CK(hipMalloc(&dbound, (size_t)N * 4));
    CK(hipMemcpy(dq, hq.data(), hq.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dt, ht.data(), ht.size(), hipMemcpyHostToDevice));
#define RUNALL(R, PF, FILT, MINW, SH, ...) for (int bpc : {__VA_ARGS__}) run<R, PF, FILT, MINW, SH>("R" #R " pf" #PF " filt" #FILT " minw" #MINW " sh" #SH, bpc)
#define RUNW(R, FILT, SH, WT, ...) for (int bpc : {__VA_ARGS__}) runw<R, FILT, SH, WT>("W R" #R " filt" #FILT " sh" #SH " wt" #WT, bpc)
    RUNALL(2, 1, 3, 1, 1, 16, 32);
    RUNW(2, 3, 1, 64, 8, 16, 32); RUNW(2, 3, 1, 128, 8, 16, 32); RUNW(2, 2, 0, 64, 16, 32);
    RUNALL(4, 1, 3, 1, 1, 12, 24);
    RUNW(4, 3, 1, 64, 6, 12, 24); RUNW(4, 3, 1, 128, 12, 24); RUNW(4, 2, 0, 64, 12, 24);
    RUNW(8, 3, 1, 64, 4, 8, 16);
    return 0;
}
