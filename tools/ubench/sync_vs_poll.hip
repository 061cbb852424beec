// How a host thread learns that a short kernel is done: hipStreamSynchronize against polling a word the kernel stores into
// pinned host memory (system-scope release store behind its results).  Development aid (host calls).
//     hipcc --offload-arch=gfx950 -O3 tools/ubench/sync_vs_poll.hip -o /tmp/sync_vs_poll && /tmp/sync_vs_poll
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void work_then_flag(int* out, volatile int* flag, int v, int spin) {
    // a little work so that the kernel is not literally empty: `spin` dependent adds per thread, then the result, then the flag
    int a = threadIdx.x;
    for (int i = 0; i < spin; i++) a = a * 3 + 1;
    out[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x == 0 && flag) {
        __threadfence_system();
        __hip_atomic_store((int*)flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int main() {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    int *flag, *out_host, *out_dev;
    CK(hipHostMalloc((void**)&flag, 64, hipHostMallocMapped));
    CK(hipHostMalloc((void**)&out_host, 1024, hipHostMallocMapped));
    CK(hipMalloc((void**)&out_dev, 1024));
    *flag = 0;
    const int reps = 5000;
    for (int spin : {0, 2000}) {
        for (int w = 0; w < 200; w++) { work_then_flag<<<1, 256, 0, s>>>(out_dev, nullptr, 0, spin); CK(hipStreamSynchronize(s)); }
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; i++) { work_then_flag<<<1, 256, 0, s>>>(out_dev, nullptr, 0, spin); CK(hipStreamSynchronize(s)); }
        auto t1 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; i++) {
            work_then_flag<<<1, 256, 0, s>>>(out_host, flag, i + 1, spin);
            while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != i + 1) { }
        }
        auto t2 = std::chrono::steady_clock::now();
        CK(hipStreamSynchronize(s));
        // polling, but a hipStreamSynchronize every 64 calls (so that the runtime's queue of completion signals stays short)
        *flag = 0;
        auto t3 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; i++) {
            work_then_flag<<<1, 256, 0, s>>>(out_host, flag, i + 1, spin);
            while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != i + 1) { }
            if ((i & 63) == 63) CK(hipStreamSynchronize(s));
        }
        auto t4 = std::chrono::steady_clock::now();
        auto us = [&](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count() / reps; };
        printf("kernel of %d dependent adds: launch + hipStreamSynchronize %.2f us per call; launch + poll a pinned word %.2f us; "
               "poll + a synchronize every 64th call %.2f us\n", spin, us(t0, t1), us(t1, t2), us(t3, t4));
    }
    return 0;
}
