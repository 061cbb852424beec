// lds_hog.hip - development aid for the SLAM_ERR_BUSY path of the one-launch window adjustment (tools/stress_ba_busy.sh):
//   lds_hog hog MS      holds 2 x 64 KiB of LDS on every compute unit for MS milliseconds of wall clock (one wave per
//                       workgroup, so wave slots, registers and the memory system stay free): a kernel that needs 58 KiB of LDS
//                       per workgroup cannot become resident meanwhile.  The spin is bounded by the 100 MHz wall clock.
//   lds_hog polls N     what N polls of round 3's grid barrier cost (s_sleep 2 + a relaxed agent-scope load each) on an
//                       idle chip: the "bounded by count" wait that round 4 replaced by a bound in time.
//     hipcc --offload-arch=gfx950 -O3 -o tools/ubench/lds_hog tools/ubench/lds_hog.hip
//     hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/ubench/liblds_hog.so tools/ubench/lds_hog.hip   (lds_hog_start / lds_hog_wait for ctypes)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

__global__ void hog_kernel(unsigned long long ticks, unsigned int* sink) {
    extern __shared__ unsigned int lds[];
    lds[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (lds[threadIdx.x] == 0xFFFFFFFFu) *sink = 1;              // keeps the LDS allocation alive
}

__global__ void poll_kernel(unsigned int* flag, unsigned int n, unsigned long long* out) {
    const unsigned long long t0 = wall_clock64();
    unsigned int polls = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        if (++polls > n) break;
        __builtin_amdgcn_s_sleep(2);
    }
    *out = wall_clock64() - t0;
}

// the same hog from inside another process's address space (ctypes): launched on a stream of its own, returns at once
extern "C" __attribute__((visibility("default"))) int lds_hog_start(double ms) {
    static hipStream_t st = nullptr;
    static unsigned int* sink = nullptr;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) return -1;
    if (!st && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return -2;
    if (!sink && hipMalloc(&sink, 64) != hipSuccess) return -3;
    if (ms <= 0 || ms > 5000) return -4;
    if (hipFuncSetAttribute((const void*)hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536) != hipSuccess) return -5;
    hog_kernel<<<p.multiProcessorCount * 3, 64, 49152, st>>>((unsigned long long)(ms * 1e5), sink);   // 3 x 48 KiB of the 160 KiB
    return hipGetLastError() == hipSuccess ? p.multiProcessorCount : -6;
}
extern "C" __attribute__((visibility("default"))) int lds_hog_wait() { return (int)hipDeviceSynchronize(); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: lds_hog hog MS | polls N\n"); return 2; }
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    unsigned int* d = nullptr;
    CK(hipMalloc(&d, 64));
    CK(hipMemset(d, 0, 64));
    if (!strcmp(argv[1], "hog")) {
        const double ms = atof(argv[2]);
        if (ms <= 0 || ms > 5000) { fprintf(stderr, "MS must be in (0, 5000]\n"); return 2; }
        CK(hipFuncSetAttribute((const void*)hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        hog_kernel<<<p.multiProcessorCount * 2, 64, 65536>>>((unsigned long long)(ms * 1e5), d);
        CK(hipGetLastError());
        printf("hog: 2 x 64 KiB of LDS on each of %d compute units for %.0f ms\n", p.multiProcessorCount, ms);
        fflush(stdout);
        CK(hipDeviceSynchronize());
        printf("hog: done\n");
    } else {
        const unsigned int n = (unsigned int)strtoul(argv[2], nullptr, 0);
        unsigned long long* out = nullptr;
        CK(hipMalloc(&out, 8));
        poll_kernel<<<1, 1>>>(d, n, out);
        CK(hipDeviceSynchronize());
        unsigned long long t = 0;
        CK(hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost));
        printf("%u polls (s_sleep 2 + relaxed agent-scope load) took %.1f us of wall clock: %.1f ns each\n", n, t / 100.0, t * 10.0 / n);
    }
    return 0;
}
