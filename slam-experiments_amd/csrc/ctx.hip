// ctx.hip — context, device memory, error reporting and stream timers of
// libslamhip.so.  One slam_ctx = one HIP device + one stream; nothing global
// is mutable except the thread-local error string (SURVEY.md §8b threading);
// tuning overrides and all scratch state live in the context.
#include "internal.h"
#include <stdio.h>
#include <string.h>
#include <new>

static thread_local char g_err[512] = "";

int slam_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* slam_last_error(void) { return g_err; }
extern "C" const char* slam_version(void) { return "slamhip 0.3 (gfx950)"; }

extern "C" int slam_device_count(int* count) {
    SLAM_REQUIRE(count, "slam_device_count: null out pointer");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return slam_set_error(SLAM_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return SLAM_OK;
}

extern "C" int slam_ctx_create(int device, slam_ctx** out) {
    SLAM_REQUIRE(out, "slam_ctx_create: null out pointer");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
        return slam_set_error(SLAM_ERR_NO_DEVICE, "no HIP device visible");
    SLAM_REQUIRE(device >= 0 && device < n, "device %d out of range [0,%d)", device, n);
    SLAM_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    SLAM_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return slam_set_error(SLAM_ERR_NO_DEVICE, "device %d is %s, library is built for gfx950 only",
                              device, prop.gcnArchName);
    slam_ctx* c = new (std::nothrow) slam_ctx();
    SLAM_REQUIRE(c, "out of host memory");
    c->device = device;
    c->num_cu = prop.multiProcessorCount;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_start);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_stop);
    if (e == hipSuccess) e = hipMalloc(&c->scratch, 4096);
    // on the context's own (non-blocking) stream: a null-stream memset is not ordered against the first kernels there
    if (e == hipSuccess) e = hipMemsetAsync(c->scratch, 0, 4096, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        delete c;
        return slam_set_error(SLAM_ERR_HIP, "context setup failed: %s", hipGetErrorString(e));
    }
    *out = c;
    return SLAM_OK;
}

extern "C" int slam_ctx_destroy(slam_ctx* ctx) {
    if (!ctx) return SLAM_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm) slam_comm_destroy(ctx);
    slam_second_stream_destroy(ctx);
    for (auto& kv : ctx->allocs) (void)hipFree(kv.first);
    if (ctx->workspace) (void)hipFree(ctx->workspace);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->io_dev) (void)hipFree(ctx->io_dev);
    if (ctx->io_host) (void)hipHostFree(ctx->io_host);
    if (ctx->sel_host) (void)hipHostFree(ctx->sel_host);
    if (ctx->bf_state_mem) (void)hipFree(ctx->bf_state_mem);
    if (ctx->bf_tbl_ready) {
        (void)hipFree(ctx->bf_tbl_dev);
        (void)hipHostFree(ctx->bf_tbl_host);
        for (int i = 0; i < SLAM_BF_TBL_RING; i++) (void)hipEventDestroy(ctx->bf_tbl_ev[i]);
    }
    if (ctx->prof_ev) {
        for (int i = 0; i < 2 * slam_ctx::PROF_MAX; i++) (void)hipEventDestroy(ctx->prof_ev[i]);
        delete[] ctx->prof_ev;
    }
    (void)hipEventDestroy(ctx->ev_start);
    (void)hipEventDestroy(ctx->ev_stop);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return SLAM_OK;
}

extern "C" int slam_ctx_device(slam_ctx* ctx, int* device) {
    SLAM_REQUIRE(ctx && device, "slam_ctx_device: null argument");
    *device = ctx->device;
    return SLAM_OK;
}

extern "C" int slam_sync(slam_ctx* ctx) {
    SLAM_REQUIRE(ctx, "slam_sync: null ctx");
    SLAM_HIP(hipSetDevice(ctx->device));
    SLAM_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->comm_stream) SLAM_HIP(hipStreamSynchronize(ctx->comm_stream));
    return SLAM_OK;
}

extern "C" int slam_malloc(slam_ctx* ctx, uint64_t bytes, void** d_ptr) {
    SLAM_REQUIRE(ctx && d_ptr, "slam_malloc: null argument");
    *d_ptr = nullptr;
    SLAM_HIP(hipSetDevice(ctx->device));
    void* p = nullptr;
    SLAM_HIP(hipMalloc(&p, bytes ? bytes : 16));
    std::lock_guard<std::mutex> g(ctx->mu);
    ctx->allocs[p] = bytes;
    *d_ptr = p;
    return SLAM_OK;
}

extern "C" int slam_free(slam_ctx* ctx, void* d_ptr) {
    SLAM_REQUIRE(ctx, "slam_free: null ctx");
    if (!d_ptr) return SLAM_OK;
    {
        std::lock_guard<std::mutex> g(ctx->mu);
        auto it = ctx->allocs.find(d_ptr);
        SLAM_REQUIRE(it != ctx->allocs.end(), "slam_free: pointer %p not owned by this context", d_ptr);
        ctx->allocs.erase(it);
    }
    SLAM_HIP(hipSetDevice(ctx->device));
    SLAM_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->comm_stream) SLAM_HIP(hipStreamSynchronize(ctx->comm_stream));   // the buffer may be one side of a gather in flight
    SLAM_HIP(hipFree(d_ptr));
    return SLAM_OK;
}

extern "C" int slam_memset(slam_ctx* ctx, void* d_ptr, int value, uint64_t bytes) {
    SLAM_REQUIRE(ctx && (d_ptr || !bytes), "slam_memset: null argument");
    if (!bytes) return SLAM_OK;
    SLAM_HIP(hipSetDevice(ctx->device));
    SLAM_HIP(hipMemsetAsync(d_ptr, value, bytes, ctx->stream));
    return SLAM_OK;
}

extern "C" int slam_copy(slam_ctx* ctx, void* d_dst, const void* d_src, uint64_t bytes) {
    SLAM_REQUIRE(ctx && ((d_dst && d_src) || !bytes), "slam_copy: null argument");
    if (!bytes) return SLAM_OK;
    SLAM_HIP(hipSetDevice(ctx->device));
    SLAM_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return SLAM_OK;
}

extern "C" int slam_upload(slam_ctx* ctx, void* d_dst, const void* h_src, uint64_t bytes) {
    SLAM_REQUIRE(ctx && ((d_dst && h_src) || !bytes), "slam_upload: null argument");
    if (!bytes) return SLAM_OK;
    SLAM_HIP(hipSetDevice(ctx->device));
    SLAM_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    SLAM_HIP(hipStreamSynchronize(ctx->stream));  // caller's buffer is only valid during the call
    return SLAM_OK;
}

extern "C" int slam_download(slam_ctx* ctx, void* h_dst, const void* d_src, uint64_t bytes) {
    SLAM_REQUIRE(ctx && ((h_dst && d_src) || !bytes), "slam_download: null argument");
    if (!bytes) return SLAM_OK;
    SLAM_HIP(hipSetDevice(ctx->device));
    SLAM_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP(hipStreamSynchronize(ctx->stream));
    return SLAM_OK;
}

int slam_workspace(slam_ctx* ctx, uint64_t bytes, void** out) {
    std::lock_guard<std::mutex> g(ctx->mu);
    if (bytes > ctx->workspace_bytes) {
        SLAM_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->workspace) SLAM_HIP(hipFree(ctx->workspace));
        ctx->workspace = nullptr;
        ctx->workspace_bytes = 0;
        uint64_t want = bytes + (bytes >> 2);  // headroom so repeated slightly larger calls do not realloc
        SLAM_HIP(hipMalloc(&ctx->workspace, want));
        ctx->workspace_bytes = want;
    }
    *out = ctx->workspace;
    return SLAM_OK;
}

int slam_io_arena(slam_ctx* ctx, uint64_t dev_bytes, uint64_t host_bytes, void** dev, void** host) {
    std::lock_guard<std::mutex> g(ctx->mu);
    if (dev_bytes > ctx->io_dev_bytes || host_bytes > ctx->io_host_bytes) SLAM_HIP(hipStreamSynchronize(ctx->stream));
    if (dev_bytes > ctx->io_dev_bytes) {
        if (ctx->io_dev) SLAM_HIP(hipFree(ctx->io_dev));
        ctx->io_dev = nullptr;
        ctx->io_dev_bytes = 0;
        const uint64_t want = dev_bytes + (dev_bytes >> 2) + 4096;
        SLAM_HIP(hipMalloc(&ctx->io_dev, want));
        ctx->io_dev_bytes = want;
    }
    if (host_bytes > ctx->io_host_bytes) {
        if (ctx->io_host) SLAM_HIP(hipHostFree(ctx->io_host));
        ctx->io_host = nullptr;
        ctx->io_host_bytes = 0;
        const uint64_t want = host_bytes + (host_bytes >> 2) + 4096;
        SLAM_HIP(hipHostMalloc(&ctx->io_host, want, hipHostMallocDefault));
        ctx->io_host_bytes = want;
    }
    *dev = ctx->io_dev;
    *host = ctx->io_host;
    return SLAM_OK;
}

extern "C" int slam_index_errors(slam_ctx* ctx, int64_t* count) {
    SLAM_REQUIRE(ctx && count, "slam_index_errors: null argument");
    SLAM_HIP(hipSetDevice(ctx->device));
    unsigned int h = 0;
    SLAM_HIP(hipMemcpyAsync(&h, slam_index_error_counter(ctx), sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP(hipMemsetAsync(slam_index_error_counter(ctx), 0, sizeof(h), ctx->stream));
    SLAM_HIP(hipStreamSynchronize(ctx->stream));
    *count = h;
    return SLAM_OK;
}

extern "C" int slam_io_counters(slam_ctx* ctx, uint64_t* h2d_bytes, uint64_t* d2h_bytes) {
    SLAM_REQUIRE(ctx, "slam_io_counters: null ctx");
    std::lock_guard<std::mutex> lk(ctx->io_mu);
    if (h2d_bytes) *h2d_bytes = ctx->io_h2d_bytes;
    if (d2h_bytes) *d2h_bytes = ctx->io_d2h_bytes;
    return SLAM_OK;
}

extern "C" int slam_timer_start(slam_ctx* ctx) {
    SLAM_REQUIRE(ctx, "slam_timer_start: null ctx");
    SLAM_HIP(hipSetDevice(ctx->device));
    SLAM_HIP(hipEventRecord(ctx->ev_start, ctx->stream));
    return SLAM_OK;
}

extern "C" int slam_timer_stop(slam_ctx* ctx, float* ms) {
    SLAM_REQUIRE(ctx && ms, "slam_timer_stop: null argument");
    SLAM_HIP(hipSetDevice(ctx->device));
    SLAM_HIP(hipEventRecord(ctx->ev_stop, ctx->stream));
    SLAM_HIP(hipEventSynchronize(ctx->ev_stop));
    SLAM_HIP(hipEventElapsedTime(ms, ctx->ev_start, ctx->ev_stop));
    return SLAM_OK;
}

extern "C" int slam_prof_enable(slam_ctx* ctx, int on) {
    SLAM_REQUIRE(ctx, "slam_prof_enable: null ctx");
    SLAM_HIP(hipSetDevice(ctx->device));
    if (on && !ctx->prof_ev) {
        ctx->prof_ev = new (std::nothrow) hipEvent_t[2 * slam_ctx::PROF_MAX];
        SLAM_REQUIRE(ctx->prof_ev, "out of host memory");
        for (int i = 0; i < 2 * slam_ctx::PROF_MAX; i++) SLAM_HIP(hipEventCreate(&ctx->prof_ev[i]));
    }
    ctx->prof_on = on ? 1 : 0;
    ctx->prof_n = 0;
    return SLAM_OK;
}

int slam_prof_begin(slam_ctx* ctx) {
    if (!ctx->prof_on || ctx->prof_n >= slam_ctx::PROF_MAX) return SLAM_OK;
    SLAM_HIP(hipEventRecord(ctx->prof_ev[2 * ctx->prof_n], ctx->stream));
    return SLAM_OK;
}

int slam_prof_end(slam_ctx* ctx) {
    if (!ctx->prof_on || ctx->prof_n >= slam_ctx::PROF_MAX) return SLAM_OK;
    SLAM_HIP(hipEventRecord(ctx->prof_ev[2 * ctx->prof_n + 1], ctx->stream));
    ctx->prof_n++;
    return SLAM_OK;
}

extern "C" int slam_prof_read(slam_ctx* ctx, int64_t* launches, double* total_ms) {
    SLAM_REQUIRE(ctx && launches && total_ms, "slam_prof_read: null argument");
    SLAM_HIP(hipSetDevice(ctx->device));
    SLAM_HIP(hipStreamSynchronize(ctx->stream));
    double sum = 0;
    for (int i = 0; i < ctx->prof_n; i++) {
        float ms = 0;
        SLAM_HIP(hipEventElapsedTime(&ms, ctx->prof_ev[2 * i], ctx->prof_ev[2 * i + 1]));
        sum += ms;
    }
    *launches = ctx->prof_n;
    *total_ms = sum;
    ctx->prof_n = 0;
    return SLAM_OK;
}
