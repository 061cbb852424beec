// comm.hip — RCCL all-gather / broadcast of result rows between the ranks of a
// query-sharded run (one process per GPU, xGMI underneath).  The reference has
// no counterpart (single process, slam.py:22-35); sharding is SURVEY.md §8e.
// librccl is dlopen'ed on first use so the library also loads on hosts
// without RCCL; every entry point fails loudly if it is missing.
#include "internal.h"
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

namespace {
struct rccl_api {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    bool ok = false;
};
rccl_api g_rccl;
std::mutex g_rccl_mu;

int load_rccl() {
    std::lock_guard<std::mutex> g(g_rccl_mu);
    if (g_rccl.ok) return SLAM_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.handle) break;
    }
    if (!g_rccl.handle) return slam_set_error(SLAM_ERR_RCCL, "cannot dlopen librccl: %s", dlerror());
#define SYM(field, name)                                                                  \
    *(void**)(&g_rccl.field) = dlsym(g_rccl.handle, name);                                \
    if (!g_rccl.field) return slam_set_error(SLAM_ERR_RCCL, "librccl lacks symbol %s", name)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllGather, "ncclAllGather");
    SYM(Broadcast, "ncclBroadcast");
    SYM(GetErrorString, "ncclGetErrorString");
    SYM(GetVersion, "ncclGetVersion");
#undef SYM
    g_rccl.ok = true;
    return SLAM_OK;
}
}  // namespace

#define SLAM_NCCL(call)                                                                        \
    do {                                                                                       \
        ncclResult_t _r = (call);                                                              \
        if (_r != ncclSuccess)                                                                 \
            return slam_set_error(SLAM_ERR_RCCL, "%s failed: %s", #call, g_rccl.GetErrorString(_r)); \
    } while (0)

static_assert(sizeof(ncclUniqueId) == SLAM_COMM_ID_BYTES, "ncclUniqueId size changed");

extern "C" int slam_comm_unique_id(void* h_id) {
    SLAM_REQUIRE(h_id, "slam_comm_unique_id: null pointer");
    if (int rc = load_rccl()) return rc;
    ncclUniqueId id;
    SLAM_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(h_id, &id, sizeof(id));
    return SLAM_OK;
}

extern "C" int slam_comm_version(int* version) {
    SLAM_REQUIRE(version, "slam_comm_version: null pointer");
    *version = 0;
    if (int rc = load_rccl()) return rc;
    SLAM_NCCL(g_rccl.GetVersion(version));
    return SLAM_OK;
}

extern "C" int slam_comm_init(slam_ctx* ctx, int nranks, int rank, const void* h_id) {
    SLAM_REQUIRE(ctx && h_id, "slam_comm_init: null argument");
    SLAM_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank %d of %d", rank, nranks);
    if (ctx->comm) return slam_set_error(SLAM_ERR_STATE, "communicator already initialised");
    if (int rc = load_rccl()) return rc;
    SLAM_HIP(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, h_id, sizeof(id));
    ncclComm_t comm = nullptr;
    SLAM_NCCL(g_rccl.CommInitRank(&comm, nranks, id, rank));
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_nranks = nranks;
    return slam_second_stream(ctx);
}

extern "C" int slam_comm_destroy(slam_ctx* ctx) {
    SLAM_REQUIRE(ctx, "slam_comm_destroy: null ctx");
    if (!ctx->comm) return SLAM_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    slam_second_stream_destroy(ctx);
    ncclResult_t r = g_rccl.CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_rank = -1;
    ctx->comm_nranks = 0;
    if (r != ncclSuccess) return slam_set_error(SLAM_ERR_RCCL, "ncclCommDestroy: %s", g_rccl.GetErrorString(r));
    return SLAM_OK;
}

extern "C" int slam_comm_allgather(slam_ctx* ctx, const void* d_send, void* d_recv, uint64_t bytes_per_rank) {
    SLAM_REQUIRE(ctx, "slam_comm_allgather: null ctx");
    if (!ctx->comm) return slam_set_error(SLAM_ERR_STATE, "slam_comm_allgather: communicator not initialised");
    if (bytes_per_rank == 0) return SLAM_OK;
    SLAM_REQUIRE(d_send && d_recv, "slam_comm_allgather: null device pointer");
    SLAM_HIP(hipSetDevice(ctx->device));
    SLAM_NCCL(g_rccl.AllGather(d_send, d_recv, (size_t)bytes_per_rank, ncclInt8, (ncclComm_t)ctx->comm, ctx->stream));
    return SLAM_OK;
}

extern "C" int slam_comm_broadcast(slam_ctx* ctx, void* d_buf, uint64_t bytes, int root) {
    SLAM_REQUIRE(ctx, "slam_comm_broadcast: null ctx");
    if (!ctx->comm) return slam_set_error(SLAM_ERR_STATE, "slam_comm_broadcast: communicator not initialised");
    SLAM_REQUIRE(root >= 0 && root < ctx->comm_nranks, "bad root %d", root);
    if (bytes == 0) return SLAM_OK;
    SLAM_REQUIRE(d_buf, "slam_comm_broadcast: null device pointer");
    SLAM_HIP(hipSetDevice(ctx->device));
    SLAM_NCCL(g_rccl.Broadcast(d_buf, d_buf, (size_t)bytes, ncclInt8, root, (ncclComm_t)ctx->comm, ctx->stream));
    return SLAM_OK;
}

extern "C" int slam_comm_allgather_overlapped(slam_ctx* ctx, const void* d_send, void* d_recv,
                                              uint64_t bytes_per_rank, int buffer_id) {
    SLAM_REQUIRE(ctx, "slam_comm_allgather_overlapped: null ctx");
    if (!ctx->comm) return slam_set_error(SLAM_ERR_STATE, "slam_comm_allgather_overlapped: communicator not initialised");
    SLAM_REQUIRE(buffer_id == 0 || buffer_id == 1, "buffer_id must be 0 or 1");
    SLAM_REQUIRE(bytes_per_rank == 0 || (d_send && d_recv), "slam_comm_allgather_overlapped: null device pointer");
    SLAM_HIP(hipSetDevice(ctx->device));
    // comm stream starts after everything issued so far on the main stream (the search that filled d_send)
    SLAM_HIP(hipEventRecord(ctx->comm_ready, ctx->stream));
    SLAM_HIP(hipStreamWaitEvent(ctx->comm_stream, ctx->comm_ready, 0));
    if (bytes_per_rank)
        SLAM_NCCL(g_rccl.AllGather(d_send, d_recv, (size_t)bytes_per_rank, ncclInt8, (ncclComm_t)ctx->comm,
                                   ctx->comm_stream));
    SLAM_HIP(hipEventRecord(ctx->comm_done[buffer_id], ctx->comm_stream));
    ctx->comm_done_valid[buffer_id] = true;
    return SLAM_OK;
}

extern "C" int slam_comm_wait_buffer(slam_ctx* ctx, int buffer_id) {
    SLAM_REQUIRE(ctx, "slam_comm_wait_buffer: null ctx");
    SLAM_REQUIRE(buffer_id == 0 || buffer_id == 1, "buffer_id must be 0 or 1");
    if (!ctx->comm_done_valid[buffer_id]) return SLAM_OK;   // nothing in flight for this buffer
    SLAM_HIP(hipSetDevice(ctx->device));
    SLAM_HIP(hipStreamWaitEvent(ctx->stream, ctx->comm_done[buffer_id], 0));
    return SLAM_OK;
}
