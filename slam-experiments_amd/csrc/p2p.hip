// p2p.hip — direct all-gather over xGMI peer mappings, no communicator library.
//
// The result tables are sharded by query rows: every rank owns one slot of a gathered buffer that
// exists on every GPU.  With the peers' buffers mapped through HIP IPC a rank simply writes its slot into each of
// them: one point-to-point copy per xGMI link, all links busy at once, one step (a ring would take nranks-1).
// Nobody else writes that slot and nobody reads a gathered buffer before the launcher's barrier, so the copies
// need no per-pass synchronisation between processes; they run on the context's second stream behind the search
// that produced the slot, exactly like the RCCL path (slam_comm_allgather_overlapped), and share its events.
// bench.py uses this path when the RCCL communicator cannot be created.
#include "internal.h"

int slam_second_stream(slam_ctx* ctx) {
    if (ctx->comm_stream) return SLAM_OK;
    SLAM_HIP(hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
    SLAM_HIP(hipEventCreateWithFlags(&ctx->comm_ready, hipEventDisableTiming));
    for (int i = 0; i < 2; i++) {
        SLAM_HIP(hipEventCreateWithFlags(&ctx->comm_done[i], hipEventDisableTiming));
        ctx->comm_done_valid[i] = false;
    }
    return SLAM_OK;
}

void slam_second_stream_destroy(slam_ctx* ctx) {
    if (ctx->comm_stream) {
        (void)hipStreamSynchronize(ctx->comm_stream);
        (void)hipStreamDestroy(ctx->comm_stream);
        ctx->comm_stream = nullptr;
    }
    if (ctx->comm_ready) { (void)hipEventDestroy(ctx->comm_ready); ctx->comm_ready = nullptr; }
    for (int i = 0; i < 2; i++) {
        if (ctx->comm_done[i]) { (void)hipEventDestroy(ctx->comm_done[i]); ctx->comm_done[i] = nullptr; }
        ctx->comm_done_valid[i] = false;
    }
}

static_assert(sizeof(hipIpcMemHandle_t) == SLAM_P2P_HANDLE_BYTES, "hipIpcMemHandle_t size changed");

extern "C" int slam_p2p_export(slam_ctx* ctx, void* d_ptr, void* h_handle) {
    SLAM_REQUIRE(ctx && d_ptr && h_handle, "slam_p2p_export: null argument");
    {
        std::lock_guard<std::mutex> g(ctx->mu);
        SLAM_REQUIRE(ctx->allocs.count(d_ptr), "slam_p2p_export: pointer is not the base of a slam_malloc allocation");
    }
    SLAM_HIP(hipSetDevice(ctx->device));
    hipIpcMemHandle_t h;
    SLAM_HIP(hipIpcGetMemHandle(&h, d_ptr));
    memcpy(h_handle, &h, sizeof(h));
    return SLAM_OK;
}

extern "C" int slam_p2p_open(slam_ctx* ctx, const void* h_handle, void** d_peer_ptr) {
    SLAM_REQUIRE(ctx && h_handle && d_peer_ptr, "slam_p2p_open: null argument");
    SLAM_HIP(hipSetDevice(ctx->device));
    hipIpcMemHandle_t h;
    memcpy(&h, h_handle, sizeof(h));
    void* p = nullptr;
    SLAM_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    *d_peer_ptr = p;
    return SLAM_OK;
}

extern "C" int slam_p2p_close(slam_ctx* ctx, void* d_peer_ptr) {
    SLAM_REQUIRE(ctx, "slam_p2p_close: null ctx");
    if (!d_peer_ptr) return SLAM_OK;
    SLAM_HIP(hipSetDevice(ctx->device));
    SLAM_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->comm_stream) SLAM_HIP(hipStreamSynchronize(ctx->comm_stream));
    SLAM_HIP(hipIpcCloseMemHandle(d_peer_ptr));
    return SLAM_OK;
}

extern "C" int slam_p2p_allgather_overlapped(slam_ctx* ctx, const void* d_send, uint64_t bytes_per_rank, int rank,
                                             void* const* h_peer_bufs, int nranks, int buffer_id) {
    SLAM_REQUIRE(ctx, "slam_p2p_allgather_overlapped: null ctx");
    SLAM_REQUIRE(nranks >= 1 && nranks <= 64 && rank >= 0 && rank < nranks, "bad rank %d of %d", rank, nranks);
    SLAM_REQUIRE(buffer_id == 0 || buffer_id == 1, "buffer_id must be 0 or 1");
    SLAM_REQUIRE(h_peer_bufs && (bytes_per_rank == 0 || d_send), "slam_p2p_allgather_overlapped: null pointer");
    SLAM_HIP(hipSetDevice(ctx->device));
    if (int rc = slam_second_stream(ctx)) return rc;
    // the copies start after everything issued so far on the main stream (the search that filled d_send)
    SLAM_HIP(hipEventRecord(ctx->comm_ready, ctx->stream));
    SLAM_HIP(hipStreamWaitEvent(ctx->comm_stream, ctx->comm_ready, 0));
    if (bytes_per_rank) {
        for (int step = 1; step < nranks; step++) {
            const int p = (rank + step) % nranks;          // staggered: at any moment the ranks target different peers
            SLAM_REQUIRE(h_peer_bufs[p], "slam_p2p_allgather_overlapped: peer %d is not mapped", p);
            char* dst = (char*)h_peer_bufs[p] + (uint64_t)rank * bytes_per_rank;
            SLAM_HIP(hipMemcpyAsync(dst, d_send, bytes_per_rank, hipMemcpyDefault, ctx->comm_stream));   // kind inferred: the target may live on another GPU
        }
    }
    SLAM_HIP(hipEventRecord(ctx->comm_done[buffer_id], ctx->comm_stream));
    ctx->comm_done_valid[buffer_id] = true;
    return SLAM_OK;
}
