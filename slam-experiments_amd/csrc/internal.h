// internal.h — shared state of libslamhip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <string.h>
#include <unordered_map>
#include <atomic>
#include <mutex>
#include "slamhip.h"

#define SLAM_BF_TBL_RING 8
#define SLAM_BF_TBL_SLOT 4096

struct slam_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    std::mutex mu;                                  // guards allocs / workspace growth
    std::unordered_map<void*, uint64_t> allocs;     // device pointers handed out by slam_malloc
    void* workspace = nullptr;                      // partial top-2 tables etc.
    uint64_t workspace_bytes = 0;
    int num_cu = 0;
    void* bf_state_mem = nullptr;                   // matcher merge state (best/bound/arrivals), clean between launches
    int64_t bf_state_rows = 0;
    int bf_knob[SLAM_BF_KNOBS] = {};                 // matcher tuning overrides (slam_bf_set_tuning), 0 = heuristic
    // chunk boundary tables of the recent searches: a ring of small slots + one big slot, device and pinned host
    void* bf_tbl_dev = nullptr;
    void* bf_tbl_host = nullptr;
    bool bf_tbl_ready = false;                      // device block, pinned block and all ring events exist
    int bf_tbl_n[SLAM_BF_TBL_RING + 1] = {};        // entries held by each slot
    bool bf_tbl_busy[SLAM_BF_TBL_RING + 1] = {};    // an event is pending behind the slot's last launch
    hipEvent_t bf_tbl_ev[SLAM_BF_TBL_RING] = {};
    int bf_tbl_cur = 0;                             // slot of the most recent search
    void* scratch = nullptr;                        // 4 KiB device scratch (filter counters, reductions)
    void* sel_host = nullptr;                       // pinned host block the search's fused selection writes its per-wave counts to
    uint64_t sel_host_bytes = 0;
    std::atomic<unsigned> done_epoch{0}, polled_calls{0};   // searches waited for by polling the block's completion words (bf_wait_done)
    void* io_dev = nullptr;                         // device arena of the host-buffer entry points (grow-only)
    uint64_t io_dev_bytes = 0;
    void* io_host = nullptr;                        // pinned host staging for the same (grow-only)
    uint64_t io_host_bytes = 0;
    std::mutex io_mu;                               // one host-buffer call at a time per context (they share the arena)
    uint64_t io_h2d_bytes = 0, io_d2h_bytes = 0;    // bytes the host-buffer entry points moved over PCIe (copies and zero-copy), under io_mu
    // profiling of the dominant kernel
    int prof_on = 0;
    static const int PROF_MAX = 4096;
    hipEvent_t* prof_ev = nullptr;                  // 2*PROF_MAX events, created lazily
    int prof_n = 0;
    // RCCL
    void* comm = nullptr;                           // ncclComm_t
    int comm_rank = -1, comm_nranks = 0;
    hipStream_t comm_stream = nullptr;              // second stream: overlapped all-gathers
    hipEvent_t comm_ready = nullptr;                // main stream -> comm stream hand-off
    hipEvent_t comm_done[2] = {nullptr, nullptr};   // last gather of result buffer 0 / 1
    bool comm_done_valid[2] = {false, false};
};

int slam_set_error(int code, const char* fmt, ...);

#define SLAM_HIP(call)                                                              \
    do {                                                                            \
        hipError_t _e = (call);                                                     \
        if (_e != hipSuccess)                                                       \
            return slam_set_error(SLAM_ERR_HIP, "%s failed: %s (%s:%d)", #call,     \
                                  hipGetErrorString(_e), __FILE__, __LINE__);       \
    } while (0)

#define SLAM_REQUIRE(cond, ...)                                                     \
    do {                                                                            \
        if (!(cond)) return slam_set_error(SLAM_ERR_INVALID, __VA_ARGS__);          \
    } while (0)

// grows ctx->workspace to at least `bytes` (stream-synchronising when it grows)
int slam_workspace(slam_ctx* ctx, uint64_t bytes, void** out);
// device arena + pinned staging of at least these sizes for one host-buffer call (stream-synchronising when they grow)
int slam_io_arena(slam_ctx* ctx, uint64_t dev_bytes, uint64_t host_bytes, void** dev, void** host);
// slam_bf_knn2_u256 that also leaves a copy of the query rows at d_keep (32*N bytes of device memory; M must be > 0
// for the copy to happen: with no train rows no kernel reads the queries)
int slam_bf_knn2_keep(slam_ctx* ctx, const void* d_query, int64_t N, const void* d_train, int64_t M, int64_t train_base,
                      int32_t* d_idx, int32_t* d_dist, void* d_keep);
// slam_bf_knn2_select_u256 with an optional copy of the query rows (d_query_keep) and an optional count (h_count null: no wait)
int slam_bf_knn2_select(slam_ctx* ctx, const void* d_query, int64_t N, const void* d_train, int64_t M, int64_t train_base,
                        int32_t* d_idx, int32_t* d_dist, void* d_query_keep, int mode, double param, uint8_t* d_sel_keep,
                        int64_t* h_count);
// slam_bf_knn2_batch_u256 with optional per-search copies of the query rows (h_keep[i] or null; h_keep may be null)
int slam_bf_knn2_batch_keep(slam_ctx* ctx, int64_t B, const slam_bf_search* h_searches, void* const* h_keep, bool wait = false);
// Completion by polling (bf_hamming.hip, slam_wait_done): the context's pinned block starts with SLAM_BF_DONE_FLAGS words that
// the last kernel of a frame-sized call stores its epoch into once the results are visible to the host.
#define SLAM_BF_DONE_FLAGS 64
#define SLAM_BF_DONE_BYTES (SLAM_BF_DONE_FLAGS * 4)
int slam_done_block(slam_ctx* ctx, uint64_t extra);
unsigned slam_done_epoch(slam_ctx* ctx);
int slam_wait_done(slam_ctx* ctx, const unsigned* flags, int count, unsigned epoch);
#define SLAM_POSE_STAGE 512      // edges the pose refinement keeps in LDS (pose_opt.hip PO_STAGE): up to here a host call is zero-copy
int slam_pose_optimize_polled(slam_ctx* ctx, const double* d_pose_in, const double* d_points, const double* d_meas, int64_t O,
                              double fx, double fy, double cx, double cy, int rounds, int iterations, double chi2_threshold,
                              double huber_delta, double* d_pose_out, uint8_t* d_inlier, double* d_chi2, int32_t* d_stats,
                              unsigned* done, unsigned epoch);
// the filter kernels of slam_bf_match_filter without the read-back (asynchronous on the ctx stream)
int slam_filter_launch(slam_ctx* ctx, const int32_t* d_idx, const int32_t* d_dist, int64_t N, int mode, double param,
                       uint8_t* d_keep, unsigned* done = nullptr, unsigned epoch = 0, bool* polled = nullptr);
// the crossCheck kernels of slam_bf_cross_check without the read-back (asynchronous on the ctx stream)
int slam_cross_launch(slam_ctx* ctx, const int32_t* d_fwd_idx, const int32_t* d_fwd_dist, int64_t N,
                      const int32_t* d_rev_idx, int64_t M, int32_t* d_out_idx, int32_t* d_out_dist);
// second stream + hand-off events shared by the RCCL and the peer-copy all-gathers (p2p.hip)
int slam_second_stream(slam_ctx* ctx);
void slam_second_stream_destroy(slam_ctx* ctx);
// sticky device counter of out-of-range pose / point indices met by the residual kernels (in ctx->scratch)
static inline unsigned int* slam_index_error_counter(slam_ctx* ctx) { return (unsigned int*)((char*)ctx->scratch + 1024); }
// event bracket around the dominant kernel when profiling is on
int slam_prof_begin(slam_ctx* ctx);
int slam_prof_end(slam_ctx* ctx);
