// host_calls.hip — the per-frame calls of the reference on caller-owned HOST buffers, one entry point each.
//
// At the reference's own sizes (<= 200 features per frame, slam.py:23) the kernels take tens of
// microseconds and what a call costs is the number of PCIe round trips and synchronisations around them.
// Each function here packs its inputs into one pinned staging block, does ONE host-to-device copy,
// launches everything on the context stream, does ONE device-to-host copy and synchronises once - or, frame-sized, does no
// copy at all (the kernels read and write the pinned block) and waits for the last kernel's completion word (slam_wait_done).
// The staging block belongs to the context, so these calls serialise per context (ctypes drops the GIL:
// the reference's tracking thread and a backend thread may both be in here).
//   slam_bf_knn2_u256_host       cv2.BFMatcher.knnMatch(k=2) (the search behind feature_matchers.py:39)
//   slam_bf_match_host           BruteForceFeatureMatcher.match (feature_matchers.py:36-44)
//   slam_pose_optimize_host_f64  Frontend._correct_current_pose (frontend.py:298-393)
#include "internal.h"
#include <string.h>
#include <vector>

static inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

// Pinned host memory is device-accessible: for frame-sized inputs the kernels pull the few KB over PCIe and push
// the result back themselves, which measured 3-6 us (7-13 %) faster per call than two more copy calls
// (200 x 200: 30 -> 26 us, 2000 x 2000: 57 -> 51 us, profiles/r01_latency_small_calls.log).
static bool zero_copy(int64_t N, int64_t M) { return N <= 4096 && M <= 4096; }

extern "C" int slam_bf_knn2_u256_host(slam_ctx* ctx, const uint8_t* h_query, int64_t N, const uint8_t* h_train,
                                      int64_t M, int32_t* h_idx, int32_t* h_dist) {
    SLAM_REQUIRE(ctx, "slam_bf_knn2_u256_host: null ctx");
    SLAM_REQUIRE(N >= 0 && M >= 0 && N <= (1ll << 28) && M <= (1ll << 28), "bad sizes N=%lld M=%lld", (long long)N, (long long)M);
    if (N == 0) return SLAM_OK;
    SLAM_REQUIRE(h_query && h_idx && h_dist && (h_train || M == 0), "slam_bf_knn2_u256_host: null host pointer");
    std::lock_guard<std::mutex> lk(ctx->io_mu);
    SLAM_HIP(hipSetDevice(ctx->device));
    const uint64_t qbytes = (uint64_t)N * SLAM_DESC_BYTES, tbytes = (uint64_t)M * SLAM_DESC_BYTES;
    const uint64_t off_t = align_up(qbytes, 256), off_i = off_t + align_up(tbytes, 256);
    const uint64_t off_d = off_i + (uint64_t)N * 8, off_k = off_d + (uint64_t)N * 8, total = align_up(off_k + (uint64_t)N, 256);
    void *dev = nullptr, *host = nullptr;
    if (int rc = slam_io_arena(ctx, total, total, &dev, &host)) return rc;
    uint8_t* hb = (uint8_t*)host;
    uint8_t* db = (uint8_t*)dev;
    memcpy(hb, h_query, qbytes);
    if (tbytes) memcpy(hb + off_t, h_train, tbytes);
    ctx->io_h2d_bytes += qbytes + tbytes;
    ctx->io_d2h_bytes += (uint64_t)N * 16;
    if (zero_copy(N, M) && M > 0) {
        // frame-sized: the kernels read the pinned block and write the result into it over PCIe themselves, and the call
        // waits for the search's completion words instead of the stream (slam_bf_knn2_select with a count: ~4 us less; the
        // has-a-neighbour flags it leaves behind the tables are a by-product)
        int64_t with_neighbour = 0;
        if (int rc = slam_bf_knn2_select(ctx, hb, N, hb + off_t, M, 0, (int32_t*)(hb + off_i), (int32_t*)(hb + off_d), nullptr, 0, 0.0,
                                         hb + off_k, &with_neighbour)) return rc;
    } else if (zero_copy(N, M)) {
        if (int rc = slam_bf_knn2_u256(ctx, hb, N, hb + off_t, M, 0, (int32_t*)(hb + off_i), (int32_t*)(hb + off_d))) return rc;
        SLAM_HIP(hipStreamSynchronize(ctx->stream));
    } else {
        SLAM_HIP(hipMemcpyAsync(db, hb, off_t + tbytes, hipMemcpyHostToDevice, ctx->stream));
        if (int rc = slam_bf_knn2_u256(ctx, db, N, db + off_t, M, 0, (int32_t*)(db + off_i), (int32_t*)(db + off_d))) return rc;
        SLAM_HIP(hipMemcpyAsync(hb + off_i, db + off_i, (uint64_t)N * 16, hipMemcpyDeviceToHost, ctx->stream));
        SLAM_HIP(hipStreamSynchronize(ctx->stream));
    }
    memcpy(h_idx, hb + off_i, (uint64_t)N * 8);
    memcpy(h_dist, hb + off_d, (uint64_t)N * 8);
    return SLAM_OK;
}

extern "C" int slam_bf_match_host(slam_ctx* ctx, const uint8_t* h_query, int64_t N, const uint8_t* h_train,
                                  const void* d_train, int64_t M, void* d_query_keep, int mode, double param,
                                  int32_t* h_query_idx, int32_t* h_train_idx, float* h_distance, int64_t* h_count) {
    SLAM_REQUIRE(ctx, "slam_bf_match_host: null ctx");
    SLAM_REQUIRE(h_count, "slam_bf_match_host: null h_count");
    *h_count = 0;
    SLAM_REQUIRE(N >= 0 && M >= 0 && N <= (1ll << 28) && M <= (1ll << 28), "bad sizes N=%lld M=%lld", (long long)N, (long long)M);
    SLAM_REQUIRE(mode >= 0 && mode <= 3, "mode %d not in {0,1,2,3}", mode);
    SLAM_REQUIRE(!(h_train && d_train), "pass the train descriptors either as h_train or as d_train, not both");
    SLAM_REQUIRE(N == 0 || h_query, "slam_bf_match_host: null h_query");
    SLAM_REQUIRE(N == 0 || (h_query_idx && h_train_idx && h_distance), "slam_bf_match_host: null output pointer");
    SLAM_REQUIRE(M == 0 || h_train || d_train, "slam_bf_match_host: no train descriptors");
    std::lock_guard<std::mutex> lk(ctx->io_mu);
    SLAM_HIP(hipSetDevice(ctx->device));
    const uint64_t qbytes = (uint64_t)N * SLAM_DESC_BYTES, tbytes = h_train ? (uint64_t)M * SLAM_DESC_BYTES : 0;
    if (N == 0 || M == 0) {
        // nothing to report (OpenCV: no candidate -> no match); still hand the query rows over if asked to
        if (N && d_query_keep) {
            ctx->io_h2d_bytes += qbytes;
            SLAM_HIP(hipMemcpyAsync(d_query_keep, h_query, qbytes, hipMemcpyHostToDevice, ctx->stream));
            SLAM_HIP(hipStreamSynchronize(ctx->stream));
        }
        return SLAM_OK;
    }
    // device arena: [query | train | idx int32[N,2] | dist int32[N,2] | keep u8[N]]; staging mirrors it
    // (crossCheck, mode 3: ... | dist int32[N,2] | reverse tables int32[M,2] x 2 | idx int32[N] | dist int32[N])
    const uint64_t off_t = align_up(qbytes, 256), off_i = off_t + align_up(tbytes, 256);
    const uint64_t off_d = off_i + (uint64_t)N * 8, off_k = off_d + (uint64_t)N * 8;
    const uint64_t off_r = align_up(off_k, 256), off_o = align_up(off_r + (uint64_t)M * 16, 256);
    const uint64_t total = mode == 3 ? align_up(off_o + (uint64_t)N * 8, 256) : align_up(off_k + N, 256);
    void *dev = nullptr, *host = nullptr;
    if (int rc = slam_io_arena(ctx, total, total, &dev, &host)) return rc;
    uint8_t* hb = (uint8_t*)host;
    uint8_t* db = (uint8_t*)dev;
    memcpy(hb, h_query, qbytes);
    if (tbytes) memcpy(hb + off_t, h_train, tbytes);
    ctx->io_h2d_bytes += qbytes + tbytes;                      // a train side passed as d_train crosses nothing
    ctx->io_d2h_bytes += mode == 3 ? (uint64_t)N * 8 : (uint64_t)N * (mode ? 17 : 16);
    // frame-sized calls skip the copies: the kernels read the pinned block and write the result into it directly
    const bool zc = zero_copy(N, M);
    uint8_t* io = zc ? hb : db;
    const void* dq = io;
    void* keep_in_kernel = nullptr;
    if (d_query_keep && zc) {
        // frame-sized: the search reads the query rows from the pinned block and its first chunk's blocks, which hold
        // them in registers, leave the device copy the caller asked for (no copy command, no DMA latency before the launch)
        keep_in_kernel = d_query_keep;
    } else if (d_query_keep) {
        // the caller keeps this frame's rows on the device as the next call's train side
        SLAM_HIP(hipMemcpyAsync(d_query_keep, hb, qbytes, hipMemcpyHostToDevice, ctx->stream));
        if (tbytes) SLAM_HIP(hipMemcpyAsync(db + off_t, hb + off_t, tbytes, hipMemcpyHostToDevice, ctx->stream));
        dq = d_query_keep;
    } else if (!zc) {
        SLAM_HIP(hipMemcpyAsync(db, hb, off_t + tbytes, hipMemcpyHostToDevice, ctx->stream));
    }
    const void* dt = h_train ? (const void*)(io + off_t) : d_train;
    if (mode == 3) {
        // cv2.BFMatcher(crossCheck=True).match: the forward and the reverse search (train rows as queries) stay on
        // the device, only the per-query (idx, dist) pair of the mutual nearest neighbours comes back
        // Frame-sized (zero-copy) calls write both searches' tables straight into the pinned block and the HOST does the
        // mutual-nearest test (one gather of the reverse table through the forward one, <= 4096 rows: less than the gap in
        // front of another kernel): ONE launch per call.  Larger calls keep the tables on the device and run cross_emit_kernel.
        int32_t* fwd_idx = (int32_t*)(io + off_i);
        int32_t* fwd_dist = (int32_t*)(io + off_d);
        int32_t* rev_idx = (int32_t*)(io + off_r);
        int32_t* rev_dist = (int32_t*)(io + off_r + (uint64_t)M * 8);
        int32_t* o_idx = (int32_t*)(io + off_o);
        int32_t* o_dist = (int32_t*)(io + off_o + (uint64_t)N * 4);
        if (N <= SLAM_MAX_TRAIN_PER_PASS && M <= SLAM_MAX_TRAIN_PER_PASS) {
            // both searches in ONE launch: at frame size each of them is a cold, half-empty grid, and two of those back
            // to back cost two launch + drain latencies (200 features: 0.057 -> 0.04 ms per call)
            const slam_bf_search both[2] = {{dq, N, dt, M, 0, fwd_idx, fwd_dist}, {dt, M, dq, N, 0, rev_idx, rev_dist}};
            void* const keeps[2] = {keep_in_kernel, nullptr};
            if (int rc = slam_bf_knn2_batch_keep(ctx, 2, both, keeps, zc)) return rc;    // frame-sized: waits by polling
        } else {
            if (int rc = slam_bf_knn2_keep(ctx, dq, N, dt, M, 0, fwd_idx, fwd_dist, keep_in_kernel)) return rc;
            if (int rc = slam_bf_knn2_u256(ctx, dt, M, dq, N, 0, rev_idx, rev_dist)) return rc;
        }
        int64_t c = 0;
        if (zc) {
            ctx->io_d2h_bytes += (uint64_t)N * 8 + (uint64_t)M * 16;     // the four tables instead of the emitted pair
            if (!(N <= SLAM_MAX_TRAIN_PER_PASS && M <= SLAM_MAX_TRAIN_PER_PASS)) SLAM_HIP(hipStreamSynchronize(ctx->stream));
            for (int64_t n = 0; n < N; n++) {
                const int32_t t = fwd_idx[2 * n];
                if (t < 0 || t >= M || rev_idx[2 * (int64_t)t] != (int32_t)n) continue;
                h_query_idx[c] = (int32_t)n;
                h_train_idx[c] = t;
                h_distance[c] = (float)fwd_dist[2 * n];
                c++;
            }
        } else {
            if (int rc = slam_cross_launch(ctx, fwd_idx, fwd_dist, N, rev_idx, M, o_idx, o_dist)) return rc;
            SLAM_HIP(hipMemcpyAsync(hb + off_o, db + off_o, (uint64_t)N * 8, hipMemcpyDeviceToHost, ctx->stream));
            SLAM_HIP(hipStreamSynchronize(ctx->stream));
            const int32_t* ri = (const int32_t*)(hb + off_o);
            const int32_t* rd = (const int32_t*)(hb + off_o + (uint64_t)N * 4);
            for (int64_t n = 0; n < N; n++) {
                if (ri[n] < 0) continue;
                h_query_idx[c] = (int32_t)n;
                h_train_idx[c] = ri[n];
                h_distance[c] = (float)rd[n];
                c++;
            }
        }
        *h_count = c;
        return SLAM_OK;
    }
    int32_t* d_idx = (int32_t*)(io + off_i);
    int32_t* d_dist = (int32_t*)(io + off_d);
    // mode 0 (the reference's own call, dist_threshold=None, frontend.py:187) keeps every query that has a neighbour:
    // that needs no reduction over the queries, so no selection kernel is launched for it (one launch less per frame);
    // the Lowe ratio test (mode 2) needs none either and is made by the search's own decode (slam_bf_knn2_select);
    // the min-distance filter (mode 1, feature_matchers.py:41-43) needs the global minimum: its own kernel
    const bool select = mode != 0;
    bool waited = false;
    if (zc && mode != 1) {
        // frame-sized (the reference's own call): the results land in the pinned block and the call waits for the search's
        // completion words instead of the stream (bf_wait_done: ~4 us of a 22 us call); mode 0's flags say "has a neighbour"
        int64_t kept = 0;
        if (int rc = slam_bf_knn2_select(ctx, dq, N, dt, M, 0, d_idx, d_dist, keep_in_kernel, mode, param, io + off_k, &kept)) return rc;
        waited = true;
    } else if (mode == 2) {
        if (int rc = slam_bf_knn2_select(ctx, dq, N, dt, M, 0, d_idx, d_dist, keep_in_kernel, 2, param, io + off_k, nullptr)) return rc;
    } else {
        if (int rc = slam_bf_knn2_keep(ctx, dq, N, dt, M, 0, d_idx, d_dist, keep_in_kernel)) return rc;
        if (select && zc) {
            // the min-distance filter's one-workgroup kernel is the call's last: it stores the completion word
            if (int rc = slam_done_block(ctx, 0)) return rc;
            unsigned* done = (unsigned*)ctx->sel_host;
            const unsigned epoch = slam_done_epoch(ctx);
            if (int rc = slam_filter_launch(ctx, d_idx, d_dist, N, mode, param, io + off_k, done, epoch, &waited)) return rc;
            if (waited)
                if (int rc = slam_wait_done(ctx, done, 1, epoch)) return rc;
        } else if (select) {
            if (int rc = slam_filter_launch(ctx, d_idx, d_dist, N, mode, param, io + off_k)) return rc;
        }
    }
    if (!zc) SLAM_HIP(hipMemcpyAsync(hb + off_i, db + off_i, (uint64_t)N * (select ? 17 : 16), hipMemcpyDeviceToHost, ctx->stream));
    if (!waited) SLAM_HIP(hipStreamSynchronize(ctx->stream));
    // compact the kept rows (for modes 1 and 2 the selection itself was made on the device)
    const int32_t* ri = (const int32_t*)(hb + off_i);
    const int32_t* rd = (const int32_t*)(hb + off_d);
    const uint8_t* rk = hb + off_k;
    int64_t c = 0;
    for (int64_t n = 0; n < N; n++) {
        if (select ? !rk[n] : ri[2 * n] < 0) continue;
        h_query_idx[c] = (int32_t)n;
        h_train_idx[c] = ri[2 * n];
        h_distance[c] = (float)rd[2 * n];   // cv2 reports CV_32S distances converted to float32
        c++;
    }
    *h_count = c;
    return SLAM_OK;
}

extern "C" int slam_pose_optimize_host_f64(slam_ctx* ctx, const double* h_pose_in, const double* h_points,
                                           const double* h_meas, int64_t O, double fx, double fy, double cx,
                                           double cy, int rounds, int iterations, double chi2_threshold,
                                           double huber_delta, double* h_pose_out, uint8_t* h_inlier,
                                           double* h_chi2, int32_t* h_stats) {
    SLAM_REQUIRE(ctx, "slam_pose_optimize_host_f64: null ctx");
    SLAM_REQUIRE(O >= 0 && O <= (1 << 24), "O=%lld out of range [0, 2^24]", (long long)O);
    SLAM_REQUIRE(h_pose_in && h_pose_out && h_stats && (O == 0 || (h_points && h_meas && h_inlier && h_chi2)),
                 "slam_pose_optimize_host_f64: null host pointer");
    std::lock_guard<std::mutex> lk(ctx->io_mu);
    SLAM_HIP(hipSetDevice(ctx->device));
    // in: [pose 12 | points 3O | pad | meas 2O]   out: [pose 12 | chi2 O | stats (2 x int32) | inlier u8[O]]
    const uint64_t o = (uint64_t)O;
    const uint64_t off_p = 96, off_m = align_up(off_p + 24 * o, 16), in_bytes = align_up(off_m + 16 * o, 256);
    const uint64_t out_pose = in_bytes, out_chi2 = out_pose + 96, out_stats = out_chi2 + 8 * o, out_inl = out_stats + 8;
    const uint64_t total = align_up(out_inl + o + 1, 256);
    void *dev = nullptr, *host = nullptr;
    if (int rc = slam_io_arena(ctx, total, total, &dev, &host)) return rc;
    uint8_t* hb = (uint8_t*)host;
    uint8_t* db = (uint8_t*)dev;
    memcpy(hb, h_pose_in, 96);
    if (O) {
        memcpy(hb + off_p, h_points, 24 * o);
        memcpy(hb + off_m, h_meas, 16 * o);
    }
    ctx->io_h2d_bytes += off_m + 16 * o;
    ctx->io_d2h_bytes += out_inl + o - out_pose;
    if (O <= SLAM_POSE_STAGE) {
        // A frame's problem (<= 200 edges, slam.py:23) is staged into LDS by the kernel's first pass and its results are written
        // once at the end: the kernel reads the pinned block and writes into it directly (no copy commands), and the call waits
        // for the kernel's completion word instead of the stream (slam_wait_done).
        if (int rc = slam_done_block(ctx, 0)) return rc;
        unsigned* done = (unsigned*)ctx->sel_host;
        const unsigned epoch = slam_done_epoch(ctx);
        if (int rc = slam_pose_optimize_polled(ctx, (const double*)hb, (const double*)(hb + off_p), (const double*)(hb + off_m),
                                               O, fx, fy, cx, cy, rounds, iterations, chi2_threshold, huber_delta,
                                               (double*)(hb + out_pose), hb + out_inl, (double*)(hb + out_chi2),
                                               (int32_t*)(hb + out_stats), done, epoch))
            return rc;
        if (int rc = slam_wait_done(ctx, done, 1, epoch)) return rc;
    } else {
        SLAM_HIP(hipMemcpyAsync(db, hb, off_m + 16 * o, hipMemcpyHostToDevice, ctx->stream));
        if (int rc = slam_pose_optimize_f64(ctx, (const double*)db, (const double*)(db + off_p), (const double*)(db + off_m),
                                            O, fx, fy, cx, cy, rounds, iterations, chi2_threshold, huber_delta,
                                            (double*)(db + out_pose), db + out_inl, (double*)(db + out_chi2),
                                            (int32_t*)(db + out_stats)))
            return rc;
        SLAM_HIP(hipMemcpyAsync(hb + out_pose, db + out_pose, out_inl + o - out_pose, hipMemcpyDeviceToHost, ctx->stream));
        SLAM_HIP(hipStreamSynchronize(ctx->stream));
    }
    memcpy(h_pose_out, hb + out_pose, 96);
    memcpy(h_stats, hb + out_stats, 8);
    if (O) {
        memcpy(h_chi2, hb + out_chi2, 8 * o);
        memcpy(h_inlier, hb + out_inl, o);
    }
    return SLAM_OK;
}

// slam_ba_optimize_f64 on HOST buffers: the window goes up in ONE copy, the index tables the kernel wants (observations
// grouped by point and by pose) are built here by two stable counting sorts, the persistent kernel runs, the optimised
// state comes back in ONE copy.  The Python wrapper did the same with two argsorts, two bincounts, a uniqueness check, a
// bytearray, three allocations and three downloads: 0.45 ms of a 1.0 ms call at the reference's window size
// (backend.py:11: 7 keyframes).  Staging block, device arena and workspace all belong to the context (grow-only).
extern "C" int slam_ba_optimize_host_f64(slam_ctx* ctx, int64_t K, int64_t L, int64_t O, const double* h_poses /*[K,12]*/,
                                         const double* h_points /*[L,3]*/, const int32_t* h_obs_pose, const int32_t* h_obs_point,
                                         const double* h_meas /*[O,2]*/, const uint8_t* h_pose_fixed /*[K]*/, double fx, double fy,
                                         double cx, double cy, double huber_delta, int iterations, double* h_poses_out,
                                         double* h_points_out, double* h_stats /*[8]*/) {
    SLAM_REQUIRE(ctx, "slam_ba_optimize_host_f64: null ctx");
    SLAM_REQUIRE(K >= 1 && K <= 64 && L >= 1 && L <= (1 << 24) && O >= 0 && O <= SLAM_BA_LM_MAX_OBS, "bad sizes (K=%lld, L=%lld, O=%lld)",
                 (long long)K, (long long)L, (long long)O);
    SLAM_REQUIRE(h_poses && h_points && h_pose_fixed && h_poses_out && h_points_out && h_stats &&
                     (O == 0 || (h_obs_pose && h_obs_point && h_meas)), "slam_ba_optimize_host_f64: null host pointer");
    int64_t n_free = 0;
    for (int64_t k = 0; k < K; k++) n_free += h_pose_fixed[k] ? 0 : 1;
    SLAM_REQUIRE(n_free <= SLAM_BA_LM_MAX_FREE, "%lld moving poses: the one-launch form takes at most %d", (long long)n_free,
                 SLAM_BA_LM_MAX_FREE);
    for (int64_t o = 0; o < O; o++)
        SLAM_REQUIRE(h_obs_pose[o] >= 0 && h_obs_pose[o] < K && h_obs_point[o] >= 0 && h_obs_point[o] < L,
                     "observation %lld: index out of range (pose %d, point %d)", (long long)o, h_obs_pose[o], h_obs_point[o]);
    std::lock_guard<std::mutex> lk(ctx->io_mu);
    SLAM_HIP(hipSetDevice(ctx->device));
    // staging layout (16-byte aligned pieces): obs_pose | obs_point | meas | pt_ptr | pt_obs | ps_ptr | ps_obs | free | poses2 | points2 | stats
    const uint64_t o4 = align_up((uint64_t)(O ? O : 1) * 4, 16);
    const uint64_t off_op = 0, off_ol = off_op + o4, off_m = off_ol + o4, off_ptp = off_m + align_up((uint64_t)(O ? O : 1) * 16, 16);
    const uint64_t off_pto = off_ptp + align_up((uint64_t)(L + 1) * 4, 16), off_psp = off_pto + o4;
    const uint64_t off_pso = off_psp + align_up((uint64_t)(K + 1) * 4, 16), off_fr = off_pso + o4;
    const uint64_t off_T = off_fr + align_up((uint64_t)(n_free ? n_free : 1) * 4, 16), off_X = off_T + align_up((uint64_t)K * 192, 16);
    const uint64_t off_st = off_X + align_up((uint64_t)L * 48, 16), in_bytes = off_st, total = off_st + 64;
    uint64_t work = 0;
    if (int rc = slam_ba_optimize_workspace(K, L, O, &work)) return rc;
    void *dev = nullptr, *host = nullptr, *ws = nullptr;
    if (int rc = slam_io_arena(ctx, total, total, &dev, &host)) return rc;
    if (int rc = slam_workspace(ctx, work, &ws)) return rc;
    uint8_t* hb = (uint8_t*)host;
    uint8_t* db = (uint8_t*)dev;
    memset(hb, 0, in_bytes);
    int32_t* op = (int32_t*)(hb + off_op);   int32_t* ol = (int32_t*)(hb + off_ol);
    int32_t* ptp = (int32_t*)(hb + off_ptp); int32_t* pto = (int32_t*)(hb + off_pto);
    int32_t* psp = (int32_t*)(hb + off_psp); int32_t* pso = (int32_t*)(hb + off_pso);
    int32_t* fr = (int32_t*)(hb + off_fr);
    if (O) {
        memcpy(op, h_obs_pose, (size_t)O * 4);
        memcpy(ol, h_obs_point, (size_t)O * 4);
        memcpy(hb + off_m, h_meas, (size_t)O * 16);
    }
    // observations grouped by point / by pose, each group in ascending observation order (stable counting sorts)
    for (int64_t o = 0; o < O; o++) { ptp[ol[o] + 1]++; psp[op[o] + 1]++; }
    for (int64_t l = 0; l < L; l++) ptp[l + 1] += ptp[l];
    for (int64_t k = 0; k < K; k++) psp[k + 1] += psp[k];
    {
        std::vector<int32_t> at_pt(ptp, ptp + L), at_ps(psp, psp + K);
        for (int64_t o = 0; o < O; o++) { pto[at_pt[ol[o]]++] = (int32_t)o; pso[at_ps[op[o]]++] = (int32_t)o; }
    }
    // a (pose, point) pair may be observed once: within a point's group no pose may repeat (K <= 64: one mask per point)
    for (int64_t l = 0; l < L; l++) {
        uint64_t seen = 0;
        for (int32_t i = ptp[l]; i < ptp[l + 1]; i++) {
            const uint64_t bit = 1ull << op[pto[i]];
            SLAM_REQUIRE(!(seen & bit), "point %lld is observed twice by pose %d", (long long)l, op[pto[i]]);
            seen |= bit;
        }
    }
    int64_t nf = 0;
    for (int64_t k = 0; k < K; k++)
        if (!h_pose_fixed[k]) fr[nf++] = (int32_t)k;
    memcpy(hb + off_T, h_poses, (size_t)K * 96);            // the state sits in the first half of [2][K,12] / [2][L,3]
    memcpy(hb + off_X, h_points, (size_t)L * 24);
    ctx->io_h2d_bytes += in_bytes;
    ctx->io_d2h_bytes += (uint64_t)K * 96 + (uint64_t)L * 24 + 64;
    SLAM_HIP(hipMemcpyAsync(db, hb, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = slam_ba_optimize_f64(ctx, K, L, O, (const int32_t*)(db + off_op), (const int32_t*)(db + off_ol),
                                      (const double*)(db + off_m), (const int32_t*)(db + off_ptp), (const int32_t*)(db + off_pto),
                                      (const int32_t*)(db + off_psp), (const int32_t*)(db + off_pso), (const int32_t*)(db + off_fr), n_free,
                                      fx, fy, cx, cy, huber_delta, iterations, (double*)(db + off_T), (double*)(db + off_X), ws, work,
                                      (double*)(db + off_st)))
        return rc;
    // which half holds the result is only known afterwards: both come back (K * 192 + L * 48 bytes: 70 KB at the reference's window)
    SLAM_HIP(hipMemcpyAsync(hb + off_T, db + off_T, total - off_T, hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(h_stats, hb + off_st, 64);
    if (h_stats[5] == 2.0)      // the device-side table checks (the host checks above make this unreachable from here)
        return slam_set_error(SLAM_ERR_INVALID, "slam_ba_optimize_f64: an index outside the window or a malformed index table");
    if (!(h_stats[5] == 0.0 && h_stats[6] == h_stats[6]))
        // not the caller's fault: the launch could not get all its workgroups resident within the barriers' time limit (other
        // work holds compute units).  The window is untouched on the host; slam_ba_reduce_f64 / slam_ba_backsub_f64 (no
        // residency requirement) do the same adjustment.
        return slam_set_error(SLAM_ERR_BUSY, "slam_ba_optimize_f64 gave up at a grid barrier (device busy): status %g after %g trials",
                              h_stats[5], h_stats[3]);
    const int half = h_stats[6] != 0.0 ? 1 : 0;
    memcpy(h_poses_out, hb + off_T + (uint64_t)half * K * 96, (size_t)K * 96);
    memcpy(h_points_out, hb + off_X + (uint64_t)half * L * 24, (size_t)L * 24);
    return SLAM_OK;
}

