/*
 * slamhip.h — C ABI of libslamhip.so, the MI355X (gfx950) drop-in for the one
 * data-parallel hot path of ViV99/slam-experiments.
 *
 * The reference is pure Python; its "FFI" for this path is the call into the
 * cv2 / g2o wheels.  Each entry point below names the reference call site it
 * replaces (file:line in /root/reference):
 *
 *   slam_bf_knn2_u256      cv2.BFMatcher(NORM_HAMMING).match / knnMatch(k=2)
 *                          behind BruteForceFeatureMatcher.match
 *                          (feature_matchers.py:33-39), called from
 *                          Frontend._match_features (frontend.py:181-187)
 *   slam_bf_match_filter   the post-match filter of feature_matchers.py:41-43
 *                          (+ Lowe ratio / crossCheck, OpenCV semantics)
 *   slam_reproj_rj_f64     EdgeProjectionPoseOnly.compute_error /
 *                          linearize_oplus (frontend.py:272-291), driven by
 *                          Frontend._correct_current_pose (frontend.py:298-393)
 *   slam_pose_normal_eq_f64 / slam_pose_optimize_f64
 *                          the g2o graph + LM loop of Frontend._correct_current_pose
 *                          (frontend.py:298-393)
 *   slam_comm_*            no reference counterpart (the reference is single
 *                          process); RCCL all-gather of per-shard top-2 rows
 *
 * Conventions
 *   - every function returns 0 on success or a negative slam_status; the
 *     message of the last failure on the calling thread is slam_last_error().
 *   - no exceptions cross the boundary; no torch / numpy types in signatures.
 *   - pointers named d_* are DEVICE pointers obtained from slam_malloc on the
 *     same context; pointers named h_* are host pointers owned by the caller
 *     and only read/written during the call.
 *   - a slam_ctx owns one HIP device + one HIP stream; calls on one ctx are
 *     stream-ordered, different ctxs are independent (one per thread / GPU).
 *     The host-buffer entry points (*_host) may be called from several threads
 *     on one ctx (they serialise on its staging block); the device-pointer
 *     entry points share the ctx's workspace and merge state and expect ONE
 *     calling thread per ctx at a time.
 *   - empty inputs are not errors: N == 0 is a no-op; M == 0 yields idx = -1,
 *     dist = INT32_MAX (OpenCV's "no neighbour").
 */
#ifndef SLAMHIP_H
#define SLAMHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct slam_ctx slam_ctx;

#define SLAM_API __attribute__((visibility("default")))

enum slam_status {
    SLAM_OK = 0,
    SLAM_ERR_INVALID = -1,   /* bad argument (null pointer, negative size, limit exceeded) */
    SLAM_ERR_HIP = -2,       /* a HIP runtime call failed */
    SLAM_ERR_NO_DEVICE = -3, /* no gfx950 device visible */
    SLAM_ERR_RCCL = -4,      /* RCCL missing or a collective failed */
    SLAM_ERR_STATE = -5,     /* call made in the wrong state (e.g. comm not initialised) */
    SLAM_ERR_BUSY = -6       /* a resource condition, not a caller error: a kernel that needs all its workgroups resident at
                                once (slam_ba_optimize_f64) could not get them - retry, or use the multi-launch form */
};

/* descriptor geometry: 256-bit ORB descriptors, one row = 32 bytes
 * (Frame.get_descriptors -> (n,32) uint8, primitives.py:200-205) */
#define SLAM_DESC_BYTES 32
/* top-2 keys pack (distance << 23 | index): one launch covers at most 2^23
 * train rows; larger train sets are looped by the library */
#define SLAM_MAX_TRAIN_PER_PASS (1 << 23)
#define SLAM_NO_MATCH_IDX (-1)
#define SLAM_NO_MATCH_DIST 2147483647

/* ---- library / context ------------------------------------------------- */
SLAM_API const char* slam_last_error(void);
SLAM_API const char* slam_version(void);
SLAM_API int slam_device_count(int* count);
SLAM_API int slam_ctx_create(int device, slam_ctx** out);
SLAM_API int slam_ctx_destroy(slam_ctx* ctx);
SLAM_API int slam_ctx_device(slam_ctx* ctx, int* device);
SLAM_API int slam_sync(slam_ctx* ctx);

/* ---- device memory (caller keeps the pointer, library tracks it) ------- */
SLAM_API int slam_malloc(slam_ctx* ctx, uint64_t bytes, void** d_ptr);
SLAM_API int slam_free(slam_ctx* ctx, void* d_ptr);
SLAM_API int slam_memset(slam_ctx* ctx, void* d_ptr, int value, uint64_t bytes);
SLAM_API int slam_copy(slam_ctx* ctx, void* d_dst, const void* d_src, uint64_t bytes); /* device to device, asynchronous on the ctx stream */
SLAM_API int slam_upload(slam_ctx* ctx, void* d_dst, const void* h_src, uint64_t bytes);
SLAM_API int slam_download(slam_ctx* ctx, void* h_dst, const void* d_src, uint64_t bytes);

/* ---- stream timers (HIP events on the ctx stream) ----------------------- */
SLAM_API int slam_timer_start(slam_ctx* ctx);
SLAM_API int slam_timer_stop(slam_ctx* ctx, float* ms); /* synchronises */
/* when enabled, every slam_bf_knn2_u256 / slam_reproj_rj_f64 brackets its
 * dominant kernel with events; slam_prof_read returns count and total ms */
SLAM_API int slam_prof_enable(slam_ctx* ctx, int on);
SLAM_API int slam_prof_read(slam_ctx* ctx, int64_t* launches, double* total_ms); /* synchronises, then resets */

/* ---- hot path 1: brute-force Hamming top-2 ------------------------------ */
/* For each of N query rows, the two nearest of M train rows under
 * popcount(q xor t), ordered by (distance asc, train index asc) — the order
 * cv2.BFMatcher.knnMatch(k=2) produces.  d_idx/d_dist are int32 [N,2];
 * column 0 is the 1-NN (what feature_matchers.py:39 returns), column 1 the
 * 2-NN; missing neighbours are (-1, INT32_MAX).  train_base is added to
 * every reported index (used by shards). */
SLAM_API int slam_bf_knn2_u256(slam_ctx* ctx, const void* d_query, int64_t N,
                      const void* d_train, int64_t M, int64_t train_base,
                      int32_t* d_idx, int32_t* d_dist);

/* Several INDEPENDENT searches in ONE launch: search i is slam_bf_knn2_u256(d_query, N, d_train, M, train_base, d_idx,
 * d_dist) of the i-th entry, bit for bit.  Frame-sized searches (the reference matches <= 200 x 200 per frame,
 * frontend.py:181-187; BASELINE configs[1] is 4096 x 4096) are bound by launch and drain latency on a half-empty chip,
 * not by their scan: crossCheck's forward and reverse search, or a handful of candidate verifications, fill one grid
 * instead of queueing cold ones.  At most SLAM_BF_BATCH_MAX searches per call, each over at most 2^23 train rows;
 * h_searches is a HOST array (the descriptors travel in the kernel arguments; nothing is uploaded).  N == 0 entries are
 * skipped, M == 0 entries report "no neighbour".  Asynchronous on the ctx stream. */
#define SLAM_BF_BATCH_MAX 32
typedef struct slam_bf_search {
    const void* d_query;
    int64_t N;
    const void* d_train;
    int64_t M;
    int64_t train_base;
    int32_t* d_idx;
    int32_t* d_dist;
} slam_bf_search;
SLAM_API int slam_bf_knn2_batch_u256(slam_ctx* ctx, int64_t B, const slam_bf_search* h_searches);

/* Merge G partial top-2 tables ([G][N][2] idx and dist, already holding
 * global indices) into one, by (dist, idx) order.  Used by train-sharded
 * runs and by passes over train sets larger than 2^23 rows. */
SLAM_API int slam_bf_merge_top2(slam_ctx* ctx, const int32_t* d_idx_parts,
                       const int32_t* d_dist_parts, int64_t G, int64_t N,
                       int32_t* d_idx, int32_t* d_dist);

/* Host-pointer convenience (uploads, runs, downloads; PCIe inclusive). */
SLAM_API int slam_bf_knn2_u256_host(slam_ctx* ctx, const uint8_t* h_query, int64_t N,
                           const uint8_t* h_train, int64_t M,
                           int32_t* h_idx, int32_t* h_dist);

/* Tuning overrides for experiments, per context.  h_knobs is an int32 array of up to SLAM_BF_KNOBS entries (missing
 * entries and a NULL array mean 0 = the shipped choice):
 *   [0] R             queries per lane: 1, 2, 4 or 8 (shipped: 1)
 *   [1] blocks_per_cu grid size target (shipped: 16 while there are no more query blocks than CUs, 32 above; train sets
 *                     below 16384 rows follow the chunk rule of [7] instead)
 *   [2] lead_rows     train rows given to the leader chunks: short chunks at the head of the dispatch order that
 *                     publish exact per-query bounds early (shipped: M/8 up to 8192 for M >= 16384, none below);
 *                     -1 = no leaders
 *   [3] lead_chunk    rows per leader chunk, a multiple of 32 (shipped: about one leader block per CU)
 *   [4] tail          number of linearly shrinking chunks at the end of the grid (shipped: up to 32, none for grids of
 *                     at most two blocks per CU); -1 = none
 *   [5] feed          how train rows reach the lanes at R = 1: 1 = through SGPRs (scalar loads, no LDS), -1 = through an LDS
 *                     tile (shipped: SGPRs whenever the rows are in device memory, and for chunks of at least 512 rows; the
 *                     LDS tile for rows in pinned host memory - the zero-copy frame-sized host calls)
 *   [6] cold          rows a chunk folds in WITHOUT a filter when it starts before anybody has published a bound for its
 *                     queries, a multiple of 16 (shipped: 128, and the whole chunk for train sets below 16384 rows whose
 *                     chunks have at most 384 rows); -1 = none
 *   [7] chunk         rows per uniform chunk for train sets below 16384 rows, a multiple of 32 (shipped: one block per CU
 *                     up to 128 rows a chunk, about 8 sqrt(that) beyond, at most 512)
 *   [8] queue         1 = a queue plan: as many worker blocks per query block as are resident at once, whose waves draw
 *                     the chunks by ticket and keep their top-2 from chunk to chunk; -1 = one block per chunk (the plans
 *                     above).  Shipped: a queue plan for train sets of at least 16384 rows when every query block gets at
 *                     least two and at most 256 workers with at least 1024 rows apiece (four, for train sets beyond 131072 rows) and they fill at least 96 % of
 *                     the resident slots; then [7] forces the rows of a
 *                     uniform chunk, [4] > 0 the shortest chunk at the end of the queue, [4] = -1 no shrinking chunks
 *   [9] merge         how the workers of a queue plan exchange what they know: 1 = by merging their best two rows into the
 *                     per-query slot and reading its 2nd row back (the exact 2nd-best distance of everything folded in so
 *                     far), -1 = through a per-query bound (the minimum of the workers' own 2nd-best distances, as the
 *                     one-block-per-chunk plans do).  Shipped: merging from 12 to 96 workers per query block */
#define SLAM_BF_KNOBS 10
SLAM_API int slam_bf_set_tuning(slam_ctx* ctx, const int32_t* h_knobs, int count);
/* The launch plan slam_bf_knn2_u256 would use for N x M on this context: h_plan int32 [10] =
 * {R, query blocks, uniform chunk rows, chunks, leader rows, leader chunks, shrinking tail chunks, CUs,
 *  feed (1 = train rows through SGPRs, 0 = through an LDS tile), unfiltered rows at a cold chunk start}. */
SLAM_API int slam_bf_plan_info(slam_ctx* ctx, int64_t N, int64_t M, int32_t* h_plan);
/* The same plan WITHOUT a device: a pure function of the CU count, the knobs (as slam_bf_set_tuning; NULL / 0 = shipped)
 * and the shape, so that the planner can be held to its invariants on a host without a GPU.  h_plan int32 [14] =
 * slam_bf_plan_info's ten entries (h_plan[7] = num_cu) + {table-free: the kernel computes its chunk from the block index
 * and no boundary table is uploaded, bound-free: no block reads or writes a bound, workers: worker blocks per query block
 * of a queue plan (knob [8]; 0 = one block per chunk), resident: blocks per CU the planner counts on for that, + 256 when the workers exchange by merging (knob [9])}.  The chunk boundary table (chunks + 1
 * ascending row indices from 0 to M) goes to h_tbl (up to tbl_cap entries; may be NULL) and its length to *tbl_len.
 * rows_on_host: the train rows lie in pinned host memory (frame-sized host calls).  qb_all: the query blocks of all the
 * searches that share the launch (slam_bf_knn2_batch_u256), 0 for a search that has the grid to itself. */
SLAM_API int slam_bf_plan_describe(int num_cu, const int32_t* h_knobs, int count, int64_t N, int64_t M, int64_t qb_all,
                                   int rows_on_host, int32_t* h_plan, int32_t* h_tbl, int64_t tbl_cap, int64_t* tbl_len);
/* Restore the matcher's per-context merge state to its idle values.  Every search leaves it clean by itself;
 * call this after a search failed part-way (the library does so on a failed launch).  Stream-ordered. */
SLAM_API int slam_bf_reset_state(slam_ctx* ctx);
/* Diagnostics: waits for the context's stream, then counts the 32-bit words of the merge state that are not at their idle
 * value (*h_words; 0 after every completed search, whatever its plan).  A leftover would corrupt the next search silently,
 * so the tests and tools/stress_state.py assert on this directly. */
SLAM_API int slam_bf_state_dirty(slam_ctx* ctx, int64_t* h_words);

/* slam_bf_knn2_u256 and a selection in ONE launch, for the selections that need no reduction over the queries: the block
 * that decodes a query block's result flags its queries as it writes them.
 *   mode 0: 1-NN of every query that has one (bf.match, feature_matchers.py:39,44)
 *   mode 2: Lowe ratio, kept iff dist0 < param * dist1 (strict; needs 2 neighbours) - BASELINE configs[1]
 * d_keep uint8 [N] (1 = kept); *h_count = rows kept, once the search is done (h_count NULL: asynchronous, no count; with a count and <= 16384 queries the call waits by polling completion words in pinned memory, see slam_bf_match_host, otherwise it synchronises the stream).  Same
 * flags and count as slam_bf_knn2_u256 + slam_bf_match_filter; mode 1 (the min-distance filter) needs the global minimum
 * and stays there.  Train sets of more than 2^23 rows and empty ones run the two steps internally. */
SLAM_API int slam_bf_knn2_select_u256(slam_ctx* ctx, const void* d_query, int64_t N, const void* d_train, int64_t M,
                                      int64_t train_base, int32_t* d_idx, int32_t* d_dist, int mode, double param,
                                      uint8_t* d_keep, int64_t* h_count);

/* Post-match selection on the device (feature_matchers.py:41-43 and the
 * OpenCV knn / ratio semantics).  Input: the [N,2] tables above.
 *   mode 0: 1-NN of every query that has one                 (bf.match, feature_matchers.py:39,44)
 *   mode 1: 1-NN kept iff dist < max(2*min_dist, param)      (feature_matchers.py:42-43, strict <)
 *   mode 2: Lowe ratio: kept iff dist0 < param * dist1       (strict; needs 2 neighbours)
 * d_keep is uint8 [N]; *h_count receives the number kept; *h_min_dist the
 * minimum 1-NN distance over all queries (INT32_MAX if none).  Synchronises. */
SLAM_API int slam_bf_match_filter(slam_ctx* ctx, const int32_t* d_idx, const int32_t* d_dist,
                         int64_t N, int mode, double param,
                         uint8_t* d_keep, int64_t* h_count, int32_t* h_min_dist);

/* crossCheck=True selection (cv2.BFMatcher(normType, crossCheck=True).match; OpenCV 4.x batch_distance.cpp
 * crosscheck branch): given the FORWARD search (every query row's nearest train row, [N,2] tables from
 * slam_bf_knn2_u256(query, train)) and the REVERSE search (every train row's nearest query row, [M,2] idx table
 * from slam_bf_knn2_u256(train as query, query as train)), query q keeps its nearest train row t iff q is also
 * t's nearest query row (mutual nearest neighbours, ties to the lowest index on both sides); all other queries
 * get (-1, INT32_MAX).  d_out_idx / d_out_dist are int32 [N].  Synchronises. */
SLAM_API int slam_bf_cross_check(slam_ctx* ctx, const int32_t* d_fwd_idx, const int32_t* d_fwd_dist, int64_t N,
                        const int32_t* d_rev_idx, int64_t M, int32_t* d_out_idx, int32_t* d_out_dist,
                        int64_t* h_count);

/* Multi-image train sets (cv2.BFMatcher.add([...]) + knnMatch, the loop-closure layout; the reference constructs and
 * queries the matcher at feature_matchers.py:34,39, the collection semantics are OpenCV's): the search runs over the
 * concatenated rows of all images - whose order is (imgIdx, trainIdx), so ties resolve as OpenCV's do - and this turns
 * every reported global train row back into the pair.  d_offsets int32 [images + 1]: first row of each image,
 * ascending, d_offsets[0] = 0, d_offsets[images] = total rows; images <= 8191 and fewer than 2^18 rows per image
 * (OpenCV's imgIdx << 18 encoding).  count entries of d_global_idx (e.g. 2 N for an [N,2] table) -> d_img_idx,
 * d_train_idx; -1 ("no neighbour") stays -1 in both.  Asynchronous on the ctx stream. */
SLAM_API int slam_bf_split_index(slam_ctx* ctx, const int32_t* d_global_idx, int64_t count, const int32_t* d_offsets,
                                 int64_t images, int32_t* d_img_idx, int32_t* d_train_idx);

/* ---- hot path 2: reprojection residual + Jacobians (f64) ---------------- */
/* Per observation o with pose k = d_obs_pose[o], point l = d_obs_point[o]:
 *   p_c = R_k p_l + t_k                      (T * pos3d,  frontend.py:275)
 *   e   = meas_o - (fx X/Z + cx, fy Y/Z + cy)            (frontend.py:275-277)
 *   Zinv = 1/(Z + 1e-18); J_pose = 2x6, rotation columns first
 *                                                        (frontend.py:284-291)
 *   J_point = -dproj/dp_c * R_k (2x3)  — extension, not in the reference
 * d_poses: [K,12] f64 = row-major R (9) then t (3), i.e. the top 3x4 of Tcw
 * flattened as [R|t] rows: (r00 r01 r02 tx r10 r11 r12 ty r20 r21 r22 tz).
 * d_points [L,3], d_meas [O,2], outputs d_e [O,2], d_Jpose [O,12] (row-major
 * 2x6), d_Jpoint [O,6] (row-major 2x3) or NULL to skip it. */
SLAM_API int slam_reproj_rj_f64(slam_ctx* ctx, const double* d_poses, int64_t K,
                       const double* d_points, int64_t L,
                       const int32_t* d_obs_pose, const int32_t* d_obs_point,
                       const double* d_meas, int64_t O,
                       double fx, double fy, double cx, double cy,
                       double* d_e, double* d_Jpose, double* d_Jpoint);

/* d_obs_pose / d_obs_point entries must lie in [0, K) / [0, L).  slam_reproj_rj_f64 and the per-observation stage
 * of slam_ba_reduce_f64 never dereference an index outside that range: the observation is computed from row 0
 * with a NaN measurement (its e / J come out NaN) and a per-context counter is bumped.  slam_index_errors returns
 * how many such observations were met since the last call and clears the counter (synchronises).  The derived
 * tables of slam_ba_reduce_f64 (pt_ptr / pt_obs / ps_ptr / ps_obs / lookup) are the caller's to build correctly. */
SLAM_API int slam_index_errors(slam_ctx* ctx, int64_t* count);

/* Pose-only normal equations for one pose (frontend.py:298-365 inner build):
 * H = sum w J^T J (6x6, row-major), b = sum w J^T e (6), chi2[o] = e.e,
 * w = Huber weight with delta (delta <= 0: w = 1); observations whose
 * d_active[o] == 0 are skipped (g2o "level 1" edges, frontend.py:372-377).
 * d_pose [12] as above; outputs d_H [36], d_b [6], d_chi2 [O] on device. */
SLAM_API int slam_pose_normal_eq_f64(slam_ctx* ctx, const double* d_pose,
                            const double* d_points, const double* d_meas,
                            const uint8_t* d_active, int64_t O,
                            double fx, double fy, double cx, double cy,
                            double huber_delta,
                            double* d_H, double* d_b, double* d_chi2);

/* The whole of Frontend._correct_current_pose (frontend.py:298-393) as one launch: `rounds` outer rounds
 * (reference: 4) of `iterations` LM iterations (reference: 10) on one pose against O fixed points, every
 * round restarting from d_pose_in, edges with chi2 > chi2_threshold (reference: 5.991**2) leaving the
 * optimisation after each round, the Huber kernel (delta; reference: 1.0) dropped after round index 2.
 * Outputs: d_pose_out [12], d_inlier uint8 [O] (1 = chi2 <= threshold at the end), d_chi2 [O],
 * d_stats int32 [2] = {inlier count, accepted LM steps}.  Asynchronous on the ctx stream. */
SLAM_API int slam_pose_optimize_f64(slam_ctx* ctx, const double* d_pose_in, const double* d_points,
                                    const double* d_meas, int64_t O, double fx, double fy, double cx, double cy,
                                    int rounds, int iterations, double chi2_threshold, double huber_delta,
                                    double* d_pose_out, uint8_t* d_inlier, double* d_chi2, int32_t* d_stats);

/* The same for B independent frames in ONE launch (one workgroup per frame): frame b owns the edges
 * [d_offsets[b], d_offsets[b+1]) of the concatenated d_points [O_total,3] / d_meas [O_total,2] (d_offsets int32 [B+1],
 * ascending, d_offsets[0] = 0, d_offsets[B] = O_total; a table that is not ascending or leaves [0, O_total] never
 * causes an access outside the arrays: the frame shrinks to the part inside and slam_index_errors counts it), d_pose_in / d_pose_out [B,12],
 * d_inlier / d_chi2 [O_total], d_stats int32 [B,2].  Use: the keyframes of a window against the fixed map, or several
 * relocalisation candidates; the reference refines one frame at a time (frontend.py:298-393). */
SLAM_API int slam_pose_optimize_batch_f64(slam_ctx* ctx, int64_t B, const double* d_pose_in, const double* d_points,
                                          const double* d_meas, const int32_t* d_offsets, int64_t O_total, double fx,
                                          double fy, double cx, double cy, int rounds, int iterations,
                                          double chi2_threshold, double huber_delta, double* d_pose_out,
                                          uint8_t* d_inlier, double* d_chi2, int32_t* d_stats);

/* ---- per-frame calls on caller-owned host buffers: one upload, one download, one wait (frame-sized: zero-copy, polled) ---- */
/* BruteForceFeatureMatcher.match (feature_matchers.py:36-44; cv2.BFMatcher.match + the min-distance filter) in
 * one call.  Query rows h_query [N,32]; train rows either h_train [M,32] (host) or d_train (device, e.g. the
 * previous frame kept by an earlier call) - exactly one of them when M > 0.  If d_query_keep is non-null the
 * query rows are uploaded there (32*N bytes, caller-allocated with slam_malloc) so the next frame can pass it
 * as d_train.  mode/param as slam_bf_match_filter, plus mode 3 = crossCheck (cv2.BFMatcher(normType,
 * crossCheck=True).match; the reverse search runs in the same call, param unused).  Outputs (caller-allocated, N entries each): the kept
 * matches in ascending query order as query index, train index and distance (float32, integer-valued like
 * cv2's); *h_count = how many.  N == 0 or M == 0: no matches, not an error.
 * Frame-sized calls (<= 4096 rows a side) return as soon as the call's last kernel has stored its completion word into
 * pinned host memory behind its results - the host polls that word instead of synchronising the stream (about 4 us less
 * per call); a call that is not complete after 2 ms of polling waits for the stream as before.  The outputs are complete
 * when the function returns either way. */
SLAM_API int slam_bf_match_host(slam_ctx* ctx, const uint8_t* h_query, int64_t N, const uint8_t* h_train,
                                const void* d_train, int64_t M, void* d_query_keep, int mode, double param,
                                int32_t* h_query_idx, int32_t* h_train_idx, float* h_distance, int64_t* h_count);
/* Bytes the host-buffer entry points (slam_bf_knn2_u256_host, slam_bf_match_host, slam_pose_optimize_host_f64)
 * have moved over PCIe on this context since it was created: inputs copied or read in place by the kernels
 * (h2d), results copied or written in place (d2h).  Rows passed as device pointers (d_train) count nothing. */
SLAM_API int slam_io_counters(slam_ctx* ctx, uint64_t* h2d_bytes, uint64_t* d2h_bytes);
/* slam_pose_optimize_f64 on host buffers (Frontend._correct_current_pose, frontend.py:298-393): h_pose_in [12],
 * h_points [O,3], h_meas [O,2] -> h_pose_out [12], h_inlier uint8 [O], h_chi2 [O], h_stats int32 [2].  Up to 512 edges the
 * kernel reads and writes the pinned staging block itself and the call waits for its completion word (see slam_bf_match_host). */
SLAM_API int slam_pose_optimize_host_f64(slam_ctx* ctx, const double* h_pose_in, const double* h_points,
                                         const double* h_meas, int64_t O, double fx, double fy, double cx,
                                         double cy, int rounds, int iterations, double chi2_threshold,
                                         double huber_delta, double* h_pose_out, uint8_t* h_inlier,
                                         double* h_chi2, int32_t* h_stats);

/* Reduced camera system of a keyframe-window bundle adjustment (extension: the reference's Backend is an
 * empty class, backend.py:101-103; residual/Jacobian arithmetic as frontend.py:272-291).  One call =
 * linearise all O observations, Schur-eliminate the L points with damping `lambda`, and leave on device:
 *   d_rec  [O,SLAM_BA_REC] per-observation blocks (Hpl, Y = Hpl E, ...), d_E [L,9] = (Hll+lambda I)^-1,
 *   d_bl [L,3], d_Hpp [K,21] (upper triangles), d_bp [K,6], d_ybl [K,6] = sum Y bl, d_cost [K] (robust cost
 *   per pose), d_W [K,K,36] with W[k1,k2] = sum_l Y_(k1,l) Hpl_(k2,l)^T for k1 <= k2 (other blocks untouched),
 *   d_hll_diag [L,3] = diagonal of the undamped Hll (optional, may be null; for the initial damping).
 * The caller assembles S = blockdiag(Hpp + lambda I) - W (symmetric), rhs = -bp + ybl, solves for dp [K,6] and
 * calls slam_ba_backsub_f64 for the point updates dl [L,3].
 * Index tables (int32, device): obs_pose/obs_point [O]; pt_ptr [L+1]/pt_obs [O] = observations grouped by
 * point; ps_ptr [K+1]/ps_obs [O] = grouped by pose; lookup [K,L] = observation index of (pose, point) or -1.
 * All reductions run in a fixed order: results are bit-identical from run to run. */
#define SLAM_BA_REC 73
SLAM_API int slam_ba_reduce_f64(slam_ctx* ctx, const double* d_poses, int64_t K, const double* d_points,
                                int64_t L, const int32_t* d_obs_pose, const int32_t* d_obs_point,
                                const double* d_meas, int64_t O, const int32_t* d_pt_ptr, const int32_t* d_pt_obs,
                                const int32_t* d_ps_ptr, const int32_t* d_ps_obs, const int32_t* d_lookup,
                                double fx, double fy, double cx, double cy, double huber_delta, double lambda,
                                double* d_rec, double* d_E, double* d_bl, double* d_Hpp, double* d_bp,
                                double* d_ybl, double* d_cost, double* d_W, double* d_hll_diag);
/* Robust cost of a candidate state only, per pose (d_cost [K]); same tables as slam_ba_reduce_f64. */
SLAM_API int slam_ba_cost_f64(slam_ctx* ctx, const double* d_poses, int64_t K, const double* d_points,
                              const int32_t* d_obs_point, const double* d_meas, const int32_t* d_ps_ptr,
                              const int32_t* d_ps_obs, double fx, double fy, double cx, double cy,
                              double huber_delta, double* d_cost);
SLAM_API int slam_ba_backsub_f64(slam_ctx* ctx, int64_t L, const int32_t* d_pt_ptr, const int32_t* d_pt_obs,
                                 const int32_t* d_obs_pose, const double* d_rec, const double* d_E,
                                 const double* d_bl, const double* d_dp, double* d_dl);

/* A whole window bundle adjustment in ONE launch: Levenberg-Marquardt with Schur complement as a persistent kernel of a
 * few dozen workgroups - the elimination of slam_ba_reduce_f64 as four phases separated by grid barriers (no per-observation
 * records: every phase linearises the observations it touches again), the dense L D L^T solve of the reduced camera system
 * in LDS, the exp() update, the candidate's cost and the accept / reject decisions all on the device; nothing crosses PCIe
 * between trials (extension: backend.py:101-103 is an empty class over a Map of
 * NUM_ACTIVE_KEYFRAMES = 7 keyframes, backend.py:11).  For windows of at most SLAM_BA_LM_MAX_FREE moving poses (a 96 x 96
 * system), 64 poses and SLAM_BA_LM_MAX_OBS observations; larger ones use slam_ba_reduce_f64 / slam_ba_backsub_f64 with
 * a host solve.
 *   d_poses2  [2][K,12]  in: the state in the first half;  d_points2 [2][L,3] likewise.  out: the optimised state is in
 *             half `d_stats[6]` of both (the halves swap roles on every accepted step).
 *   index tables as slam_ba_reduce_f64 (obs_pose / obs_point [O], pt_ptr [L+1] / pt_obs [O], ps_ptr [K+1] / ps_obs [O]);
 *             a (pose, point) pair may be observed once; the (pose, point) -> observation table is built on the device.  A
 *             pose or point index outside the window, a free list that is not ascending below K, or pose list heads that
 *             are not 0 = ps_ptr[0] <= ... <= ps_ptr[K] = O are counted (slam_index_errors) and end the launch (status 2);
 *             pt_ptr and the two observation lists are trusted (slam_ba_optimize_host_f64 builds all of them itself).
 *   d_free_poses int32 [n_free], ascending: the poses that move (the others hold the gauge).
 *   d_work    scratch of slam_ba_optimize_workspace(K, L, O) bytes, 16-byte aligned.
 *   d_stats   double [8]: initial cost, final cost, accepted steps, trials, final lambda, status (0 = ok; all NaN until the
 *             launch has completed; 1 = abandoned at a grid barrier: device busy; 2 = bad index / table), result half,
 *             workgroups used.
 * Schedule: lambda0 = 1e-5 max diag(H of the free poses and of the points); `iterations` iterations of up to 10 trials;
 * rho = (cost - cost_new) / (dx.(lambda dx - b) + 1e-3); accepted: lambda *= max(1/3, min(1 - (2 rho - 1)^3, 2/3));
 * rejected (or a factorisation that fails): lambda *= ni, ni *= 2.  Every sum is formed in a fixed order: two runs
 * give identical bits.  Asynchronous on the ctx stream.  The launch holds one compute unit per workgroup (at most 128, see
 * d_stats[7]) from start to end and needs all of them resident at once: launches of other contexts run beside it while
 * their workgroups fit as well (two of the largest do).  A launch larger than the device could ever hold is refused
 * (SLAM_ERR_BUSY); one that cannot get its workgroups resident NOW - other work holds compute units - gives up at a barrier
 * after 50 ms of wall clock and reports status 1, which slam_ba_optimize_host_f64 returns as SLAM_ERR_BUSY: the window is
 * unchanged and slam_ba_reduce_f64 / slam_ba_backsub_f64 (no residency requirement) do the same job. */
#define SLAM_BA_LM_MAX_FREE 16
#define SLAM_BA_LM_MAX_OBS (1 << 17)
SLAM_API int slam_ba_optimize_workspace(int64_t K, int64_t L, int64_t O, uint64_t* bytes);
SLAM_API int slam_ba_optimize_f64(slam_ctx* ctx, int64_t K, int64_t L, int64_t O, const int32_t* d_obs_pose,
                                  const int32_t* d_obs_point, const double* d_meas, const int32_t* d_pt_ptr,
                                  const int32_t* d_pt_obs, const int32_t* d_ps_ptr, const int32_t* d_ps_obs,
                                  const int32_t* d_free_poses, int64_t n_free, double fx, double fy, double cx, double cy,
                                  double huber_delta, int iterations, double* d_poses2, double* d_points2, void* d_work,
                                  uint64_t work_bytes, double* d_stats);
/* slam_ba_optimize_f64 on HOST buffers: poses [K,12] and points [L,3] in, the optimised ones out; the index tables the kernel
 * wants are built inside (two stable counting sorts), a (pose, point) pair observed twice or an index out of range is
 * refused (SLAM_ERR_INVALID) before anything is launched; SLAM_ERR_BUSY = the launch gave up at a grid barrier (see above),
 * the outputs are not written.  h_pose_fixed [K]: non-zero = the pose holds the gauge.  One
 * upload, one launch, one download; staging, device arena and workspace belong to the context.  h_stats as d_stats above.
 * Serialises with the other host-buffer calls of the context. */
SLAM_API int slam_ba_optimize_host_f64(slam_ctx* ctx, int64_t K, int64_t L, int64_t O, const double* h_poses,
                                       const double* h_points, const int32_t* h_obs_pose, const int32_t* h_obs_point,
                                       const double* h_meas, const uint8_t* h_pose_fixed, double fx, double fy, double cx,
                                       double cy, double huber_delta, int iterations, double* h_poses_out,
                                       double* h_points_out, double* h_stats);

/* ---- multi-GPU: RCCL all-gather of per-shard result rows ---------------- */
#define SLAM_COMM_ID_BYTES 128
SLAM_API int slam_comm_version(int* version); /* ncclGetVersion of the librccl that was loaded (e.g. 22703); needs no GPU */
SLAM_API int slam_comm_unique_id(void* h_id /*[128]*/);
SLAM_API int slam_comm_init(slam_ctx* ctx, int nranks, int rank, const void* h_id);
SLAM_API int slam_comm_destroy(slam_ctx* ctx);
/* every rank contributes bytes_per_rank bytes; d_recv holds nranks*bytes_per_rank.
 * in-place allowed when d_send == d_recv + rank*bytes_per_rank. */
SLAM_API int slam_comm_allgather(slam_ctx* ctx, const void* d_send, void* d_recv, uint64_t bytes_per_rank);
SLAM_API int slam_comm_broadcast(slam_ctx* ctx, void* d_buf, uint64_t bytes, int root);
/* Double-buffered form for back-to-back passes: the all-gather of result buffer `buffer_id` (0 or 1) runs on a
 * second stream of the ctx, ordered after everything issued so far on the main stream, so the next pass (which
 * writes the OTHER buffer) overlaps it.  slam_comm_wait_buffer makes the main stream wait until the last
 * gather of that buffer has finished (call it before overwriting the buffer); slam_sync waits for both streams. */
SLAM_API int slam_comm_allgather_overlapped(slam_ctx* ctx, const void* d_send, void* d_recv, uint64_t bytes_per_rank,
                                            int buffer_id);
SLAM_API int slam_comm_wait_buffer(slam_ctx* ctx, int buffer_id);

/* ---- multi-GPU without a communicator library: direct all-gather over xGMI peer mappings (HIP IPC) ----
 * Each rank exports its gathered buffer(s) (slam_p2p_export on the base pointer of a slam_malloc allocation),
 * the launcher ships the 64-byte handles around, every rank maps the peers' buffers (slam_p2p_open) and
 * slam_p2p_allgather_overlapped then copies this rank's slot (bytes_per_rank at offset rank*bytes_per_rank) into
 * every peer's buffer on the context's second stream, behind the work queued so far on the main stream.
 * h_peer_bufs is a HOST array of nranks device pointers (entry [rank] unused).  slam_comm_wait_buffer(buffer_id)
 * orders later main-stream work after these copies; arrival on the peers is the launcher's barrier to guarantee.
 * No reference counterpart (the reference is single-process). */
#define SLAM_P2P_HANDLE_BYTES 64
SLAM_API int slam_p2p_export(slam_ctx* ctx, void* d_ptr, void* h_handle /*[64]*/);
SLAM_API int slam_p2p_open(slam_ctx* ctx, const void* h_handle, void** d_peer_ptr);
SLAM_API int slam_p2p_close(slam_ctx* ctx, void* d_peer_ptr);
SLAM_API int slam_p2p_allgather_overlapped(slam_ctx* ctx, const void* d_send, uint64_t bytes_per_rank, int rank,
                                           void* const* h_peer_bufs, int nranks, int buffer_id);

#ifdef __cplusplus
}
#endif
#endif /* SLAMHIP_H */
