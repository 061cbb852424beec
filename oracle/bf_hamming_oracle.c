/*
 * bf_hamming_oracle.c — CPU restatement of the reference's descriptor-matching
 * path.  TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg; never by the product path.
 *
 * PARITY UNPINNED: the reference (feature_matchers.py:33-44) delegates to
 * cv2.BFMatcher, i.e. opencv-python 4.9.0.80 (poetry.lock:1798-1799), which is
 * not vendored under /root/reference and not installable here; the reference
 * has no tests or golden vectors.  This file restates OpenCV's published
 * algorithm and is pinned only by the hand-derived known-answer vectors in
 * tests/golden/ (SURVEY.md §8c).
 *
 * What is restated, and from where:
 *   - BruteForceFeatureMatcher.match(source, query, dist_threshold)
 *       feature_matchers.py:36-44: bf.match(query, source), then the optional
 *       "distance < max(2*min_dist, dist_threshold)" filter (strict <).
 *   - cv2.BFMatcher.match/knnMatch(normType=NORM_HAMMING, crossCheck=False)
 *       OpenCV 4.9.0 modules/features2d/src/matchers.cpp (BFMatcher::knnMatchImpl)
 *       and modules/core/src/batch_distance.cpp (BatchDistInvoker): distances as
 *       int32, K best kept per query by insertion while scanning train rows in
 *       ascending order: enter iff d < dist[K-1] (strict), shift while
 *       dist[k] > d (strict).  Hence order (distance asc, train index asc), ties
 *       to the lowest index; dist initialised INT_MAX, idx -1.
 *   - multi-image train sets: images scanned in order, index encoded
 *       imgIdx << 18 | trainIdx (IMGIDX_SHIFT = 18 in matchers.cpp).
 *   - crossCheck=True: batch_distance.cpp crosscheck branch (K == 1), OpenCV
 *       4.x: the reverse 1-NN per train row (tdist/tidx) AND the forward 1-NN
 *       per query row (sdist/sidx) are computed; train rows are scanned in
 *       ascending order with "if d < dist[q]: dist[q] = d, idx[q] = t"; then
 *       every query i with tidx[sidx[i]] != i has its idx cleared to -1.  The
 *       result is the documented contract of BFMatcher(crossCheck=True): pair
 *       (i, j) is returned iff j is the nearest train row of i AND i is the
 *       nearest query row of j (ties -> lowest index on both sides).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

#define DESC_BYTES 32
#define IMGIDX_SHIFT 18
#define IMGIDX_ONE (1 << IMGIDX_SHIFT)

/* normHamming over one 256-bit row (OpenCV hal::normHamming) */
static inline int hamming256(const uint8_t* a, const uint8_t* b) {
    uint64_t x[4], y[4];
    memcpy(x, a, 32);
    memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) +
           __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
}

/* batchDistHamming: distances of one query row to rows [0, n) of a train block, written to buf
 * (OpenCV fills a per-row distance buffer the same way before the K-best pass).  Two builds of the
 * same loop: portable, and AVX-512 VPOPCNTDQ where the host has it (gcc vectorises the popcounts);
 * picked once at run time, so the .so built in the build container stays runnable on any x86-64 host. */
static void dist_block_generic(const uint8_t* q, const uint8_t* t, int64_t n, int* buf) {
    for (int64_t j = 0; j < n; j++) buf[j] = hamming256(q, t + j * DESC_BYTES);
}

__attribute__((target("avx512f,avx512vl,avx512vpopcntdq")))
static void dist_block_avx512(const uint8_t* q, const uint8_t* t, int64_t n, int* buf) {
    uint64_t x[4];
    memcpy(x, q, 32);
    for (int64_t j = 0; j < n; j++) {
        uint64_t y[4];
        memcpy(y, t + j * DESC_BYTES, 32);
        buf[j] = __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) +
                 __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
    }
}

static int have_avx512_popcnt(void) {
    static int cached = -1;
    if (cached < 0)
        cached = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl") &&
                 __builtin_cpu_supports("avx512vpopcntdq");
    return cached;
}

/* which distance loop this host runs (reported by bench.py next to the CPU baseline) */
const char* oracle_bf_simd(void) { return have_avx512_popcnt() ? "avx512-vpopcntdq" : "scalar-popcnt"; }

#define DIST_BLOCK 2048

/* BatchDistInvoker K-best insertion for one query row against rows
 * [0, M) of one train image; `update` is added to the stored index. */
static void knn_insert_row(const uint8_t* q, const uint8_t* t, int64_t M, int K, int update, int* idx, int* dist) {
    int buf[DIST_BLOCK];
    const int wide = have_avx512_popcnt();
    for (int64_t j0 = 0; j0 < M; j0 += DIST_BLOCK) {
        const int64_t n = M - j0 < DIST_BLOCK ? M - j0 : DIST_BLOCK;
        if (wide) dist_block_avx512(q, t + j0 * DESC_BYTES, n, buf);
        else dist_block_generic(q, t + j0 * DESC_BYTES, n, buf);
        for (int64_t j = 0; j < n; j++) {
            const int d = buf[j];
            if (d < dist[K - 1]) {
                int k;
                for (k = K - 2; k >= 0 && dist[k] > d; k--) {
                    idx[k + 1] = idx[k];
                    dist[k + 1] = dist[k];
                }
                idx[k + 1] = (int)(j0 + j) + update;
                dist[k + 1] = d;
            }
        }
    }
}

/* knnMatch(query, train, k=K) as raw tables: idx/dist are int32 [N,K], missing
 * neighbours are (-1, INT_MAX).  Threads split the query rows (OpenCV uses
 * parallel_for_ over query rows the same way). */
int oracle_bf_knn_u256(const uint8_t* q, int64_t N, const uint8_t* t, int64_t M, int K, int32_t* idx,
                       int32_t* dist, int threads) {
    if (K < 1 || N < 0 || M < 0) return -1;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int64_t i = 0; i < N; i++) {
        int* ip = idx + i * K;
        int* dp = dist + i * K;
        for (int k = 0; k < K; k++) { ip[k] = -1; dp[k] = INT_MAX; }
        knn_insert_row(q + i * DESC_BYTES, t, M, K, 0, ip, dp);
    }
    return 0;
}

/* knnMatch against a collection of train images (BFMatcher.add([...])): rows
 * of all images are concatenated in `t`; img_rows[c] gives each image's row
 * count.  Stored index = imgIdx << 18 | trainIdx, as OpenCV encodes it. */
int oracle_bf_knn_multi_u256(const uint8_t* q, int64_t N, const uint8_t* t, const int64_t* img_rows, int n_img,
                             int K, int32_t* idx, int32_t* dist, int threads) {
    if (K < 1 || N < 0 || n_img < 0) return -1;
    for (int c = 0; c < n_img; c++)
        if (img_rows[c] >= IMGIDX_ONE) return -2; /* CV_Assert(trainDescCollection[iIdx].rows < IMGIDX_ONE) */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int64_t i = 0; i < N; i++) {
        int* ip = idx + i * K;
        int* dp = dist + i * K;
        for (int k = 0; k < K; k++) { ip[k] = -1; dp[k] = INT_MAX; }
        const uint8_t* tp = t;
        int update = 0;
        for (int c = 0; c < n_img; c++) {
            knn_insert_row(q + i * DESC_BYTES, tp, img_rows[c], K, update, ip, dp);
            tp += img_rows[c] * DESC_BYTES;
            update += IMGIDX_ONE;
        }
    }
    return 0;
}

/* feature_matchers.py:39-44 on top of a [N,K] table (K >= 1): out arrays get one
 * entry per returned DMatch (queryIdx, trainIdx, distance as float), in
 * ascending queryIdx.  dist_threshold <= 0 or NaN-free "falsy" (0.0) means
 * "no filter", as `if dist_threshold and ...` does.  Returns the match count. */
int64_t oracle_bf_match(const int32_t* idx, const int32_t* dist, int64_t N, int K, double dist_threshold,
                        int has_threshold, int32_t* out_query, int32_t* out_train, float* out_dist) {
    int64_t n = 0;
    /* bf.match(): one DMatch per query that has a neighbour */
    for (int64_t i = 0; i < N; i++) {
        if (idx[i * K] < 0) continue;
        out_query[n] = (int32_t)i;
        out_train[n] = idx[i * K];
        out_dist[n] = (float)dist[i * K];
        n++;
    }
    if (has_threshold && dist_threshold != 0.0 && n != 0) {
        float min_dist = out_dist[0];
        for (int64_t i = 1; i < n; i++)
            if (out_dist[i] < min_dist) min_dist = out_dist[i];
        const double lim = 2.0 * (double)min_dist > dist_threshold ? 2.0 * (double)min_dist : dist_threshold;
        int64_t m = 0;
        for (int64_t i = 0; i < n; i++) {
            if ((double)out_dist[i] < lim) {
                out_query[m] = out_query[i];
                out_train[m] = out_train[i];
                out_dist[m] = out_dist[i];
                m++;
            }
        }
        n = m;
    }
    return n;
}

/* Lowe ratio on a [N,2] table: keep[i] = has two neighbours && d0 < ratio * d1
 * (the usual `m.distance < ratio * n.distance` idiom on float32 distances). */
int64_t oracle_bf_ratio_test(const int32_t* idx, const int32_t* dist, int64_t N, double ratio, uint8_t* keep) {
    int64_t n = 0;
    for (int64_t i = 0; i < N; i++) {
        const int k = idx[2 * i] >= 0 && idx[2 * i + 1] >= 0 &&
                      (double)(float)dist[2 * i] < ratio * (double)(float)dist[2 * i + 1];
        keep[i] = (uint8_t)k;
        n += k;
    }
    return n;
}

/* crossCheck=True match(query, train): out_idx/out_dist int32 [N], (-1, INT_MAX)
 * where the query has no mutual nearest neighbour.  Follows batchDistance's
 * crosscheck branch step by step: reverse table, forward table, one-pass
 * scatter over ascending train rows (strict <), then the forward-consistency
 * pass "if (tidx[sidx[i]] != i) nidx[i] = -1".  (OpenCV leaves the scattered
 * distance behind for cleared rows; it is never reported because knnMatchImpl
 * stops at nidx < 0, so INT_MAX is stored here instead.) */
int oracle_bf_cross_check_u256(const uint8_t* q, int64_t N, const uint8_t* t, int64_t M, int32_t* out_idx,
                               int32_t* out_dist, int threads) {
    const size_t mm = (size_t)(M > 0 ? M : 1), nn = (size_t)(N > 0 ? N : 1);
    int32_t* tidx = (int32_t*)malloc(mm * sizeof(int32_t));
    int32_t* tdist = (int32_t*)malloc(mm * sizeof(int32_t));
    int32_t* sidx = (int32_t*)malloc(nn * sizeof(int32_t));
    int32_t* sdist = (int32_t*)malloc(nn * sizeof(int32_t));
    if (!tidx || !tdist || !sidx || !sdist) { free(tidx); free(tdist); free(sidx); free(sdist); return -1; }
    /* batchDistance(src2, src1, tdist, tidx, K=1): every train row's nearest query */
    oracle_bf_knn_u256(t, M, q, N, 1, tidx, tdist, threads);
    /* batchDistance(src1, src2, sdist, sidx, K=1): every query row's nearest train row */
    oracle_bf_knn_u256(q, N, t, M, 1, sidx, sdist, threads);
    for (int64_t i = 0; i < N; i++) { out_idx[i] = -1; out_dist[i] = INT_MAX; }
    for (int64_t i = 0; i < M; i++) {
        const int qi = tidx[i];
        if (qi < 0) continue;
        const int d = tdist[i], d0 = out_dist[qi];
        if (d < d0) {
            out_dist[qi] = d;
            out_idx[qi] = (int32_t)i;
        }
    }
    for (int64_t i = 0; i < N; i++) {
        const int si = sidx[i];
        if (si < 0 || tidx[si] != (int32_t)i) {
            out_idx[i] = -1;
            out_dist[i] = INT_MAX;
        }
    }
    free(tidx);
    free(tdist);
    free(sidx);
    free(sdist);
    return 0;
}
