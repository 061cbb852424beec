// bf_filter.hip — post-match selection on the device: the reference's
// "distance < max(2*min_dist, dist_threshold)" filter
// (feature_matchers.py:41-43), plus Lowe ratio and OpenCV crossCheck, which
// BASELINE.json names as extensions (the reference itself never uses them).
#include "internal.h"

struct filter_scratch {        // lives in the 4 KiB per-context device scratch (ctx->scratch)
    int min_dist;              // min 1-NN distance over all queries
    int pad;
    unsigned long long count;  // rows kept
};

__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
    return v;
}

__global__ __launch_bounds__(256) void filter_init_kernel(filter_scratch* s) {
    s->min_dist = SLAM_NO_MATCH_DIST;
    s->pad = 0;
    s->count = 0ull;
}

// min over queries of the 1-NN distance: wavefront min reduction, one atomic per wave
__global__ __launch_bounds__(256) void filter_min_kernel(const int2* __restrict__ idx, const int2* __restrict__ dist,
                                                         int N, filter_scratch* s) {
    int m = SLAM_NO_MATCH_DIST;
    for (int n = blockIdx.x * 256 + threadIdx.x; n < N; n += gridDim.x * 256)
        if (idx[n].x >= 0) m = min(m, dist[n].x);
    m = wave_min_i32(m);
    if ((threadIdx.x & 63) == 0 && m != SLAM_NO_MATCH_DIST) atomicMin(&s->min_dist, m);
}

__global__ __launch_bounds__(256) void filter_keep_kernel(const int2* __restrict__ idx, const int2* __restrict__ dist,
                                                          int N, int mode, double param, filter_scratch* s,
                                                          uint8_t* __restrict__ keep) {
    const int min_dist = s->min_dist;
    unsigned int kept = 0;
    for (int n = blockIdx.x * 256 + threadIdx.x; n < N; n += gridDim.x * 256) {
        const int2 i = idx[n], d = dist[n];
        bool k = i.x >= 0;
        if (mode == 1) {
            // feature_matchers.py:43: m.distance < max(2 * min_dist, dist_threshold); distance is the
            // float32 image of an integer <= 256, so the comparison is exact in f64
            const double lim = fmax(2.0 * (double)min_dist, param);
            k = k && (double)d.x < lim;
        } else if (mode == 2) {
            // Lowe: m.distance < ratio * n.distance in float32-valued doubles (what the Python idiom computes)
            k = k && i.y >= 0 && (double)d.x < param * (double)d.y;
        }
        keep[n] = k ? 1 : 0;
        kept += k ? 1u : 0u;
    }
    // ballot-free wave sum, then one atomic per wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) kept += __shfl_xor(kept, off, 64);
    if ((threadIdx.x & 63) == 0 && kept) atomicAdd(&s->count, (unsigned long long)kept);
}

// Both steps in one workgroup for frame-sized inputs: saves two launches on the path the reference runs per frame.
#define FILTER_SMALL_MAX 16384
__global__ __launch_bounds__(1024) void filter_small_kernel(const int2* __restrict__ idx, const int2* __restrict__ dist,
                                                            int N, int mode, double param, filter_scratch* s,
                                                            uint8_t* __restrict__ keep, unsigned* done, unsigned epoch) {
    __shared__ int s_min[16];
    __shared__ unsigned int s_cnt[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int m = SLAM_NO_MATCH_DIST;
    for (int n = threadIdx.x; n < N; n += 1024)
        if (idx[n].x >= 0) m = min(m, dist[n].x);
    m = wave_min_i32(m);
    if (lane == 0) s_min[wave] = m;
    __syncthreads();
    int min_dist = SLAM_NO_MATCH_DIST;
#pragma unroll
    for (int w = 0; w < 16; w++) min_dist = min(min_dist, s_min[w]);
    unsigned int kept = 0;
    for (int n = threadIdx.x; n < N; n += 1024) {
        const int2 i = idx[n], d = dist[n];
        bool k = i.x >= 0;
        if (mode == 1) k = k && (double)d.x < fmax(2.0 * (double)min_dist, param);       // feature_matchers.py:43
        else if (mode == 2) k = k && i.y >= 0 && (double)d.x < param * (double)d.y;       // Lowe ratio
        keep[n] = k ? 1 : 0;
        kept += k ? 1u : 0u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) kept += __shfl_xor(kept, off, 64);
    if (lane == 0) s_cnt[wave] = kept;
    if (done) __threadfence_system();      // the caller polls done[0] (slam_wait_done): the flags above are visible to the host first
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long c = 0;
        for (int w = 0; w < 16; w++) c += s_cnt[w];
        s->min_dist = min_dist;
        s->pad = 0;
        s->count = c;
        if (done) __hip_atomic_store(done, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static int filter_scratch_ptr(slam_ctx* ctx, filter_scratch** out) {
    *out = (filter_scratch*)ctx->scratch;  // 4 KiB per-context device scratch, allocated at ctx creation
    return SLAM_OK;
}

// done / epoch: the one-workgroup form stores `epoch` into done[0] (pinned host memory) behind its flags, for a caller that
// polls (slam_wait_done); *polled says whether it did (inputs beyond FILTER_SMALL_MAX take three kernels and do not).
int slam_filter_launch(slam_ctx* ctx, const int32_t* d_idx, const int32_t* d_dist, int64_t N, int mode, double param,
                       uint8_t* d_keep, unsigned* done, unsigned epoch, bool* polled) {
    filter_scratch* s = nullptr;
    if (int rc = filter_scratch_ptr(ctx, &s)) return rc;
    if (polled) *polled = false;
    if (N <= FILTER_SMALL_MAX) {
        filter_small_kernel<<<1, 1024, 0, ctx->stream>>>((const int2*)d_idx, (const int2*)d_dist, (int)N, mode, param,
                                                         s, d_keep, done, epoch);
        if (polled) *polled = done != nullptr;
    } else {
        const int blocks = (int)((N + 255) / 256 < 2048 ? (N + 255) / 256 : 2048);
        filter_init_kernel<<<1, 1, 0, ctx->stream>>>(s);
        filter_min_kernel<<<blocks, 256, 0, ctx->stream>>>((const int2*)d_idx, (const int2*)d_dist, (int)N, s);
        filter_keep_kernel<<<blocks, 256, 0, ctx->stream>>>((const int2*)d_idx, (const int2*)d_dist, (int)N, mode,
                                                            param, s, d_keep);
    }
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}

extern "C" int slam_bf_match_filter(slam_ctx* ctx, const int32_t* d_idx, const int32_t* d_dist, int64_t N,
                                    int mode, double param, uint8_t* d_keep, int64_t* h_count,
                                    int32_t* h_min_dist) {
    SLAM_REQUIRE(ctx, "slam_bf_match_filter: null ctx");
    SLAM_REQUIRE(mode >= 0 && mode <= 2, "mode %d not in {0,1,2}", mode);
    SLAM_REQUIRE(N >= 0 && N <= (1ll << 30), "bad N=%lld", (long long)N);
    if (h_count) *h_count = 0;
    if (h_min_dist) *h_min_dist = SLAM_NO_MATCH_DIST;
    if (N == 0) return SLAM_OK;
    SLAM_REQUIRE(d_idx && d_dist && d_keep, "slam_bf_match_filter: null device pointer");
    SLAM_HIP(hipSetDevice(ctx->device));
    if (int rc = slam_filter_launch(ctx, d_idx, d_dist, N, mode, param, d_keep)) return rc;
    filter_scratch h;
    SLAM_HIP(hipMemcpyAsync(&h, ctx->scratch, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP(hipStreamSynchronize(ctx->stream));
    if (h_count) *h_count = (int64_t)h.count;
    if (h_min_dist) *h_min_dist = h.min_dist;
    return SLAM_OK;
}

// ---- crossCheck ------------------------------------------------------------
// cv2.BFMatcher(normType, crossCheck=True).match: pair (q, t) is returned iff t is q's nearest train row AND q is
// t's nearest query row (ties to the lowest index on both sides).  OpenCV 4.x batch_distance.cpp gets there by a
// scatter of the reverse table followed by "if (tidx[sidx[i]] != i) nidx[i] = -1"; whenever that test passes, the
// scattered row is sidx[i] itself (it has the smallest distance of all train rows and the lowest index among equals),
// so the scatter is not needed: one gather of the reverse table through the forward one decides each query.
__global__ __launch_bounds__(256) void cross_emit_kernel(const int2* __restrict__ fwd_idx,
                                                         const int2* __restrict__ fwd_dist, int N,
                                                         const int2* __restrict__ rev_idx, int M,
                                                         int* __restrict__ out_idx, int* __restrict__ out_dist,
                                                         filter_scratch* s) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    unsigned int kept = 0;
    if (n < N) {
        const int t = fwd_idx[n].x;
        const bool has = t >= 0 && t < M && rev_idx[t].x == n;
        out_idx[n] = has ? t : SLAM_NO_MATCH_IDX;
        out_dist[n] = has ? fwd_dist[n].x : SLAM_NO_MATCH_DIST;
        kept = has ? 1u : 0u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) kept += __shfl_xor(kept, off, 64);
    if ((threadIdx.x & 63) == 0 && kept) atomicAdd(&s->count, (unsigned long long)kept);
}

int slam_cross_launch(slam_ctx* ctx, const int32_t* d_fwd_idx, const int32_t* d_fwd_dist, int64_t N,
                      const int32_t* d_rev_idx, int64_t M, int32_t* d_out_idx, int32_t* d_out_dist) {
    filter_scratch* s = nullptr;
    if (int rc = filter_scratch_ptr(ctx, &s)) return rc;
    filter_init_kernel<<<1, 1, 0, ctx->stream>>>(s);
    cross_emit_kernel<<<(unsigned)((N + 255) / 256), 256, 0, ctx->stream>>>(
        (const int2*)d_fwd_idx, (const int2*)d_fwd_dist, (int)N, (const int2*)d_rev_idx, (int)M, d_out_idx, d_out_dist, s);
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}

extern "C" int slam_bf_cross_check(slam_ctx* ctx, const int32_t* d_fwd_idx, const int32_t* d_fwd_dist, int64_t N,
                                   const int32_t* d_rev_idx, int64_t M, int32_t* d_out_idx, int32_t* d_out_dist,
                                   int64_t* h_count) {
    SLAM_REQUIRE(ctx, "slam_bf_cross_check: null ctx");
    SLAM_REQUIRE(N >= 0 && M >= 0 && N <= (1ll << 30) && M <= (1ll << 30), "bad sizes");
    if (h_count) *h_count = 0;
    if (N == 0) return SLAM_OK;
    SLAM_REQUIRE(d_out_idx && d_out_dist && d_fwd_idx && d_fwd_dist && (M == 0 || d_rev_idx),
                 "slam_bf_cross_check: null device pointer");
    SLAM_HIP(hipSetDevice(ctx->device));
    if (int rc = slam_cross_launch(ctx, d_fwd_idx, d_fwd_dist, N, d_rev_idx, M, d_out_idx, d_out_dist)) return rc;
    filter_scratch h;
    SLAM_HIP(hipMemcpyAsync(&h, ctx->scratch, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP(hipStreamSynchronize(ctx->stream));
    if (h_count) *h_count = (int64_t)h.count;
    return SLAM_OK;
}

// ---- multi-image train sets: global train row -> (imgIdx, trainIdx) ----------------------------------------------
// cv2.BFMatcher.add([...]) + knnMatch reports, per neighbour, the image of the collection and the row inside it
// (feature_matchers.py:34,39 construct and query the matcher; the collection semantics are OpenCV's, matchers.cpp).
// The search runs over the concatenated rows, whose order IS (imgIdx, trainIdx), so ties resolve as OpenCV's do; this
// turns the global row back into the pair.  offsets: int32 [images + 1], ascending, offsets[0] = 0.
__global__ __launch_bounds__(256) void split_index_kernel(const int* __restrict__ gidx, long long n, const int* __restrict__ offsets,
                                                          int images, int* __restrict__ img, int* __restrict__ local) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) {
        const int g = gidx[i];
        int im = SLAM_NO_MATCH_IDX, lo = SLAM_NO_MATCH_IDX;
        if (g >= 0 && g < offsets[images]) {
            int a = 0, b = images;                      // last image whose first row is <= g (empty images are skipped over)
            while (b - a > 1) {
                const int mid = (a + b) >> 1;
                if (offsets[mid] <= g) a = mid; else b = mid;
            }
            im = a;
            lo = g - offsets[im];
        }
        img[i] = im;
        local[i] = lo;
    }
}

extern "C" int slam_bf_split_index(slam_ctx* ctx, const int32_t* d_global_idx, int64_t count, const int32_t* d_offsets,
                                   int64_t images, int32_t* d_img_idx, int32_t* d_train_idx) {
    SLAM_REQUIRE(ctx, "slam_bf_split_index: null ctx");
    SLAM_REQUIRE(count >= 0 && count <= (1ll << 32) && images >= 1 && images <= (1 << 13),
                 "bad sizes (count=%lld, images=%lld; at most 8191 images, OpenCV's imgIdx << 18 encoding)", (long long)count, (long long)images);
    if (count == 0) return SLAM_OK;
    SLAM_REQUIRE(d_global_idx && d_offsets && d_img_idx && d_train_idx, "slam_bf_split_index: null device pointer");
    SLAM_HIP(hipSetDevice(ctx->device));
    unsigned blocks = (unsigned)((count + 255) / 256);
    if (blocks > (unsigned)ctx->num_cu * 16) blocks = (unsigned)ctx->num_cu * 16;
    split_index_kernel<<<dim3(blocks), dim3(256), 0, ctx->stream>>>(d_global_idx, (long long)count, d_offsets, (int)images,
                                                                  d_img_idx, d_train_idx);
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}
