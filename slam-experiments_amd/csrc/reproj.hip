// reproj.hip — per-observation reprojection residual + Jacobians in f64 on
// gfx950.  Restates Frontend.EdgeProjectionPoseOnly.compute_error /
// linearize_oplus (reference frontend.py:272-291): e = meas - proj(T p),
// 2x6 pose Jacobian with rotation columns first and Zinv = 1/(Z + 1e-18).
// The 2x3 point Jacobian is an extension (SURVEY.md §8a last note) for
// Backend.optimize(); the reference has no counterpart.
//
// HBM-write bound: 24 B read + 112..160 B written per observation.  One
// thread per observation computes in registers; the block's outputs are
// transposed through LDS so every global store is a full 16-B-per-lane
// coalesced write of a contiguous region (a lane's 96-B Jacobian row would
// otherwise be written as 6 strided partial lines).
#include "internal.h"

#define RJ_BLOCK 256

struct cam4 { double fx, fy, cx, cy; };
typedef double f64x2 __attribute__((ext_vector_type(2)));

// projection + Jacobian core shared by both kernels
struct proj_out {
    double e0, e1;
    double jp[12];  // row-major 2x6
    double A[6];    // dproj/dp_c rows: (fx*Zinv, 0, -fx*X*Zinv2), (0, fy*Zinv, -fy*Y*Zinv2)
};

__device__ __forceinline__ void project(const double* __restrict__ P /*12: [R|t] rows*/, double px, double py,
                                        double pz, double mu, double mv, const cam4 c, proj_out& o) {
    const double X = P[0] * px + P[1] * py + P[2] * pz + P[3];
    const double Y = P[4] * px + P[5] * py + P[6] * pz + P[7];
    const double Z = P[8] * px + P[9] * py + P[10] * pz + P[11];
    // frontend.py:275-277: pos_pixel = K @ p_c; pos_pixel /= pos_pixel[2]; e = meas - pos_pixel[:2]
    o.e0 = mu - (c.fx * X + c.cx * Z) / Z;
    o.e1 = mv - (c.fy * Y + c.cy * Z) / Z;
    // frontend.py:284-291
    const double Zinv = 1.0 / (Z + 1e-18);
    const double Zinv2 = Zinv * Zinv;
    o.jp[0] = c.fx * X * Y * Zinv2;
    o.jp[1] = -c.fx - c.fx * X * X * Zinv2;
    o.jp[2] = c.fx * Y * Zinv;
    o.jp[3] = -c.fx * Zinv;
    o.jp[4] = 0.0;
    o.jp[5] = c.fx * X * Zinv2;
    o.jp[6] = c.fy + c.fy * Y * Y * Zinv2;
    o.jp[7] = -c.fy * X * Y * Zinv2;
    o.jp[8] = -c.fy * X * Zinv;
    o.jp[9] = 0.0;
    o.jp[10] = -c.fy * Zinv;
    o.jp[11] = c.fy * Y * Zinv2;
    o.A[0] = c.fx * Zinv; o.A[1] = 0.0; o.A[2] = -c.fx * X * Zinv2;
    o.A[3] = 0.0; o.A[4] = c.fy * Zinv; o.A[5] = -c.fy * Y * Zinv2;
}

// 16-byte streaming store, non-temporal and written through (see store_rows)
__device__ __forceinline__ void rj_store16(f64x2* p, f64x2 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
}

// cooperative coalesced store of a block's [cnt][W] f64 rows staged in LDS
template <int W>
__device__ __forceinline__ void store_rows(const double* __restrict__ s, double* __restrict__ g, int cnt) {
    // cnt*W doubles contiguous; g is 16-B aligned because the block base is a multiple of 256 rows
    // Non-temporal: the outputs are written once and read by a later kernel at the earliest, so
    // they should not displace the pose/point tables from L2 (measured 0.341 -> 0.314 ms at 1e7 observations).
    // "sc1 nt" = non-temporal AND written through: nothing of the 1.6 GB stays dirty in the XCD L2s (same box, alternating
    // builds, 1e7 observations: plain 307 us, sc1 307, nt 257.8, sc1 nt 254.7 us).
    const int n2 = cnt * W / 2;  // W even
    const f64x2* s2 = (const f64x2*)s;
    f64x2* g2 = (f64x2*)g;
    for (int i = threadIdx.x; i < n2; i += RJ_BLOCK) rj_store16(&g2[i], s2[i]);
}

template <bool WITH_POINT>
__global__ __launch_bounds__(RJ_BLOCK) void reproj_rj_kernel(const double* __restrict__ poses,
                                                             const double* __restrict__ points,
                                                             const int* __restrict__ obs_pose,
                                                             const int* __restrict__ obs_point,
                                                             const double2* __restrict__ meas, long long O, cam4 cam,
                                                             int K, int L, unsigned int* __restrict__ index_errors,
                                                             double* __restrict__ e, double* __restrict__ Jpose,
                                                             double* __restrict__ Jpoint) {
    __shared__ double sJ[RJ_BLOCK * 12];  // 24 KiB, reused for Jpoint (12 KiB) after Jpose
    const long long base = (long long)blockIdx.x * RJ_BLOCK;
    const long long o = base + threadIdx.x;
    const int cnt = (int)((O - base) < RJ_BLOCK ? (O - base) : RJ_BLOCK);
    proj_out r;
    double Rm[9];
    if (o < O) {
        int k = obs_pose[o], l = obs_point[o];
        double2 m = meas[o];
        if ((unsigned)k >= (unsigned)K || (unsigned)l >= (unsigned)L) {
            // misuse is reported, never dereferenced: the observation reads row 0 and comes out as NaN
            atomicAdd(index_errors, 1u);
            k = 0; l = 0;
            m.x = m.y = __builtin_nan("");
        }
        double P[12];
        const double2* pp = (const double2*)(poses + (size_t)k * 12);  // 96-B rows, 16-B aligned
#pragma unroll
        for (int i = 0; i < 6; i++) { const double2 v = pp[i]; P[2 * i] = v.x; P[2 * i + 1] = v.y; }
        const double px = points[(size_t)l * 3], py = points[(size_t)l * 3 + 1], pz = points[(size_t)l * 3 + 2];
        project(P, px, py, pz, m.x, m.y, cam, r);
        Rm[0] = P[0]; Rm[1] = P[1]; Rm[2] = P[2]; Rm[3] = P[4]; Rm[4] = P[5]; Rm[5] = P[6];
        Rm[6] = P[8]; Rm[7] = P[9]; Rm[8] = P[10];
        f64x2 ev; ev.x = r.e0; ev.y = r.e1;
        rj_store16(&((f64x2*)e)[o], ev);                   // 16 B per lane, already coalesced
#pragma unroll
        for (int i = 0; i < 6; i++) ((double2*)sJ)[threadIdx.x * 6 + i] = make_double2(r.jp[2 * i], r.jp[2 * i + 1]);
    }
    __syncthreads();
    store_rows<12>(sJ, Jpose + (size_t)base * 12, cnt);
    if (WITH_POINT) {
        __syncthreads();
        if (o < O) {
            // J_point = -A * R
            double jq[6];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                jq[c] = -(r.A[0] * Rm[c] + r.A[2] * Rm[6 + c]);
                jq[3 + c] = -(r.A[4] * Rm[3 + c] + r.A[5] * Rm[6 + c]);
            }
#pragma unroll
            for (int i = 0; i < 3; i++) ((double2*)sJ)[threadIdx.x * 3 + i] = make_double2(jq[2 * i], jq[2 * i + 1]);
        }
        __syncthreads();
        store_rows<6>(sJ, Jpoint + (size_t)base * 6, cnt);
    }
}

extern "C" int slam_reproj_rj_f64(slam_ctx* ctx, const double* d_poses, int64_t K, const double* d_points,
                                  int64_t L, const int32_t* d_obs_pose, const int32_t* d_obs_point,
                                  const double* d_meas, int64_t O, double fx, double fy, double cx, double cy,
                                  double* d_e, double* d_Jpose, double* d_Jpoint) {
    SLAM_REQUIRE(ctx, "slam_reproj_rj_f64: null ctx");
    SLAM_REQUIRE(O >= 0 && K >= 0 && L >= 0, "negative size");
    SLAM_REQUIRE(O <= (1ll << 40), "O too large");
    if (O == 0) return SLAM_OK;
    SLAM_REQUIRE(K > 0 && L > 0, "observations given but no poses/points");
    SLAM_REQUIRE(K <= 0x7FFFFFFF && L <= 0x7FFFFFFF, "pose / point tables are indexed by int32");
    SLAM_REQUIRE(d_poses && d_points && d_obs_pose && d_obs_point && d_meas && d_e && d_Jpose,
                 "slam_reproj_rj_f64: null device pointer");
    SLAM_REQUIRE((((uintptr_t)d_poses | (uintptr_t)d_meas | (uintptr_t)d_e | (uintptr_t)d_Jpose |
                   (uintptr_t)d_Jpoint) & 15) == 0, "f64 arrays must be 16-byte aligned");
    SLAM_HIP(hipSetDevice(ctx->device));
    const cam4 cam = {fx, fy, cx, cy};
    const unsigned blocks = (unsigned)((O + RJ_BLOCK - 1) / RJ_BLOCK);
    if (int rc = slam_prof_begin(ctx)) return rc;
    if (d_Jpoint)
        reproj_rj_kernel<true><<<blocks, RJ_BLOCK, 0, ctx->stream>>>(d_poses, d_points, d_obs_pose, d_obs_point,
                                                                      (const double2*)d_meas, O, cam, (int)K, (int)L,
                                                                      slam_index_error_counter(ctx), d_e, d_Jpose, d_Jpoint);
    else
        reproj_rj_kernel<false><<<blocks, RJ_BLOCK, 0, ctx->stream>>>(d_poses, d_points, d_obs_pose, d_obs_point,
                                                                       (const double2*)d_meas, O, cam, (int)K, (int)L,
                                                                       slam_index_error_counter(ctx), d_e, d_Jpose, nullptr);
    if (int rc = slam_prof_end(ctx)) return rc;
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}

// ---- pose-only normal equations (one pose, O observations) -----------------
// 27 sums: upper triangle of H (21) + b (6).  Two deterministic stages: each
// block reduces its grid-stride slice in a fixed order (wave shuffle tree, then
// waves in order); a single block then adds the block partials in index order.
#define NE_TERMS 27
#define NE_MAX_BLOCKS 128   // 128 * 27 * 8 B = 27 KiB of workspace

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(256) void pose_ne_partial_kernel(const double* __restrict__ pose,
                                                              const double* __restrict__ points,
                                                              const double2* __restrict__ meas,
                                                              const uint8_t* __restrict__ active, long long O,
                                                              cam4 cam, double huber_delta,
                                                              double* __restrict__ chi2,
                                                              double* __restrict__ partial) {
    __shared__ double sw[4][NE_TERMS];
    double P[12];
#pragma unroll
    for (int i = 0; i < 12; i++) P[i] = pose[i];
    double acc[NE_TERMS];
#pragma unroll
    for (int i = 0; i < NE_TERMS; i++) acc[i] = 0.0;
    for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < O; o += (long long)gridDim.x * 256) {
        const double2 m = meas[o];
        proj_out r;
        project(P, points[o * 3], points[o * 3 + 1], points[o * 3 + 2], m.x, m.y, cam, r);
        const double c2 = r.e0 * r.e0 + r.e1 * r.e1;  // information = I2 (frontend.py:349)
        chi2[o] = c2;
        if (active && !active[o]) continue;           // level-1 edges are left out (frontend.py:372-377)
        // g2o RobustKernelHuber: rho'(e2) = 1 if sqrt(e2) <= delta else delta / sqrt(e2)
        double w = 1.0;
        if (huber_delta > 0.0) {
            const double en = sqrt(c2);
            if (en > huber_delta) w = huber_delta / en;
        }
        int t = 0;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = a; b < 6; b++) acc[t++] += w * (r.jp[a] * r.jp[b] + r.jp[6 + a] * r.jp[6 + b]);
#pragma unroll
        for (int a = 0; a < 6; a++) acc[21 + a] += w * (r.jp[a] * r.e0 + r.jp[6 + a] * r.e1);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NE_TERMS; i++) {
        const double s = wave_sum_f64(acc[i]);
        if (lane == 0) sw[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < NE_TERMS)
        partial[(size_t)blockIdx.x * NE_TERMS + threadIdx.x] =
            ((sw[0][threadIdx.x] + sw[1][threadIdx.x]) + sw[2][threadIdx.x]) + sw[3][threadIdx.x];
}

__global__ __launch_bounds__(64) void pose_ne_final_kernel(const double* __restrict__ partial, int nblocks,
                                                           double* __restrict__ H, double* __restrict__ b) {
    __shared__ double s[NE_TERMS];
    if (threadIdx.x < NE_TERMS) {
        double v = 0.0;
        for (int i = 0; i < nblocks; i++) v += partial[(size_t)i * NE_TERMS + threadIdx.x];
        s[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x < 36) {
        const int a = threadIdx.x / 6, c = threadIdx.x % 6;
        const int lo = a < c ? a : c, hi = a < c ? c : a;
        // index of (lo,hi) in the packed upper triangle
        const int t = lo * 6 - lo * (lo - 1) / 2 + (hi - lo);
        H[threadIdx.x] = s[t];
    }
    if (threadIdx.x < 6) b[threadIdx.x] = s[21 + threadIdx.x];
}

extern "C" int slam_pose_normal_eq_f64(slam_ctx* ctx, const double* d_pose, const double* d_points,
                                       const double* d_meas, const uint8_t* d_active, int64_t O, double fx,
                                       double fy, double cx, double cy, double huber_delta, double* d_H,
                                       double* d_b, double* d_chi2) {
    SLAM_REQUIRE(ctx, "slam_pose_normal_eq_f64: null ctx");
    SLAM_REQUIRE(O >= 0, "negative size");
    SLAM_REQUIRE(d_pose && d_H && d_b && (O == 0 || (d_points && d_meas && d_chi2)),
                 "slam_pose_normal_eq_f64: null device pointer");
    SLAM_REQUIRE(((uintptr_t)d_meas & 15) == 0, "d_meas must be 16-byte aligned");
    SLAM_HIP(hipSetDevice(ctx->device));
    int nblocks = (int)((O + 255) / 256);
    if (nblocks > NE_MAX_BLOCKS) nblocks = NE_MAX_BLOCKS;
    if (nblocks < 1) nblocks = 1;
    void* ws = nullptr;
    if (int rc = slam_workspace(ctx, (uint64_t)NE_MAX_BLOCKS * NE_TERMS * sizeof(double), &ws)) return rc;
    const cam4 cam = {fx, fy, cx, cy};
    pose_ne_partial_kernel<<<nblocks, 256, 0, ctx->stream>>>(d_pose, d_points, (const double2*)d_meas, d_active, O,
                                                             cam, huber_delta, d_chi2, (double*)ws);
    pose_ne_final_kernel<<<1, 64, 0, ctx->stream>>>((const double*)ws, nblocks, d_H, d_b);
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}
