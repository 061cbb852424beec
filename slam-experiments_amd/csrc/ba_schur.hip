// ba_schur.hip — reduced camera system of a keyframe-window bundle adjustment on gfx950.
//
// Extension: the reference has no bundle adjustment at all
// (Backend is an empty class, backend.py:101-103; Map keeps NUM_ACTIVE_KEYFRAMES = 7
// keyframes, backend.py:11).  Residuals and Jacobians are the arithmetic of
// frontend.py:272-291 (+ the 2x3 point Jacobian), the same as reproj.hip.
//
// One LM step needs, with J = [Jp | Jq] per observation and w its Huber weight:
//   Hpp_k = sum Jp^T w Jp, bp_k = sum Jp^T w e            (per pose)
//   Hll_l = sum Jq^T w Jq, bl_l = sum Jq^T w e            (per point)
//   Hpl_o = Jp^T w Jq                                     (per observation, 6x3)
//   E_l   = (Hll_l + lam I)^-1,  Y_o = Hpl_o E_l
//   S[k1,k2] = [k1==k2](Hpp_k1 + lam I) - sum_{l seen by k1 and k2} Y_(k1,l) Hpl_(k2,l)^T
//   rhs_k    = -bp_k + sum_{o of pose k} Y_o bl_point(o)
// then (host, 6K x 6K) S dp = rhs, and dl_l = E_l (-bl_l - sum_{o of l} Hpl_o^T dp_pose(o)).
//
// Everything is a gather + fixed-order reduction (no float atomics), so results are
// run-to-run identical:
//   ba_obs_kernel     one thread per observation : e, Jp, Jq, w -> Hpl, per-obs products
//   ba_point_kernel   one thread per point       : Hll, bl, E, Y_o (over the point's CSR row)
//   ba_pose_kernel    one block per pose         : Hpp, bp, rhs, cost (over the pose's obs list)
//   ba_pair_kernel    one block per (k1 <= k2)   : the S block (over k1's obs list, looking k2's up)
//   ba_backsub_kernel one thread per point       : dl
#include "internal.h"
#include <math.h>

#define BA_THREADS 256

struct ba_cam { double fx, fy, cx, cy; };

// per-observation record written by ba_obs_kernel / ba_point_kernel (all f64)
//   [0..17]  Hpl (6x3 row-major)      [18..35] Y = Hpl E (6x3)
//   [36..56] upper triangle of Jp^T w Jp (21)   [57..62] Jp^T w e (6)
//   [63..68] upper triangle of Jq^T w Jq (6)    [69..71] Jq^T w e (3)    [72] rho(e.e)
#define BA_REC 73

__device__ __forceinline__ double ba_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(BA_THREADS) void ba_obs_kernel(const double* __restrict__ poses,
                                                            const double* __restrict__ points,
                                                            const int* __restrict__ obs_pose,
                                                            const int* __restrict__ obs_point,
                                                            const double2* __restrict__ meas, int O, ba_cam cam,
                                                            double delta, int K, int L,
                                                            unsigned int* __restrict__ index_errors,
                                                            double* __restrict__ rec) {
    const int o = blockIdx.x * BA_THREADS + threadIdx.x;
    if (o >= O) return;
    int k = obs_pose[o], l = obs_point[o];
    bool bad = false;
    if ((unsigned)k >= (unsigned)K || (unsigned)l >= (unsigned)L) {   // reported (slam_index_errors), never dereferenced
        atomicAdd(index_errors, 1u);
        k = 0; l = 0; bad = true;
    }
    const double* P = poses + (size_t)k * 12;
    const double* p = points + (size_t)l * 3;
    const double X = P[0] * p[0] + P[1] * p[1] + P[2] * p[2] + P[3];
    const double Y = P[4] * p[0] + P[5] * p[1] + P[6] * p[2] + P[7];
    const double Z = P[8] * p[0] + P[9] * p[1] + P[10] * p[2] + P[11];
    double2 m = meas[o];
    if (bad) m.x = m.y = __builtin_nan("");
    const double e0 = m.x - (cam.fx * X + cam.cx * Z) / Z;      // frontend.py:275-277
    const double e1 = m.y - (cam.fy * Y + cam.cy * Z) / Z;
    const double Zinv = 1.0 / (Z + 1e-18), Zinv2 = Zinv * Zinv;  // frontend.py:284-291
    const double jp[2][6] = {{cam.fx * X * Y * Zinv2, -cam.fx - cam.fx * X * X * Zinv2, cam.fx * Y * Zinv,
                              -cam.fx * Zinv, 0.0, cam.fx * X * Zinv2},
                             {cam.fy + cam.fy * Y * Y * Zinv2, -cam.fy * X * Y * Zinv2, -cam.fy * X * Zinv, 0.0,
                              -cam.fy * Zinv, cam.fy * Y * Zinv2}};
    const double A[2][3] = {{cam.fx * Zinv, 0.0, -cam.fx * X * Zinv2}, {0.0, cam.fy * Zinv, -cam.fy * Y * Zinv2}};
    double jq[2][3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        jq[0][c] = -(A[0][0] * P[c] + A[0][2] * P[8 + c]);
        jq[1][c] = -(A[1][1] * P[4 + c] + A[1][2] * P[8 + c]);
    }
    const double c2 = e0 * e0 + e1 * e1;
    double w = 1.0, rho = c2;
    if (delta > 0.0) {
        const double en = sqrt(c2);
        if (en > delta) { w = delta / en; rho = 2.0 * delta * en - delta * delta; }
    }
    double* r = rec + (size_t)o * BA_REC;
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
        for (int c = 0; c < 3; c++) r[a * 3 + c] = w * (jp[0][a] * jq[0][c] + jp[1][a] * jq[1][c]);
    int t = 36;
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
        for (int b = a; b < 6; b++) r[t++] = w * (jp[0][a] * jp[0][b] + jp[1][a] * jp[1][b]);
#pragma unroll
    for (int a = 0; a < 6; a++) r[57 + a] = w * (jp[0][a] * e0 + jp[1][a] * e1);
    t = 63;
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = a; b < 3; b++) r[t++] = w * (jq[0][a] * jq[0][b] + jq[1][a] * jq[1][b]);
#pragma unroll
    for (int a = 0; a < 3; a++) r[69 + a] = w * (jq[0][a] * e0 + jq[1][a] * e1);
    r[72] = rho;
}

// per point: Hll, bl over its observations (CSR row, ascending), E = (Hll + lam I)^-1, Y_o = Hpl_o E
__global__ __launch_bounds__(BA_THREADS) void ba_point_kernel(const int* __restrict__ pt_ptr,
                                                              const int* __restrict__ pt_obs, int L, double lam,
                                                              double* __restrict__ rec, double* __restrict__ E,
                                                              double* __restrict__ bl,
                                                              double* __restrict__ hll_diag /*[L,3] or null*/) {
    const int l = blockIdx.x * BA_THREADS + threadIdx.x;
    if (l >= L) return;
    double h[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
    const int a0 = pt_ptr[l], a1 = pt_ptr[l + 1];
    for (int i = a0; i < a1; i++) {
        const double* r = rec + (size_t)pt_obs[i] * BA_REC;
#pragma unroll
        for (int k = 0; k < 6; k++) h[k] += r[63 + k];
#pragma unroll
        for (int k = 0; k < 3; k++) b[k] += r[69 + k];
    }
    // symmetric 3x3 inverse of H + lam I (identity for a point nobody observes)
    double m00 = h[0] + lam, m01 = h[1], m02 = h[2], m11 = h[3] + lam, m12 = h[4], m22 = h[5] + lam;
    if (a1 == a0) { m00 = m11 = m22 = 1.0; m01 = m02 = m12 = 0.0; }
    const double c00 = m11 * m22 - m12 * m12, c01 = m02 * m12 - m01 * m22, c02 = m01 * m12 - m02 * m11;
    const double det = m00 * c00 + m01 * c01 + m02 * c02;
    const double id = 1.0 / det;
    double e[9];
    e[0] = c00 * id; e[1] = c01 * id; e[2] = c02 * id;
    e[3] = e[1]; e[4] = (m00 * m22 - m02 * m02) * id; e[5] = (m01 * m02 - m00 * m12) * id;
    e[6] = e[2]; e[7] = e[5]; e[8] = (m00 * m11 - m01 * m01) * id;
#pragma unroll
    for (int k = 0; k < 9; k++) E[(size_t)l * 9 + k] = e[k];
#pragma unroll
    for (int k = 0; k < 3; k++) bl[(size_t)l * 3 + k] = b[k];
    if (hll_diag) {
        hll_diag[(size_t)l * 3] = h[0];
        hll_diag[(size_t)l * 3 + 1] = h[3];
        hll_diag[(size_t)l * 3 + 2] = h[5];
    }
    for (int i = a0; i < a1; i++) {
        double* r = rec + (size_t)pt_obs[i] * BA_REC;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int c = 0; c < 3; c++)
                r[18 + a * 3 + c] = r[a * 3] * e[c] + r[a * 3 + 1] * e[3 + c] + r[a * 3 + 2] * e[6 + c];
    }
}

// block-wide fixed-order sum of NT per-thread accumulators into out[NT] (shared), all threads return after it
template <int NT>
__device__ __forceinline__ void ba_block_sum(const double (&acc)[NT], double (*sw)[NT], double* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NT; i++) {
        const double s = ba_wave_sum(acc[i]);
        if (lane == 0) sw[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < NT) out[threadIdx.x] = ((sw[0][threadIdx.x] + sw[1][threadIdx.x]) + sw[2][threadIdx.x]) + sw[3][threadIdx.x];
    __syncthreads();
}

// one block per pose k: Hpp (21), bp (6), y = sum Y_o bl (6), cost (1) over the pose's observation list
__global__ __launch_bounds__(BA_THREADS) void ba_pose_kernel(const int* __restrict__ ps_ptr,
                                                             const int* __restrict__ ps_obs,
                                                             const int* __restrict__ obs_point,
                                                             const double* __restrict__ rec,
                                                             const double* __restrict__ bl,
                                                             double* __restrict__ Hpp /*[K,21]*/,
                                                             double* __restrict__ bp /*[K,6]*/,
                                                             double* __restrict__ ybl /*[K,6]*/,
                                                             double* __restrict__ cost /*[K]*/) {
    __shared__ double sw[4][34];
    __shared__ double out[34];
    const int k = blockIdx.x;
    double acc[34];
#pragma unroll
    for (int i = 0; i < 34; i++) acc[i] = 0.0;
    for (int i = ps_ptr[k] + threadIdx.x; i < ps_ptr[k + 1]; i += BA_THREADS) {
        const int o = ps_obs[i];
        const double* r = rec + (size_t)o * BA_REC;
        const double* b = bl + (size_t)obs_point[o] * 3;
#pragma unroll
        for (int t = 0; t < 27; t++) acc[t] += r[36 + t];
#pragma unroll
        for (int a = 0; a < 6; a++) acc[27 + a] += r[18 + a * 3] * b[0] + r[18 + a * 3 + 1] * b[1] + r[18 + a * 3 + 2] * b[2];
        acc[33] += r[72];
    }
    ba_block_sum<34>(acc, sw, out);
    if (threadIdx.x < 21) Hpp[(size_t)k * 21 + threadIdx.x] = out[threadIdx.x];
    if (threadIdx.x < 6) {
        bp[(size_t)k * 6 + threadIdx.x] = out[21 + threadIdx.x];
        ybl[(size_t)k * 6 + threadIdx.x] = out[27 + threadIdx.x];
    }
    if (threadIdx.x == 0) cost[k] = out[33];
}

// one block per pose pair (k1 <= k2): W[k1,k2] = sum over points seen by both of Y_(k1,l) Hpl_(k2,l)^T (6x6)
__global__ __launch_bounds__(BA_THREADS) void ba_pair_kernel(const int* __restrict__ ps_ptr,
                                                             const int* __restrict__ ps_obs,
                                                             const int* __restrict__ obs_point,
                                                             const int* __restrict__ lookup /*[K,L] obs of (pose, point) or -1*/,
                                                             int K, int L, const double* __restrict__ rec,
                                                             double* __restrict__ W /*[K,K,36]*/) {
    __shared__ double sw[4][36];
    __shared__ double out[36];
    const int k1 = blockIdx.x, k2 = blockIdx.y;
    if (k2 < k1) return;
    double acc[36];
#pragma unroll
    for (int i = 0; i < 36; i++) acc[i] = 0.0;
    for (int i = ps_ptr[k1] + threadIdx.x; i < ps_ptr[k1 + 1]; i += BA_THREADS) {
        const int o1 = ps_obs[i];
        const int o2 = lookup[(size_t)k2 * L + obs_point[o1]];
        if (o2 < 0) continue;
        const double* y = rec + (size_t)o1 * BA_REC + 18;   // Y of (k1, l)
        const double* h = rec + (size_t)o2 * BA_REC;        // Hpl of (k2, l)
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = 0; b < 6; b++)
                acc[a * 6 + b] += y[a * 3] * h[b * 3] + y[a * 3 + 1] * h[b * 3 + 1] + y[a * 3 + 2] * h[b * 3 + 2];
    }
    ba_block_sum<36>(acc, sw, out);
    if (threadIdx.x < 36) W[((size_t)k1 * K + k2) * 36 + threadIdx.x] = out[threadIdx.x];
}

// per point: dl = E (-bl - sum_{o of l} Hpl_o^T dp_pose(o))
__global__ __launch_bounds__(BA_THREADS) void ba_backsub_kernel(const int* __restrict__ pt_ptr,
                                                                const int* __restrict__ pt_obs,
                                                                const int* __restrict__ obs_pose, int L,
                                                                const double* __restrict__ rec,
                                                                const double* __restrict__ E,
                                                                const double* __restrict__ bl,
                                                                const double* __restrict__ dp /*[K,6]*/,
                                                                double* __restrict__ dl /*[L,3]*/) {
    const int l = blockIdx.x * BA_THREADS + threadIdx.x;
    if (l >= L) return;
    double t[3] = {-bl[(size_t)l * 3], -bl[(size_t)l * 3 + 1], -bl[(size_t)l * 3 + 2]};
    const int a0 = pt_ptr[l], a1 = pt_ptr[l + 1];
    for (int i = a0; i < a1; i++) {
        const int o = pt_obs[i];
        const double* r = rec + (size_t)o * BA_REC;
        const double* d = dp + (size_t)obs_pose[o] * 6;
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int a = 0; a < 6; a++) t[c] -= r[a * 3 + c] * d[a];
    }
    const double* e = E + (size_t)l * 9;
    const bool seen = a1 > a0;
#pragma unroll
    for (int c = 0; c < 3; c++) dl[(size_t)l * 3 + c] = seen ? e[c * 3] * t[0] + e[c * 3 + 1] * t[1] + e[c * 3 + 2] * t[2] : 0.0;
}

// robust cost only (an LM trial needs nothing else from the candidate state): one block per pose
__global__ __launch_bounds__(BA_THREADS) void ba_cost_kernel(const double* __restrict__ poses,
                                                             const double* __restrict__ points,
                                                             const int* __restrict__ obs_point,
                                                             const double2* __restrict__ meas,
                                                             const int* __restrict__ ps_ptr,
                                                             const int* __restrict__ ps_obs, ba_cam cam, double delta,
                                                             double* __restrict__ cost /*[K]*/) {
    __shared__ double sw[4][1];
    __shared__ double out[1];
    const int k = blockIdx.x;
    const double* P = poses + (size_t)k * 12;
    double acc[1] = {0.0};
    for (int i = ps_ptr[k] + threadIdx.x; i < ps_ptr[k + 1]; i += BA_THREADS) {
        const int o = ps_obs[i];
        const double* p = points + (size_t)obs_point[o] * 3;
        const double X = P[0] * p[0] + P[1] * p[1] + P[2] * p[2] + P[3];
        const double Y = P[4] * p[0] + P[5] * p[1] + P[6] * p[2] + P[7];
        const double Z = P[8] * p[0] + P[9] * p[1] + P[10] * p[2] + P[11];
        const double2 m = meas[o];
        const double e0 = m.x - (cam.fx * X + cam.cx * Z) / Z;
        const double e1 = m.y - (cam.fy * Y + cam.cy * Z) / Z;
        const double c2 = e0 * e0 + e1 * e1;
        double rho = c2;
        if (delta > 0.0) {
            const double en = sqrt(c2);
            if (en > delta) rho = 2.0 * delta * en - delta * delta;
        }
        acc[0] += rho;
    }
    ba_block_sum<1>(acc, sw, out);
    if (threadIdx.x == 0) cost[k] = out[0];
}

extern "C" int slam_ba_cost_f64(slam_ctx* ctx, const double* d_poses, int64_t K, const double* d_points,
                                const int32_t* d_obs_point, const double* d_meas, const int32_t* d_ps_ptr,
                                const int32_t* d_ps_obs, double fx, double fy, double cx, double cy,
                                double huber_delta, double* d_cost) {
    SLAM_REQUIRE(ctx, "slam_ba_cost_f64: null ctx");
    SLAM_REQUIRE(K >= 1 && K <= 1024, "bad K");
    SLAM_REQUIRE(d_poses && d_points && d_obs_point && d_meas && d_ps_ptr && d_ps_obs && d_cost,
                 "slam_ba_cost_f64: null device pointer");
    SLAM_REQUIRE(((uintptr_t)d_meas & 15) == 0, "d_meas must be 16-byte aligned");
    SLAM_HIP(hipSetDevice(ctx->device));
    const ba_cam cam = {fx, fy, cx, cy};
    ba_cost_kernel<<<(unsigned)K, BA_THREADS, 0, ctx->stream>>>(d_poses, d_points, d_obs_point, (const double2*)d_meas,
                                                                d_ps_ptr, d_ps_obs, cam, huber_delta, d_cost);
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}

extern "C" int slam_ba_reduce_f64(slam_ctx* ctx, const double* d_poses, int64_t K, const double* d_points,
                                  int64_t L, const int32_t* d_obs_pose, const int32_t* d_obs_point,
                                  const double* d_meas, int64_t O, const int32_t* d_pt_ptr, const int32_t* d_pt_obs,
                                  const int32_t* d_ps_ptr, const int32_t* d_ps_obs, const int32_t* d_lookup,
                                  double fx, double fy, double cx, double cy, double huber_delta, double lambda,
                                  double* d_rec, double* d_E, double* d_bl, double* d_Hpp, double* d_bp,
                                  double* d_ybl, double* d_cost, double* d_W, double* d_hll_diag) {
    SLAM_REQUIRE(ctx, "slam_ba_reduce_f64: null ctx");
    SLAM_REQUIRE(K >= 1 && K <= 1024 && L >= 1 && O >= 0 && L <= (1 << 28) && O <= (1 << 28), "bad sizes");
    SLAM_REQUIRE(d_poses && d_points && d_obs_pose && d_obs_point && d_meas && d_pt_ptr && d_pt_obs && d_ps_ptr &&
                     d_ps_obs && d_lookup && d_rec && d_E && d_bl && d_Hpp && d_bp && d_ybl && d_cost && d_W,
                 "slam_ba_reduce_f64: null device pointer");
    SLAM_REQUIRE(((uintptr_t)d_meas & 15) == 0, "d_meas must be 16-byte aligned");
    SLAM_HIP(hipSetDevice(ctx->device));
    const ba_cam cam = {fx, fy, cx, cy};
    if (O)
        ba_obs_kernel<<<(unsigned)((O + BA_THREADS - 1) / BA_THREADS), BA_THREADS, 0, ctx->stream>>>(
            d_poses, d_points, d_obs_pose, d_obs_point, (const double2*)d_meas, (int)O, cam, huber_delta, (int)K, (int)L,
            slam_index_error_counter(ctx), d_rec);
    ba_point_kernel<<<(unsigned)((L + BA_THREADS - 1) / BA_THREADS), BA_THREADS, 0, ctx->stream>>>(
        d_pt_ptr, d_pt_obs, (int)L, lambda, d_rec, d_E, d_bl, d_hll_diag);
    ba_pose_kernel<<<(unsigned)K, BA_THREADS, 0, ctx->stream>>>(d_ps_ptr, d_ps_obs, d_obs_point, d_rec, d_bl, d_Hpp,
                                                                 d_bp, d_ybl, d_cost);
    ba_pair_kernel<<<dim3((unsigned)K, (unsigned)K), BA_THREADS, 0, ctx->stream>>>(d_ps_ptr, d_ps_obs, d_obs_point,
                                                                                    d_lookup, (int)K, (int)L, d_rec, d_W);
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}

extern "C" int slam_ba_backsub_f64(slam_ctx* ctx, int64_t L, const int32_t* d_pt_ptr, const int32_t* d_pt_obs,
                                   const int32_t* d_obs_pose, const double* d_rec, const double* d_E,
                                   const double* d_bl, const double* d_dp, double* d_dl) {
    SLAM_REQUIRE(ctx, "slam_ba_backsub_f64: null ctx");
    SLAM_REQUIRE(L >= 1 && L <= (1 << 28), "bad sizes");
    SLAM_REQUIRE(d_pt_ptr && d_pt_obs && d_obs_pose && d_rec && d_E && d_bl && d_dp && d_dl,
                 "slam_ba_backsub_f64: null device pointer");
    SLAM_HIP(hipSetDevice(ctx->device));
    ba_backsub_kernel<<<(unsigned)((L + BA_THREADS - 1) / BA_THREADS), BA_THREADS, 0, ctx->stream>>>(
        d_pt_ptr, d_pt_obs, d_obs_pose, (int)L, d_rec, d_E, d_bl, d_dp, d_dl);
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}
