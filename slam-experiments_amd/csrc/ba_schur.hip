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
#include <atomic>
#include <math.h>

#define BA_THREADS 256

struct ba_cam { double fx, fy, cx, cy; };

// per-observation record written by ba_obs_kernel / ba_point_kernel (all f64)
//   [0..17]  Hpl (6x3 row-major)      [18..35] Y = Hpl E (6x3)
//   [36..56] upper triangle of Jp^T w Jp (21)   [57..62] Jp^T w e (6)
//   [63..68] upper triangle of Jq^T w Jq (6)    [69..71] Jq^T w e (3)    [72] rho(e.e)
#define BA_REC 73
// The records are kept observation-major (73 consecutive doubles per observation).  Two field-major layouts - field f
// of observation o at rec[f * O + o], and the same inside tiles of 64 observations - were measured while the one-launch
// form at the end of this file still passed records from phase to phase (tools/ba_phase_probe.py) and were SLOWER,
// although they make a wave's loads contiguous: its pose and pair phases went from 20 to 48 us at the reference's window;
// a thread that walks its own contiguous record keeps more reads in flight than 46 separate streams do.  That form has
// since dropped the records altogether (it linearises again wherever it needs a Jacobian); the per-phase kernels below,
// which the host drives for windows beyond its limits, keep them.
struct ba_rec_ref {
    double* base;
    __device__ __forceinline__ double& operator[](int f) const { return base[f]; }
};
struct ba_recs {
    double* p;
    __device__ __forceinline__ ba_rec_ref of(int o) const { return {p + (size_t)o * BA_REC}; }
};

__device__ __forceinline__ double ba_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// (the *_body functions take the block index as an argument: blockIdx.x of the per-phase kernels below, a loop variable
// of the persistent kernel in ba_lm.hip's grid form at the end of this file)
// one observation linearised: residual e, robust weight w and cost rho, the 2x6 pose Jacobian (rotation first) and the 2x3
// point Jacobian, at pose P (3x4 row-major, any address space) and point p
struct ba_lin { double e0, e1, w, rho, jp[2][6], jq[2][3]; };

__device__ __forceinline__ void ba_linearise(const double* P, const double* p, const double2 m, const ba_cam& cam, const double delta,
                                             ba_lin& q) {
    const double X = P[0] * p[0] + P[1] * p[1] + P[2] * p[2] + P[3];
    const double Y = P[4] * p[0] + P[5] * p[1] + P[6] * p[2] + P[7];
    const double Z = P[8] * p[0] + P[9] * p[1] + P[10] * p[2] + P[11];
    q.e0 = m.x - (cam.fx * X + cam.cx * Z) / Z;      // frontend.py:275-277
    q.e1 = m.y - (cam.fy * Y + cam.cy * Z) / Z;
    const double Zinv = 1.0 / (Z + 1e-18), Zinv2 = Zinv * Zinv;  // frontend.py:284-291
    q.jp[0][0] = cam.fx * X * Y * Zinv2; q.jp[0][1] = -cam.fx - cam.fx * X * X * Zinv2; q.jp[0][2] = cam.fx * Y * Zinv;
    q.jp[0][3] = -cam.fx * Zinv; q.jp[0][4] = 0.0; q.jp[0][5] = cam.fx * X * Zinv2;
    q.jp[1][0] = cam.fy + cam.fy * Y * Y * Zinv2; q.jp[1][1] = -cam.fy * X * Y * Zinv2; q.jp[1][2] = -cam.fy * X * Zinv;
    q.jp[1][3] = 0.0; q.jp[1][4] = -cam.fy * Zinv; q.jp[1][5] = cam.fy * Y * Zinv2;
    const double A[2][3] = {{cam.fx * Zinv, 0.0, -cam.fx * X * Zinv2}, {0.0, cam.fy * Zinv, -cam.fy * Y * Zinv2}};
#pragma unroll
    for (int c = 0; c < 3; c++) {
        q.jq[0][c] = -(A[0][0] * P[c] + A[0][2] * P[8 + c]);
        q.jq[1][c] = -(A[1][1] * P[4 + c] + A[1][2] * P[8 + c]);
    }
    const double c2 = q.e0 * q.e0 + q.e1 * q.e1;
    q.w = 1.0; q.rho = c2;
    if (delta > 0.0) {
        const double en = sqrt(c2);
        if (en > delta) { q.w = delta / en; q.rho = 2.0 * delta * en - delta * delta; }
    }
}

__device__ __forceinline__ void ba_obs_body(const double* __restrict__ poses, const double* __restrict__ points,
                                            const int* __restrict__ obs_pose, const int* __restrict__ obs_point,
                                            const double2* __restrict__ meas, int O, ba_cam cam, double delta, int K, int L,
                                            unsigned int* __restrict__ index_errors, const ba_recs rec, const int o) {
    if (o >= O) return;
    int k = obs_pose[o], l = obs_point[o];
    bool bad = false;
    if ((unsigned)k >= (unsigned)K || (unsigned)l >= (unsigned)L) {   // reported (slam_index_errors), never dereferenced
        atomicAdd(index_errors, 1u);
        k = 0; l = 0; bad = true;
    }
    double2 m = meas[o];
    if (bad) m.x = m.y = __builtin_nan("");
    ba_lin q;
    ba_linearise(poses + (size_t)k * 12, points + (size_t)l * 3, m, cam, delta, q);
    const double e0 = q.e0, e1 = q.e1, w = q.w, rho = q.rho;
    const double (&jp)[2][6] = q.jp;
    const double (&jq)[2][3] = q.jq;
    const ba_rec_ref r = rec.of(o);
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

__global__ __launch_bounds__(BA_THREADS) void ba_obs_kernel(const double* __restrict__ poses,
                                                            const double* __restrict__ points,
                                                            const int* __restrict__ obs_pose,
                                                            const int* __restrict__ obs_point,
                                                            const double2* __restrict__ meas, int O, ba_cam cam,
                                                            double delta, int K, int L,
                                                            unsigned int* __restrict__ index_errors,
                                                            double* __restrict__ rec) {
    ba_obs_body(poses, points, obs_pose, obs_point, meas, O, cam, delta, K, L, index_errors, ba_recs{rec}, (int)(blockIdx.x * BA_THREADS + threadIdx.x));
}

// symmetric 3x3 inverse of H + lam I, H as its packed upper triangle (the identity for a point nobody observes)
__device__ __forceinline__ void ba_damped_inverse(const double (&h)[6], const double lam, const bool seen, double (&e)[9]) {
    double m00 = h[0] + lam, m01 = h[1], m02 = h[2], m11 = h[3] + lam, m12 = h[4], m22 = h[5] + lam;
    if (!seen) { m00 = m11 = m22 = 1.0; m01 = m02 = m12 = 0.0; }
    const double c00 = m11 * m22 - m12 * m12, c01 = m02 * m12 - m01 * m22, c02 = m01 * m12 - m02 * m11;
    const double det = m00 * c00 + m01 * c01 + m02 * c02;
    const double id = 1.0 / det;
    e[0] = c00 * id; e[1] = c01 * id; e[2] = c02 * id;
    e[3] = e[1]; e[4] = (m00 * m22 - m02 * m02) * id; e[5] = (m01 * m02 - m00 * m12) * id;
    e[6] = e[2]; e[7] = e[5]; e[8] = (m00 * m11 - m01 * m01) * id;
}

// per point: Hll, bl over its observations (CSR row, ascending), E = (Hll + lam I)^-1, Y_o = Hpl_o E
__device__ __forceinline__ void ba_point_body(const int* __restrict__ pt_ptr, const int* __restrict__ pt_obs, int L, double lam,
                                              const ba_recs rec, double* __restrict__ E, double* __restrict__ bl,
                                              double* __restrict__ hll_diag /*[L,3] or null*/, const int l) {
    if (l >= L) return;
    double h[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
    const int a0 = pt_ptr[l], a1 = pt_ptr[l + 1];
    for (int i = a0; i < a1; i++) {
        const ba_rec_ref r = rec.of(pt_obs[i]);
#pragma unroll
        for (int k = 0; k < 6; k++) h[k] += r[63 + k];
#pragma unroll
        for (int k = 0; k < 3; k++) b[k] += r[69 + k];
    }
    double e[9];
    ba_damped_inverse(h, lam, a1 > a0, e);
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
        const ba_rec_ref r = rec.of(pt_obs[i]);
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int c = 0; c < 3; c++)
                r[18 + a * 3 + c] = r[a * 3] * e[c] + r[a * 3 + 1] * e[3 + c] + r[a * 3 + 2] * e[6 + c];
    }
}

__global__ __launch_bounds__(BA_THREADS) void ba_point_kernel(const int* __restrict__ pt_ptr,
                                                              const int* __restrict__ pt_obs, int L, double lam,
                                                              double* __restrict__ rec, double* __restrict__ E,
                                                              double* __restrict__ bl,
                                                              double* __restrict__ hll_diag /*[L,3] or null*/) {
    ba_point_body(pt_ptr, pt_obs, L, lam, ba_recs{rec}, E, bl, hll_diag, (int)(blockIdx.x * BA_THREADS + threadIdx.x));
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

// one block per pose k, or per slice [begin, end) of its observation list: Hpp (21), bp (6), y = sum Y_o bl (6) and the
// cost (1) over those entries, stored as block `slot` of the four output arrays (the pose itself when the block takes the
// whole list; slice s of nsub, slot k * nsub + s, in the one-launch form, whose consumers add the slices in order)
__device__ __forceinline__ void ba_pose_body(const int* __restrict__ ps_obs, const int* __restrict__ obs_point, const ba_recs rec,
                                             const double* __restrict__ bl, double* __restrict__ Hpp /*[.,21]*/,
                                             double* __restrict__ bp /*[.,6]*/, double* __restrict__ ybl /*[.,6]*/,
                                             double* __restrict__ cost /*[.]*/, double (*sw)[34], double* out, const int begin,
                                             const int end, const int slot) {
    double acc[34];
#pragma unroll
    for (int i = 0; i < 34; i++) acc[i] = 0.0;
    for (int i = begin + threadIdx.x; i < end; i += BA_THREADS) {
        const int o = ps_obs[i];
        const ba_rec_ref r = rec.of(o);
        const double* b = bl + (size_t)obs_point[o] * 3;
#pragma unroll
        for (int t = 0; t < 27; t++) acc[t] += r[36 + t];
#pragma unroll
        for (int a = 0; a < 6; a++) acc[27 + a] += r[18 + a * 3] * b[0] + r[18 + a * 3 + 1] * b[1] + r[18 + a * 3 + 2] * b[2];
        acc[33] += r[72];
    }
    ba_block_sum<34>(acc, sw, out);
    if (threadIdx.x < 21) Hpp[(size_t)slot * 21 + threadIdx.x] = out[threadIdx.x];
    if (threadIdx.x < 6) {
        bp[(size_t)slot * 6 + threadIdx.x] = out[21 + threadIdx.x];
        ybl[(size_t)slot * 6 + threadIdx.x] = out[27 + threadIdx.x];
    }
    if (threadIdx.x == 0) cost[slot] = out[33];
}

__global__ __launch_bounds__(BA_THREADS) void ba_pose_kernel(const int* __restrict__ ps_ptr,
                                                             const int* __restrict__ ps_obs,
                                                             const int* __restrict__ obs_point,
                                                             double* __restrict__ rec,
                                                             const double* __restrict__ bl,
                                                             double* __restrict__ Hpp /*[K,21]*/,
                                                             double* __restrict__ bp /*[K,6]*/,
                                                             double* __restrict__ ybl /*[K,6]*/,
                                                             double* __restrict__ cost /*[K]*/) {
    __shared__ double sw[4][34];
    __shared__ double out[34];
    const int k = (int)blockIdx.x;
    ba_pose_body(ps_obs, obs_point, ba_recs{rec}, bl, Hpp, bp, ybl, cost, sw, out, ps_ptr[k], ps_ptr[k + 1], k);
}

// this thread's share of W[k1,k2] over the entries [begin, end) of k1's observation list, the partner observation of k2
// looked up per point
__device__ __forceinline__ void ba_pair_accumulate(const int* __restrict__ ps_obs, const int* __restrict__ obs_point,
                                                   const int* __restrict__ lookup, int L, const ba_recs rec, int k1, int k2,
                                                   int begin, int end, double (&acc)[36]) {
    for (int i = begin + threadIdx.x; i < end; i += BA_THREADS) {
        const int o1 = ps_obs[i];
        const int o2 = k1 == k2 ? o1 : lookup[(size_t)k2 * L + obs_point[o1]];
        if (o2 < 0) continue;
        const ba_rec_ref y = rec.of(o1);   // [18 + ..]: Y of (k1, l)
        const ba_rec_ref h = rec.of(o2);   // [0 + ..]: Hpl of (k2, l)
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = 0; b < 6; b++)
                acc[a * 6 + b] += y[18 + a * 3] * h[b * 3] + y[18 + a * 3 + 1] * h[b * 3 + 1] + y[18 + a * 3 + 2] * h[b * 3 + 2];
    }
}

// one block per pose pair (k1 <= k2): W[k1,k2] = sum over points seen by both of Y_(k1,l) Hpl_(k2,l)^T (6x6).
// `pair` runs over the K (K + 1) / 2 upper blocks in row-major order (round 2 launched K x K blocks and half returned).
__device__ __forceinline__ void ba_pair_body(const int* __restrict__ ps_ptr, const int* __restrict__ ps_obs,
                                             const int* __restrict__ obs_point,
                                             const int* __restrict__ lookup /*[K,L] obs of (pose, point) or -1*/, int K, int L,
                                             const ba_recs rec, double* __restrict__ W /*[K,K,36]*/,
                                             double (*sw)[36], double* out, int pair) {
    int k1 = 0;
    while (pair >= K - k1) { pair -= K - k1; k1++; }
    const int k2 = k1 + pair;
    double acc[36];
#pragma unroll
    for (int i = 0; i < 36; i++) acc[i] = 0.0;
    ba_pair_accumulate(ps_obs, obs_point, lookup, L, rec, k1, k2, ps_ptr[k1], ps_ptr[k1 + 1], acc);
    ba_block_sum<36>(acc, sw, out);
    if (threadIdx.x < 36) W[((size_t)k1 * K + k2) * 36 + threadIdx.x] = out[threadIdx.x];
}

__global__ __launch_bounds__(BA_THREADS) void ba_pair_kernel(const int* __restrict__ ps_ptr,
                                                             const int* __restrict__ ps_obs,
                                                             const int* __restrict__ obs_point,
                                                             const int* __restrict__ lookup, int K, int L,
                                                             double* __restrict__ rec, double* __restrict__ W) {
    __shared__ double sw[4][36];
    __shared__ double out[36];
    ba_pair_body(ps_ptr, ps_obs, obs_point, lookup, K, L, ba_recs{rec}, W, sw, out, (int)blockIdx.x);
}

// per point: dl = E (-bl - sum_{o of l} Hpl_o^T dp_pose(o))
__device__ __forceinline__ void ba_backsub_body(const int* __restrict__ pt_ptr, const int* __restrict__ pt_obs,
                                                const int* __restrict__ obs_pose, int L, const ba_recs rec,
                                                const double* __restrict__ E, const double* __restrict__ bl,
                                                const double* __restrict__ dp /*[K,6]*/, double* __restrict__ dl /*[L,3]*/,
                                                int block) {
    const int l = block * BA_THREADS + threadIdx.x;
    if (l >= L) return;
    double t[3] = {-bl[(size_t)l * 3], -bl[(size_t)l * 3 + 1], -bl[(size_t)l * 3 + 2]};
    const int a0 = pt_ptr[l], a1 = pt_ptr[l + 1];
    for (int i = a0; i < a1; i++) {
        const int o = pt_obs[i];
        const ba_rec_ref r = rec.of(o);
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

__global__ __launch_bounds__(BA_THREADS) void ba_backsub_kernel(const int* __restrict__ pt_ptr,
                                                                const int* __restrict__ pt_obs,
                                                                const int* __restrict__ obs_pose, int L,
                                                                double* __restrict__ rec,
                                                                const double* __restrict__ E,
                                                                const double* __restrict__ bl,
                                                                const double* __restrict__ dp /*[K,6]*/,
                                                                double* __restrict__ dl /*[L,3]*/) {
    ba_backsub_body(pt_ptr, pt_obs, obs_pose, L, ba_recs{rec}, E, bl, dp, dl, (int)blockIdx.x);
}

// robust cost only (an LM trial needs nothing else from the candidate state): one block per pose
__device__ __forceinline__ void ba_cost_body(const double* __restrict__ poses, const double* __restrict__ points,
                                             const int* __restrict__ obs_point, const double2* __restrict__ meas,
                                             const int* __restrict__ ps_ptr, const int* __restrict__ ps_obs, ba_cam cam,
                                             double delta, double* __restrict__ cost /*[K]*/, double (*sw)[1], double* out,
                                             const int k) {
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

__global__ __launch_bounds__(BA_THREADS) void ba_cost_kernel(const double* __restrict__ poses,
                                                             const double* __restrict__ points,
                                                             const int* __restrict__ obs_point,
                                                             const double2* __restrict__ meas,
                                                             const int* __restrict__ ps_ptr,
                                                             const int* __restrict__ ps_obs, ba_cam cam, double delta,
                                                             double* __restrict__ cost /*[K]*/) {
    __shared__ double sw[4][1];
    __shared__ double out[1];
    ba_cost_body(poses, points, obs_point, meas, ps_ptr, ps_obs, cam, delta, cost, sw, out, (int)blockIdx.x);
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
    ba_pair_kernel<<<(unsigned)(K * (K + 1) / 2), BA_THREADS, 0, ctx->stream>>>(d_ps_ptr, d_ps_obs, d_obs_point, d_lookup, (int)K,
                                                                                 (int)L, d_rec, d_W);
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
        d_pt_ptr, d_pt_obs, d_obs_pose, (int)L, (double*)d_rec, d_E, d_bl, d_dp, d_dl);
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}

// =====================================================================================================================
// The whole window LM as ONE persistent launch of a few dozen workgroups (slam_ba_optimize_f64).
//
// Nothing returns to the host between the phases: a grid barrier separates them, workgroup 0 assembles the reduced system
// of the free poses in LDS and factors it, and the Levenberg-Marquardt bookkeeping is replicated in every workgroup
// (same arithmetic on the same values), so a verdict costs no barrier of its own.
//
// Unlike the per-phase kernels above this form keeps NO per-observation records: every phase linearises the observations
// it touches again (ba_linearise: ~150 flops from the pose in LDS, the point and the measurement - 7 doubles read instead
// of up to 46 of a 584-byte record, and no phase that writes them).  Behind a grid barrier every read comes from memory,
// and the record traffic was what the phases cost (~17 ns per observation and compute unit, tools/ba_phase_probe.py); the
// arithmetic is free beside it.  Four barriers per trial:
//   points         four lanes per point: Hll, bl over its observations, E = (Hll + lambda I)^-1
//   camera blocks  a workgroup per slice of a pose's observation list: Hpp, bp, y = sum Y bl, cost; and per slice of a pair
//                  of free poses: W = sum Y_(k1,l) Hpl_(k2,l)^T, both observations linearised on the spot
//   solve          workgroup 0: the reduced system from the slices, L D L^T, the pose steps
//   step           four lanes per point: dl by back-substitution, the candidate point, the candidate's robust cost
//
// The grid barrier is the counter hand-off of cdna_hip_programming.md G16: every wave drains its stores, the
// workgroup's barrier, then ONE lane does an agent-scope release, takes a ticket, and either opens the next generation
// (last arriver) or polls the generation word; an agent-scope acquire and the workgroup's barrier follow.  That is what
// makes plain loads of another workgroup's results valid afterwards (per-CU L1s and per-XCD L2s are not coherent by
// themselves).  It needs every workgroup of the grid to be resident at once: the grid is at most 128 workgroups of 256
// threads on a 256-CU device, far below what fits, and every poll is bounded - a workgroup that waits 2^22 polls raises
// `abort`, which every workgroup checks behind every barrier, so a fault ends the launch instead of hanging the GPU.
// =====================================================================================================================
struct bg_ctl {
    unsigned int arrive; unsigned int pad0[31];   // the grid barrier: the ticket counter and the generation word the waiters poll
    unsigned int gen; unsigned int pad1[31];      // live in cache lines of their own
    int abort;                                    // launch abandoned
};

struct bg_args {
    int K, L, O, iterations, nfree, nblocks, nsub;
    const int* obs_pose; const int* obs_point; const double2* meas;
    const int* pt_ptr; const int* pt_obs; const int* ps_ptr; const int* ps_obs; const int* free_list;
    int* lookup;
    double* T; double* X;                                        // [2][K*12], [2][L*3]
    double* E; double* bl;                                       // [L*9], [L*3]
    // per pose and slice (slot k * nsub + s): [.*21] [.*6] [.*6] [.]; per pair of free poses and slice: [.*36]; bpc [K*6]:
    // the slices of bp added up (workgroup 0, with the solve), for the gain ratio's denominator
    double* Hpp; double* bp; double* ybl; double* costk; double* bpc; double* W;
    double* dp; double* dl; double* part;                        // [K*6 + 1] (the steps, then the factorisation's outcome), [L*3], [nblocks][2] (gain-ratio denominator, candidate cost)
    bg_ctl* ctl;
    double* stats;
    unsigned int* index_errors;
    ba_cam cam; double delta;
};

#define BG_MAX_BLOCKS 128
#define BG_MAX_SLICES 8
#define BG_MAX_PAIRS (SLAM_BA_LM_MAX_FREE * (SLAM_BA_LM_MAX_FREE + 1) / 2)
#define BG_TRI(i, j) ((i) * ((i) + 1) / 2 + (j))   // packed lower triangle, j <= i
#define BG_MAXN (SLAM_BA_LM_MAX_FREE * 6)

template <typename T>
__device__ __forceinline__ T bg_load(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Why a launch ends early (bg_ctl::abort, d_stats[5]): 1 = a grid barrier was abandoned (the workgroups did not all become
// resident in time: the device is busy with somebody else's work - a RESOURCE condition, SLAM_ERR_BUSY), 2 = an index
// outside the window or a malformed index table (caller error, SLAM_ERR_INVALID).  The larger code wins.
#define BG_ABORT_BUSY 1
#define BG_ABORT_INDEX 2
// A barrier gives up after BG_BARRIER_TICKS of the 100 MHz wall clock (50 ms; a whole window adjustment at the reference's
// size takes 0.3 ms) - bounded by TIME, not by a poll count whose duration depends on what else the memory system is doing
// (the 2^22 polls of round 3 measured 0.56-0.67 s on an idle chip, 133-160 ns each, and more under load:
// profiles/r04_ba_busy.log) - with a poll cap behind it.
#define BG_BARRIER_TICKS 5000000ull
__device__ __forceinline__ void bg_abort(bg_ctl* c, int why) { atomicMax(&c->abort, why); }

// returns false when the launch is being abandoned
__device__ __forceinline__ bool bg_grid_sync(bg_ctl* c, unsigned int nblocks, unsigned int& gen) {
    __syncthreads();                                         // every wave's stores are issued and drained
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned int ticket = __hip_atomic_fetch_add(&c->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket == nblocks - 1) {
            __hip_atomic_store(&c->arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&c->gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned int polls = 0;
            const unsigned long long t0 = wall_clock64();
            while (__hip_atomic_load(&c->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                ++polls;
                if ((polls & 63) == 0) {
                    if (__hip_atomic_load(&c->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                    if (wall_clock64() - t0 > BG_BARRIER_TICKS || polls > (1u << 26)) {
                        bg_abort(c, BG_ABORT_BUSY);
                        break;
                    }
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    gen++;
    __syncthreads();
    return bg_load(&c->abort) == 0;
}

// Hpl = w Jp^T Jq (6x3) of a linearised observation
__device__ __forceinline__ void bg_hpl(const ba_lin& q, double (&h)[6][3]) {
#pragma unroll
    for (int x = 0; x < 6; x++)
#pragma unroll
        for (int c = 0; c < 3; c++) h[x][c] = q.w * (q.jp[0][x] * q.jq[0][c] + q.jp[1][x] * q.jq[1][c]);
}

// slice [begin, end) of pose k's observation list: Hpp (21), bp (6), y = sum Y_o bl (6), cost (1), stored as block `slot`
// (what ba_obs_body + ba_pose_body do through the records)
__device__ __forceinline__ void bg_pose(const bg_args& a, const double* __restrict__ X, const double* sT, int k, int begin, int end, int slot,
                                        const bool with_y, double (*sw)[34], double* out) {
    double acc[34];
#pragma unroll
    for (int i = 0; i < 34; i++) acc[i] = 0.0;
    // Four strides of the list at a time, level by level: entry -> observation -> (point, measurement) -> (point, E, bl).
    // With a few dozen entries per thread the phase is a chain of dependent memory latencies (every read behind the grid
    // barrier comes from memory), and walking one entry at a time pays the chain once per entry.  Indices are clamped into
    // the slice, the surplus is masked where it is added; the sums take their terms in the list's order either way.
    for (int base = begin + threadIdx.x; base < end; base += 4 * BA_THREADS) {
        int o[4], l[4];
        double2 m[4];
#pragma unroll
        for (int u = 0; u < 4; u++) o[u] = a.ps_obs[min(base + u * BA_THREADS, end - 1)];
#pragma unroll
        for (int u = 0; u < 4; u++) { l[u] = a.obs_point[o[u]]; m[u] = a.meas[o[u]]; }
        double x[4][3], e[4][9], b[4][3];
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int c = 0; c < 3; c++) x[u][c] = X[(size_t)l[u] * 3 + c];
            if (with_y) {                       // (the launch's first pass runs beside the point phase: E and bl are not there yet)
#pragma unroll
                for (int c = 0; c < 3; c++) b[u][c] = a.bl[(size_t)l[u] * 3 + c];
#pragma unroll
                for (int c = 0; c < 9; c++) e[u][c] = a.E[(size_t)l[u] * 9 + c];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (base + u * BA_THREADS >= end) break;
            ba_lin q;
            ba_linearise(sT + (size_t)k * 12, x[u], m[u], a.cam, a.delta, q);
            int t = 0;
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
                for (int c = r; c < 6; c++) acc[t++] += q.w * (q.jp[0][r] * q.jp[0][c] + q.jp[1][r] * q.jp[1][c]);
#pragma unroll
            for (int r = 0; r < 6; r++) acc[21 + r] += q.w * (q.jp[0][r] * q.e0 + q.jp[1][r] * q.e1);
            if (with_y) {
                double h[6][3];
                bg_hpl(q, h);
#pragma unroll
                for (int r = 0; r < 6; r++) {
                    const double y0 = h[r][0] * e[u][0] + h[r][1] * e[u][3] + h[r][2] * e[u][6], y1 = h[r][0] * e[u][1] + h[r][1] * e[u][4] + h[r][2] * e[u][7],
                                 y2 = h[r][0] * e[u][2] + h[r][1] * e[u][5] + h[r][2] * e[u][8];
                    acc[27 + r] += y0 * b[u][0] + y1 * b[u][1] + y2 * b[u][2];
                }
            }
            acc[33] += q.rho;
        }
    }
    ba_block_sum<34>(acc, sw, out);
    if (threadIdx.x < 21) a.Hpp[(size_t)slot * 21 + threadIdx.x] = out[threadIdx.x];
    if (threadIdx.x < 6) {
        a.bp[(size_t)slot * 6 + threadIdx.x] = out[21 + threadIdx.x];
        a.ybl[(size_t)slot * 6 + threadIdx.x] = out[27 + threadIdx.x];
    }
    if (threadIdx.x == 0) a.costk[slot] = out[33];
}

// slice [begin, end) of k1's list towards the W block of one pair of poses (k1 <= k2), see ba_pair_body; stored as block `slot`
__device__ __forceinline__ void bg_pair(const bg_args& a, const double* __restrict__ X, const double* sT, int k1, int k2, int begin, int end,
                                        int slot, double (*sw)[36], double* out) {
    double acc[36];
#pragma unroll
    for (int i = 0; i < 36; i++) acc[i] = 0.0;
    // four strides at a time, level by level (see bg_pose): entry -> observation -> (point, measurement) -> (partner, point, E)
    // -> the partner's measurement
    for (int base = begin + threadIdx.x; base < end; base += 4 * BA_THREADS) {
        int o1[4], l[4], o2[4];
        double2 m1[4], m2[4];
#pragma unroll
        for (int u = 0; u < 4; u++) o1[u] = a.ps_obs[min(base + u * BA_THREADS, end - 1)];
#pragma unroll
        for (int u = 0; u < 4; u++) { l[u] = a.obs_point[o1[u]]; m1[u] = a.meas[o1[u]]; }
        double x[4][3], e[4][9];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            o2[u] = k1 == k2 ? o1[u] : a.lookup[(size_t)k2 * a.L + l[u]];
#pragma unroll
            for (int c = 0; c < 3; c++) x[u][c] = X[(size_t)l[u] * 3 + c];
#pragma unroll
            for (int c = 0; c < 9; c++) e[u][c] = a.E[(size_t)l[u] * 9 + c];
        }
        if (k1 != k2) {
#pragma unroll
            for (int u = 0; u < 4; u++) m2[u] = a.meas[max(o2[u], 0)];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (base + u * BA_THREADS >= end) break;
            if (o2[u] < 0) continue;
            double y[6][3], h2[6][3];
            {
                ba_lin q;
                ba_linearise(sT + (size_t)k1 * 12, x[u], m1[u], a.cam, a.delta, q);
                double h1[6][3];
                bg_hpl(q, h1);
#pragma unroll
                for (int r = 0; r < 6; r++)
#pragma unroll
                    for (int c = 0; c < 3; c++) y[r][c] = h1[r][0] * e[u][c] + h1[r][1] * e[u][3 + c] + h1[r][2] * e[u][6 + c];
                if (k1 == k2) {
#pragma unroll
                    for (int r = 0; r < 6; r++)
#pragma unroll
                        for (int c = 0; c < 3; c++) h2[r][c] = h1[r][c];
                }
            }
            if (k1 != k2) {
                ba_lin q;
                ba_linearise(sT + (size_t)k2 * 12, x[u], m2[u], a.cam, a.delta, q);
                bg_hpl(q, h2);
            }
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
                for (int b = 0; b < 6; b++) acc[r * 6 + b] += y[r][0] * h2[b][0] + y[r][1] * h2[b][1] + y[r][2] * h2[b][2];
        }
    }
    ba_block_sum<36>(acc, sw, out);
    if (threadIdx.x < 36) a.W[(size_t)slot * 36 + threadIdx.x] = out[threadIdx.x];
}

// sum of the nsub slices of one entry, in slice order (stride = entries per slice block)
__device__ __forceinline__ double bg_slices(const double* __restrict__ p, int nsub, int stride) {
    // (every slice's load is issued before the first is added: one memory latency for the entry, not one per slice; nsub is
    // the same in every thread, so the two loops end on scalar branches)
    double q[BG_MAX_SLICES];
#pragma unroll
    for (int s = 0; s < BG_MAX_SLICES; s++) {
        if (s >= nsub) break;
        q[s] = p[(size_t)s * stride];
    }
    double v = q[0];
#pragma unroll
    for (int s = 1; s < BG_MAX_SLICES; s++) {
        if (s >= nsub) break;
        v += q[s];
    }
    return v;
}

// one value of wave-uniform lane `src` to every lane (two v_readlane_b32, no LDS round trip)
__device__ __forceinline__ double bg_lane_bcast(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// 1 / d for a pivot: v_rcp_f64 (about 24 good bits) and two Newton steps, five dependent instructions where the IEEE
// division sequence has about thirty - the 6 n pivots of a factorisation are one dependent chain, and at n = 30 the
// divisions were a third of it.  The result is within an ulp or two of 1 / d; the factors feed an iterative LM step.
__device__ __forceinline__ double bg_reciprocal(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    return fma(fma(-d, r, 1.0), r, r);
}

// Workgroup 0's dense solve of the reduced system: S (LDS, lower triangle packed by rows, n = 6 * free poses <= 96) and
// rhs (LDS) in, the solution in rhs out; false when S is not positive definite or the solution is not finite.
//
// L D L^T in place, right-looking, in PANELS of six columns (one pose block), two workgroup barriers per panel:
//   (a) every thread factors the panel's 6 x 6 diagonal block for itself in registers (the same 21 LDS words, the same
//       arithmetic: no barrier to hand it round), the thread that owns row r below the block solves that row's six L
//       entries against it, stores them over S[r][panel] and keeps t = L D of the row in Tp (the unscaled column values
//       the trailing update multiplies by);
//   (b) the trailing triangle takes S[i][k] -= sum_c L[i][c] t[k][c] on a 16 x 16 thread tile, each thread holding the t
//       rows of its (up to six) columns in registers.
// The operations on every element, and their order, are those of the column-by-column form this replaces (one column and
// two barriers per step: 15 us at n = 30, 81 us at n = 84, measured with tools/ba_phase_probe.py), so the factors are
// the same bits.  The two substitutions run in wave 0 with the right-hand side in registers (lane i holds rows i and
// i + 64) and one v_readlane pair per step instead of an LDS write + read.
__device__ __forceinline__ bool bg_factor_solve(double* __restrict__ S, double* __restrict__ rhs, double* __restrict__ Tp, const int n) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ty = tid >> 4, tx = tid & 15;
    bool spd = true;
    for (int c0 = 0; c0 < n; c0 += 6) {
        // (a) the diagonal block: D[a][b], b <= a, becomes (unscaled column values | pivots); inv[] = 1 / pivot
        double D[6][6], inv[6];
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = 0; b <= a; b++) D[a][b] = S[BG_TRI(c0 + a, c0 + b)];
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const double d = D[j][j];
            spd = spd && d > 0.0 && isfinite(d);
            inv[j] = bg_reciprocal(d);
#pragma unroll
            for (int i = j + 1; i < 6; i++) {
                const double lij = D[i][j] * inv[j];
#pragma unroll
                for (int k = j + 1; k <= i; k++) D[i][k] -= lij * D[k][j];
            }
        }
        const int r = c0 + tid;
        if (r < n && tid >= 6) {
            {
                double A[6];
#pragma unroll
                for (int b = 0; b < 6; b++) A[b] = S[BG_TRI(r, c0 + b)];
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    const double l = A[c] * inv[c];
                    Tp[r * 6 + c] = A[c];
                    S[BG_TRI(r, c0 + c)] = l;
#pragma unroll
                    for (int b = c + 1; b < 6; b++) A[b] -= l * D[b][c];
                }
            }
        }
        __syncthreads();
        // the block's own rows: the pivot, and what lies left of it scaled (only now: every thread has read the block above)
#pragma unroll
        for (int a = 1; a < 6; a++)
            if (tid == a) {
#pragma unroll
                for (int b = 0; b < a; b++) S[BG_TRI(c0 + a, c0 + b)] = D[a][b] * inv[b];
                S[BG_TRI(c0 + a, c0 + a)] = D[a][a];
            }
        // (b) the trailing triangle, rows and columns from c0 + 6
        const int m0 = c0 + 6;
        if (m0 < n) {
            // (the loop bounds below are wave-uniform - a tile row or column that starts beyond the matrix ends the loop for
            // every thread alike, and a column tile right of the row tile holds no element of the lower triangle - so a small
            // system skips the tiles it does not have instead of evaluating 36 empty predicates)
            double tk[6][6];
#pragma unroll
            for (int b = 0; b < 6; b++) {
                if (m0 + 16 * b >= n) break;
                const int k = min(m0 + tx + 16 * b, n - 1);
#pragma unroll
                for (int c = 0; c < 6; c++) tk[b][c] = Tp[k * 6 + c];
            }
#pragma unroll
            for (int a = 0; a < 6; a++) {
                if (m0 + 16 * a >= n) break;
                const int i = m0 + ty + 16 * a;
                if (i < n) {
                    double l[6];
#pragma unroll
                    for (int c = 0; c < 6; c++) l[c] = S[BG_TRI(i, c0 + c)];
#pragma unroll
                    for (int b = 0; b <= a; b++) {
                        const int k = m0 + tx + 16 * b;
                        if (k <= i) {
                            double v = S[BG_TRI(i, k)];
#pragma unroll
                            for (int c = 0; c < 6; c++) v -= l[c] * tk[b][c];
                            S[BG_TRI(i, k)] = v;
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    bool ok = true;
    if (wave == 0) {
        const int i0 = lane, i1 = lane + 64;
        double r0 = i0 < n ? rhs[i0] : 0.0, r1 = i1 < n ? rhs[i1] : 0.0;
        // (every lane loads from a clamped, always valid address and masks the value: twelve predicated LDS reads per panel,
        // each in an exec-mask region of its own, were most of a step's time; likewise the row that holds step j is picked
        // with a select, not a branch)
        const int c0i = min(i0, n - 1), c1i = min(i1, n - 1);
        for (int j0 = 0; j0 < n; j0 += 6) {                   // forward: L y = rhs
            double s0[6], s1[6];
#pragma unroll
            for (int c = 0; c < 6; c++) {
                const int j = j0 + c;
                const double v0 = S[BG_TRI(c0i, min(j, c0i))], v1 = S[BG_TRI(c1i, min(j, c1i))];
                s0[c] = (i0 > j && i0 < n) ? v0 : 0.0;
                s1[c] = (i1 > j && i1 < n) ? v1 : 0.0;
            }
#pragma unroll
            for (int c = 0; c < 6; c++) {
                const int j = j0 + c;
                const double yj = bg_lane_bcast(j < 64 ? r0 : r1, j & 63);
                r0 -= s0[c] * yj;
                r1 -= s1[c] * yj;
            }
        }
        if (i0 < n) r0 /= S[BG_TRI(i0, i0)];
        if (i1 < n) r1 /= S[BG_TRI(i1, i1)];
        for (int j0 = n - 6; j0 >= 0; j0 -= 6) {              // backward: L^T x = y
            double s0[6], s1[6];
#pragma unroll
            for (int c = 0; c < 6; c++) {
                const int j = j0 + c;
                const double v0 = S[BG_TRI(j, min(i0, j))], v1 = S[BG_TRI(j, min(i1, j))];
                s0[c] = i0 < j ? v0 : 0.0;
                s1[c] = i1 < j ? v1 : 0.0;
            }
#pragma unroll
            for (int c = 5; c >= 0; c--) {
                const int j = j0 + c;
                const double xj = bg_lane_bcast(j < 64 ? r0 : r1, j & 63);
                r0 -= s0[c] * xj;
                r1 -= s1[c] * xj;
            }
        }
        const bool finite = (i0 >= n || isfinite(r0)) && (i1 >= n || isfinite(r1));
        ok = __ballot(!(spd && finite)) == 0ull;
        if (i0 < n) rhs[i0] = r0;
        if (i1 < n) rhs[i1] = r1;
    }
    return ok;     // meaningful in wave 0 (thread 0 publishes it)
}

// exp([w, v]) * T for a 3x4 row-major pose (rotation first): the update the Jacobian of frontend.py:288-291 is the derivative for
__device__ void bg_apply_update(const double* dx, const double* T, double* Tn) {
    const double wx = dx[0], wy = dx[1], wz = dx[2];
    const double th2 = wx * wx + wy * wy + wz * wz, th = sqrt(th2);
    double sa, sb, sc;  // sin(th)/th, (1-cos)/th^2, (th-sin)/th^3
    if (th < 1e-10) { sa = 1.0; sb = 0.5; sc = 1.0 / 6.0; }
    else {
        double sn, cs;
        sincos(th, &sn, &cs);
        sa = sn / th; sb = (1.0 - cs) / th2; sc = (th - sn) / (th2 * th);
    }
    const double Wm[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double W2[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) W2[i * 3 + j] = Wm[i * 3] * Wm[j] + Wm[i * 3 + 1] * Wm[3 + j] + Wm[i * 3 + 2] * Wm[6 + j];
    double R[9], V[9];
    for (int i = 0; i < 9; i++) {
        const double I = (i % 4 == 0) ? 1.0 : 0.0;
        R[i] = I + sa * Wm[i] + sb * W2[i];
        V[i] = I + sb * Wm[i] + sc * W2[i];
    }
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 4; j++) Tn[i * 4 + j] = R[i * 3] * T[j] + R[i * 3 + 1] * T[4 + j] + R[i * 3 + 2] * T[8 + j];
        Tn[i * 4 + 3] += V[i * 3] * dx[3] + V[i * 3 + 1] * dx[4] + V[i * 3 + 2] * dx[5];
    }
}

// sum over the four lanes of a quad (DPP quad_perm, no LDS): every lane gets (v0 + v1) + (v2 + v3) or its mirror image -
// the same bits, addition being commutative.  All four lanes must be active.
__device__ __forceinline__ double bg_quad_sum(double v) {
    long long b = __double_as_longlong(v);
    int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0xB1, 0xF, 0xF, true);            // quad_perm [1,0,3,2]
    int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0xB1, 0xF, 0xF, true);
    v += __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    b = __double_as_longlong(v);
    lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x4E, 0xF, 0xF, true);                // quad_perm [2,3,0,1]
    hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x4E, 0xF, 0xF, true);
    return v + __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// a grid barrier; an abandoned launch says so in its status word (the other seven stay NaN) and ends
#define BG_SYNC_OR_QUIT()                                   \
    do {                                                    \
        if (!bg_grid_sync(c, G, gen)) {                     \
            if (blk == 0 && tid == 0) a.stats[5] = (double)bg_load(&c->abort);   /* 1 = busy, 2 = bad index / table */ \
            return;                                         \
        }                                                   \
    } while (0)

__global__ __launch_bounds__(BA_THREADS) void ba_lm_grid_kernel(const bg_args a) {
    __shared__ double S[BG_TRI(BG_MAXN, 0)];       // block 0: the reduced system, lower triangle packed by rows (37 KiB at 96 x 96)
    __shared__ double rhs[BG_MAXN];
    __shared__ double Tp[BG_MAXN * 6];              // block 0: L D of the panel being eliminated
    __shared__ double sw[4][36];
    __shared__ double out[36];
    __shared__ int s_solved;
    __shared__ int s_ps_ptr[65], s_free[SLAM_BA_LM_MAX_FREE];   // the poses' list heads and the free poses, read once
    __shared__ double sT[64 * 12];                 // the poses of the state, per workgroup (K <= 64)
    __shared__ double sTn[64 * 12];                // the candidate poses
    __shared__ double sdp[64 * 6];                 // and the pose steps they come from
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, blk = (int)blockIdx.x, G = a.nblocks;
    const int K = a.K, L = a.L, O = a.O, nf = a.nfree, n = 6 * nf, nsub = a.nsub;
    // the one-thread-per-item phases give every workgroup an equal share of the items (a phase is bound by what ONE
    // compute unit can pull from memory behind the barrier, so it is spread over all of them, not packed 256 to a workgroup)
    const int perL = (L + G - 1) / G;
    bg_ctl* c = a.ctl;
    unsigned int gen = 0;
    // The Levenberg-Marquardt state is REPLICATED: every thread of every workgroup holds it and updates it with the same
    // arithmetic on the same device-memory values (read behind a grid barrier), so no verdict has to be published and
    // waited for - only the factorisation's outcome (workgroup 0's) travels, with the pose steps it produced.
    int cur = 0, accepted = 0, trials = 0, iter = 0, trial = 0, done = 0;
    double lambda = -1.0, ni = 2.0, cost = 0.0;      // lambda < 0: "not measured yet"

    // ---- phase 0: the (pose, point) -> observation table, on the device; an index outside the window is counted
    //      (slam_index_errors) and ends the launch: every later phase trusts the two index arrays -------------------------------
    for (long long i = (long long)blk * BA_THREADS + tid; i < (long long)K * L; i += (long long)G * BA_THREADS) a.lookup[i] = -1;
    BG_SYNC_OR_QUIT();
    for (int o = blk * BA_THREADS + tid; o < O; o += G * BA_THREADS) {
        const int k = a.obs_pose[o], l = a.obs_point[o];
        if ((unsigned)k >= (unsigned)K || (unsigned)l >= (unsigned)L) {
            atomicAdd(a.index_errors, 1u);
            bg_abort(c, BG_ABORT_INDEX);
        } else {
            a.lookup[(size_t)k * L + l] = o;
        }
    }
    // the small tables that index LDS arrays and the per-pose blocks: the free list (ascending, below K) and the poses' list
    // heads (0 = ps_ptr[0] <= ... <= ps_ptr[K] = O); a device-array caller's mistake there would be an out-of-bounds access
    // (ADVICE r03).  The per-point heads pt_ptr / the two observation lists stay trusted: O + L entries, and
    // slam_ba_optimize_host_f64 builds them itself.
    if (blk == 0) {
        for (int i = tid; i < nf; i += BA_THREADS) {
            const int f = a.free_list[i];
            if ((unsigned)f >= (unsigned)K || (i > 0 && a.free_list[i - 1] >= f)) { atomicAdd(a.index_errors, 1u); bg_abort(c, BG_ABORT_INDEX); }
        }
        for (int i = tid; i <= K; i += BA_THREADS) {
            const int v = a.ps_ptr[i];
            const bool bad = v < 0 || v > O || (i == 0 && v != 0) || (i == K && v != O) || (i > 0 && a.ps_ptr[i - 1] > v);
            if (bad) { atomicAdd(a.index_errors, 1u); bg_abort(c, BG_ABORT_INDEX); }
        }
    }
    BG_SYNC_OR_QUIT();

    // The small tables every workgroup reads in every pass are read once, here.  The poses of the state live in LDS from
    // here on: every workgroup computes the candidate poses anyway and takes them over when a step is accepted.
    for (int i = tid; i <= K; i += BA_THREADS) s_ps_ptr[i] = a.ps_ptr[i];
    for (int i = tid; i < nf; i += BA_THREADS) s_free[i] = a.free_list[i];
    for (int i = tid; i < K * 12; i += BA_THREADS) sT[i] = a.T[i];
    __syncthreads();

    while (!done) {
        const double lam = lambda < 0.0 ? 1.0 : lambda;             // the very first pass only measures the diagonal
        const double* X = a.X + (size_t)cur * L * 3;
        double* Tn = a.T + (size_t)(1 - cur) * K * 12;
        double* Xn = a.X + (size_t)(1 - cur) * L * 3;

        // ---- points: Hll, bl over the point's observations, E = (Hll + lambda I)^-1 ---------------------------------------------
        // (FOUR lanes per point, each taking every fourth observation of the point's row, their partial sums added across
        // the quad: a thread's observations are a serial chain - loads, then ~40 dependent f64 operations each - and a
        // point seen by all the keyframes made that chain the phase)
        double hmax = 0.0;                                           // the largest diagonal entry of this thread's points' Hll
        for (int qd = tid; qd < 4 * perL; qd += BA_THREADS) {
            const int l = blk * perL + (qd >> 2), sub = qd & 3;
            if (l >= min(L, (blk + 1) * perL)) continue;                 // (a whole quad at a time)
            const int a0 = a.pt_ptr[l], a1 = a.pt_ptr[l + 1];
            double x[3], h[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
#pragma unroll
            for (int cI = 0; cI < 3; cI++) x[cI] = X[(size_t)l * 3 + cI];
            for (int i = a0 + sub; i < a1; i += 4) {
                const int o = a.pt_obs[i];
                ba_lin q;
                ba_linearise(sT + (size_t)a.obs_pose[o] * 12, x, a.meas[o], a.cam, a.delta, q);
                int t = 0;
#pragma unroll
                for (int r = 0; r < 3; r++)
#pragma unroll
                    for (int cI = r; cI < 3; cI++) h[t++] += q.w * (q.jq[0][r] * q.jq[0][cI] + q.jq[1][r] * q.jq[1][cI]);
#pragma unroll
                for (int r = 0; r < 3; r++) b[r] += q.w * (q.jq[0][r] * q.e0 + q.jq[1][r] * q.e1);
            }
#pragma unroll
            for (int i = 0; i < 6; i++) h[i] = bg_quad_sum(h[i]);
#pragma unroll
            for (int i = 0; i < 3; i++) b[i] = bg_quad_sum(b[i]);
            double e[9];
            ba_damped_inverse(h, lam, a1 > a0, e);
            if (sub == 0) {
#pragma unroll
                for (int i = 0; i < 9; i++) a.E[(size_t)l * 9 + i] = e[i];
#pragma unroll
                for (int i = 0; i < 3; i++) a.bl[(size_t)l * 3 + i] = b[i];
                hmax = fmax(hmax, fmax(h[0], fmax(h[3], h[5])));
            }
        }
        if (lambda < 0.0) {                                          // the first pass measures the diagonal: this workgroup's share
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) hmax = fmax(hmax, __shfl_xor(hmax, off, 64));
            __syncthreads();
            if (lane == 0) sw[wave][0] = hmax;
            __syncthreads();
            if (tid == 0) a.part[blk] = fmax(fmax(sw[0][0], sw[1][0]), fmax(sw[2][0], sw[3][0]));
            __syncthreads();                                         // (sw is used again at once)
        } else {
            BG_SYNC_OR_QUIT();                                       // the first pass's pose tasks need nothing from the points
        }
        // ---- per pose: Hpp, bp, y, cost; per pair of free poses: W.  A task (a pose's, or a pair's first pose's, observation
        //      list) is cut into nsub slices, one workgroup each; the consumers add the slices up in slice order.  The very
        //      first pass only measures the diagonal and the cost: it runs the pose tasks alone --------------------------------
        {
            const int npair = nf * (nf + 1) / 2, ntask = lambda < 0.0 ? K : K + npair;
            for (int t = blk; t < ntask * nsub; t += G) {
                const int task = t % ntask, sl = t / ntask;
                int k1, k2 = -1;
                if (task < K) k1 = task;
                else {
                    int pi = task - K, f1 = 0;
                    while (pi >= nf - f1) { pi -= nf - f1; f1++; }
                    k1 = s_free[f1]; k2 = s_free[f1 + pi];
                }
                const int p0 = s_ps_ptr[k1], p1 = s_ps_ptr[k1 + 1], len = (p1 - p0 + nsub - 1) / nsub;
                const int begin = min(p1, p0 + sl * len), end = min(p1, begin + len);
                if (task < K) bg_pose(a, X, sT, k1, begin, end, sl * K + k1, lambda >= 0.0, (double(*)[34])sw, out);
                else bg_pair(a, X, sT, k1, k2, begin, end, sl * npair + (task - K), sw, out);
                __syncthreads();
            }
        }
        BG_SYNC_OR_QUIT();

        if (lambda < 0.0) {
            // initial damping: tau * the largest diagonal entry of the free poses' Hpp and of every Hll (every workgroup
            // computes it for itself: a maximum does not depend on the order)
            // (the points' share comes as one maximum per workgroup from the point phase, and the poses' costs are added by a
            // block-wide reduction: a loop over 3 L diagonal entries, or over the K costs, in every thread is a chain of
            // dependent loads at the head of the launch)
            double v = 0.0;
            for (int i = tid; i < G; i += BA_THREADS) v = fmax(v, a.part[i]);
            for (int i = tid; i < nf * 6; i += BA_THREADS) {
                const int d = i % 6;
                v = fmax(v, bg_slices(a.Hpp + (size_t)s_free[i / 6] * 21 + d * 6 - d * (d - 1) / 2, nsub, K * 21));   // diagonal entry d of the packed upper triangle
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
            __syncthreads();
            if (lane == 0) sw[wave][0] = v;
            __syncthreads();
            lambda = 1e-5 * fmax(fmax(fmax(sw[0][0], sw[1][0]), fmax(sw[2][0], sw[3][0])), 1e-12);
            double pc[1] = {0.0};
            for (int k = tid; k < K; k += BA_THREADS) pc[0] += bg_slices(a.costk + k, nsub, K);
            ba_block_sum<1>(pc, (double(*)[1])sw, out);
            cost = out[0];
            if (blk == 0 && tid == 0) a.stats[0] = cost;
            if (a.iterations <= 0 || nf == 0) done = 1;
            __syncthreads();
            continue;                                  // points / blocks again, now with the real lambda
        }

        // ---- workgroup 0: assemble the reduced system of the free poses in LDS and solve it -----------------------------------
        if (blk == 0) {
            const int npair = nf * (nf + 1) / 2;
            for (int idx = tid; idx < BG_TRI(n, 0); idx += BA_THREADS) {     // S is symmetric: the lower triangle is kept, packed by rows
                int i = (int)((sqrtf(8.0f * (float)idx + 1.0f) - 1.0f) * 0.5f);
                while (BG_TRI(i, 0) > idx) i--;
                while (BG_TRI(i + 1, 0) <= idx) i++;
                const int j = idx - BG_TRI(i, 0), fa = i / 6, fb = j / 6, ii = i % 6, jj = j % 6;
                const int pidx = fb * nf - fb * (fb - 1) / 2 + (fa - fb);          // the pair (fb <= fa) in the order the tasks run
                double v;
                if (fa == fb) {
                    const int tri = jj * 6 - jj * (jj - 1) / 2 + (ii - jj);        // jj <= ii: index in the packed upper triangle of Hpp
                    v = bg_slices(a.Hpp + (size_t)s_free[fa] * 21 + tri, nsub, K * 21) + (ii == jj ? lam : 0.0) -
                        bg_slices(a.W + (size_t)pidx * 36 + ii * 6 + jj, nsub, npair * 36);
                } else {
                    v = -bg_slices(a.W + (size_t)pidx * 36 + jj * 6 + ii, nsub, npair * 36);   // fa > fb: the transpose of W[kb, ka]
                }
                S[idx] = v;
            }
            for (int i = tid; i < K * 6; i += BA_THREADS) a.bpc[i] = bg_slices(a.bp + i, nsub, K * 6);
            for (int i = tid; i < n; i += BA_THREADS) {
                const size_t e = (size_t)s_free[i / 6] * 6 + i % 6;
                rhs[i] = -bg_slices(a.bp + e, nsub, K * 6) + bg_slices(a.ybl + e, nsub, K * 6);
            }
            __syncthreads();
            const bool ok = bg_factor_solve(S, rhs, Tp, n);
            if (tid == 0) s_solved = ok ? 1 : 0;
            __syncthreads();
            // the steps of all K poses (zero for the fixed ones), then the factorisation's outcome
            for (int i = tid; i < K * 6; i += BA_THREADS) a.dp[i] = 0.0;
            __syncthreads();
            if (s_solved)
                for (int i = tid; i < n; i += BA_THREADS) a.dp[(size_t)s_free[i / 6] * 6 + i % 6] = rhs[i];
            if (tid == 0) a.dp[K * 6] = (double)s_solved;
        }
        BG_SYNC_OR_QUIT();
        for (int i = tid; i <= K * 6; i += BA_THREADS) {
            const double v = a.dp[i];
            if (i < K * 6) sdp[i] = v;
            else s_solved = v != 0.0;
        }
        __syncthreads();
        if (!s_solved) {                               // a factorisation that fails counts as a trial
            lambda = lam * ni; ni *= 2.0; trials++; trial++;
            if (trial >= 10 || !isfinite(lambda)) done = 1;
            continue;
        }

        // ---- the step and its verdict's ingredients, four lanes per point: dl by back-substitution, the candidate point, its
        //      share of the gain ratio's denominator, and the candidate's robust cost over the point's observations (every
        //      workgroup keeps the K candidate poses in LDS; workgroup 0 also stores them) --------------------------------------
        {
            for (int k = tid; k < K; k += BA_THREADS) {
                bg_apply_update(sdp + (size_t)k * 6, sT + (size_t)k * 12, sTn + (size_t)k * 12);
                if (blk == 0)
                    for (int x = 0; x < 12; x++) Tn[(size_t)k * 12 + x] = sTn[(size_t)k * 12 + x];
            }
            __syncthreads();
            double sc = 0.0, cc = 0.0;
            for (int qd = tid; qd < 4 * perL; qd += BA_THREADS) {    // four lanes per point, as in the point phase
                const int l = blk * perL + (qd >> 2), sub = qd & 3;
                if (l >= min(L, (blk + 1) * perL)) continue;
                // dl = E (-bl - sum_{o of l} Hpl_o^T dp_pose(o)) as ba_backsub_body has it, Hpl linearised again, the pose steps from LDS
                const int a0 = a.pt_ptr[l], a1 = a.pt_ptr[l + 1];
                double t[3] = {0.0, 0.0, 0.0}, bl3[3], x0[3];
#pragma unroll
                for (int x = 0; x < 3; x++) { bl3[x] = a.bl[(size_t)l * 3 + x]; x0[x] = X[(size_t)l * 3 + x]; }
                for (int i = a0 + sub; i < a1; i += 4) {
                    const int o = a.pt_obs[i], k = a.obs_pose[o];
                    ba_lin q;
                    ba_linearise(sT + (size_t)k * 12, x0, a.meas[o], a.cam, a.delta, q);
                    double h[6][3];
                    bg_hpl(q, h);
                    const double* d = sdp + (size_t)k * 6;
#pragma unroll
                    for (int cI = 0; cI < 3; cI++)
#pragma unroll
                        for (int aI = 0; aI < 6; aI++) t[cI] -= h[aI][cI] * d[aI];
                }
#pragma unroll
                for (int x = 0; x < 3; x++) t[x] = bg_quad_sum(t[x]) - bl3[x];
                const double* e = a.E + (size_t)l * 9;
                double p[3];
#pragma unroll
                for (int x = 0; x < 3; x++) {
                    const double d = a1 > a0 ? e[x * 3] * t[0] + e[x * 3 + 1] * t[1] + e[x * 3 + 2] * t[2] : 0.0;
                    p[x] = x0[x] + d;
                    if (sub == 0) {
                        a.dl[(size_t)l * 3 + x] = d;
                        Xn[(size_t)l * 3 + x] = p[x];
                        sc += d * (lam * d - bl3[x]);
                    }
                }
                for (int i = a0 + sub; i < a1; i += 4) {
                    const int o = a.pt_obs[i];
                    const double* P = sTn + (size_t)a.obs_pose[o] * 12;
                    const double Xc = P[0] * p[0] + P[1] * p[1] + P[2] * p[2] + P[3];
                    const double Yc = P[4] * p[0] + P[5] * p[1] + P[6] * p[2] + P[7];
                    const double Zc = P[8] * p[0] + P[9] * p[1] + P[10] * p[2] + P[11];
                    const double2 m = a.meas[o];
                    const double e0 = m.x - (a.cam.fx * Xc + a.cam.cx * Zc) / Zc;
                    const double e1 = m.y - (a.cam.fy * Yc + a.cam.cy * Zc) / Zc;
                    const double c2 = e0 * e0 + e1 * e1;
                    double rho = c2;
                    if (a.delta > 0.0) {
                        const double en = sqrt(c2);
                        if (en > a.delta) rho = 2.0 * a.delta * en - a.delta * a.delta;
                    }
                    cc += rho;
                }
            }
            if (blk == 0)
                for (int k = tid; k < K; k += BA_THREADS)
                    for (int x = 0; x < 6; x++) sc += sdp[(size_t)k * 6 + x] * (lam * sdp[(size_t)k * 6 + x] - a.bpc[(size_t)k * 6 + x]);
            double acc[2] = {sc, cc};
            ba_block_sum<2>(acc, (double(*)[2])sw, out);
            if (tid == 0) { a.part[2 * blk] = out[0]; a.part[2 * blk + 1] = out[1]; }
        }
        BG_SYNC_OR_QUIT();
        // ---- verdict (replicated) ------------------------------------------------------------------------------------------------------
        {
            // (the workgroups' partial sums are added by a block-wide reduction, the same one in every workgroup: a loop over
            // them in every thread is a chain of G dependent loads and additions - 9 us of a trial at 121 workgroups)
            double pr[2] = {0.0, 0.0};
            for (int b = tid; b < G; b += BA_THREADS) { pr[0] += a.part[2 * b]; pr[1] += a.part[2 * b + 1]; }
            ba_block_sum<2>(pr, (double(*)[2])sw, out);
            const double scale = 1e-3 + out[0], cc = out[1];
            const double rho = (cost - cc) / scale;
            trials++;
            if (rho > 0.0 && isfinite(cc)) {
                cur = 1 - cur;                             // the candidate buffers become the state
                for (int i = tid; i < K * 12; i += BA_THREADS) sT[i] = sTn[i];
                cost = cc;
                const double g = 2.0 * rho - 1.0;
                lambda = lam * fmax(1.0 / 3.0, fmin(1.0 - g * g * g, 2.0 / 3.0));
                ni = 2.0;
                accepted++; iter++; trial = 0;
                if (iter >= a.iterations) done = 1;
            } else {
                lambda = lam * ni;
                ni *= 2.0;
                trial++;
                if (trial >= 10 || !isfinite(lambda)) done = 1;   // an iteration without an accepted step ends the run
            }
            __syncthreads();
        }
    }
    if (blk == 0 && tid == 0) {
        a.stats[1] = cost;
        a.stats[2] = (double)accepted;
        a.stats[3] = (double)trials;
        a.stats[4] = lambda;
        a.stats[5] = (double)bg_load(&c->abort);
        a.stats[6] = (double)cur;
        a.stats[7] = (double)G;
    }
}

static inline uint64_t bg_round16(uint64_t b) { return (b + 15) / 16 * 16; }


// Number of workgroups of the persistent launch, and the slices a pose / pair task is cut into.  Measured at the
// reference's window (7 keyframes, 5792 observations, 22 tasks; tools/ba_phase_probe.py, us per accepted step incl. its
// four barriers): 22 workgroups 51.2, 44 (two slices per task) 51.7, 66 53.3, 88 54.0 - the camera-block phase shrinks
// with the slices (14.8 -> 12.7 -> 12.0 us), but workgroup 0 adds the slices up before it can solve (17 -> 18 -> 20 us) and
// every barrier grows with the launch (2.3 -> 3.0 us).  Hence one workgroup per task, more only when the lists are long
// (about 512 observations per workgroup), at most 128 (half the device: all of them must be resident at once), and as
// many slices as that width gives a workgroup each (<= 8).
static void bg_shape(int64_t K, int64_t O, int64_t n_free, int* blocks, int* slices) {
    const int64_t ntask = K + n_free * (n_free + 1) / 2;
    int64_t want = (O + 511) / 512;
    if (want < ntask) want = ntask;
    want = want < 8 ? 8 : (want > BG_MAX_BLOCKS ? BG_MAX_BLOCKS : want);
    int64_t ns = want / ntask;
    *blocks = (int)want;
    *slices = (int)(ns < 1 ? 1 : (ns > BG_MAX_SLICES ? BG_MAX_SLICES : ns));
}

extern "C" int slam_ba_optimize_workspace(int64_t K, int64_t L, int64_t O, uint64_t* bytes) {
    SLAM_REQUIRE(bytes, "slam_ba_optimize_workspace: null pointer");
    SLAM_REQUIRE(K >= 1 && K <= 64 && L >= 1 && L <= (1 << 24) && O >= 0 && O <= SLAM_BA_LM_MAX_OBS, "bad sizes");
    const uint64_t ks = (uint64_t)K * BG_MAX_SLICES;
    *bytes = bg_round16(sizeof(bg_ctl)) + bg_round16((uint64_t)K * L * 4) +
             bg_round16((uint64_t)L * 72) + bg_round16((uint64_t)L * 24) + bg_round16(ks * 168) + 2 * bg_round16(ks * 48) +
             bg_round16(ks * 8) + bg_round16((uint64_t)K * 48) + bg_round16((uint64_t)BG_MAX_PAIRS * BG_MAX_SLICES * 288) +
             bg_round16((uint64_t)(K * 6 + 1) * 8) + bg_round16((uint64_t)L * 24) + bg_round16(2 * BG_MAX_BLOCKS * 8);
    return SLAM_OK;
}

extern "C" int slam_ba_optimize_f64(slam_ctx* ctx, int64_t K, int64_t L, int64_t O, const int32_t* d_obs_pose,
                                    const int32_t* d_obs_point, const double* d_meas, const int32_t* d_pt_ptr,
                                    const int32_t* d_pt_obs, const int32_t* d_ps_ptr, const int32_t* d_ps_obs,
                                    const int32_t* d_free_poses, int64_t n_free, double fx, double fy, double cx, double cy,
                                    double huber_delta, int iterations, double* d_poses2, double* d_points2, void* d_work,
                                    uint64_t work_bytes, double* d_stats) {
    SLAM_REQUIRE(ctx, "slam_ba_optimize_f64: null ctx");
    SLAM_REQUIRE(K >= 1 && K <= 64 && L >= 1 && L <= (1 << 24) && O >= 0 && O <= SLAM_BA_LM_MAX_OBS, "bad sizes (K=%lld, L=%lld, O=%lld)",
                 (long long)K, (long long)L, (long long)O);
    SLAM_REQUIRE(n_free >= 0 && n_free <= SLAM_BA_LM_MAX_FREE && n_free <= K, "n_free=%lld: at most %d free poses", (long long)n_free,
                 SLAM_BA_LM_MAX_FREE);
    SLAM_REQUIRE(iterations >= 0 && iterations <= 1000, "bad iteration count");
    SLAM_REQUIRE(d_obs_pose && d_obs_point && d_meas && d_pt_ptr && d_pt_obs && d_ps_ptr && d_ps_obs && (d_free_poses || n_free == 0) &&
                     d_poses2 && d_points2 && d_work && d_stats, "slam_ba_optimize_f64: null device pointer");
    SLAM_REQUIRE(((uintptr_t)d_meas & 15) == 0 && ((uintptr_t)d_work & 15) == 0, "d_meas and d_work must be 16-byte aligned");
    uint64_t need = 0;
    if (int rc = slam_ba_optimize_workspace(K, L, O, &need)) return rc;
    SLAM_REQUIRE(work_bytes >= need, "workspace of %llu bytes, %llu needed (slam_ba_optimize_workspace)", (unsigned long long)work_bytes,
                 (unsigned long long)need);
    SLAM_HIP(hipSetDevice(ctx->device));
    bg_args a;
    a.K = (int)K; a.L = (int)L; a.O = (int)O; a.iterations = iterations; a.nfree = (int)n_free;
    bg_shape(K, O, n_free, &a.nblocks, &a.nsub);
    a.obs_pose = d_obs_pose; a.obs_point = d_obs_point; a.meas = (const double2*)d_meas;
    a.pt_ptr = d_pt_ptr; a.pt_obs = d_pt_obs; a.ps_ptr = d_ps_ptr; a.ps_obs = d_ps_obs; a.free_list = d_free_poses;
    a.T = d_poses2; a.X = d_points2;
    char* w = (char*)d_work;
    auto take = [&](uint64_t bytes) { char* p = w; w += bg_round16(bytes); return p; };
    a.ctl = (bg_ctl*)take(sizeof(bg_ctl));
    a.lookup = (int*)take((uint64_t)K * L * 4);
    a.E = (double*)take((uint64_t)L * 72); a.bl = (double*)take((uint64_t)L * 24);
    const uint64_t ks = (uint64_t)K * BG_MAX_SLICES;
    a.Hpp = (double*)take(ks * 168); a.bp = (double*)take(ks * 48); a.ybl = (double*)take(ks * 48);
    a.costk = (double*)take(ks * 8); a.bpc = (double*)take((uint64_t)K * 48);
    a.W = (double*)take((uint64_t)BG_MAX_PAIRS * BG_MAX_SLICES * 288);
    a.dp = (double*)take((uint64_t)(K * 6 + 1) * 8); a.dl = (double*)take((uint64_t)L * 24);
    a.part = (double*)take(2 * BG_MAX_BLOCKS * 8);
    SLAM_REQUIRE((uint64_t)(w - (char*)d_work) <= need, "slam_ba_optimize_f64: the workspace layout outgrew slam_ba_optimize_workspace");
    a.stats = d_stats;
    a.index_errors = slam_index_error_counter(ctx);
    a.cam = {fx, fy, cx, cy};
    a.delta = huber_delta;
    // Every workgroup has to be resident at once (hand-made grid barriers): refuse what this device could never hold; what
    // it cannot hold RIGHT NOW (other work on the compute units) is found out by the barriers themselves, in bounded time.
    static std::atomic<int> per_cu_once{0};        // a property of the kernel and the architecture: computed once per process
    int blocks_per_cu = per_cu_once.load(std::memory_order_relaxed);
    if (!blocks_per_cu) {
        int occ = 0;
        SLAM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, ba_lm_grid_kernel, BA_THREADS, 0));
        blocks_per_cu = occ > 0 ? occ : -1;
        per_cu_once.store(blocks_per_cu, std::memory_order_relaxed);
    }
    if (blocks_per_cu < 0 || (int64_t)blocks_per_cu * ctx->num_cu < a.nblocks)
        return slam_set_error(SLAM_ERR_BUSY, "slam_ba_optimize_f64: %d workgroups cannot be resident at once on %d compute units",
                              a.nblocks, ctx->num_cu);
    // the control block starts as: barrier idle, no abort, state in half 0
    SLAM_HIP(hipMemsetAsync(a.ctl, 0, sizeof(bg_ctl), ctx->stream));                           // (workgroup 0 fills in the rest before the first barrier)
    SLAM_HIP(hipMemsetAsync(d_stats, 0xFF, 64, ctx->stream));                                  // all-NaN until the launch completes
    ba_lm_grid_kernel<<<(unsigned)a.nblocks, BA_THREADS, 0, ctx->stream>>>(a);
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}
