/* ba_lm_oracle.c - TEST INFRASTRUCTURE (CPU oracle / CPU baseline), never part of the product.
 *
 * The window bundle adjustment of the product (slam_ba_optimize_f64, csrc/ba_schur.hip; an extension: the reference's
 * Backend is an empty class over a Map of NUM_ACTIVE_KEYFRAMES = 7 keyframes, backend.py:10-12,101-103) as a plain C
 * loop on ONE host core: a second oracle-side statement of oracle.ba_lm_np (the CPU suite holds the two against each
 * other) and the CPU baseline of that path (tools/ba_time.py prints it beside the device time).
 *
 * Same problem statement as oracle.ba_lm_np: residuals and Jacobians of frontend.py:272-291 (oracle_reproj_rj_f64, plus
 * the 2x3 point Jacobian), Huber weights, points eliminated by a damped Schur complement, the Levenberg-Marquardt schedule
 * of g2o's OptimizationAlgorithmLevenberg (tau = 1e-5, rho with the +1e-3 scale, lambda * max(1/3, min(1 - (2 rho - 1)^3,
 * 2/3)), ni doubling, ten trials).  Different arithmetic where a choice exists: 3x3 inverses by cofactors, the reduced
 * system by Gaussian elimination with partial pivoting, the exponential by a scaling-and-squaring series. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

int oracle_reproj_rj_f64(const double* poses, int64_t K, const double* points, int64_t L, const int32_t* obs_pose,
                         const int32_t* obs_point, const double* meas, int64_t O, double fx, double fy, double cx,
                         double cy, double* e, double* Jpose, double* Jpoint, int threads);

/* exp of the 4x4 twist generator [[W, v], [0, 0]] (rotation first) by scaling and squaring, Taylor to order 18 */
static void ba_se3_exp(const double* xi, double* E) {
    double G[16] = {0}, term[16], tmp[16];
    G[1] = -xi[2]; G[2] = xi[1]; G[3] = xi[3];
    G[4] = xi[2]; G[6] = -xi[0]; G[7] = xi[4];
    G[8] = -xi[1]; G[9] = xi[0]; G[11] = xi[5];
    double nrm = 0.0;
    for (int i = 0; i < 16; i++) nrm += fabs(G[i]);
    int s = 0;
    while (nrm > 0.5 && s < 60) { nrm *= 0.5; s++; }
    const double sc = ldexp(1.0, -s);
    for (int i = 0; i < 16; i++) G[i] *= sc;
    memset(E, 0, 16 * sizeof(double));
    memset(term, 0, sizeof(term));
    for (int i = 0; i < 4; i++) E[i * 5] = term[i * 5] = 1.0;
    for (int k = 1; k <= 18; k++) {
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                double v = 0.0;
                for (int l = 0; l < 4; l++) v += term[i * 4 + l] * G[l * 4 + j];
                tmp[i * 4 + j] = v / (double)k;
            }
        memcpy(term, tmp, sizeof(term));
        for (int i = 0; i < 16; i++) E[i] += term[i];
    }
    for (; s > 0; s--) {
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                double v = 0.0;
                for (int l = 0; l < 4; l++) v += E[i * 4 + l] * E[l * 4 + j];
                tmp[i * 4 + j] = v;
            }
        memcpy(E, tmp, 16 * sizeof(double));
    }
}

/* inverse of a symmetric 3x3 by cofactors; 0 if singular */
static int inv3(const double* A, double* B) {
    const double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
    const double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
    if (!(fabs(det) > 0.0) || !isfinite(det)) return 0;
    const double r = 1.0 / det;
    B[0] = c00 * r; B[1] = (A[2] * A[7] - A[1] * A[8]) * r; B[2] = (A[1] * A[5] - A[2] * A[4]) * r;
    B[3] = c01 * r; B[4] = (A[0] * A[8] - A[2] * A[6]) * r; B[5] = (A[2] * A[3] - A[0] * A[5]) * r;
    B[6] = c02 * r; B[7] = (A[1] * A[6] - A[0] * A[7]) * r; B[8] = (A[0] * A[4] - A[1] * A[3]) * r;
    return 1;
}

/* solve A x = b (n x n, row-major, A and b destroyed) by Gaussian elimination with partial pivoting; 0 if singular */
static int solve_dense(double* A, double* b, int n) {
    for (int c = 0; c < n; c++) {
        int piv = c;
        for (int r = c + 1; r < n; r++)
            if (fabs(A[r * n + c]) > fabs(A[piv * n + c])) piv = r;
        if (!(fabs(A[piv * n + c]) > 0.0) || !isfinite(A[piv * n + c])) return 0;
        if (piv != c) {
            for (int j = 0; j < n; j++) { const double t = A[c * n + j]; A[c * n + j] = A[piv * n + j]; A[piv * n + j] = t; }
            const double t = b[c]; b[c] = b[piv]; b[piv] = t;
        }
        const double inv = 1.0 / A[c * n + c];
        for (int r = c + 1; r < n; r++) {
            const double f = A[r * n + c] * inv;
            if (f == 0.0) continue;
            for (int j = c; j < n; j++) A[r * n + j] -= f * A[c * n + j];
            b[r] -= f * b[c];
        }
    }
    for (int r = n - 1; r >= 0; r--) {
        double v = b[r];
        for (int j = r + 1; j < n; j++) v -= A[r * n + j] * b[j];
        b[r] = v / A[r * n + r];
        if (!isfinite(b[r])) return 0;
    }
    return 1;
}

static double robust_cost(const double* e, int64_t O, double delta) {
    double c = 0.0;
    for (int64_t o = 0; o < O; o++) {
        const double c2 = e[2 * o] * e[2 * o] + e[2 * o + 1] * e[2 * o + 1];
        const double s = sqrt(c2);
        c += (delta > 0.0 && s > delta) ? 2.0 * delta * s - delta * delta : c2;
    }
    return c;
}

/* poses12 [K,12] (3x4 row-major, rotation first), points [L,3]; pose_fixed [K] non-zero = the pose holds the gauge.
 * stats [4] = initial cost, final cost, accepted steps, trials.  Returns 0, or -1 when memory runs out. */
int oracle_ba_lm_f64(int64_t K, int64_t L, int64_t O, const double* poses_in, const double* points_in, const int32_t* obs_pose,
                     const int32_t* obs_point, const double* meas, const uint8_t* pose_fixed, double fx, double fy, double cx,
                     double cy, double delta, int iterations, double* poses_out, double* points_out, double* stats) {
    int nf = 0;
    int* slot = (int*)malloc((size_t)K * sizeof(int));      /* pose -> row block of the reduced system, or -1 */
    if (!slot) return -1;
    for (int64_t k = 0; k < K; k++) slot[k] = pose_fixed[k] ? -1 : nf++;
    const int n = 6 * nf;
    const size_t o1 = (size_t)(O ? O : 1);
    double* T = (double*)malloc((size_t)K * 12 * 2 * sizeof(double));
    double* X = (double*)malloc((size_t)L * 3 * 2 * sizeof(double));
    double* e = (double*)malloc(o1 * 2 * sizeof(double));
    double* Jp = (double*)malloc(o1 * 12 * sizeof(double));
    double* Jq = (double*)malloc(o1 * 6 * sizeof(double));
    double* Hpl = (double*)malloc(o1 * 18 * sizeof(double));
    double* Hpp = (double*)calloc((size_t)K * 36 + (size_t)K * 6 * 2, sizeof(double));
    double* Hll = (double*)calloc((size_t)L * (9 + 9 + 3 + 3), sizeof(double));
    double* S = (double*)malloc(((size_t)n * n + n + 1) * sizeof(double));
    int64_t* ptr = (int64_t*)calloc((size_t)L + 1, sizeof(int64_t));
    int32_t* by_point = (int32_t*)malloc(o1 * sizeof(int32_t));
    if (!T || !X || !e || !Jp || !Jq || !Hpl || !Hpp || !Hll || !S || !ptr || !by_point) {
        free(slot); free(T); free(X); free(e); free(Jp); free(Jq); free(Hpl); free(Hpp); free(Hll); free(S); free(ptr); free(by_point);
        return -1;
    }
    double* bp = Hpp + (size_t)K * 36;
    double* dp = bp + (size_t)K * 6;
    double* E = Hll + (size_t)L * 9;
    double* bl = E + (size_t)L * 9;
    double* dl = bl + (size_t)L * 3;
    double* rhs = S + (size_t)n * n;
    double* Tn = T + (size_t)K * 12;
    double* Xn = X + (size_t)L * 3;
    memcpy(T, poses_in, (size_t)K * 12 * sizeof(double));
    memcpy(X, points_in, (size_t)L * 3 * sizeof(double));
    /* observations grouped by point (stable) */
    for (int64_t o = 0; o < O; o++) ptr[obs_point[o] + 1]++;
    for (int64_t l = 0; l < L; l++) ptr[l + 1] += ptr[l];
    {
        int64_t* at = (int64_t*)malloc(((size_t)L + 1) * sizeof(int64_t));
        if (!at) return -1;
        memcpy(at, ptr, ((size_t)L + 1) * sizeof(int64_t));
        for (int64_t o = 0; o < O; o++) by_point[at[obs_point[o]]++] = (int32_t)o;
        free(at);
    }
    double lambda = -1.0, ni = 2.0, cost = 0.0, cost0 = 0.0;
    int accepted = 0, trials = 0, need_lin = 1, done = nf == 0;
    int iter = 0, trial = 0;
    /* cost at the start even when nothing moves */
    oracle_reproj_rj_f64(T, K, X, L, obs_pose, obs_point, meas, O, fx, fy, cx, cy, e, Jp, Jq, 1);
    cost0 = cost = robust_cost(e, O, delta);
    while (!done && iter < iterations) {
        if (need_lin) {
            oracle_reproj_rj_f64(T, K, X, L, obs_pose, obs_point, meas, O, fx, fy, cx, cy, e, Jp, Jq, 1);
            memset(Hpp, 0, ((size_t)K * 36 + (size_t)K * 6) * sizeof(double));
            memset(Hll, 0, (size_t)L * 9 * sizeof(double));
            memset(bl, 0, (size_t)L * 3 * sizeof(double));
            for (int64_t o = 0; o < O; o++) {
                const int k = obs_pose[o], l = obs_point[o];
                const double* a = Jp + 12 * o;   /* [2][6] */
                const double* q = Jq + 6 * o;    /* [2][3] */
                const double e0 = e[2 * o], e1 = e[2 * o + 1];
                const double s = sqrt(e0 * e0 + e1 * e1);
                const double w = (delta > 0.0 && s > delta) ? delta / s : 1.0;
                for (int i = 0; i < 6; i++) {
                    for (int j = 0; j < 6; j++) Hpp[(size_t)k * 36 + i * 6 + j] += w * (a[i] * a[j] + a[6 + i] * a[6 + j]);
                    bp[(size_t)k * 6 + i] += w * (a[i] * e0 + a[6 + i] * e1);
                    for (int j = 0; j < 3; j++) Hpl[(size_t)o * 18 + i * 3 + j] = w * (a[i] * q[j] + a[6 + i] * q[3 + j]);
                }
                for (int i = 0; i < 3; i++) {
                    for (int j = 0; j < 3; j++) Hll[(size_t)l * 9 + i * 3 + j] += w * (q[i] * q[j] + q[3 + i] * q[3 + j]);
                    bl[(size_t)l * 3 + i] += w * (q[i] * e0 + q[3 + i] * e1);
                }
            }
            if (lambda < 0.0) {   /* tau * the largest diagonal entry of the blocks that move */
                double dmax = 0.0;
                for (int64_t k = 0; k < K; k++)
                    if (slot[k] >= 0)
                        for (int i = 0; i < 6; i++) dmax = fmax(dmax, Hpp[(size_t)k * 36 + i * 7]);
                for (int64_t l = 0; l < L; l++)
                    for (int i = 0; i < 3; i++) dmax = fmax(dmax, Hll[(size_t)l * 9 + i * 4]);
                lambda = 1e-5 * fmax(dmax, 1e-12);
            }
            need_lin = 0;
        }
        /* E = (Hll + lambda I)^-1, identity for points nobody observes */
        for (int64_t l = 0; l < L; l++) {
            double A[9];
            memcpy(A, Hll + (size_t)l * 9, sizeof(A));
            A[0] += lambda; A[4] += lambda; A[8] += lambda;
            double* El = E + (size_t)l * 9;
            if (ptr[l + 1] == ptr[l] || !inv3(A, El)) { memset(El, 0, 9 * sizeof(double)); El[0] = El[4] = El[8] = 1.0; }
        }
        /* reduced camera system over the poses that move */
        memset(S, 0, ((size_t)n * n + n) * sizeof(double));
        for (int64_t k = 0; k < K; k++) {
            const int a = slot[k];
            if (a < 0) continue;
            for (int i = 0; i < 6; i++) {
                for (int j = 0; j < 6; j++) S[(size_t)(6 * a + i) * n + 6 * a + j] = Hpp[(size_t)k * 36 + i * 6 + j] + (i == j ? lambda : 0.0);
                rhs[6 * a + i] = -bp[(size_t)k * 6 + i];
            }
        }
        for (int64_t l = 0; l < L; l++) {
            const double* El = E + (size_t)l * 9;
            for (int64_t i1 = ptr[l]; i1 < ptr[l + 1]; i1++) {
                const int o1i = by_point[i1], a = slot[obs_pose[o1i]];
                if (a < 0) continue;
                double Y[18];   /* Hpl[o1] E : 6x3 */
                for (int i = 0; i < 6; i++)
                    for (int j = 0; j < 3; j++) {
                        double v = 0.0;
                        for (int m = 0; m < 3; m++) v += Hpl[(size_t)o1i * 18 + i * 3 + m] * El[m * 3 + j];
                        Y[i * 3 + j] = v;
                    }
                for (int i = 0; i < 6; i++)
                    for (int m = 0; m < 3; m++) rhs[6 * a + i] += Y[i * 3 + m] * bl[(size_t)l * 3 + m];
                for (int64_t i2 = ptr[l]; i2 < ptr[l + 1]; i2++) {
                    const int o2i = by_point[i2], b = slot[obs_pose[o2i]];
                    if (b < 0) continue;
                    for (int i = 0; i < 6; i++)
                        for (int j = 0; j < 6; j++) {
                            double v = 0.0;
                            for (int m = 0; m < 3; m++) v += Y[i * 3 + m] * Hpl[(size_t)o2i * 18 + j * 3 + m];
                            S[(size_t)(6 * a + i) * n + 6 * b + j] -= v;
                        }
                }
            }
        }
        trials++;
        if (!solve_dense(S, rhs, n)) {   /* a system that cannot be solved counts as a trial */
            lambda *= ni; ni *= 2.0; trial++;
            if (trial >= 10 || !isfinite(lambda)) done = 1;
            continue;
        }
        memset(dp, 0, (size_t)K * 6 * sizeof(double));
        for (int64_t k = 0; k < K; k++)
            if (slot[k] >= 0) memcpy(dp + (size_t)k * 6, rhs + 6 * slot[k], 6 * sizeof(double));
        /* dl = E (-bl - sum_k Hpl^T dp), zero for unseen points */
        for (int64_t l = 0; l < L; l++) {
            double t[3] = {-bl[(size_t)l * 3], -bl[(size_t)l * 3 + 1], -bl[(size_t)l * 3 + 2]};
            for (int64_t i1 = ptr[l]; i1 < ptr[l + 1]; i1++) {
                const int o = by_point[i1];
                const double* d = dp + (size_t)obs_pose[o] * 6;
                for (int m = 0; m < 3; m++)
                    for (int i = 0; i < 6; i++) t[m] -= Hpl[(size_t)o * 18 + i * 3 + m] * d[i];
            }
            const double* El = E + (size_t)l * 9;
            for (int m = 0; m < 3; m++)
                dl[(size_t)l * 3 + m] = ptr[l + 1] == ptr[l] ? 0.0 : El[m * 3] * t[0] + El[m * 3 + 1] * t[1] + El[m * 3 + 2] * t[2];
        }
        /* candidate state */
        double scale = 1e-3;
        for (int64_t k = 0; k < K; k++) {
            double Ex[16];
            ba_se3_exp(dp + (size_t)k * 6, Ex);
            const double* P = T + (size_t)k * 12;
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 4; j++)
                    Tn[(size_t)k * 12 + i * 4 + j] = Ex[i * 4] * P[j] + Ex[i * 4 + 1] * P[4 + j] + Ex[i * 4 + 2] * P[8 + j] + (j == 3 ? Ex[i * 4 + 3] : 0.0);
            for (int i = 0; i < 6; i++) scale += dp[(size_t)k * 6 + i] * (lambda * dp[(size_t)k * 6 + i] - bp[(size_t)k * 6 + i]);
        }
        for (int64_t l = 0; l < L; l++)
            for (int m = 0; m < 3; m++) {
                const double d = dl[(size_t)l * 3 + m];
                Xn[(size_t)l * 3 + m] = X[(size_t)l * 3 + m] + d;
                scale += d * (lambda * d - bl[(size_t)l * 3 + m]);
            }
        oracle_reproj_rj_f64(Tn, K, Xn, L, obs_pose, obs_point, meas, O, fx, fy, cx, cy, e, Jp, NULL, 1);
        const double cand = robust_cost(e, O, delta);
        const double rho = (cost - cand) / scale;
        if (rho > 0.0 && isfinite(cand)) {
            memcpy(T, Tn, (size_t)K * 12 * sizeof(double));
            memcpy(X, Xn, (size_t)L * 3 * sizeof(double));
            cost = cand;
            need_lin = 1;
            const double g = 2.0 * rho - 1.0;
            lambda *= fmax(1.0 / 3.0, fmin(1.0 - g * g * g, 2.0 / 3.0));
            ni = 2.0;
            accepted++; iter++; trial = 0;
        } else {
            lambda *= ni; ni *= 2.0; trial++;
            if (trial >= 10 || !isfinite(lambda)) done = 1;   /* an iteration without an accepted step ends the run */
        }
    }
    memcpy(poses_out, T, (size_t)K * 12 * sizeof(double));
    memcpy(points_out, X, (size_t)L * 3 * sizeof(double));
    stats[0] = cost0; stats[1] = cost; stats[2] = (double)accepted; stats[3] = (double)trials;
    free(slot); free(T); free(X); free(e); free(Jp); free(Jq); free(Hpl); free(Hpp); free(Hll); free(S); free(ptr); free(by_point);
    return 0;
}
