/*
 * pose_lm_oracle.c — CPU restatement (f64, plain C, one thread) of the pose-only
 * refinement Frontend._correct_current_pose (frontend.py:298-393).
 * TEST INFRASTRUCTURE ONLY (tests/, tools/latency.py's CPU baseline leg); never
 * imported by the product path.
 *
 * PARITY UNPINNED: the loop it restates runs inside g2o-python 0.0.12
 * (poetry.lock:667-668), which is not installable here, and the reference holds
 * no tests or golden vectors.  What is followed, line by line:
 *   graph            one pose vertex, one 2-D edge per feature with a map point,
 *                    information I2, RobustKernelHuber      frontend.py:310-354
 *   rounds           4 x optimize(10), every round restarts from the frame's
 *                    pose                                    frontend.py:358-365
 *   outliers         chi2 > 5.991**2 -> level 1 (leaves the optimisation),
 *                    else level 0                            frontend.py:356,371-377
 *   robust kernel    removed after round index 2             frontend.py:378-379
 *   result           pose written back, inlier count         frontend.py:384-393
 *   optimiser        g2o's published OptimizationAlgorithmLevenberg: lambda0 =
 *                    1e-5 max diag(H); up to 10 trials per iteration of
 *                    (H + lambda I) dx = -b; rho = (chi - chi_new) /
 *                    (dx.(lambda dx - b) + 1e-3); accepted iff rho > 0 and finite:
 *                    lambda *= max(1/3, min(1 - (2 rho - 1)^3, 2/3)), ni = 2;
 *                    rejected: lambda *= ni, ni *= 2
 *   update           T <- exp([w, v]) T, the update the Jacobian of
 *                    frontend.py:288-291 is the derivative for
 * Residuals, Jacobians and Huber weights come from oracle_pose_normal_eq_f64
 * (reproj_oracle.c).  Deliberately different from the product's arithmetic where
 * a choice exists: the 6x6 system is solved by Gaussian elimination with partial
 * pivoting (product: LDL^T), the SE(3) exponential is the scaling-and-squaring
 * Taylor series of the 4x4 generator (product: closed-form Rodrigues / V matrix).
 * The numpy statement of the same loop is oracle.pose_lm_np; the CPU suite holds
 * the two against each other.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

int oracle_pose_normal_eq_f64(const double* pose, const double* points, const double* meas, const uint8_t* active,
                              int64_t O, double fx, double fy, double cx, double cy, double huber_delta, double* H,
                              double* b, double* chi2);

/* exp of the 4x4 twist generator [[W, v], [0, 0]] by scaling and squaring: G / 2^s has norm < 1/2, Taylor to order 18 */
static void se3_exp(const double* xi, double* E /* 4x4 */) {
    double G[16] = {0};
    G[1] = -xi[2]; G[2] = xi[1]; G[3] = xi[3];
    G[4] = xi[2]; G[6] = -xi[0]; G[7] = xi[4];
    G[8] = -xi[1]; G[9] = xi[0]; G[11] = xi[5];
    double nrm = 0.0;
    for (int i = 0; i < 16; i++) nrm += fabs(G[i]);
    int s = 0;
    while (nrm > 0.5 && s < 60) { nrm *= 0.5; s++; }
    const double sc = ldexp(1.0, -s);
    for (int i = 0; i < 16; i++) G[i] *= sc;
    double term[16], tmp[16];
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

/* solve A x = rhs (6x6, A destroyed) by Gaussian elimination with partial pivoting; 0 if singular */
static int solve6(double* A, double* rhs, double* x) {
    for (int c = 0; c < 6; c++) {
        int p = c;
        for (int r = c + 1; r < 6; r++)
            if (fabs(A[r * 6 + c]) > fabs(A[p * 6 + c])) p = r;
        if (!(fabs(A[p * 6 + c]) > 0.0) || !isfinite(A[p * 6 + c])) return 0;
        if (p != c) {
            for (int j = 0; j < 6; j++) { double t = A[c * 6 + j]; A[c * 6 + j] = A[p * 6 + j]; A[p * 6 + j] = t; }
            double t = rhs[c]; rhs[c] = rhs[p]; rhs[p] = t;
        }
        for (int r = c + 1; r < 6; r++) {
            const double f = A[r * 6 + c] / A[c * 6 + c];
            for (int j = c; j < 6; j++) A[r * 6 + j] -= f * A[c * 6 + j];
            rhs[r] -= f * rhs[c];
        }
    }
    for (int r = 5; r >= 0; r--) {
        double v = rhs[r];
        for (int j = r + 1; j < 6; j++) v -= A[r * 6 + j] * x[j];
        x[r] = v / A[r * 6 + r];
    }
    return 1;
}

/* g2o RobustKernelHuber::robustify, rho[0] summed over the active edges */
static double robust_cost(const double* chi2, const uint8_t* active, int64_t O, double delta) {
    double s = 0.0;
    for (int64_t o = 0; o < O; o++) {
        if (!active[o]) continue;
        const double en = sqrt(chi2[o]);
        s += (delta > 0.0 && en > delta) ? 2.0 * delta * en - delta * delta : chi2[o];
    }
    return s;
}

/* pose_in / pose_out: 12 doubles = rows of [R|t]; inlier uint8 [O]; chi2 [O]; stats[0] = inliers, stats[1] = accepted steps.
 * work: 2 * O doubles of scratch. */
int oracle_pose_lm_f64(const double* pose_in, const double* points, const double* meas, int64_t O, double fx, double fy,
                       double cx, double cy, int rounds, int iterations, double chi2_threshold, double huber_delta,
                       double* pose_out, uint8_t* inlier, double* chi2, int32_t* stats, double* work) {
    double T0[16] = {0}, T[16], Tn[16], H[36], b[6], Hn[36], bn[6];
    memcpy(T0, pose_in, 12 * sizeof(double));
    T0[15] = 1.0;
    memcpy(T, T0, sizeof(T));
    for (int64_t o = 0; o < O; o++) { inlier[o] = 1; chi2[o] = 0.0; }
    double* chi2n = work;
    double delta = huber_delta;
    int accepted = 0;
    int64_t nactive = O;
    for (int rnd = 0; rnd < rounds; rnd++) {
        memcpy(T, T0, sizeof(T));
        oracle_pose_normal_eq_f64(T, points, meas, inlier, O, fx, fy, cx, cy, delta, H, b, chi2);
        double cur = robust_cost(chi2, inlier, O, delta);
        double dmax = 0.0;
        for (int i = 0; i < 6; i++) dmax = fmax(dmax, H[i * 7]);
        double lam = 1e-5 * fmax(dmax, 1e-12), ni = 2.0;
        for (int it = 0; it < iterations && nactive > 0; it++) {
            int stepped = 0;
            for (int trial = 0; trial < 10; trial++) {
                double A[36], rhs[6], dx[6];
                memcpy(A, H, sizeof(A));
                for (int i = 0; i < 6; i++) { A[i * 7] += lam; rhs[i] = -b[i]; }
                if (!solve6(A, rhs, dx)) { lam *= ni; ni *= 2.0; continue; }
                double E[16];
                se3_exp(dx, E);
                for (int i = 0; i < 4; i++)
                    for (int j = 0; j < 4; j++) {
                        double v = 0.0;
                        for (int l = 0; l < 4; l++) v += E[i * 4 + l] * T[l * 4 + j];
                        Tn[i * 4 + j] = v;
                    }
                oracle_pose_normal_eq_f64(Tn, points, meas, inlier, O, fx, fy, cx, cy, delta, Hn, bn, chi2n);
                const double nw = robust_cost(chi2n, inlier, O, delta);
                double scale = 1e-3;
                for (int i = 0; i < 6; i++) scale += dx[i] * (lam * dx[i] - b[i]);
                const double rho = (cur - nw) / scale;
                if (rho > 0.0 && isfinite(nw)) {
                    memcpy(T, Tn, sizeof(T));
                    memcpy(H, Hn, sizeof(H));
                    memcpy(b, bn, sizeof(b));
                    memcpy(chi2, chi2n, (size_t)O * sizeof(double));
                    cur = nw;
                    const double g = 2.0 * rho - 1.0;
                    lam *= fmax(1.0 / 3.0, fmin(1.0 - g * g * g, 2.0 / 3.0));
                    ni = 2.0;
                    stepped = 1;
                    accepted++;
                    break;
                }
                lam *= ni;
                ni *= 2.0;
                if (!isfinite(lam)) break;
            }
            if (!stepped) break;
        }
        /* chi2 holds e.e at T (the last accepted evaluation, or the round's first one) */
        nactive = 0;
        for (int64_t o = 0; o < O; o++) {
            inlier[o] = chi2[o] <= chi2_threshold ? 1 : 0;
            nactive += inlier[o];
        }
        if (rnd == 2) delta = 0.0;
    }
    memcpy(pose_out, T, 12 * sizeof(double));
    stats[0] = (int32_t)(rounds > 0 ? nactive : O);
    stats[1] = accepted;
    return 0;
}
