/*
 * reproj_oracle.c — CPU restatement (f64) of the reference's reprojection
 * residual / Jacobian.  TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's
 * cpu_baseline leg); never imported by the product path.
 *
 * PARITY UNPINNED: the reference code is Frontend.EdgeProjectionPoseOnly
 * (frontend.py:262-291), callbacks of g2o-python 0.0.12 (poetry.lock:667-668),
 * which is not installable here; the reference holds no tests or golden
 * vectors.  The arithmetic below follows the Python text line by line and is
 * pinned by hand-computed vectors + finite differences (tests/golden/).
 *
 *   compute_error      frontend.py:272-277
 *       p_c = T * pos3d;  px = K @ p_c;  px /= px[2];  e = meas - px[:2]
 *   linearize_oplus    frontend.py:279-291
 *       Zinv = 1/(Z + 1e-18); Zinv2 = Zinv**2; 2x6 rows, rotation columns first
 *   K layout           primitives.py:25-29  [[fx,0,cx],[0,fy,cy],[0,0,1]]
 *   measurement        frontend.py:348 (int-truncated pixel as f64)
 *   point Jacobian     NOT in the reference: -dproj/dp_c * R
 *   normal equations   the quadratic form g2o builds for one VertexSE3 with
 *                      information I2 (frontend.py:349) and RobustKernelHuber
 *                      (frontend.py:350, delta = 1): H = sum rho' J^T J,
 *                      b = sum rho' J^T e, chi2 = e.e; level-1 edges skipped
 *                      (frontend.py:372-377)
 * Pose layout: 12 doubles = rows of [R|t] (the top 3x4 of Tcw, row-major).
 */
#include <stdint.h>
#include <math.h>

static void project_one(const double* P, const double* p, const double* m, double fx, double fy, double cx,
                        double cy, double* e, double* J, double* A, double* Xc) {
    const double X = P[0] * p[0] + P[1] * p[1] + P[2] * p[2] + P[3];
    const double Y = P[4] * p[0] + P[5] * p[1] + P[6] * p[2] + P[7];
    const double Z = P[8] * p[0] + P[9] * p[1] + P[10] * p[2] + P[11];
    /* K @ p_c, then divide by the third component */
    const double px0 = fx * X + 0.0 * Y + cx * Z;
    const double px1 = 0.0 * X + fy * Y + cy * Z;
    const double px2 = Z;
    e[0] = m[0] - px0 / px2;
    e[1] = m[1] - px1 / px2;
    const double Zinv = 1.0 / (Z + 1e-18);
    const double Zinv2 = Zinv * Zinv;
    J[0] = fx * X * Y * Zinv2;
    J[1] = -fx - fx * X * X * Zinv2;
    J[2] = fx * Y * Zinv;
    J[3] = -fx * Zinv;
    J[4] = 0.0;
    J[5] = fx * X * Zinv2;
    J[6] = fy + fy * Y * Y * Zinv2;
    J[7] = -fy * X * Y * Zinv2;
    J[8] = -fy * X * Zinv;
    J[9] = 0.0;
    J[10] = -fy * Zinv;
    J[11] = fy * Y * Zinv2;
    if (A) {
        A[0] = fx * Zinv; A[1] = 0.0; A[2] = -fx * X * Zinv2;
        A[3] = 0.0; A[4] = fy * Zinv; A[5] = -fy * Y * Zinv2;
    }
    if (Xc) { Xc[0] = X; Xc[1] = Y; Xc[2] = Z; }
}

int oracle_reproj_rj_f64(const double* poses, int64_t K, const double* points, int64_t L, const int32_t* obs_pose,
                         const int32_t* obs_point, const double* meas, int64_t O, double fx, double fy, double cx,
                         double cy, double* e, double* Jpose, double* Jpoint, int threads) {
    (void)K; (void)L;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int64_t o = 0; o < O; o++) {
        const double* P = poses + (int64_t)obs_pose[o] * 12;
        const double* p = points + (int64_t)obs_point[o] * 3;
        double A[6];
        project_one(P, p, meas + 2 * o, fx, fy, cx, cy, e + 2 * o, Jpose + 12 * o, A, 0);
        if (Jpoint) {
            double* Jq = Jpoint + 6 * o;
            for (int c = 0; c < 3; c++) {
                Jq[c] = -(A[0] * P[c] + A[2] * P[8 + c]);
                Jq[3 + c] = -(A[4] * P[4 + c] + A[5] * P[8 + c]);
            }
        }
    }
    return 0;
}

int oracle_pose_normal_eq_f64(const double* pose, const double* points, const double* meas, const uint8_t* active,
                              int64_t O, double fx, double fy, double cx, double cy, double huber_delta, double* H,
                              double* b, double* chi2) {
    for (int i = 0; i < 36; i++) H[i] = 0.0;
    for (int i = 0; i < 6; i++) b[i] = 0.0;
    for (int64_t o = 0; o < O; o++) {
        double e[2], J[12];
        project_one(pose, points + 3 * o, meas + 2 * o, fx, fy, cx, cy, e, J, 0, 0);
        const double c2 = e[0] * e[0] + e[1] * e[1];
        chi2[o] = c2;
        if (active && !active[o]) continue;
        double w = 1.0;
        if (huber_delta > 0.0) {
            const double en = sqrt(c2);
            if (en > huber_delta) w = huber_delta / en;
        }
        for (int a = 0; a < 6; a++) {
            for (int c = 0; c < 6; c++) H[a * 6 + c] += w * (J[a] * J[c] + J[6 + a] * J[6 + c]);
            b[a] += w * (J[a] * e[0] + J[6 + a] * e[1]);
        }
    }
    return 0;
}
