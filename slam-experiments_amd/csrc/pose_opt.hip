// pose_opt.hip — the whole pose-only refinement of Frontend._correct_current_pose
// (reference frontend.py:298-393) as ONE kernel launch on gfx950.
//
// The reference builds a g2o graph with one pose vertex and one 2-D reprojection
// edge per feature that has a map point, then runs four outer rounds of ten
// Levenberg-Marquardt iterations; every residual and Jacobian evaluation is a
// Python callback from g2o's C++ (frontend.py:272-291), <= 200 edges x <= 40
// iterations x 2 callbacks.  Here one 256-thread workgroup keeps the problem
// on chip: every evaluation is a block-wide pass over the observations
// (residual, 2x6 Jacobian, Huber weight, 28 sums by one DPP pairing + LDS in a fixed
// order, so results are run-to-run identical), lane 0 solves the damped 6x6 system by
// LDL^T and drives g2o's published LM schedule.
//
// Measured and dropped (round 3, tools/pose_lm_probe.py, profiles/r03_pose_lm.log): the whole refinement on ONE wave
// for frames of <= 256 edges (up to 4 edges per lane in registers, no barrier at all, every lane running the solve
// redundantly on wave-uniform values): 338 / 394 / 409 us at 50 / 200 / 256 edges against 317 / 339 / 284 us for this
// 256-thread form - a lone wave exposes the latency of every dependent f64 instruction, which costs more than the ten
// barriers per trial it saves.  Where the time goes: the schedule makes ~70 trials (24 accepted, and every round ends
// with ten rejected ones: g2o gives up after maxTrialsAfterFailure), a trial is one evaluation (2.4 us) plus solve,
// exponential and bookkeeping (2.2 us).  One host core running the same loop in plain C (the test suite's CPU statement) takes
// 102 / 372 / 576 / 2728 us at 50 / 200 / 256 / 1000 edges: at the reference's frame size (<= 200 edges, slam.py:23)
// a single refinement gains nothing from the GPU; the batch form below (one workgroup per frame) is where it does:
// 16 frames x 200 edges in 0.68 ms = 42 us per frame.
//
// Same structure and constants as slamhip/pose_opt.py (the host-driven version
// the tests compare against): every round restarts from the input pose
// (frontend.py:360), chi2 > threshold marks an edge as outlier / level 1
// (frontend.py:371-377), the robust kernel is dropped after round index 2
// (frontend.py:378-379).  PARITY UNPINNED against g2o itself (absent here; the
// reference also mixes a VertexSE3 with an Expmap Jacobian): the update is the
// one the Jacobian of frontend.py:288-291 is the derivative for, T <- exp([w,v]) T.
#include "internal.h"
#include <math.h>

#define PO_THREADS 256
#define PO_TERMS 28   // 21 (upper H) + 6 (b) + 1 (robust chi2 of the active edges)
#define PO_RED_STRIDE (PO_THREADS / 2 + 8)   // doubles per term in the reduction buffer: padded, so the 16-value reads of
                                             // thread (term, segment) fall on different LDS banks for different terms
static_assert(SLAM_POSE_STAGE == 512, "host_calls.hip decides by it");
#define PO_STAGE SLAM_POSE_STAGE  // observations kept in LDS (3 + 2 + 1 doubles and a flag each: 24.5 KiB)

struct po_cam { double fx, fy, cx, cy; };

struct po_params {
    int rounds, iterations;
    double chi2_threshold, huber_delta;
};

// exp([w, v]) * T for a 3x4 row-major pose (rotation first, g2o SE3Quat::exp ordering)
__device__ __forceinline__ void po_apply_update(const double* dx, const double* T, double* Tn) {
    const double wx = dx[0], wy = dx[1], wz = dx[2];
    const double th2 = wx * wx + wy * wy + wz * wz, th = sqrt(th2);
    double a, b, c;  // sin(th)/th, (1-cos)/th^2, (th-sin)/th^3
    if (th < 1e-10) { a = 1.0; b = 0.5; c = 1.0 / 6.0; }
    else {
        double sn, cs;
        sincos(th, &sn, &cs);
        a = sn / th; b = (1.0 - cs) / th2; c = (th - sn) / (th2 * th);
    }
    const double W[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double W2[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) W2[i * 3 + j] = W[i * 3] * W[j] + W[i * 3 + 1] * W[3 + j] + W[i * 3 + 2] * W[6 + j];
    double R[9], V[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const double I = (i % 4 == 0) ? 1.0 : 0.0;
        R[i] = I + a * W[i] + b * W2[i];
        V[i] = I + b * W[i] + c * W2[i];
    }
    double t[3];
#pragma unroll
    for (int i = 0; i < 3; i++) t[i] = V[i * 3] * dx[3] + V[i * 3 + 1] * dx[4] + V[i * 3 + 2] * dx[5];
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 4; j++)
            Tn[i * 4 + j] = R[i * 3] * T[j] + R[i * 3 + 1] * T[4 + j] + R[i * 3 + 2] * T[8 + j];
        Tn[i * 4 + 3] += t[i];
    }
}

// solve (H + lam I) x = -b by LDL^T (six reciprocals, no square roots: this runs on one lane, so the length of
// the dependent f64 chain is what it costs); H given as packed upper triangle s[0..20]; false if not SPD
__device__ __forceinline__ bool po_solve(const double* s, const double* b, double lam, double* x) {
    double A[36];
    int t = 0;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = i; j < 6; j++) { A[i * 6 + j] = A[j * 6 + i] = s[t++]; }
#pragma unroll
    for (int i = 0; i < 6; i++) A[i * 6 + i] += lam;
    double L[36], d[6], dinv[6];
    bool spd = true;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        double v = A[j * 6 + j];
#pragma unroll
        for (int k = 0; k < j; k++) v -= L[j * 6 + k] * L[j * 6 + k] * d[k];
        spd = spd && (v > 0.0) && isfinite(v);
        d[j] = v;
        dinv[j] = 1.0 / v;
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            double u = A[i * 6 + j];
#pragma unroll
            for (int k = 0; k < j; k++) u -= L[i * 6 + k] * L[j * 6 + k] * d[k];
            L[i * 6 + j] = u * dinv[j];
        }
    }
    if (!spd) return false;
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double v = -b[i];
#pragma unroll
        for (int k = 0; k < i; k++) v -= L[i * 6 + k] * y[k];
        y[i] = v;
    }
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        double v = y[i] * dinv[i];
#pragma unroll
        for (int k = i + 1; k < 6; k++) v -= L[k * 6 + i] * x[k];
        x[i] = v;
    }
    return true;
}

// Residual and robust cost of one edge at pose T (frontend.py:272-277 and g2o's RobustKernelHuber).  Written with explicit
// fma() for every multiply-add so that the compiler has no contraction left to decide: po_evaluate and po_costs call it in
// different surroundings and must round identically - at convergence the gain ratio's numerator is a difference of two
// sums that agree to the last bits, and a reduction or contraction that differed between the two passes would turn every
// such trial into a coin toss (measured: 29 -> 33-37 "accepted" steps at 200 edges).
struct po_edge { double X, Y, Z, e0, e1, c2, w, rho; };
__device__ __forceinline__ po_edge po_edge_at(const double* T, double px, double py, double pz, double2 m, po_cam cam,
                                              double delta) {
    po_edge r;
    r.X = fma(T[0], px, fma(T[1], py, fma(T[2], pz, T[3])));
    r.Y = fma(T[4], px, fma(T[5], py, fma(T[6], pz, T[7])));
    r.Z = fma(T[8], px, fma(T[9], py, fma(T[10], pz, T[11])));
    r.e0 = m.x - fma(cam.fx, r.X, cam.cx * r.Z) / r.Z;      // frontend.py:275-277
    r.e1 = m.y - fma(cam.fy, r.Y, cam.cy * r.Z) / r.Z;
    r.c2 = fma(r.e0, r.e0, r.e1 * r.e1);
    r.w = 1.0;                                               // Huber: rho' and rho
    r.rho = r.c2;
    if (delta > 0.0) {
        const double en = sqrt(r.c2);
        if (en > delta) { r.w = delta / en; r.rho = fma(2.0 * delta, en, -(delta * delta)); }
    }
    return r;
}

// sum over the block of `nterms` per-thread values each (acc[0..nterms)), run-to-run identical: adjacent lanes are paired
// by a DPP swap, the even lane of each pair parks the pair's sum in LDS (128 values per term), thread (term, segment) adds
// its 16 values in a rotated but fixed order (bank-conflict free through the padded term stride; the rotation depends on the segment only, so a term
// rounds the same whatever its index), then one thread per term adds the 8 partials.  Three barriers; every thread returns
// with the sums in out[0..nterms).
__device__ __forceinline__ double po_swap_adjacent(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0xB1, 0xF, 0xF, true);          // quad_perm [1,0,3,2]
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0xB1, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int NT>
__device__ __forceinline__ void po_block_sums(double (&acc)[NT], double* red, double* part, double* out) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NT; i++) acc[i] += po_swap_adjacent(acc[i]);
    if (!(tid & 1)) {
#pragma unroll
        for (int i = 0; i < NT; i++) red[i * PO_RED_STRIDE + (tid >> 1)] = acc[i];
    }
    __syncthreads();
    if (tid < NT * 8) {
        const double* r = red + (tid >> 3) * PO_RED_STRIDE + (tid & 7) * 16;
        double sum = 0.0;
#pragma unroll
        for (int k = 0; k < 16; k++) sum += r[(k + (tid & 7)) & 15];
        part[tid] = sum;
    }
    __syncthreads();
    if (tid < NT) {
        const double* q = part + tid * 8;
        out[tid] = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
    }
    __syncthreads();
}

// One block-wide evaluation at pose T: sums[0..20] = upper H, [21..26] = b, [27] = robust chi2 over the
// active edges; chi2[o] = e.e for every edge.  Every thread returns with the sums in `out` (shared).
// points / meas / active / chi2 are generic pointers: the LDS copies when the problem is staged, else global.
__device__ __forceinline__ void po_evaluate(const double* T, const double* points, const double2* meas, const uint8_t* active, int O,
                            po_cam cam, double delta, double* chi2, double* red, double* part, double* out) {
    double acc[PO_TERMS];
#pragma unroll
    for (int i = 0; i < PO_TERMS; i++) acc[i] = 0.0;
    for (int o = threadIdx.x; o < O; o += PO_THREADS) {
        const po_edge g = po_edge_at(T, points[o * 3], points[o * 3 + 1], points[o * 3 + 2], meas[o], cam, delta);
        chi2[o] = g.c2;
        if (!active[o]) continue;
        const double X = g.X, Y = g.Y, Z = g.Z, e0 = g.e0, e1 = g.e1, w = g.w;
        const double Zinv = 1.0 / (Z + 1e-18), Zinv2 = Zinv * Zinv;  // frontend.py:284-291
        const double j0[6] = {cam.fx * X * Y * Zinv2, -cam.fx - cam.fx * X * X * Zinv2, cam.fx * Y * Zinv,
                              -cam.fx * Zinv, 0.0, cam.fx * X * Zinv2};
        const double j1[6] = {cam.fy + cam.fy * Y * Y * Zinv2, -cam.fy * X * Y * Zinv2, -cam.fy * X * Zinv, 0.0,
                              -cam.fy * Zinv, cam.fy * Y * Zinv2};
        int t = 0;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = a; b < 6; b++) acc[t++] += w * (j0[a] * j0[b] + j1[a] * j1[b]);
#pragma unroll
        for (int a = 0; a < 6; a++) acc[21 + a] += w * (j0[a] * e0 + j1[a] * e1);
        acc[27] += g.rho;
    }
    po_block_sums<PO_TERMS>(acc, red, part, out);
}

// The robust cost of the active edges at up to PO_SPEC candidate poses in ONE block-wide pass (no Jacobians, no normal
// equations): costs[j] for every j with solved[j].  Same per-edge arithmetic as po_evaluate; the sums travel through LDS
// in the same rotated fixed order.
#define PO_SPEC 9     // trials 1..9 of an iteration, evaluated together after trial 0 was turned down
__device__ __forceinline__ void po_costs(const double* Tk /*[PO_SPEC][12]*/, const int* solved, const double* points,
                                         const double2* meas, const uint8_t* active, int O, po_cam cam, double delta,
                                         double* red, double* part, double* costs) {
    double acc[PO_SPEC];
#pragma unroll
    for (int j = 0; j < PO_SPEC; j++) acc[j] = 0.0;
    for (int o = threadIdx.x; o < O; o += PO_THREADS) {
        if (!active[o]) continue;
        const double px = points[o * 3], py = points[o * 3 + 1], pz = points[o * 3 + 2];
        const double2 m = meas[o];
#pragma unroll
        for (int j = 0; j < PO_SPEC; j++) {
            if (!solved[j]) continue;                                   // block-uniform
            acc[j] += po_edge_at(Tk + 12 * j, px, py, pz, m, cam, delta).rho;
        }
    }
    po_block_sums<PO_SPEC>(acc, red, part, costs);
}

#define PO_TRIALS (PO_SPEC + 1)   // maxTrialsAfterFailure of g2o's Levenberg-Marquardt: ten trials per iteration

struct po_shared {
    double red[PO_TERMS * PO_RED_STRIDE];
    double part[PO_TERMS * 8];
    double sums[2][PO_TERMS];                       // the evaluation at the current pose / at the candidate: swapped, not copied
    double T0[12], Tcur[12];
    double pose[2][PO_TRIALS * 12];                 // the candidates of an iteration; the buffers alternate, so an accepted
                                                    // candidate stays where it is and becomes the current pose by pointer
    double scale_k[PO_TRIALS], cost_k[PO_SPEC];
    int solved_k[PO_TRIALS];
    int nactive;
};

// The LM loop proper.  Force-inlined into both call sites of the kernel so that the staged instance addresses
// points / meas / active / chi2 as LDS (ds_read) and the large-problem instance as global memory, instead of
// both going through flat loads on generic pointers.
//
// The Levenberg-Marquardt state (damping, its growth factor, the accepted-step count, which buffers hold the current
// pose / sums / chi2) lives in REGISTERS of every thread: all threads read the same sums behind the same barrier and
// take the same decision, so a verdict needs no lane-0 section and no barrier of its own (per iteration: one barrier
// behind the proposals + the three of the evaluation; the first form had seven and two serial sections).
// chi2_a / chi2_b: where evaluations leave e.e per edge.  With two buffers (the staged form) an evaluation writes the
// one that is NOT current and an accepted pose makes it current by swapping the pointers, so the chi2 of the current pose
// is always at hand and the outlier decision at the end of a round needs no evaluation of its own; with one buffer
// (chi2_a == chi2_b: problems too large to stage) the round ends with an evaluation, as before.
__device__ __forceinline__ int po_run(po_shared& sh, const double* points, const double2* meas, uint8_t* active,
                                      double* chi2_a, double* chi2_b, int O, po_cam cam, po_params prm, int* accepted_out,
                                      const double** pose_out, double** chi2_out) {
    const int tid = threadIdx.x;
    double* red = sh.red; double* part = sh.part;
    double* cur = sh.sums[0]; double* cand = sh.sums[1];
    const double* T0 = sh.T0;
    const double* T = sh.Tcur;
    double* chi2 = chi2_a; double* chi2_w = chi2_b;          // current pose's / where the next evaluation writes
    const bool two_chi2 = chi2_a != chi2_b;
    int p = 0, accepted = 0;
    for (int o = tid; o < O; o += PO_THREADS) active[o] = 1;
    __syncthreads();
    int nactive = O;
    double delta = prm.huber_delta;
    for (int round = 0; round < prm.rounds; round++) {
        if (tid < 12) sh.Tcur[tid] = T0[tid];           // every round restarts from the frame's pose (frontend.py:360)
        T = sh.Tcur;
        __syncthreads();
        po_evaluate(T, points, meas, active, O, cam, delta, chi2_w, red, part, cur);
        { double* t = chi2; chi2 = chi2_w; chi2_w = t; }
        double lambda, ni = 2.0;
        {
            double dmax = 0.0;
            const int diag[6] = {0, 6, 11, 15, 18, 20};
#pragma unroll
            for (int i = 0; i < 6; i++) dmax = fmax(dmax, cur[diag[i]]);
            lambda = 1e-5 * fmax(dmax, 1e-12);          // tau * max diagonal
        }
        for (int it = 0; it < prm.iterations && nactive > 0; it++) {
            // ---- the ten trials of this iteration are PROPOSED together.  A trial that is turned down changes nothing but
            // the damping (lambda *= ni, ni *= 2), so the candidates of all ten trials are known in advance: lane j of
            // wave 0 solves with the damping trial j would meet and builds its pose - ten solves in the instruction stream
            // of one.  Trial 0 is then evaluated in full (cost, H, b); only when it is turned down does ONE more pass give
            // the costs of trials 1..9, and they are walked in order: the first one the sequential loop would have accepted
            // is accepted, with the damping that loop would have had.  g2o's schedule ends every round on ten rejected
            // trials (maxTrialsAfterFailure): ten sequential solve + evaluation pairs (38 us of a 67 us round at 200
            // edges) become one solve and two passes.
            double* cands = sh.pose[p];
            if (tid < PO_TRIALS) {
                double lam = lambda, n2 = ni;
                for (int k = 0; k < tid; k++) { lam *= n2; n2 *= 2.0; }
                double dx[6];
                const bool ok_j = po_solve(cur, cur + 21, lam, dx);
                double sc = 1e-3;
                for (int i = 0; i < 6; i++) sc += dx[i] * (lam * dx[i] - cur[21 + i]);
                sh.solved_k[tid] = ok_j ? 1 : 0;
                if (ok_j) { po_apply_update(dx, T, cands + 12 * tid); sh.scale_k[tid] = sc; }
            }
            __syncthreads();
            const int solved = sh.solved_k[0];          // read now: the next iteration's proposals overwrite them
            const double scale = sh.scale_k[0];
            if (solved) po_evaluate(cands, points, meas, active, O, cam, delta, chi2_w, red, part, cand);
            const double rho0 = solved ? (cur[27] - cand[27]) / scale : -1.0;
            if (solved && rho0 > 0.0 && isfinite(cand[27])) {
                { double* t = cur; cur = cand; cand = t; }
                { double* t = chi2; chi2 = chi2_w; chi2_w = t; }
                T = cands;
                p ^= 1;
                const double g = 2.0 * rho0 - 1.0;
                lambda *= fmax(1.0 / 3.0, fmin(1.0 - g * g * g, 2.0 / 3.0));
                ni = 2.0;
                accepted++;
                continue;
            }
            lambda *= ni;
            ni *= 2.0;
            if (solved && !isfinite(lambda)) break;     // (an unsolvable system never ends the iteration by itself)
            // ---- trial 0 was turned down: the costs of trials 1..9 in one pass
            po_costs(cands + 12, sh.solved_k + 1, points, meas, active, O, cam, delta, red, part, sh.cost_k);
            int taken = -1;
            for (int j = 0; j < PO_SPEC; j++) {
                if (sh.solved_k[j + 1]) {
                    const double rho = (cur[27] - sh.cost_k[j]) / sh.scale_k[j + 1];
                    if (rho > 0.0 && isfinite(sh.cost_k[j])) {
                        const double g = 2.0 * rho - 1.0;
                        lambda *= fmax(1.0 / 3.0, fmin(1.0 - g * g * g, 2.0 / 3.0));
                        ni = 2.0;
                        taken = j + 1;
                        break;
                    }
                }
                lambda *= ni;
                ni *= 2.0;
                if (sh.solved_k[j + 1] && !isfinite(lambda)) break;
            }
            if (taken < 0) break;                       // ten trials turned down (or the damping overflowed): the round's iterations end
            accepted++;
            T = cands + 12 * taken;
            p ^= 1;
            po_evaluate(T, points, meas, active, O, cam, delta, chi2_w, red, part, cur);   // H, b, cost at the accepted pose
            { double* t = chi2; chi2 = chi2_w; chi2_w = t; }
        }
        // chi2 at the pose this round ended on, then the outlier / level decision (frontend.py:366-379)
        if (!two_chi2) po_evaluate(T, points, meas, active, O, cam, delta, chi2, red, part, cand);
        int mine = 0;
        for (int o = tid; o < O; o += PO_THREADS) {
            const uint8_t in = chi2[o] <= prm.chi2_threshold ? 1 : 0;
            active[o] = in;
            mine += in;
        }
        if (mine) atomicAdd(&sh.nactive, mine);          // integer: order does not matter
        if (round == 2) delta = 0.0;
        __syncthreads();
        nactive = sh.nactive;
        __syncthreads();
        if (tid == 0) sh.nactive = 0;
    }
    *accepted_out = accepted;
    *pose_out = T;
    *chi2_out = chi2;
    return nactive;
}

// One workgroup per problem.  offsets == nullptr: a single problem of O edges.  Otherwise problem b = blockIdx.x owns the
// edges [offsets[b], offsets[b+1]) of the concatenated point / pixel arrays, pose b of pose_in / pose_out and stats[2b..]:
// independent frames (a window's keyframes against the fixed map, relocalisation candidates) refined in ONE launch.
__global__ __launch_bounds__(PO_THREADS) void pose_opt_kernel(const double* __restrict__ pose_in,
                                                              const double* __restrict__ g_points,
                                                              const double2* __restrict__ g_meas, int O,
                                                              const int* __restrict__ offsets, po_cam cam,
                                                              po_params prm, double* __restrict__ pose_out,
                                                              uint8_t* __restrict__ g_active,
                                                              double* __restrict__ g_chi2, int* __restrict__ stats,
                                                              unsigned int* __restrict__ index_errors,
                                                              unsigned* __restrict__ done, unsigned epoch) {
    if (offsets) {
        // O is the length of the concatenated arrays here.  A table that is not ascending or leaves [0, O] is the
        // caller's bug; it is reported (slam_index_errors) and the frame shrinks to what lies inside, never a wild access.
        const int b = blockIdx.x, total = O, lo = offsets[b], hi = offsets[b + 1];
        const int first = min(max(lo, 0), total), last = min(max(hi, first), total);
        if ((first != lo || last != hi) && threadIdx.x == 0) atomicAdd(index_errors, 1u);
        O = last - first;
        pose_in += 12 * b; pose_out += 12 * b; stats += 2 * b;
        g_points += 3 * (size_t)first; g_meas += first; g_active += first; g_chi2 += first;
    }
    // A frame has <= 200 edges (slam.py:23): the whole problem lives in LDS (24.5 KiB) and no evaluation touches
    // global memory.  Larger problems run the same code on the global arrays.
    __shared__ double s_points[PO_STAGE * 3];
    __shared__ double2 s_meas[PO_STAGE];
    __shared__ double s_chi2[2][PO_STAGE];
    __shared__ uint8_t s_active[PO_STAGE];
    __shared__ po_shared sh;
    const int tid = threadIdx.x;
    if (tid < 12) sh.T0[tid] = sh.Tcur[tid] = pose_in[tid];
    if (tid == 0) sh.nactive = 0;
    int nactive, accepted = 0;
    const double* pose = sh.Tcur;
    double* chi2 = nullptr;
    if (O <= PO_STAGE) {
        for (int i = tid; i < O * 3; i += PO_THREADS) s_points[i] = g_points[i];
        for (int o = tid; o < O; o += PO_THREADS) s_meas[o] = g_meas[o];
        nactive = po_run(sh, s_points, s_meas, s_active, s_chi2[0], s_chi2[1], O, cam, prm, &accepted, &pose, &chi2);
        if (prm.rounds > 0)
            for (int o = tid; o < O; o += PO_THREADS) { g_active[o] = s_active[o]; g_chi2[o] = chi2[o]; }
        else
            for (int o = tid; o < O; o += PO_THREADS) { g_active[o] = 1; g_chi2[o] = 0.0; }
    } else {
        nactive = po_run(sh, g_points, g_meas, g_active, g_chi2, g_chi2, O, cam, prm, &accepted, &pose, &chi2);
    }
    __syncthreads();
    if (tid < 12) pose_out[tid] = pose[tid];
    if (tid == 0) {
        stats[0] = prm.rounds > 0 ? nactive : O;   // inlier count: what _correct_current_pose returns (frontend.py:393)
        stats[1] = accepted;                        // accepted LM steps over all rounds
    }
    if (done) {
        // the host call polls this word instead of synchronising the stream (host_calls.hip; slam_wait_done): every thread's
        // result stores - they went straight into pinned host memory - are visible to the host before one lane releases it
        __threadfence_system();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(done, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

extern "C" int slam_pose_optimize_f64(slam_ctx* ctx, const double* d_pose_in, const double* d_points,
                                      const double* d_meas, int64_t O, double fx, double fy, double cx, double cy,
                                      int rounds, int iterations, double chi2_threshold, double huber_delta,
                                      double* d_pose_out, uint8_t* d_inlier, double* d_chi2, int32_t* d_stats) {
    return slam_pose_optimize_polled(ctx, d_pose_in, d_points, d_meas, O, fx, fy, cx, cy, rounds, iterations, chi2_threshold,
                                     huber_delta, d_pose_out, d_inlier, d_chi2, d_stats, nullptr, 0);
}

// slam_pose_optimize_f64 whose kernel stores `epoch` into *done (pinned host memory) behind its results, or done == nullptr
int slam_pose_optimize_polled(slam_ctx* ctx, const double* d_pose_in, const double* d_points, const double* d_meas, int64_t O,
                              double fx, double fy, double cx, double cy, int rounds, int iterations, double chi2_threshold,
                              double huber_delta, double* d_pose_out, uint8_t* d_inlier, double* d_chi2, int32_t* d_stats,
                              unsigned* done, unsigned epoch) {
    SLAM_REQUIRE(ctx, "slam_pose_optimize_f64: null ctx");
    SLAM_REQUIRE(O >= 0 && O <= (1 << 24), "O=%lld out of range [0, 2^24]", (long long)O);
    SLAM_REQUIRE(rounds >= 0 && iterations >= 0 && rounds <= 64 && iterations <= 1000, "bad rounds / iterations");
    SLAM_REQUIRE(d_pose_in && d_pose_out && d_stats && (O == 0 || (d_points && d_meas && d_inlier && d_chi2)),
                 "slam_pose_optimize_f64: null device pointer");
    SLAM_REQUIRE(((uintptr_t)d_meas & 15) == 0, "d_meas must be 16-byte aligned");
    SLAM_HIP(hipSetDevice(ctx->device));
    const po_cam cam = {fx, fy, cx, cy};
    const po_params prm = {rounds, iterations, chi2_threshold, huber_delta};
    pose_opt_kernel<<<1, PO_THREADS, 0, ctx->stream>>>(d_pose_in, d_points, (const double2*)d_meas, (int)O, nullptr, cam, prm,
                                                       d_pose_out, d_inlier, d_chi2, d_stats, slam_index_error_counter(ctx), done, epoch);
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}

extern "C" int slam_pose_optimize_batch_f64(slam_ctx* ctx, int64_t B, const double* d_pose_in, const double* d_points,
                                            const double* d_meas, const int32_t* d_offsets, int64_t O_total, double fx,
                                            double fy, double cx, double cy, int rounds, int iterations,
                                            double chi2_threshold, double huber_delta, double* d_pose_out,
                                            uint8_t* d_inlier, double* d_chi2, int32_t* d_stats) {
    SLAM_REQUIRE(ctx, "slam_pose_optimize_batch_f64: null ctx");
    SLAM_REQUIRE(B >= 0 && B <= (1 << 20) && O_total >= 0 && O_total <= (1 << 28), "bad sizes (B=%lld, O=%lld)", (long long)B, (long long)O_total);
    SLAM_REQUIRE(rounds >= 0 && iterations >= 0 && rounds <= 64 && iterations <= 1000, "bad rounds / iterations");
    if (B == 0) return SLAM_OK;
    SLAM_REQUIRE(d_pose_in && d_pose_out && d_stats && d_offsets && (O_total == 0 || (d_points && d_meas && d_inlier && d_chi2)),
                 "slam_pose_optimize_batch_f64: null device pointer");
    SLAM_REQUIRE(((uintptr_t)d_meas & 15) == 0, "d_meas must be 16-byte aligned");
    SLAM_HIP(hipSetDevice(ctx->device));
    const po_cam cam = {fx, fy, cx, cy};
    const po_params prm = {rounds, iterations, chi2_threshold, huber_delta};
    pose_opt_kernel<<<(unsigned)B, PO_THREADS, 0, ctx->stream>>>(d_pose_in, d_points, (const double2*)d_meas, (int)O_total, d_offsets, cam, prm,
                                                                 d_pose_out, d_inlier, d_chi2, d_stats, slam_index_error_counter(ctx), nullptr, 0u);
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}
