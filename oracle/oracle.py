"""CPU oracle for the descriptor-matching / reprojection hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module; the product path
(``slam-experiments_amd/``) never does and fails loudly without its HIP library.

PARITY UNPINNED: the reference delegates this path to cv2
(opencv-python 4.9.0.80), g2o-python 0.0.12 and jaxlie, none of which is
installed or installable here, and ships no tests or golden vectors.  The
oracle is therefore a restatement of the published algorithms, pinned by the
hand-derived known-answer vectors under ``tests/golden/``.

Two independent restatements live here so they can check each other:

* ``*_c``  — ctypes bindings of ``liboracle.so`` (``bf_hamming_oracle.c``,
  ``reproj_oracle.c``): OpenCV's K-best insertion loop, line by line.
* ``*_np`` — numpy: full distance matrix + stable lexicographic sort, and the
  reference's residual/Jacobian text (``frontend.py:272-291``) with numpy
  matrices exactly as the Python callbacks compute them.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None

NO_IDX = -1
NO_DIST = 2**31 - 1
IMGIDX_SHIFT = 18  # OpenCV matchers.cpp: index = imgIdx << 18 | trainIdx


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (used by __graft_entry__.build())."""
    srcs = sorted(os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h")) or f == "Makefile")   # every source there is
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs
    )
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "all"])
    return _LIB_PATH


def _load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    lib = ctypes.CDLL(_LIB_PATH)
    i64, i32, vp, dbl = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_double
    lib.oracle_bf_simd.argtypes = []
    lib.oracle_bf_simd.restype = ctypes.c_char_p
    lib.oracle_bf_knn_u256.argtypes = [vp, i64, vp, i64, i32, vp, vp, i32]
    lib.oracle_bf_knn_u256.restype = i32
    lib.oracle_bf_knn_multi_u256.argtypes = [vp, i64, vp, vp, i32, i32, vp, vp, i32]
    lib.oracle_bf_knn_multi_u256.restype = i32
    lib.oracle_bf_match.argtypes = [vp, vp, i64, i32, dbl, i32, vp, vp, vp]
    lib.oracle_bf_match.restype = i64
    lib.oracle_bf_ratio_test.argtypes = [vp, vp, i64, dbl, vp]
    lib.oracle_bf_ratio_test.restype = i64
    lib.oracle_bf_cross_check_u256.argtypes = [vp, i64, vp, i64, vp, vp, i32]
    lib.oracle_bf_cross_check_u256.restype = i32
    lib.oracle_reproj_rj_f64.argtypes = [vp, i64, vp, i64, vp, vp, vp, i64, dbl, dbl, dbl, dbl, vp, vp, vp, i32]
    lib.oracle_reproj_rj_f64.restype = i32
    lib.oracle_pose_normal_eq_f64.argtypes = [vp, vp, vp, vp, i64, dbl, dbl, dbl, dbl, dbl, vp, vp, vp]
    lib.oracle_pose_normal_eq_f64.restype = i32
    lib.oracle_pose_lm_f64.argtypes = [vp, vp, vp, i64, dbl, dbl, dbl, dbl, i32, i32, dbl, dbl, vp, vp, vp, vp, vp]
    lib.oracle_pose_lm_f64.restype = i32
    lib.oracle_ba_lm_f64.argtypes = [i64, i64, i64, vp, vp, vp, vp, vp, vp, dbl, dbl, dbl, dbl, dbl, i32, vp, vp, vp]
    lib.oracle_ba_lm_f64.restype = i32
    _lib = lib
    return lib


def _desc(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim != 2 or a.shape[1] != 32:
        a = a.reshape(-1, 32)
    return a


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def bf_simd() -> str:
    """Which distance loop liboracle runs on this host ("avx512-vpopcntdq" or "scalar-popcnt")."""
    return _load().oracle_bf_simd().decode()


# --------------------------------------------------------------------------
# matching: C restatement
# --------------------------------------------------------------------------
def bf_knn_c(query: np.ndarray, train: np.ndarray, k: int = 2, threads: int = 1) -> Tuple[np.ndarray, np.ndarray]:
    """cv2.BFMatcher(NORM_HAMMING).knnMatch(query, train, k) as (idx, dist) int32 [N,k]."""
    q, t = _desc(query), _desc(train)
    idx = np.empty((q.shape[0], k), np.int32)
    dist = np.empty((q.shape[0], k), np.int32)
    rc = _load().oracle_bf_knn_u256(_p(q), q.shape[0], _p(t), t.shape[0], k, _p(idx), _p(dist), threads)
    assert rc == 0
    return idx, dist


def bf_knn_multi_c(query: np.ndarray, train_images: Sequence[np.ndarray], k: int = 2, threads: int = 1):
    """knnMatch against BFMatcher.add([...]) collections; returns (imgIdx, trainIdx, dist) [N,k]."""
    q = _desc(query)
    imgs = [_desc(t) for t in train_images]
    rows = np.array([t.shape[0] for t in imgs], np.int64)
    cat = np.concatenate(imgs, 0) if imgs else np.zeros((0, 32), np.uint8)
    idx = np.empty((q.shape[0], k), np.int32)
    dist = np.empty((q.shape[0], k), np.int32)
    rc = _load().oracle_bf_knn_multi_u256(_p(q), q.shape[0], _p(cat), _p(rows), len(imgs), k, _p(idx), _p(dist), threads)
    assert rc == 0, rc
    img = np.where(idx >= 0, idx >> IMGIDX_SHIFT, -1).astype(np.int32)
    tr = np.where(idx >= 0, idx & ((1 << IMGIDX_SHIFT) - 1), -1).astype(np.int32)
    return img, tr, dist


def bf_match_c(source: np.ndarray, query: np.ndarray, dist_threshold: Optional[float] = None, threads: int = 1):
    """BruteForceFeatureMatcher.match(source, query, dist_threshold) (feature_matchers.py:36-44).

    Returns (queryIdx, trainIdx, distance float32) arrays, one entry per DMatch."""
    idx, dist = bf_knn_c(query, source, 1, threads)
    n = idx.shape[0]
    oq = np.empty(n, np.int32)
    ot = np.empty(n, np.int32)
    od = np.empty(n, np.float32)
    has = dist_threshold is not None
    m = _load().oracle_bf_match(_p(idx), _p(dist), n, 1, float(dist_threshold or 0.0), int(has), _p(oq), _p(ot), _p(od))
    return oq[:m].copy(), ot[:m].copy(), od[:m].copy()


def bf_ratio_c(idx: np.ndarray, dist: np.ndarray, ratio: float) -> np.ndarray:
    idx = np.ascontiguousarray(idx, np.int32)
    dist = np.ascontiguousarray(dist, np.int32)
    keep = np.empty(idx.shape[0], np.uint8)
    _load().oracle_bf_ratio_test(_p(idx), _p(dist), idx.shape[0], float(ratio), _p(keep))
    return keep.astype(bool)


def bf_cross_check_c(query: np.ndarray, train: np.ndarray, threads: int = 1):
    q, t = _desc(query), _desc(train)
    oi = np.empty(q.shape[0], np.int32)
    od = np.empty(q.shape[0], np.int32)
    rc = _load().oracle_bf_cross_check_u256(_p(q), q.shape[0], _p(t), t.shape[0], _p(oi), _p(od), threads)
    assert rc == 0
    return oi, od


# --------------------------------------------------------------------------
# matching: numpy restatement (independent of the C loop)
# --------------------------------------------------------------------------
def hamming_matrix_np(query: np.ndarray, train: np.ndarray) -> np.ndarray:
    q, t = _desc(query), _desc(train)
    out = np.empty((q.shape[0], t.shape[0]), np.int32)
    step = max(1, (1 << 24) // max(1, t.shape[0]))  # ~512 MiB of xor temporaries at most
    for a in range(0, q.shape[0], step):
        x = q[a:a + step, None, :] ^ t[None, :, :]
        out[a:a + step] = np.bitwise_count(x).sum(-1, dtype=np.int32)
    return out


def bf_knn_np(query: np.ndarray, train: np.ndarray, k: int = 2) -> Tuple[np.ndarray, np.ndarray]:
    """(distance asc, train index asc) top-k via a stable sort of the full distance matrix."""
    q, t = _desc(query), _desc(train)
    n, m = q.shape[0], t.shape[0]
    idx = np.full((n, k), NO_IDX, np.int32)
    dist = np.full((n, k), NO_DIST, np.int32)
    if n == 0 or m == 0:
        return idx, dist
    d = hamming_matrix_np(q, t)
    order = np.argsort(d, axis=1, kind="stable")[:, :k]
    kk = order.shape[1]
    idx[:, :kk] = order
    dist[:, :kk] = np.take_along_axis(d, order, 1)
    return idx, dist


def bf_match_np(source: np.ndarray, query: np.ndarray, dist_threshold: Optional[float] = None):
    idx, dist = bf_knn_np(query, source, 1)
    has = idx[:, 0] >= 0
    q = np.nonzero(has)[0].astype(np.int32)
    t = idx[has, 0]
    d = dist[has, 0].astype(np.float32)
    if dist_threshold and len(q) != 0:
        lim = max(2 * float(d.min()), dist_threshold)
        keep = d < lim
        q, t, d = q[keep], t[keep], d[keep]
    return q, t, d


def bf_cross_check_np(query: np.ndarray, train: np.ndarray):
    """crossCheck=True as its documented contract, independently of the C loop: pair (i, j) is returned iff
    j is i's nearest train row and i is j's nearest query row (first minimum = lowest index on both sides)."""
    q, t = _desc(query), _desc(train)
    n, m = q.shape[0], t.shape[0]
    oi = np.full(n, NO_IDX, np.int32)
    od = np.full(n, NO_DIST, np.int32)
    if n == 0 or m == 0:
        return oi, od
    d = hamming_matrix_np(q, t)               # [N, M]
    fwd = np.argmin(d, axis=1)                # query -> nearest train row (lowest index among ties)
    rev = np.argmin(d, axis=0)                # train -> nearest query row (lowest index among ties)
    mutual = rev[fwd] == np.arange(n)
    oi[mutual] = fwd[mutual]
    od[mutual] = d[np.arange(n), fwd][mutual]
    return oi, od


# --------------------------------------------------------------------------
# reprojection residual / Jacobian
# --------------------------------------------------------------------------
def poses_to_rt12(T: np.ndarray) -> np.ndarray:
    """[K,4,4] (or [K,3,4]) Tcw matrices -> [K,12] rows of [R|t]."""
    T = np.asarray(T, np.float64)
    return np.ascontiguousarray(T[:, :3, :4].reshape(T.shape[0], 12))


def reproj_rj_c(poses12, points, obs_pose, obs_point, meas, fx, fy, cx, cy, with_point=True, threads=1, out=None):
    poses12 = np.ascontiguousarray(poses12, np.float64)
    points = np.ascontiguousarray(points, np.float64)
    obs_pose = np.ascontiguousarray(obs_pose, np.int32)
    obs_point = np.ascontiguousarray(obs_point, np.int32)
    meas = np.ascontiguousarray(meas, np.float64)
    O = obs_pose.shape[0]
    if out is not None:                                  # caller-owned output arrays (timing without page faults)
        e, Jp, Jq = out
        assert e.shape == (O, 2) and Jp.shape == (O, 2, 6) and (Jq is None or Jq.shape == (O, 2, 3))
    else:
        e = np.empty((O, 2))
        Jp = np.empty((O, 2, 6))
        Jq = np.empty((O, 2, 3)) if with_point else None
    rc = _load().oracle_reproj_rj_f64(_p(poses12), poses12.shape[0], _p(points), points.shape[0], _p(obs_pose),
                                      _p(obs_point), _p(meas), O, fx, fy, cx, cy, _p(e), _p(Jp), _p(Jq), threads)
    assert rc == 0
    return e, Jp, Jq


def reproj_rj_np(poses12, points, obs_pose, obs_point, meas, fx, fy, cx, cy):
    """frontend.py:272-291 per observation, with numpy objects shaped as the callbacks see them."""
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=np.float64)  # primitives.py:25-29
    O = len(obs_pose)
    e = np.empty((O, 2))
    Jp = np.empty((O, 2, 6))
    Jq = np.empty((O, 2, 3))
    for o in range(O):
        P = np.asarray(poses12[obs_pose[o]], np.float64).reshape(3, 4)
        R, t = P[:, :3], P[:, 3]
        pos3d = np.asarray(points[obs_point[o]], np.float64)
        pos_cam = R @ pos3d + t                      # T * self.pos3d
        pos_pixel = K @ pos_cam
        pos_pixel = pos_pixel / pos_pixel[2]
        e[o] = np.asarray(meas[o], np.float64) - pos_pixel[:2]
        X, Y, Z = pos_cam
        Zinv = 1.0 / (Z + 1e-18)
        Zinv2 = Zinv ** 2
        Jp[o] = np.array([
            [fx * X * Y * Zinv2, -fx - fx * X * X * Zinv2, fx * Y * Zinv, -fx * Zinv, 0, fx * X * Zinv2],
            [fy + fy * Y * Y * Zinv2, -fy * X * Y * Zinv2, -fy * X * Zinv, 0, -fy * Zinv, fy * Y * Zinv2],
        ])
        A = np.array([[fx * Zinv, 0, -fx * X * Zinv2], [0, fy * Zinv, -fy * Y * Zinv2]])
        Jq[o] = -A @ R
    return e, Jp, Jq


def pose_normal_eq_c(pose12, points, meas, active, fx, fy, cx, cy, huber_delta):
    pose12 = np.ascontiguousarray(pose12, np.float64).reshape(12)
    points = np.ascontiguousarray(points, np.float64)
    meas = np.ascontiguousarray(meas, np.float64)
    act = None if active is None else np.ascontiguousarray(active, np.uint8)
    O = points.shape[0]
    H = np.empty((6, 6))
    b = np.empty(6)
    chi2 = np.empty(O)
    rc = _load().oracle_pose_normal_eq_f64(_p(pose12), _p(points), _p(meas), _p(act), O, fx, fy, cx, cy,
                                           float(huber_delta), _p(H), _p(b), _p(chi2))
    assert rc == 0
    return H, b, chi2


# --------------------------------------------------------------------------
# pose-only Levenberg-Marquardt (Frontend._correct_current_pose) and the Schur reduction of a window BA
# --------------------------------------------------------------------------
def se3_exp_np(xi) -> np.ndarray:
    """exp of the twist [w (rotation), v (translation)] as a 4x4 matrix, by the matrix exponential of its 4x4
    generator (scipy), i.e. without the closed-form Rodrigues / V-matrix expressions the product code uses."""
    from scipy.linalg import expm

    w, v = np.asarray(xi[:3], np.float64), np.asarray(xi[3:], np.float64)
    G = np.zeros((4, 4))
    G[:3, :3] = [[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]]
    G[:3, 3] = v
    return expm(G)


def _huber_rho(chi2: np.ndarray, delta: float) -> float:
    """g2o RobustKernelHuber::robustify, rho[0] summed: e2 for sqrt(e2) <= delta, 2 delta sqrt(e2) - delta^2 beyond."""
    if delta <= 0:
        return float(np.sum(chi2))
    s = np.sqrt(chi2)
    return float(np.sum(np.where(s <= delta, chi2, 2.0 * delta * s - delta * delta)))


def pose_lm_np(pose, points, meas, fx, fy, cx, cy, rounds: int = 4, iterations: int = 10,
               chi2_threshold: float = 5.991 ** 2, huber_delta: float = 1.0):
    """Frontend._correct_current_pose (frontend.py:298-393) on arrays, CPU only.

    Round logic from the reference: `rounds` outer rounds (frontend.py:358), each restarting from the frame's pose
    (:360) and running optimizer.optimize(10) (:365); after each round every edge is classified by
    chi2() > 5.991**2 (:356, :371-377: outliers go to level 1 and leave the optimisation, inliers return to level 0),
    and after round index 2 the robust kernel is removed (:378-379).  The optimiser is g2o's published
    OptimizationAlgorithmLevenberg: lambda0 = 1e-5 * max diag(H); per iteration up to 10 trials of
    (H + lambda I) dx = -b; rho = (chi - chi_new) / (dx.(lambda dx - b) + 1e-3); accepted iff rho > 0 and finite:
    lambda *= max(1/3, min(1 - (2 rho - 1)^3, 2/3)), ni = 2; rejected: lambda *= ni, ni *= 2; an iteration that ends
    without an accepted step ends the round.  Update convention: the one frontend.py:288-291's Jacobian is the
    derivative for, T <- exp([w, v]) T (rotation first).  Residuals, Jacobians and Huber weights come from the C oracle
    (oracle_pose_normal_eq_f64); nothing here touches the HIP library.

    Returns (T 4x4, inliers bool [O], chi2 [O] at T, accepted LM steps)."""
    P = np.asarray(pose, np.float64)
    T0 = np.eye(4)
    T0[:3, :4] = P.reshape(3, 4) if P.size == 12 else P.reshape(-1, 4)[:3, :4]
    points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    meas = np.ascontiguousarray(meas, np.float64).reshape(-1, 2)
    O = points.shape[0]
    active = np.ones(O, bool)
    delta = float(huber_delta)
    T, chi2, accepted = T0, np.zeros(O), 0

    def evaluate(Tm):
        return pose_normal_eq_c(Tm[:3, :4].reshape(12), points, meas, active.astype(np.uint8), fx, fy, cx, cy, delta)

    for rnd in range(rounds):
        T = T0.copy()
        H, b, chi2 = evaluate(T)
        cur = _huber_rho(chi2[active], delta)
        lam = 1e-5 * max(float(np.max(np.diag(H))), 1e-12)
        ni = 2.0
        for _ in range(iterations):
            if not active.any():
                break
            stepped = False
            for _trial in range(10):
                try:
                    dx = np.linalg.solve(H + lam * np.eye(6), -b)
                except np.linalg.LinAlgError:
                    lam, ni = lam * ni, ni * 2
                    continue
                Tn = se3_exp_np(dx) @ T
                Hn, bn, chi2n = evaluate(Tn)
                new = _huber_rho(chi2n[active], delta)
                rho = (cur - new) / (float(dx @ (lam * dx - b)) + 1e-3)
                if rho > 0 and np.isfinite(new):
                    T, H, b, chi2, cur = Tn, Hn, bn, chi2n, new
                    lam *= max(1.0 / 3.0, min(1.0 - (2.0 * rho - 1.0) ** 3, 2.0 / 3.0))
                    ni = 2.0
                    stepped = True
                    accepted += 1
                    break
                lam, ni = lam * ni, ni * 2
                if not np.isfinite(lam):
                    break
            if not stepped:
                break
        active = chi2 <= chi2_threshold
        if rnd == 2:
            delta = 0.0
    return T, active, chi2, accepted


def pose_lm_c(pose, points, meas, fx, fy, cx, cy, rounds: int = 4, iterations: int = 10,
              chi2_threshold: float = 5.991 ** 2, huber_delta: float = 1.0):
    """The same loop in plain C on one host core (oracle/pose_lm_oracle.c: Gaussian elimination instead of numpy's
    solver, a scaling-and-squaring series instead of scipy's expm): the CPU baseline of the pose refinement, and a
    second oracle-side statement that the CPU suite holds against ``pose_lm_np``.
    Returns (T 3x4 as 12 doubles, inliers bool [O], chi2 [O], accepted steps)."""
    P = np.ascontiguousarray(np.asarray(pose, np.float64).reshape(-1)[:12])
    points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    meas = np.ascontiguousarray(meas, np.float64).reshape(-1, 2)
    O = points.shape[0]
    out = np.empty(12)
    inl = np.empty(max(O, 1), np.uint8)
    chi2 = np.empty(max(O, 1))
    stats = np.zeros(2, np.int32)
    work = np.empty(2 * max(O, 1))
    rc = _load().oracle_pose_lm_f64(_p(P), _p(points), _p(meas), O, fx, fy, cx, cy, int(rounds), int(iterations),
                                    float(chi2_threshold), float(huber_delta), _p(out), _p(inl), _p(chi2), _p(stats), _p(work))
    assert rc == 0
    return out, inl[:O].astype(bool), chi2[:O], int(stats[1])


def ba_schur_np(poses12, points, obs_pose, obs_point, meas, fx, fy, cx, cy, huber_delta: float, lam: float):
    """Reduced camera system of a window bundle adjustment, observation by observation in f64 on the C oracle's
    residuals and Jacobians (reproj_rj_c, frontend.py:272-291 + the 2x3 point Jacobian).

    With w = Huber weight, per observation o = (pose k, point l):  Hpp[k] += w Jp^T Jp, bp[k] += w Jp^T e,
    Hll[l] += w Jq^T Jq, bl[l] += w Jq^T e, Hpl[o] = w Jp^T Jq.  Points are eliminated with damping lam:
    E[l] = (Hll[l] + lam I)^-1 (identity for points nobody observes),
    S[k1,k2] = delta(k1,k2) (Hpp[k1] + lam I) - sum_l Hpl[k1,l] E[l] Hpl[k2,l]^T,
    rhs[k] = -bp[k] + sum_l Hpl[k,l] E[l] bl[l];  cost = sum of Huber rho(e.e).
    Returns a dict with S [K,K,6,6], rhs [K,6], bp, bl, E, Hpl [O,6,3], cost."""
    poses12 = np.ascontiguousarray(poses12, np.float64).reshape(-1, 12)
    points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    K, L, O = poses12.shape[0], points.shape[0], len(obs_pose)
    e, Jp, Jq = reproj_rj_c(poses12, points, obs_pose, obs_point, meas, fx, fy, cx, cy, with_point=True)
    Hpp, bp = np.zeros((K, 6, 6)), np.zeros((K, 6))
    Hll, bl = np.zeros((L, 3, 3)), np.zeros((L, 3))
    Hpl = np.zeros((O, 6, 3))
    cost = 0.0
    for o in range(O):
        k, l = int(obs_pose[o]), int(obs_point[o])
        c2 = float(e[o] @ e[o])
        s = np.sqrt(c2)
        w = 1.0
        if huber_delta > 0 and s > huber_delta:
            w = huber_delta / s
            cost += 2.0 * huber_delta * s - huber_delta * huber_delta
        else:
            cost += c2
        Hpp[k] += w * Jp[o].T @ Jp[o]
        bp[k] += w * Jp[o].T @ e[o]
        Hll[l] += w * Jq[o].T @ Jq[o]
        bl[l] += w * Jq[o].T @ e[o]
        Hpl[o] = w * Jp[o].T @ Jq[o]
    seen = np.zeros(L, bool)
    seen[np.asarray(obs_point, np.int64)] = True
    E = np.zeros((L, 3, 3))
    for l in range(L):
        E[l] = np.linalg.inv(Hll[l] + lam * np.eye(3)) if seen[l] else np.eye(3)
    S = np.zeros((K, K, 6, 6))
    rhs = -bp.copy()
    for k in range(K):
        S[k, k] = Hpp[k] + lam * np.eye(6)
    by_point = [[] for _ in range(L)]
    for o in range(O):
        by_point[int(obs_point[o])].append(o)
    for l in range(L):
        for o1 in by_point[l]:
            Y = Hpl[o1] @ E[l]
            rhs[int(obs_pose[o1])] += Y @ bl[l]
            for o2 in by_point[l]:
                S[int(obs_pose[o1]), int(obs_pose[o2])] -= Y @ Hpl[o2].T
    return {"S": S, "rhs": rhs, "bp": bp, "bl": bl, "E": E, "Hpl": Hpl, "cost": cost, "seen": seen}


def ba_backsub_np(red: dict, obs_pose, obs_point, dp) -> np.ndarray:
    """Point updates after the pose solve: dl[l] = E[l] (-bl[l] - sum_k Hpl[k,l]^T dp[k]); zero for unseen points."""
    dp = np.asarray(dp, np.float64).reshape(-1, 6)
    tmp = -red["bl"].copy()
    for o in range(len(obs_pose)):
        tmp[int(obs_point[o])] -= red["Hpl"][o].T @ dp[int(obs_pose[o])]
    dl = np.einsum("lab,lb->la", red["E"], tmp)
    dl[~red["seen"]] = 0.0
    return dl


def ba_lm_np(poses12, points, obs_pose, obs_point, meas, fx, fy, cx, cy, iterations: int = 10, fixed_poses=(0,),
             huber_delta: float = 0.0):
    """The window bundle adjustment the product runs on the device, as a CPU loop around ``ba_schur_np`` /
    ``ba_backsub_np`` (C-oracle residuals and Jacobians, numpy everything else): the TRAJECTORY oracle of
    ``slam_ba_optimize_f64`` and of the host-driven device form.

    Schedule (the one the pose-only refinement takes from g2o's OptimizationAlgorithmLevenberg, applied to the joint
    problem): lambda0 = 1e-5 max(diag Hpp of the free poses, diag Hll); per iteration up to 10 trials of the damped,
    Schur-reduced system over the free poses; dl by back-substitution; T <- exp([w, v]) T, X <- X + dl;
    rho = (cost - cost_new) / (dp.(lambda dp - bp) + dl.(lambda dl - bl) + 1e-3); accepted iff rho > 0 and finite:
    lambda *= max(1/3, min(1 - (2 rho - 1)^3, 2/3)), ni = 2; rejected: lambda *= ni, ni *= 2; an iteration without an
    accepted step ends the run.  Returns (T [K,4,4], X [L,3], cost0, cost, accepted, lambdas of the accepted steps)."""
    P = np.ascontiguousarray(poses12, np.float64).reshape(-1, 12)
    K = P.shape[0]
    T = np.tile(np.eye(4), (K, 1, 1))
    T[:, :3, :4] = P.reshape(K, 3, 4)
    X = np.array(points, np.float64).reshape(-1, 3)
    op, ol = np.asarray(obs_pose, np.int32), np.asarray(obs_point, np.int32)
    free = np.ones(K, bool)
    free[list(fixed_poses)] = False
    fidx = np.flatnonzero(free)
    rt = lambda Tm: Tm[:, :3, :4].reshape(K, 12)

    def cost_at(Tm, Xm):
        e, _, _ = reproj_rj_c(rt(Tm), Xm, op, ol, meas, fx, fy, cx, cy, with_point=False)
        return _huber_rho((e * e).sum(1), huber_delta)

    red = ba_schur_np(rt(T), X, op, ol, meas, fx, fy, cx, cy, huber_delta, 1.0)
    cost0 = cost = red["cost"]
    # diag of the undamped blocks: S[k,k] + W[k,k] - lam I = Hpp[k]; rebuild Hpp / Hll diagonals from the linearisation
    e, Jp, Jq = reproj_rj_c(rt(T), X, op, ol, meas, fx, fy, cx, cy, with_point=True)
    s = np.sqrt((e * e).sum(1))
    w = np.where((huber_delta > 0) & (s > huber_delta), huber_delta / np.maximum(s, 1e-300), 1.0)
    dpp = np.zeros((K, 6)); np.add.at(dpp, op, w[:, None] * (Jp * Jp).sum(1))
    dll = np.zeros((X.shape[0], 3)); np.add.at(dll, ol, w[:, None] * (Jq * Jq).sum(1))
    lam = 1e-5 * max(float(dpp[fidx].max(initial=0.0)), float(dll.max(initial=0.0)), 1e-12)
    ni, accepted, lams = 2.0, 0, []
    if len(fidx) == 0:
        return T, X, cost0, cost, 0, lams
    for _ in range(iterations):
        ok = False
        for _trial in range(10):
            red = ba_schur_np(rt(T), X, op, ol, meas, fx, fy, cx, cy, huber_delta, lam)
            Sf = red["S"][np.ix_(fidx, fidx)].transpose(0, 2, 1, 3).reshape(6 * len(fidx), 6 * len(fidx))
            try:
                dpf = np.linalg.solve(Sf, red["rhs"][fidx].reshape(-1)).reshape(-1, 6)
                if not np.isfinite(dpf).all():
                    raise np.linalg.LinAlgError
            except np.linalg.LinAlgError:
                lam, ni = lam * ni, ni * 2
                continue
            dp = np.zeros((K, 6)); dp[fidx] = dpf
            dl = ba_backsub_np(red, op, ol, dp)
            Tn = np.stack([se3_exp_np(dp[k]) @ T[k] for k in range(K)])
            Xn = X + dl
            new = cost_at(Tn, Xn)
            scale = float((dp * (lam * dp - red["bp"])).sum() + (dl * (lam * dl - red["bl"])).sum()) + 1e-3
            rho = (red["cost"] - new) / scale
            if rho > 0 and np.isfinite(new):
                T, X, cost = Tn, Xn, new
                lams.append(lam)
                lam *= max(1.0 / 3.0, min(1.0 - (2.0 * rho - 1.0) ** 3, 2.0 / 3.0))
                ni = 2.0
                accepted += 1
                ok = True
                break
            lam, ni = lam * ni, ni * 2
            if not np.isfinite(lam):
                break
        if not ok:
            break
    return T, X, cost0, cost, accepted, lams


def ba_lm_c(poses12, points, obs_pose, obs_point, meas, fx, fy, cx, cy, iterations: int = 10, fixed_poses=(0,),
            huber_delta: float = 0.0):
    """The same window bundle adjustment as ``ba_lm_np`` in plain C on one host core (oracle/ba_lm_oracle.c: 3x3 inverses
    by cofactors, Gaussian elimination with partial pivoting for the reduced system, a scaling-and-squaring series for the
    exponential): the CPU baseline of that path and a second oracle-side statement that the CPU suite holds against
    ``ba_lm_np``.  Returns (T [K,4,4], X [L,3], cost0, cost, accepted steps, trials)."""
    P = np.ascontiguousarray(poses12, np.float64).reshape(-1, 12)
    X = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    K, L = P.shape[0], X.shape[0]
    op = np.ascontiguousarray(obs_pose, np.int32).reshape(-1)
    ol = np.ascontiguousarray(obs_point, np.int32).reshape(-1)
    m = np.ascontiguousarray(meas, np.float64).reshape(-1, 2)
    fixed = np.zeros(K, np.uint8)
    fixed[list(fixed_poses)] = 1
    Pout, Xout, stats = np.empty((K, 12)), np.empty((L, 3)), np.zeros(4)
    rc = _load().oracle_ba_lm_f64(K, L, op.shape[0], _p(P), _p(X), _p(op), _p(ol), _p(m), _p(fixed), fx, fy, cx, cy,
                                  float(huber_delta), int(iterations), _p(Pout), _p(Xout), _p(stats))
    assert rc == 0
    T = np.tile(np.eye(4), (K, 1, 1))
    T[:, :3, :4] = Pout.reshape(K, 3, 4)
    return T, Xout, float(stats[0]), float(stats[1]), int(stats[2]), int(stats[3])

