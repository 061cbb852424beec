"""CPU suite: the oracle against the hand-derived golden vectors, and its two restatements against each other."""
import json
import os

import numpy as np
import pytest

from oracle import oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")
INT_MAX = 2**31 - 1


def gold(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def d(rows):
    return np.array(rows, np.uint8).reshape(-1, 32)


@pytest.mark.parametrize("knn", [oracle.bf_knn_c, oracle.bf_knn_np])
def test_ladder_ties_short(knn):
    g = gold("kat_ladder.json")
    idx, dist = knn(d(g["query"]), d(g["train"]), 2)
    assert idx.tolist() == g["idx"] and dist.tolist() == g["dist"]
    g = gold("kat_ties.json")
    idx, dist = knn(d(g["query"]), d(g["train"]), 2)
    assert idx.tolist() == g["idx"] and dist.tolist() == g["dist"]
    g = gold("kat_short_train.json")
    idx, dist = knn(d(g["query"]), d(g["train_one"]), 2)
    assert idx.tolist() == g["idx_one"] and dist.tolist() == g["dist_one"]
    idx, dist = knn(d(g["query"]), np.zeros((0, 32), np.uint8), 2)
    assert idx.tolist() == g["idx_none"] and dist.tolist() == g["dist_none"]


@pytest.mark.parametrize("match", [oracle.bf_match_c, oracle.bf_match_np])
def test_reference_filter(match):
    g = gold("kat_filter.json")
    for case in g["cases"]:
        q, t, dist = match(d(g["source"]), d(g["query"]), case["thr"])
        assert q.tolist() == case["queryIdx"], case
        assert t.tolist() == case["trainIdx"], case
        assert dist.tolist() == case["distance"], case
    # M = 0 / N = 0: bf.match returns no DMatch at all
    assert len(match(np.zeros((0, 32), np.uint8), d(g["query"]))[0]) == 0
    assert len(match(d(g["source"]), np.zeros((0, 32), np.uint8))[0]) == 0


@pytest.mark.parametrize("cc", [oracle.bf_cross_check_c, oracle.bf_cross_check_np])
def test_cross_check(cc):
    g = gold("kat_cross_check.json")
    oi, od = cc(d(g["query"]), d(g["train"]))
    assert oi.tolist() == g["out_idx"] and od.tolist() == g["out_dist"]
    oi, od = cc(d(g["query2"]), d(g["train"]))
    assert oi.tolist() == g["out_idx2"] and od.tolist() == g["out_dist2"]
    oi, od = cc(d(g["query3"]), d(g["train3"]))
    assert oi.tolist() == g["out_idx3"] and od.tolist() == g["out_dist3"]
    # no train rows / no query rows: nothing to pair
    oi, od = cc(d(g["query"]), np.zeros((0, 32), np.uint8))
    assert oi.tolist() == [-1, -1] and od.tolist() == [INT_MAX, INT_MAX]
    assert cc(np.zeros((0, 32), np.uint8), d(g["train"]))[0].shape == (0,)


def test_ratio():
    g = gold("kat_ratio.json")
    idx, dist = oracle.bf_knn_c(d(g["query"]), d(g["train"]), 2)
    assert oracle.bf_ratio_c(idx, dist, 0.75).tolist() == g["keep_075"]
    assert oracle.bf_ratio_c(idx, dist, 0.5).tolist() == g["keep_050"]


def test_multi_image():
    g = gold("kat_multi_image.json")
    imgs = [d(i) for i in g["images"]]
    img, tr, dist = oracle.bf_knn_multi_c(d(g["query"]), imgs, 2)
    assert img.tolist() == g["img"] and tr.tolist() == g["train"] and dist.tolist() == g["dist"]


@pytest.mark.parametrize("n,m", [(1, 1), (2, 1), (63, 65), (64, 64), (65, 63), (200, 200), (513, 1031)])
def test_c_and_numpy_restatements_agree(n, m):
    rng = np.random.default_rng(n * 1000 + m)
    q = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (m, 32), dtype=np.uint8)
    if m > 10:
        t[m // 2] = t[3]            # duplicate row
        q[0] = t[3]
    for k in (1, 2, 3):
        ic, dc = oracle.bf_knn_c(q, t, k, threads=2)
        inp, dnp = oracle.bf_knn_np(q, t, k)
        assert np.array_equal(ic, inp) and np.array_equal(dc, dnp)
    a = oracle.bf_cross_check_c(q, t)
    b = oracle.bf_cross_check_np(q, t)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for thr in (None, 60.0, 100.0):
        a = oracle.bf_match_c(t, q, thr)
        b = oracle.bf_match_np(t, q, thr)
        assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_threads_do_not_change_results():
    rng = np.random.default_rng(0)
    q = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (700, 32), dtype=np.uint8)
    a = oracle.bf_knn_c(q, t, 2, threads=1)
    b = oracle.bf_knn_c(q, t, 2, threads=8)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


# ---- reprojection -----------------------------------------------------------
def test_reproj_golden():
    g = gold("kat_reproj.json")
    fx, fy, cx, cy = g["intrinsics"]
    for fn in (oracle.reproj_rj_c, oracle.reproj_rj_np):
        e, Jp, Jq = fn(np.array(g["poses12"], float), np.array(g["points"], float), g["obs_pose"], g["obs_point"],
                       np.array(g["meas"], float), fx, fy, cx, cy)[:3]
        assert np.allclose(e, g["e"], rtol=0, atol=1e-12)
        assert np.allclose(Jp, g["Jpose"], rtol=0, atol=1e-12)
        assert np.allclose(Jq, g["Jpoint"], rtol=0, atol=1e-12)


def _random_problem(rng, K=6, L=50, O=300):
    from scipy.spatial.transform import Rotation

    R = Rotation.from_rotvec(rng.uniform(-0.3, 0.3, (K, 3))).as_matrix()
    t = rng.uniform(-1, 1, (K, 3))
    poses = np.concatenate([R, t[:, :, None]], 2).reshape(K, 12)
    pts = np.c_[rng.uniform(-3, 3, (L, 2)), rng.uniform(4, 20, L)]
    op = rng.integers(0, K, O).astype(np.int32)
    ol = rng.integers(0, L, O).astype(np.int32)
    meas = rng.uniform(0, 700, (O, 2)).astype(np.int32).astype(float)
    return poses, pts, op, ol, meas


def test_reproj_c_vs_numpy_and_finite_differences():
    from scipy.spatial.transform import Rotation

    rng = np.random.default_rng(3)
    fx, fy, cx, cy = 458.654, 457.296, 367.215, 248.375
    poses, pts, op, ol, meas = _random_problem(rng)
    ec, Jpc, Jqc = oracle.reproj_rj_c(poses, pts, op, ol, meas, fx, fy, cx, cy)
    en, Jpn, Jqn = oracle.reproj_rj_np(poses, pts, op, ol, meas, fx, fy, cx, cy)
    assert np.allclose(ec, en, rtol=0, atol=1e-10) and np.allclose(Jpc, Jpn, rtol=1e-13, atol=1e-10)
    assert np.allclose(Jqc, Jqn, rtol=1e-13, atol=1e-10)

    def err(P12, p, m):
        e, _, _ = oracle.reproj_rj_c(P12[None], p[None], [0], [0], m[None], fx, fy, cx, cy)
        return e[0]

    h = 1e-6
    for o in range(0, 300, 37):
        P, p, m = poses[op[o]].copy(), pts[ol[o]].copy(), meas[o]
        # point Jacobian: plain finite differences
        for c in range(3):
            dp = np.zeros(3); dp[c] = h
            fd = (err(P, p + dp, m) - err(P, p - dp, m)) / (2 * h)
            assert np.allclose(fd, Jqc[o][:, c], rtol=1e-6, atol=1e-5)
        # pose Jacobian: left perturbation exp([w, v]) * T, rotation columns first (frontend.py:288-291)
        T = np.eye(4); T[:3, :4] = P.reshape(3, 4)
        for c in range(6):
            xi = np.zeros(6); xi[c] = h
            def pert(s):
                D = np.eye(4)
                D[:3, :3] = Rotation.from_rotvec(s * xi[:3]).as_matrix()
                D[:3, 3] = s * xi[3:]
                return (D @ T)[:3, :4].reshape(12)
            fd = (err(pert(1.0), p, m) - err(pert(-1.0), p, m)) / (2 * h)
            assert np.allclose(fd, Jpc[o][:, c], rtol=1e-5, atol=1e-4), (o, c, fd, Jpc[o][:, c])


def test_pose_normal_equations_oracle():
    rng = np.random.default_rng(5)
    fx, fy, cx, cy = 458.654, 457.296, 367.215, 248.375
    poses, pts, _, _, _ = _random_problem(rng, K=1, L=120, O=1)
    meas = rng.uniform(0, 700, (120, 2))
    act = (rng.uniform(size=120) > 0.2).astype(np.uint8)
    for delta in (0.0, 1.0):
        H, b, chi2 = oracle.pose_normal_eq_c(poses[0], pts, meas, act, fx, fy, cx, cy, delta)
        e, J, _ = oracle.reproj_rj_np(poses, pts, np.zeros(120, int), np.arange(120), meas, fx, fy, cx, cy)
        w = np.ones(120)
        if delta > 0:
            n = np.sqrt((e**2).sum(1))
            w = np.where(n > delta, delta / n, 1.0)
        w = w * act
        Href = np.einsum("o,oia,oib->ab", w, J, J)
        bref = np.einsum("o,oia,oi->a", w, J, e)
        assert np.allclose(H, Href, rtol=1e-12) and np.allclose(b, bref, rtol=1e-12)
        assert np.allclose(chi2, (e**2).sum(1), rtol=1e-13)


def test_pose_lm_c_and_numpy_statements_agree():
    """oracle/pose_lm_oracle.c (Gaussian elimination, series exponential; the CPU baseline of the pose refinement) and
    oracle.pose_lm_np (numpy solve, scipy expm) restate Frontend._correct_current_pose (frontend.py:298-393)
    independently: same pose to 1e-9, the same inlier sets, chi2 to 1e-7, on frames with gross outliers, other
    schedules, the empty frame and a frame whose every edge is an outlier."""
    from scipy.spatial.transform import Rotation

    FX, FY, CX, CY = 458.654, 457.296, 367.215, 248.375
    for seed, O, rounds, iters, thr, delta in ((228, 200, 4, 10, 5.991 ** 2, 1.0), (1, 64, 4, 10, 5.991 ** 2, 1.0),
                                               (2, 700, 2, 3, 9.0, 0.0), (3, 12, 4, 10, 5.991, 2.5), (4, 150, 3, 1, 35.89, 1.0)):
        rng = np.random.default_rng(seed)
        T = np.eye(4)
        T[:3, :3] = Rotation.from_rotvec(rng.uniform(-0.15, 0.15, 3)).as_matrix()
        T[:3, 3] = rng.uniform(-0.5, 0.5, 3)
        X = np.c_[rng.uniform(-4, 4, (O, 2)), rng.uniform(6, 15, O)]
        pc = X @ T[:3, :3].T + T[:3, 3]
        meas = np.c_[FX * pc[:, 0] / pc[:, 2] + CX, FY * pc[:, 1] / pc[:, 2] + CY] + rng.normal(0, 0.4, (O, 2))
        meas[::7] += rng.uniform(40, 120, (len(meas[::7]), 2))
        meas = meas.astype(np.int32).astype(np.float64)
        T_init = oracle.se3_exp_np(rng.normal(0, 0.02, 6)) @ T
        Tn, inl_n, chi_n, acc_n = oracle.pose_lm_np(T_init, X, meas, FX, FY, CX, CY, rounds, iters, thr, delta)
        Tc, inl_c, chi_c, acc_c = oracle.pose_lm_c(T_init[:3, :4], X, meas, FX, FY, CX, CY, rounds, iters, thr, delta)
        assert np.allclose(Tc.reshape(3, 4), Tn[:3, :4], rtol=0, atol=1e-9), (seed, np.abs(Tc.reshape(3, 4) - Tn[:3, :4]).max())
        assert np.array_equal(inl_c, inl_n) and np.allclose(chi_c, chi_n, rtol=1e-7, atol=1e-7)
        assert abs(acc_c - acc_n) <= 8
    # no edges: the pose comes back unchanged, no inliers
    Tc, inl_c, _, acc = oracle.pose_lm_c(np.eye(4)[:3, :4], np.zeros((0, 3)), np.zeros((0, 2)), FX, FY, CX, CY)
    assert np.array_equal(Tc.reshape(3, 4), np.eye(4)[:3, :4]) and inl_c.size == 0 and acc == 0


def _ba_window(rng, K, L, density, noise=0.5):
    from scipy.spatial.transform import Rotation

    FX, FY, CX, CY = 458.654, 457.296, 367.215, 248.375
    T = np.tile(np.eye(4), (K, 1, 1))
    for k in range(K):
        T[k, :3, :3] = Rotation.from_rotvec(rng.uniform(-0.1, 0.1, 3)).as_matrix()
        T[k, :3, 3] = rng.uniform(-0.6, 0.6, 3)
    X = np.c_[rng.uniform(-4, 4, (L, 2)), rng.uniform(6, 15, L)]
    vis = rng.uniform(size=(K, L)) < density
    vis[0, :] = True
    op, ol = np.nonzero(vis)
    pc = np.einsum("oij,oj->oi", T[op, :3, :3], X[ol]) + T[op, :3, 3]
    meas = np.c_[FX * pc[:, 0] / pc[:, 2] + CX, FY * pc[:, 1] / pc[:, 2] + CY] + rng.normal(0, noise, (len(op), 2))
    meas = meas.astype(np.int32).astype(np.float64)
    T0 = np.stack([oracle.se3_exp_np(rng.normal(0, 0.01, 6)) @ T[k] for k in range(K)])
    X0 = X + rng.normal(0, 0.05, X.shape)
    return T0, X0, op.astype(np.int32), ol.astype(np.int32), meas, (FX, FY, CX, CY)


def test_ba_lm_c_and_numpy_statements_agree():
    """oracle/ba_lm_oracle.c (cofactor inverses, Gaussian elimination, series exponential; the CPU baseline of the window
    bundle adjustment) and oracle.ba_lm_np (numpy inverses and solver, scipy expm) state the same Schur-complement LM
    independently: the same number of accepted steps, costs to 1e-9 relative, poses to 1e-8, points to 1e-7 - with and
    without the Huber kernel, with several fixed poses, with a point nobody observes and with nothing free to move."""
    for seed, K, L, delta, fixed, iters in ((1, 4, 60, 0.0, (0,), 6), (2, 7, 120, 1.0, (0, 1), 5), (3, 3, 25, 2.0, (0,), 10),
                                            (4, 5, 40, 0.0, (0, 1, 2, 3, 4), 4)):
        rng = np.random.default_rng(seed)
        T0, X0, op, ol, meas, (fx, fy, cx, cy) = _ba_window(rng, K, L, 0.6)
        if seed == 3:                                   # a point nobody observes stays where it is
            keep = ol != 7
            op, ol, meas = op[keep], ol[keep], meas[keep]
        P0 = np.ascontiguousarray(T0[:, :3, :4]).reshape(K, 12)
        Tn, Xn, c0n, cn, accn, _ = oracle.ba_lm_np(P0, X0, op, ol, meas, fx, fy, cx, cy, iters, fixed, delta)
        Tc, Xc, c0c, cc, accc, trials = oracle.ba_lm_c(P0, X0, op, ol, meas, fx, fy, cx, cy, iters, fixed, delta)
        assert accc == accn and trials >= accc, (seed, accc, accn)
        assert abs(c0c - c0n) <= 1e-9 * max(1.0, c0n) and abs(cc - cn) <= 1e-9 * max(1.0, cn), (seed, c0c, c0n, cc, cn)
        assert np.abs(Tc - Tn).max() <= 1e-8 and np.abs(Xc - Xn).max() <= 1e-7, (seed, np.abs(Tc - Tn).max(), np.abs(Xc - Xn).max())
        if len(fixed) < K:
            assert cc < c0c                             # it did optimise
        else:
            assert accc == 0 and np.array_equal(Tc, T0) and np.array_equal(Xc, X0)
        if seed == 3:
            assert np.array_equal(Xc[7], X0[7])
