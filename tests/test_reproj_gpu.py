"""GPU parity: residual / Jacobian kernels through the C ABI vs the CPU oracle.

Tolerance: north_star asks 1e-6 on residuals; f64 on both sides gives ~1e-10 here (pixel-scale values,
different FMA contraction), asserted at 1e-9 absolute / 1e-12 relative."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
FX, FY, CX, CY = 458.654, 457.296, 367.215, 248.375      # config/orb.yaml:1
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _problem(rng, K, L, O):
    from scipy.spatial.transform import Rotation

    R = Rotation.from_rotvec(rng.uniform(-0.3, 0.3, (K, 3))).as_matrix()
    t = rng.uniform(-5, 5, (K, 3))
    poses = np.concatenate([R, t[:, :, None]], 2).reshape(K, 12)
    pts = np.c_[rng.uniform(-10, 10, (L, 2)), rng.uniform(8, 30, L)]
    op = rng.integers(0, K, O).astype(np.int32)
    ol = rng.integers(0, L, O).astype(np.int32)
    meas = rng.uniform(0, 752, (O, 2)).astype(np.int32).astype(np.float64)    # int-truncated pixels
    return poses, pts, op, ol, meas


def test_golden_vectors(gpu_ctx):
    import slamhip

    g = json.load(open(os.path.join(GOLD, "kat_reproj.json")))
    fx, fy, cx, cy = g["intrinsics"]
    e, Jp, Jq = slamhip.build_linearization(np.array(g["poses12"], float), g["points"], g["obs_pose"], g["obs_point"],
                                            g["meas"], fx, fy, cx, cy)
    assert np.allclose(e, g["e"], rtol=0, atol=1e-12)
    assert np.allclose(Jp, g["Jpose"], rtol=0, atol=1e-12)
    assert np.allclose(Jq, g["Jpoint"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("O", [1, 255, 256, 257, 1000, 100003])
@pytest.mark.parametrize("with_point", [True, False])
def test_rj_matches_oracle(gpu_ctx, O, with_point):
    import slamhip
    from oracle import oracle

    poses, pts, op, ol, meas = _problem(np.random.default_rng(O), 7, 501, O)
    e, Jp, Jq = slamhip.build_linearization(poses, pts, op, ol, meas, FX, FY, CX, CY, with_point)
    re, rJp, rJq = oracle.reproj_rj_c(poses, pts, op, ol, meas, FX, FY, CX, CY, with_point, threads=8)
    assert np.allclose(e, re, rtol=0, atol=1e-9)
    assert np.allclose(Jp, rJp, rtol=1e-12, atol=1e-9)
    if with_point:
        assert np.allclose(Jq, rJq, rtol=1e-12, atol=1e-9)
    else:
        assert Jq is None


def test_accepts_4x4_poses_and_empty(gpu_ctx):
    import slamhip
    from backend import Backend

    poses, pts, op, ol, meas = _problem(np.random.default_rng(1), 3, 10, 20)
    T = np.tile(np.eye(4), (3, 1, 1))
    T[:, :3, :4] = poses.reshape(3, 3, 4)
    a = Backend().build_linearization(T, pts, op, ol, meas, FX, FY, CX, CY)
    b = slamhip.build_linearization(poses, pts, op, ol, meas, FX, FY, CX, CY)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    e, Jp, Jq = slamhip.build_linearization(poses, pts, [], [], np.zeros((0, 2)), FX, FY, CX, CY)
    assert e.shape == (0, 2) and Jp.shape == (0, 2, 6) and Jq.shape == (0, 2, 3)
    with pytest.raises(ValueError):
        slamhip.build_linearization(poses, pts, [5], [0], np.zeros((1, 2)), FX, FY, CX, CY)


@pytest.mark.parametrize("O", [1, 64, 200, 5000, 70001])
@pytest.mark.parametrize("delta", [0.0, 1.0])
def test_pose_normal_equations(gpu_ctx, O, delta):
    import slamhip
    from oracle import oracle

    rng = np.random.default_rng(O)
    poses, pts, _, _, _ = _problem(rng, 1, O, 1)
    e0 = oracle.reproj_rj_c(poses, pts, np.zeros(O, np.int32), np.arange(O, dtype=np.int32), np.zeros((O, 2)),
                            FX, FY, CX, CY)[0]
    meas = -e0 + rng.normal(0, 2.0, (O, 2))              # projections + noise: a mix of inliers and Huber-weighted rows
    act = (rng.uniform(size=O) > 0.1).astype(np.uint8)
    prob = slamhip.PoseOnlyProblem(gpu_ctx, pts, meas, (FX, FY, CX, CY))
    prob.set_active(act)
    H, b, chi2 = prob.normal_equations(poses[0], delta)
    prob.free()
    rH, rb, rchi2 = oracle.pose_normal_eq_c(poses[0], pts, meas, act, FX, FY, CX, CY, delta)
    scale = np.abs(rH).max()
    assert np.allclose(H, rH, rtol=1e-10, atol=1e-10 * scale)
    assert np.allclose(b, rb, rtol=1e-10, atol=1e-10 * np.abs(rb).max())
    assert np.allclose(chi2, rchi2, rtol=1e-12, atol=1e-12)
    assert np.array_equal(H, H.T)


def test_normal_equations_are_deterministic(gpu_ctx):
    import slamhip

    rng = np.random.default_rng(9)
    poses, pts, _, _, _ = _problem(rng, 1, 30000, 1)
    meas = rng.uniform(0, 700, (30000, 2))
    prob = slamhip.PoseOnlyProblem(gpu_ctx, pts, meas, (FX, FY, CX, CY))
    a = prob.normal_equations(poses[0], 1.0)
    b = prob.normal_equations(poses[0], 1.0)
    prob.free()
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_dense_ba_size_properties(gpu_ctx):
    """BASELINE configs[4] scale (200 poses x 50k points, 1e7 observations): linearity / consistency properties."""
    import slamhip
    from oracle import oracle

    rng = np.random.default_rng(228)
    K, L = 200, 50000
    poses, pts, _, _, _ = _problem(rng, K, L, 1)
    op = np.repeat(np.arange(K, dtype=np.int32), L)
    ol = np.tile(np.arange(L, dtype=np.int32), K)
    meas = np.zeros((K * L, 2))
    prob = slamhip.ReprojProblem(gpu_ctx, poses, pts, op, ol, meas, (FX, FY, CX, CY))
    prob.linearize()
    e, Jp, Jq = prob.download()
    prob.free()
    assert np.isfinite(e).all() and np.isfinite(Jp).all() and np.isfinite(Jq).all()
    # with zero measurements e = -projection: shifting the measurement shifts e by exactly the same amount
    sel = rng.choice(K * L, 4096, replace=False)
    re, rJp, rJq = oracle.reproj_rj_c(poses, pts, op[sel], ol[sel], meas[sel], FX, FY, CX, CY, threads=8)
    assert np.allclose(e[sel], re, rtol=0, atol=1e-9) and np.allclose(Jp[sel], rJp, rtol=1e-12, atol=1e-9)
    assert np.allclose(Jq[sel], rJq, rtol=1e-12, atol=1e-9)
    # structural zeros of the pose Jacobian (frontend.py:288-289) and J_point = -A R consistency: J_pose[:, 3:6] = -A
    assert (Jp[:, 0, 4] == 0).all() and (Jp[:, 1, 3] == 0).all()
    R = poses.reshape(K, 3, 4)[:, :, :3]
    A = -Jp[sel][:, :, 3:6]
    assert np.allclose(Jq[sel], -np.einsum("oij,ojk->oik", A, R[op[sel]]), rtol=1e-10, atol=1e-9)
