"""Pose-only optimisation on top of the GPU normal-equation kernel.

Mirrors the structure of ``Frontend._correct_current_pose`` (``frontend.py:298-393``): one pose
vertex, one 2-D reprojection edge per feature with a map point (information I2, Huber kernel),
four outer rounds of ten Levenberg-Marquardt iterations; every round restarts from the frame's
pose (``frontend.py:360``), then edges are classified by ``chi2 > 5.991**2`` (``:356,371-377``),
outliers leave the optimisation (g2o "level 1"), and after round index 2 the robust kernel is
dropped (``:378-379``).  Residuals, Jacobians, Huber weights and the 6x6 system are evaluated on the
GPU (``slam_pose_normal_eq_f64``); the 6x6 solve and the LM control flow stay on the host.

PARITY UNPINNED: the reference runs this inside g2o-python 0.0.12 (absent here), with a
``VertexSE3`` whose update convention does not match the Jacobian it is given (SURVEY.md §8a
notes).  This module uses the convention the Jacobian of ``frontend.py:288-291`` is the
derivative for — left perturbation ``T <- exp([w, v]) T`` with rotation first, as in
``g2o::VertexSE3Expmap`` — and g2o's published LM schedule (tau = 1e-5, rho-based lambda update,
at most ten failed trials).  Solver trajectories are therefore not claimed to match the reference;
the residual/Jacobian arithmetic underneath is what the parity tests cover.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from .device import Context, default_context
from .reproj import PoseOnlyProblem

CHI2_THRESHOLD = 5.991 ** 2     # frontend.py:356 (sic: the reference squares the 95 % chi2 quantile)
HUBER_DELTA = 1.0               # g2o RobustKernelHuber default, frontend.py:350


def _hat(w: np.ndarray) -> np.ndarray:
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


def se3_exp(xi: Sequence[float]) -> np.ndarray:
    """exp of a twist [w (rotation), v (translation)] as a 4x4 matrix (g2o SE3Quat::exp ordering)."""
    xi = np.asarray(xi, np.float64)
    w, v = xi[:3], xi[3:]
    th = float(np.linalg.norm(w))
    W = _hat(w)
    if th < 1e-10:
        R = np.eye(3) + W + 0.5 * W @ W
        V = np.eye(3) + 0.5 * W + W @ W / 6.0
    else:
        a, b, c = np.sin(th) / th, (1 - np.cos(th)) / th**2, (th - np.sin(th)) / th**3
        R = np.eye(3) + a * W + b * W @ W
        V = np.eye(3) + b * W + c * W @ W
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ v
    return T


def _robust_chi2(chi2: np.ndarray, delta: float) -> float:
    """Sum of Huber rho(e2): e2 inside delta, 2*delta*|e| - delta^2 outside (g2o RobustKernelHuber)."""
    if delta <= 0:
        return float(chi2.sum())
    e = np.sqrt(chi2)
    return float(np.where(e <= delta, chi2, 2 * delta * e - delta * delta).sum())


@dataclass
class PoseOptResult:
    pose: np.ndarray          # 4x4 Tcw after the last round
    inliers: np.ndarray       # bool [O]: chi2 <= threshold after the last round
    chi2: np.ndarray          # [O] at the returned pose
    n_inliers: int            # what _correct_current_pose returns (frontend.py:393)
    iterations: int           # accepted LM steps over all rounds


def _lm_round(prob: PoseOnlyProblem, T: np.ndarray, active: np.ndarray, delta: float, iterations: int):
    """Ten LM iterations (g2o OptimizationAlgorithmLevenberg schedule) on the active edges."""
    prob.set_active(active.astype(np.uint8))
    H, b, chi2 = prob.normal_equations(T[:3, :4].reshape(12), delta)
    cur = _robust_chi2(chi2[active], delta)
    lam = 1e-5 * max(float(np.max(np.diag(H))), 1e-12)      # tau * max diagonal
    ni = 2.0
    accepted = 0
    for _ in range(iterations):
        if not active.any():
            break
        ok = False
        for _trial in range(10):                                # maxTrialsAfterFailure
            try:
                dx = np.linalg.solve(H + lam * np.eye(6), -b)   # b = sum w J^T e, step solves H dx = -b
            except np.linalg.LinAlgError:
                lam *= ni
                ni *= 2
                continue
            Tn = se3_exp(dx) @ T
            Hn, bn, chi2n = prob.normal_equations(Tn[:3, :4].reshape(12), delta)
            new = _robust_chi2(chi2n[active], delta)
            scale = float(dx @ (lam * dx - b)) + 1e-3
            rho = (cur - new) / scale
            if rho > 0 and np.isfinite(new):
                T, H, b, chi2, cur = Tn, Hn, bn, chi2n, new
                alpha = min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0)
                lam *= max(1.0 / 3.0, alpha)
                ni = 2.0
                ok = True
                accepted += 1
                break
            lam *= ni
            ni *= 2
            if not np.isfinite(lam):
                break
        if not ok:
            break
    return T, chi2, accepted


def optimize_pose_only(pose, points, meas, intrinsics, rounds: int = 4, iterations: int = 10,
                       chi2_threshold: float = CHI2_THRESHOLD, huber_delta: float = HUBER_DELTA,
                       ctx: Optional[Context] = None) -> PoseOptResult:
    """Refine one camera pose against fixed 3-D points (the job of ``_correct_current_pose``).

    ``pose``: 4x4 (or 3x4) Tcw; ``points`` [O,3] map-point positions; ``meas`` [O,2] pixel positions
    (the reference passes int-truncated pixels, ``frontend.py:348``); ``intrinsics`` (fx, fy, cx, cy)."""
    ctx = ctx or default_context()
    T0 = np.eye(4)
    P = np.asarray(pose, np.float64)
    T0[:3, :4] = P.reshape(-1, 4)[:3, :4] if P.size != 12 else P.reshape(3, 4)
    prob = PoseOnlyProblem(ctx, points, meas, intrinsics)
    O = prob.O
    active = np.ones(O, bool)
    delta = huber_delta
    T, chi2 = T0, np.zeros(O)
    total = 0
    try:
        for it in range(rounds):
            T, chi2, acc = _lm_round(prob, T0.copy(), active, delta, iterations)   # every round restarts from the input pose
            total += acc
            active = chi2 <= chi2_threshold        # chi2 > threshold -> outlier, level 1 (frontend.py:371-377)
            if it == 2:
                delta = 0.0                        # set_robust_kernel(None) (frontend.py:378-379)
    finally:
        prob.free()
    return PoseOptResult(pose=T, inliers=active, chi2=chi2, n_inliers=int(active.sum()), iterations=total)


def optimize_pose_only_device(pose, points, meas, intrinsics, rounds: int = 4, iterations: int = 10,
                              chi2_threshold: float = CHI2_THRESHOLD, huber_delta: float = HUBER_DELTA,
                              ctx: Optional[Context] = None) -> PoseOptResult:
    """Same job as ``optimize_pose_only`` with the whole LM loop on the GPU: ONE kernel launch between one
    upload and one download (``slam_pose_optimize_host_f64``) instead of one launch + PCIe round trip per LM trial."""
    import ctypes  # noqa: F401  (ctypes types come through the binding)

    from ._lib import check

    ctx = ctx or default_context()
    T0 = np.eye(4)
    P = np.asarray(pose, np.float64)
    T0[:3, :4] = P.reshape(-1, 4)[:3, :4] if P.size != 12 else P.reshape(3, 4)
    points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    meas = np.ascontiguousarray(meas, np.float64).reshape(-1, 2)
    O = points.shape[0]
    if meas.shape[0] != O:
        raise ValueError("one measurement per point")
    fx, fy, cx, cy = (float(v) for v in intrinsics)
    T = np.eye(4)
    out12 = np.empty(12, np.float64)
    inl = np.zeros(O, np.uint8)
    chi2 = np.zeros(O, np.float64)
    stats = np.zeros(2, np.int32)
    pin = np.ascontiguousarray(T0[:3, :4].reshape(12))
    from ._lib import addr

    check(ctx.lib.slam_pose_optimize_host_f64(ctx.handle, addr(pin), addr(points) if O else None, addr(meas) if O else None, O,
                                              fx, fy, cx, cy, int(rounds), int(iterations), float(chi2_threshold), float(huber_delta),
                                              addr(out12), addr(inl) if O else None, addr(chi2) if O else None, addr(stats)))
    T[:3, :4] = out12.reshape(3, 4)
    return PoseOptResult(pose=T, inliers=inl.astype(bool), chi2=chi2, n_inliers=int(stats[0]), iterations=int(stats[1]))


def optimize_poses_batch(poses, points_list, meas_list, intrinsics, rounds: int = 4, iterations: int = 10,
                         chi2_threshold: float = CHI2_THRESHOLD, huber_delta: float = HUBER_DELTA,
                         ctx: Optional[Context] = None):
    """``optimize_pose_only_device`` for several independent frames in ONE kernel launch (one workgroup per frame,
    ``slam_pose_optimize_batch_f64``): ``poses`` [B,4,4] (or [B,3,4]), ``points_list[b]`` [O_b,3], ``meas_list[b]`` [O_b,2].
    Returns a list of ``PoseOptResult``.  The reference refines one frame per call (``frontend.py:298-393``); this is
    for the keyframes of a window against the fixed map, or relocalisation candidates."""
    from ._lib import check

    ctx = ctx or default_context()
    P = np.asarray(poses, np.float64)
    B = P.shape[0]
    if len(points_list) != B or len(meas_list) != B:
        raise ValueError("one point array and one pixel array per pose")
    if B == 0:
        return []
    pin = np.ascontiguousarray(P.reshape(B, -1, 4)[:, :3, :4].reshape(B, 12))
    pts = [np.ascontiguousarray(p, np.float64).reshape(-1, 3) for p in points_list]
    mes = [np.ascontiguousarray(m, np.float64).reshape(-1, 2) for m in meas_list]
    if any(p.shape[0] != m.shape[0] for p, m in zip(pts, mes)):
        raise ValueError("one measurement per point in every frame")
    off = np.zeros(B + 1, np.int32)
    off[1:] = np.cumsum([p.shape[0] for p in pts])
    O = int(off[-1])
    fx, fy, cx, cy = (float(v) for v in intrinsics)
    bufs = []
    try:
        d_pose = ctx.upload(pin); bufs.append(d_pose)
        d_pts = ctx.upload(np.concatenate(pts) if O else np.zeros((1, 3))); bufs.append(d_pts)
        d_mes = ctx.upload(np.concatenate(mes) if O else np.zeros((1, 2))); bufs.append(d_mes)
        d_off = ctx.upload(off); bufs.append(d_off)
        d_out = ctx.malloc(B * 96); bufs.append(d_out)
        d_inl = ctx.malloc(max(O, 1)); bufs.append(d_inl)
        d_chi = ctx.malloc(max(O, 1) * 8); bufs.append(d_chi)
        d_st = ctx.malloc(B * 8); bufs.append(d_st)
        check(ctx.lib.slam_pose_optimize_batch_f64(ctx.handle, B, d_pose.ptr, d_pts.ptr, d_mes.ptr, d_off.ptr, O, fx, fy, cx, cy,
                                                   int(rounds), int(iterations), float(chi2_threshold), float(huber_delta),
                                                   d_out.ptr, d_inl.ptr, d_chi.ptr, d_st.ptr))
        out = d_out.download(np.float64, (B, 3, 4))
        inl = d_inl.download(np.uint8, (max(O, 1),)).astype(bool)
        chi = d_chi.download(np.float64, (max(O, 1),))
        st = d_st.download(np.int32, (B, 2))
    finally:
        for b in bufs:
            b.free()
    res = []
    for b in range(B):
        T = np.eye(4)
        T[:3, :4] = out[b]
        a, e = int(off[b]), int(off[b + 1])
        res.append(PoseOptResult(pose=T, inliers=inl[a:e].copy(), chi2=chi[a:e].copy(), n_inliers=int(st[b, 0]), iterations=int(st[b, 1])))
    return res
