"""Windowed bundle adjustment on top of the GPU residual/Jacobian kernel.

The reference has no bundle adjustment (``Backend`` is an empty class, ``backend.py:101-103``; ``Map``
keeps a window of 7 keyframes, ``backend.py:11``); this is the extension BASELINE.json calls
``optimize()``.  Every LM iteration evaluates e, J_pose (2x6) and J_point (2x3) for all observations on
the GPU (``slam_reproj_rj_f64``); the Schur complement on the landmark blocks and the reduced camera
solve run on the host in numpy, which is adequate for keyframe windows (tens of poses, 10^4-10^5
observations).  No reference function or test pins this code: it is checked against synthetic scenes
and finite differences only.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from .device import Context, default_context
from .pose_opt import se3_exp
from .reproj import ReprojProblem, poses_to_rt12


@dataclass
class BAResult:
    poses: np.ndarray        # [K,4,4] Tcw
    points: np.ndarray       # [L,3]
    chi2_initial: float
    chi2_final: float
    iterations: int          # accepted LM steps


def _huber_weights(e: np.ndarray, delta: float) -> np.ndarray:
    if delta <= 0:
        return np.ones(e.shape[0])
    n = np.sqrt((e * e).sum(1))
    return np.where(n > delta, delta / np.maximum(n, 1e-300), 1.0)


def _robust_cost(e: np.ndarray, delta: float) -> float:
    c2 = (e * e).sum(1)
    if delta <= 0:
        return float(c2.sum())
    n = np.sqrt(c2)
    return float(np.where(n <= delta, c2, 2 * delta * n - delta * delta).sum())


def bundle_adjust(poses, points, obs_pose_idx, obs_point_idx, meas, intrinsics, iterations: int = 10,
                  fixed_poses: Sequence[int] = (0,), huber_delta: float = 0.0,
                  ctx: Optional[Context] = None) -> BAResult:
    ctx = ctx or default_context()
    P = np.asarray(poses, np.float64)
    T = np.tile(np.eye(4), (P.shape[0], 1, 1))
    T[:, :3, :4] = (P.reshape(-1, 12) if P.ndim == 2 else poses_to_rt12(P)).reshape(-1, 3, 4)
    X = np.array(points, np.float64).reshape(-1, 3)
    op = np.ascontiguousarray(obs_pose_idx, np.int32)
    ol = np.ascontiguousarray(obs_point_idx, np.int32)
    K, L, O = T.shape[0], X.shape[0], op.shape[0]
    free = np.ones(K, bool)
    free[list(fixed_poses)] = False
    prob = ReprojProblem(ctx, T[:, :3, :4].reshape(K, 12), X, op, ol, meas, intrinsics, with_point=True)

    def linearize(Tc, Xc):
        prob.set_poses(Tc[:, :3, :4].reshape(K, 12))
        prob.set_points(Xc)
        prob.linearize()
        return prob.download()

    try:
        e, Jp, Jq = linearize(T, X)
        cost = _robust_cost(e, huber_delta)
        cost0 = cost
        lam, ni, accepted = None, 2.0, 0
        for _ in range(iterations):
            w = _huber_weights(e, huber_delta)
            Jpw = Jp * w[:, None, None]
            Hpp = np.zeros((K, 6, 6)); bp = np.zeros((K, 6)); Hll = np.zeros((L, 3, 3)); bl = np.zeros((L, 3))
            np.add.at(Hpp, op, np.einsum("oia,oib->oab", Jpw, Jp))
            np.add.at(bp, op, np.einsum("oia,oi->oa", Jpw, e))
            np.add.at(Hll, ol, np.einsum("oia,oib->oab", Jq * w[:, None, None], Jq))
            np.add.at(bl, ol, np.einsum("oia,oi->oa", Jq * w[:, None, None], e))
            Hpl = np.einsum("oia,oib->oab", Jpw, Jq)                     # [O,6,3]
            if lam is None:
                lam = 1e-5 * max(Hpp[free].reshape(-1, 36)[:, ::7].max(initial=0.0), Hll.reshape(-1, 9)[:, ::4].max(initial=0.0), 1e-12)
            step_ok = False
            for _trial in range(10):
                Hll_d = Hll + lam * np.eye(3)
                seen = np.zeros(L, bool); seen[ol] = True
                Hll_d[~seen] = np.eye(3)
                Hll_inv = np.linalg.inv(Hll_d)
                Y = np.einsum("oab,obc->oac", Hpl, Hll_inv[ol])           # Hpl Hll^-1  [O,6,3]
                S = np.zeros((K, K, 6, 6))
                for k in range(K):
                    S[k, k] = Hpp[k] + lam * np.eye(6)
                # S[k1,k2] -= sum over landmarks seen by both of Y_o1 Hpl_o2^T : group observations by landmark
                order = np.argsort(ol, kind="stable")
                starts = np.r_[0, np.flatnonzero(np.diff(ol[order])) + 1, O]
                for a, b in zip(starts[:-1], starts[1:]):
                    obs = order[a:b]
                    blk = np.einsum("iab,jcb->ijac", Y[obs], Hpl[obs])   # [n,n,6,6]
                    S[np.ix_(op[obs], op[obs])] -= blk
                rhs = -bp + np.zeros((K, 6))
                np.add.at(rhs, op, np.einsum("oab,ob->oa", Y, bl[ol]))
                fidx = np.flatnonzero(free)
                Sf = S[np.ix_(fidx, fidx)].transpose(0, 2, 1, 3).reshape(6 * len(fidx), 6 * len(fidx))
                try:
                    dxp_f = np.linalg.solve(Sf, rhs[fidx].reshape(-1)).reshape(-1, 6)
                except np.linalg.LinAlgError:
                    lam *= ni; ni *= 2
                    continue
                dxp = np.zeros((K, 6)); dxp[fidx] = dxp_f
                tmp = -bl.copy()
                np.subtract.at(tmp, ol, np.einsum("oab,oa->ob", Hpl, dxp[op]))
                dxl = np.einsum("lab,lb->la", Hll_inv, tmp)
                dxl[~seen] = 0
                Tn = np.stack([se3_exp(dxp[k]) @ T[k] for k in range(K)])
                Xn = X + dxl
                en, Jpn, Jqn = linearize(Tn, Xn)
                new = _robust_cost(en, huber_delta)
                scale = float((dxp * (lam * dxp - bp)).sum() + (dxl * (lam * dxl - bl)).sum()) + 1e-3
                rho = (cost - new) / scale
                if rho > 0 and np.isfinite(new):
                    T, X, e, Jp, Jq, cost = Tn, Xn, en, Jpn, Jqn, new
                    lam *= max(1.0 / 3.0, min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0))
                    ni = 2.0
                    accepted += 1
                    step_ok = True
                    break
                lam *= ni; ni *= 2
            if not step_ok:
                break
    finally:
        prob.free()
    return BAResult(poses=T, points=X, chi2_initial=cost0, chi2_final=cost, iterations=accepted)
