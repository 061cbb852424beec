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

from ._lib import SLAM_ERR_INVALID, SlamHipBusy, SlamHipError, addr, check
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


class SchurProblem:
    """Device-resident window for ``slam_ba_reduce_f64`` / ``slam_ba_backsub_f64``: the observation list, its
    two groupings (by point, by pose), the (pose, point) -> observation table and all output blocks."""

    REC = 73   # SLAM_BA_REC

    def __init__(self, ctx: Context, K: int, L: int, obs_pose, obs_point, meas, intrinsics):
        self._check = check
        self.ctx = ctx
        op = np.ascontiguousarray(obs_pose, np.int32).reshape(-1)
        ol = np.ascontiguousarray(obs_point, np.int32).reshape(-1)
        meas = np.ascontiguousarray(meas, np.float64).reshape(-1, 2)
        O = op.shape[0]
        if ol.shape[0] != O or meas.shape[0] != O:
            raise ValueError("obs_pose, obs_point and meas must have one row per observation")
        if K < 1 or L < 1:
            raise ValueError("need at least one pose and one point")
        if O and (op.min() < 0 or op.max() >= K or ol.min() < 0 or ol.max() >= L):
            raise ValueError("observation index out of range")
        lookup = np.full((K, L), -1, np.int32)
        lookup[op, ol] = np.arange(O, dtype=np.int32)
        if int((lookup >= 0).sum()) != O:
            raise ValueError("a (pose, point) pair is observed more than once")
        self.K, self.L, self.O = K, L, O
        self.fx, self.fy, self.cx, self.cy = (float(v) for v in intrinsics)
        pt_obs = np.argsort(ol, kind="stable").astype(np.int32)
        ps_obs = np.argsort(op, kind="stable").astype(np.int32)
        pt_ptr = np.zeros(L + 1, np.int32); pt_ptr[1:] = np.cumsum(np.bincount(ol, minlength=L))
        ps_ptr = np.zeros(K + 1, np.int32); ps_ptr[1:] = np.cumsum(np.bincount(op, minlength=K))
        pad = lambda a: a if a.size else np.zeros(1, a.dtype)
        up = ctx.upload
        self.d_op, self.d_ol, self.d_meas = up(pad(op)), up(pad(ol)), up(meas if O else np.zeros((1, 2)))
        self.d_pt_ptr, self.d_pt_obs, self.d_ps_ptr, self.d_ps_obs = up(pt_ptr), up(pad(pt_obs)), up(ps_ptr), up(pad(ps_obs))
        self.d_lookup = up(lookup)
        self.d_poses, self.d_points = ctx.malloc(K * 96), ctx.malloc(L * 24)
        o = max(O, 1)
        self.d_rec, self.d_E, self.d_bl = ctx.malloc(o * self.REC * 8), ctx.malloc(L * 72), ctx.malloc(L * 24)
        # every reduced block in one allocation, so a reduction comes back in one download
        self._n_out = K * (21 + 6 + 6 + 1) + K * K * 36
        self.d_out = ctx.malloc(self._n_out * 8)
        self.d_Hpp, self.d_bp = self.d_out.view(0, K * 168), self.d_out.view(K * 168, K * 48)
        self.d_ybl, self.d_cost = self.d_out.view(K * 216, K * 48), self.d_out.view(K * 264, K * 8)
        self.d_W = self.d_out.view(K * 272, K * K * 288)
        self.d_dp, self.d_dl = ctx.malloc(K * 48), ctx.malloc(L * 24)
        self.d_hll = ctx.malloc(L * 24)
        self._iu = np.triu_indices(6)

    def reduce(self, poses12, points, huber_delta: float, lam: float):
        """-> (S [K,K,6,6] symmetric, rhs [K,6], bp [K,6], cost) at this state and damping."""
        c, K = self.ctx, self.K
        self.d_poses.upload(np.ascontiguousarray(poses12, np.float64).reshape(K, 12))
        self.d_points.upload(np.ascontiguousarray(points, np.float64).reshape(self.L, 3))
        self._check(c.lib.slam_ba_reduce_f64(
            c.handle, self.d_poses.ptr, K, self.d_points.ptr, self.L, self.d_op.ptr, self.d_ol.ptr, self.d_meas.ptr,
            self.O, self.d_pt_ptr.ptr, self.d_pt_obs.ptr, self.d_ps_ptr.ptr, self.d_ps_obs.ptr, self.d_lookup.ptr,
            self.fx, self.fy, self.cx, self.cy, float(huber_delta), float(lam), self.d_rec.ptr, self.d_E.ptr,
            self.d_bl.ptr, self.d_Hpp.ptr, self.d_bp.ptr, self.d_ybl.ptr, self.d_cost.ptr, self.d_W.ptr, self.d_hll.ptr))
        out = self.d_out.download(np.float64, (self._n_out,))
        tri, bp, ybl = out[:K * 21].reshape(K, 21), out[K * 21:K * 27].reshape(K, 6), out[K * 27:K * 33].reshape(K, 6)
        cost = float(out[K * 33:K * 34].sum())
        W = out[K * 34:].reshape(K, K, 6, 6)
        Hpp = np.zeros((K, 6, 6))
        Hpp[:, self._iu[0], self._iu[1]] = tri
        Hpp[:, self._iu[1], self._iu[0]] = tri
        S = np.zeros((K, K, 6, 6))
        k1, k2 = np.triu_indices(K, 1)
        S[k1, k2] = -W[k1, k2]
        S[k2, k1] = -W[k1, k2].transpose(0, 2, 1)
        kk = np.arange(K)
        S[kk, kk] = Hpp + lam * np.eye(6) - W[kk, kk]
        return S, -bp + ybl, bp, cost

    def cost(self, poses12, points, huber_delta: float) -> float:
        """Robust cost at a candidate state (``slam_ba_cost_f64``); leaves the blocks of the last ``reduce`` alone
        but replaces the device copy of the state."""
        c, K = self.ctx, self.K
        self.d_poses.upload(np.ascontiguousarray(poses12, np.float64).reshape(K, 12))
        self.d_points.upload(np.ascontiguousarray(points, np.float64).reshape(self.L, 3))
        self._check(c.lib.slam_ba_cost_f64(c.handle, self.d_poses.ptr, K, self.d_points.ptr, self.d_ol.ptr, self.d_meas.ptr,
                                           self.d_ps_ptr.ptr, self.d_ps_obs.ptr, self.fx, self.fy, self.cx, self.cy,
                                           float(huber_delta), self.d_cost.ptr))
        return float(self.d_cost.download(np.float64, (K,)).sum())

    def back_substitute(self, dp):
        """dp [K,6] -> (dl [L,3], bl [L,3]) for the system of the last ``reduce``."""
        c = self.ctx
        self.d_dp.upload(np.ascontiguousarray(dp, np.float64).reshape(self.K, 6))
        self._check(c.lib.slam_ba_backsub_f64(c.handle, self.L, self.d_pt_ptr.ptr, self.d_pt_obs.ptr, self.d_op.ptr,
                                              self.d_rec.ptr, self.d_E.ptr, self.d_bl.ptr, self.d_dp.ptr, self.d_dl.ptr))
        return self.d_dl.download(np.float64, (self.L, 3)), self.d_bl.download(np.float64, (self.L, 3))

    def diag_max(self, free=None) -> float:
        """Largest diagonal entry of the Hpp blocks of the poses ``free`` (all if None) and of every Hll of the last
        ``reduce`` (g2o-style initial damping; poses that hold the gauge are not part of the system)."""
        tri = self.d_Hpp.download(np.float64, (self.K, 21))
        if free is not None:
            tri = tri[np.asarray(free, np.int64)]
        d = tri[:, [0, 6, 11, 15, 18, 20]].max(initial=0.0)
        if self.O:
            d = max(d, self.d_hll.download(np.float64, (self.L, 3)).max(initial=0.0))
        return float(d)

    def free(self) -> None:
        for name, b in list(vars(self).items()):
            if name.startswith("d_") and b is not None:
                b.free()
                setattr(self, name, None)


# the single-launch form (slam_ba_optimize_f64) takes windows of up to 16 free poses (a 96 x 96 reduced system in LDS),
# 64 poses and 131072 observations; measured (tools/ba_time.py) it is ahead of the multi-launch form at K = 7 (1.0 vs 2.9 ms
# for five steps) and at K = 16 / 48 k observations (3.3 vs 5.4 ms)
ONE_LAUNCH_MAX_FREE, ONE_LAUNCH_MAX_OBS = 16, 131072


def bundle_adjust_one_launch(poses, points, obs_pose_idx, obs_point_idx, meas, intrinsics, iterations: int = 10,
                             fixed_poses: Sequence[int] = (0,), huber_delta: float = 0.0,
                             ctx: Optional[Context] = None) -> BAResult:
    """``bundle_adjust_device`` with the whole Levenberg-Marquardt loop in ONE kernel launch (``slam_ba_optimize_f64``):
    the window goes up once, the dense reduced system is solved on the device, the trials, their costs and the accept /
    reject decisions never leave it, the result comes back once.  For windows of at most 16 moving poses (the reference
    keeps 7 keyframes, ``backend.py:11``).

    Raises ``ValueError`` for bad arguments and ``SlamHipBusy`` when the launch gave up at one of its grid barriers because
    other work held compute units (it needs all its workgroups resident at once): nothing was changed, and
    ``bundle_adjust_device`` - per-phase kernels, no residency requirement - does the same adjustment (``bundle_adjust_auto``
    and ``Backend.optimize`` fall back to it by themselves)."""
    ctx = ctx or default_context()
    P = np.asarray(poses, np.float64)
    T12 = np.ascontiguousarray((P.reshape(-1, 12) if P.ndim == 2 else poses_to_rt12(P)).reshape(-1, 12))
    X = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    K, L = T12.shape[0], X.shape[0]
    op = np.ascontiguousarray(obs_pose_idx, np.int32).reshape(-1)
    ol = np.ascontiguousarray(obs_point_idx, np.int32).reshape(-1)
    meas = np.ascontiguousarray(meas, np.float64).reshape(-1, 2)
    O = op.shape[0]
    if ol.shape[0] != O or meas.shape[0] != O:
        raise ValueError("obs_pose, obs_point and meas must have one row per observation")
    if K < 1 or L < 1:
        raise ValueError("need at least one pose and one point")
    fixed = np.zeros(K, np.uint8)
    fixed[list(fixed_poses)] = 1
    fx, fy, cx, cy = (float(v) for v in intrinsics)
    Tout, Xout, st = np.empty((K, 12)), np.empty((L, 3)), np.empty(8)
    # ranges, the one-observation-per-(pose, point) rule and the index tables are the library's business (slam_ba_optimize_host_f64)
    try:
        check(ctx.lib.slam_ba_optimize_host_f64(ctx.handle, K, L, O, addr(T12), addr(X), addr(op), addr(ol), addr(meas), addr(fixed),
                                                fx, fy, cx, cy, float(huber_delta), int(iterations), addr(Tout), addr(Xout), addr(st)))
    except SlamHipError as exc:
        if exc.code == SLAM_ERR_INVALID:         # a bad argument, e.g. an index out of range or a pair observed twice
            raise ValueError(str(exc)) from None
        raise                                    # SlamHipBusy (device busy: not the caller's fault) and HIP errors as they are
    Tr = np.tile(np.eye(4), (K, 1, 1))
    Tr[:, :3, :4] = Tout.reshape(K, 3, 4)
    return BAResult(poses=Tr, points=Xout, chi2_initial=float(st[0]), chi2_final=float(st[1]), iterations=int(st[2]))


def bundle_adjust_auto(poses, points, obs_pose_idx, obs_point_idx, meas, intrinsics, iterations: int = 10,
                       fixed_poses: Sequence[int] = (0,), huber_delta: float = 0.0, ctx: Optional[Context] = None,
                       on_busy=None) -> BAResult:
    """``bundle_adjust_one_launch``, and when that reports a busy device (``SlamHipBusy``) the same adjustment ONCE more
    through ``bundle_adjust_device``, whose kernels need no co-residency: a GPU that is shared with the tracking thread's
    searches (``slam.py:27-35``: tracking and backend are two threads) slows the window adjustment down, it does not fail it.
    ``on_busy(exc)`` is called when the fall-back is taken."""
    args = (poses, points, obs_pose_idx, obs_point_idx, meas, intrinsics, iterations, fixed_poses, huber_delta)
    try:
        return bundle_adjust_one_launch(*args, ctx=ctx)
    except SlamHipBusy as exc:
        if on_busy is not None:
            on_busy(exc)
        return bundle_adjust_device(*args, ctx=ctx)


def bundle_adjust_device(poses, points, obs_pose_idx, obs_point_idx, meas, intrinsics, iterations: int = 10,
                         fixed_poses: Sequence[int] = (0,), huber_delta: float = 0.0,
                         ctx: Optional[Context] = None) -> BAResult:
    """``bundle_adjust`` with the linearisation, the point elimination, the reduced camera blocks and the
    back-substitution on the GPU (``slam_ba_reduce_f64`` / ``slam_ba_backsub_f64``); the host only solves the
    6K x 6K reduced system and drives the same LM schedule."""
    ctx = ctx or default_context()
    P = np.asarray(poses, np.float64)
    T = np.tile(np.eye(4), (P.shape[0], 1, 1))
    T[:, :3, :4] = (P.reshape(-1, 12) if P.ndim == 2 else poses_to_rt12(P)).reshape(-1, 3, 4)
    X = np.array(points, np.float64).reshape(-1, 3)
    K, L = T.shape[0], X.shape[0]
    free = np.ones(K, bool)
    free[list(fixed_poses)] = False
    fidx = np.flatnonzero(free)
    prob = SchurProblem(ctx, K, L, obs_pose_idx, obs_point_idx, meas, intrinsics)
    rt = lambda Tc: Tc[:, :3, :4].reshape(K, 12)
    try:
        S, rhs, bp, cost = prob.reduce(rt(T), X, huber_delta, 1.0)
        cost0 = cost
        lam = 1e-5 * max(prob.diag_max(fidx), 1e-12)
        ni, accepted = 2.0, 0
        for _ in range(iterations):
            step_ok = False
            for _trial in range(10):
                S, rhs, bp, cost = prob.reduce(rt(T), X, huber_delta, lam)
                Sf = S[np.ix_(fidx, fidx)].transpose(0, 2, 1, 3).reshape(6 * len(fidx), 6 * len(fidx))
                try:
                    dxp_f = np.linalg.solve(Sf, rhs[fidx].reshape(-1)).reshape(-1, 6)
                except np.linalg.LinAlgError:
                    lam *= ni; ni *= 2
                    continue
                dxp = np.zeros((K, 6)); dxp[fidx] = dxp_f
                dxl, bl = prob.back_substitute(dxp)
                Tn = np.stack([se3_exp(dxp[k]) @ T[k] for k in range(K)])
                Xn = X + dxl
                new = prob.cost(rt(Tn), Xn, huber_delta)
                scale = float((dxp * (lam * dxp - bp)).sum() + (dxl * (lam * dxl - bl)).sum()) + 1e-3
                rho = (cost - new) / scale
                if rho > 0 and np.isfinite(new):
                    T, X, cost = Tn, Xn, new
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
