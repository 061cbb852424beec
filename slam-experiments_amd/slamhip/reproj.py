"""Reprojection residuals / Jacobians (f64) on MI355X.

Host side of ``slam_reproj_rj_f64`` / ``slam_pose_normal_eq_f64``: the
arithmetic of ``Frontend.EdgeProjectionPoseOnly`` (``frontend.py:262-291``)
evaluated for whole observation lists at once.  GPU only; no CPU path.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from ._lib import check
from .device import Context, DeviceBuffer, default_context


def poses_to_rt12(T) -> np.ndarray:
    """[K,4,4] or [K,3,4] Tcw matrices (``Frame.pose.as_matrix()``) -> [K,12] rows of [R|t]."""
    T = np.asarray(T, np.float64)
    if T.ndim == 2:
        T = T[None]
    if T.shape[-2:] not in ((4, 4), (3, 4)):
        raise ValueError(f"poses must be [K,4,4] or [K,3,4], got {T.shape}")
    return np.ascontiguousarray(T[:, :3, :4].reshape(T.shape[0], 12))


class ReprojProblem:
    """Device-resident observation list: poses [K,12], points [L,3], (pose, point, pixel) per observation."""

    def __init__(self, ctx: Context, poses12, points, obs_pose, obs_point, meas, intrinsics, with_point: bool = True):
        self.ctx = ctx
        poses12 = np.ascontiguousarray(poses12, np.float64).reshape(-1, 12)
        points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
        obs_pose = np.ascontiguousarray(obs_pose, np.int32).reshape(-1)
        obs_point = np.ascontiguousarray(obs_point, np.int32).reshape(-1)
        meas = np.ascontiguousarray(meas, np.float64).reshape(-1, 2)
        self.K, self.L, self.O = poses12.shape[0], points.shape[0], obs_pose.shape[0]
        if obs_point.shape[0] != self.O or meas.shape[0] != self.O:
            raise ValueError("obs_pose, obs_point and meas must have one row per observation")
        if self.O and (obs_pose.min() < 0 or obs_pose.max() >= self.K or obs_point.min() < 0 or obs_point.max() >= self.L):
            raise ValueError("observation index out of range")
        self.fx, self.fy, self.cx, self.cy = (float(v) for v in intrinsics)
        self.with_point = with_point
        self.d_poses = ctx.upload(poses12)
        self.d_points = ctx.upload(points)
        self.d_obs_pose = ctx.upload(obs_pose)
        self.d_obs_point = ctx.upload(obs_point)
        self.d_meas = ctx.upload(meas)
        o = max(self.O, 1)
        self.d_e = ctx.malloc(o * 16)
        self.d_Jpose = ctx.malloc(o * 96)
        self.d_Jpoint = ctx.malloc(o * 48) if with_point else None

    def set_poses(self, poses12) -> None:
        self.d_poses.upload(np.ascontiguousarray(poses12, np.float64).reshape(self.K, 12))

    def set_points(self, points) -> None:
        self.d_points.upload(np.ascontiguousarray(points, np.float64).reshape(self.L, 3))

    def linearize(self) -> None:
        """One pass of the residual/Jacobian kernel (asynchronous on the ctx stream)."""
        c = self.ctx
        check(c.lib.slam_reproj_rj_f64(c.handle, self.d_poses.ptr, self.K, self.d_points.ptr, self.L,
                                       self.d_obs_pose.ptr, self.d_obs_point.ptr, self.d_meas.ptr, self.O,
                                       self.fx, self.fy, self.cx, self.cy, self.d_e.ptr, self.d_Jpose.ptr,
                                       self.d_Jpoint.ptr if self.d_Jpoint else None))

    def download(self) -> Tuple[np.ndarray, np.ndarray, Optional[np.ndarray]]:
        e = self.d_e.download(np.float64, (self.O, 2))
        Jp = self.d_Jpose.download(np.float64, (self.O, 2, 6))
        Jq = self.d_Jpoint.download(np.float64, (self.O, 2, 3)) if self.d_Jpoint else None
        return e, Jp, Jq

    def free(self) -> None:
        for b in (self.d_poses, self.d_points, self.d_obs_pose, self.d_obs_point, self.d_meas, self.d_e,
                  self.d_Jpose, self.d_Jpoint):
            if b is not None:
                b.free()


def build_linearization(poses, points, obs_pose_idx, obs_point_idx, meas, fx, fy, cx, cy, with_point: bool = True,
                        ctx: Optional[Context] = None):
    """(e [O,2], J_pose [O,2,6], J_point [O,2,3]) for an observation list.

    ``poses`` is [K,4,4] / [K,3,4] Tcw or already [K,12]."""
    poses = np.asarray(poses, np.float64)
    poses12 = poses.reshape(-1, 12) if poses.ndim == 2 and poses.shape[1] == 12 else poses_to_rt12(poses)
    ctx = ctx or default_context()
    prob = ReprojProblem(ctx, poses12, points, obs_pose_idx, obs_point_idx, meas, (fx, fy, cx, cy), with_point)
    try:
        prob.linearize()
        return prob.download()
    finally:
        prob.free()


class PoseOnlyProblem:
    """One camera pose against O fixed landmarks: the graph of ``_correct_current_pose`` (frontend.py:298-354)."""

    def __init__(self, ctx: Context, points, meas, intrinsics):
        self.ctx = ctx
        points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
        meas = np.ascontiguousarray(meas, np.float64).reshape(-1, 2)
        self.O = points.shape[0]
        if meas.shape[0] != self.O:
            raise ValueError("one measurement per point")
        self.fx, self.fy, self.cx, self.cy = (float(v) for v in intrinsics)
        o = max(self.O, 1)
        self.d_points = ctx.upload(points) if self.O else ctx.malloc(24)
        self.d_meas = ctx.upload(meas) if self.O else ctx.malloc(16)
        self.d_active = ctx.malloc(o)
        self.d_pose = ctx.malloc(96)
        self.d_H = ctx.malloc(36 * 8)
        self.d_b = ctx.malloc(6 * 8)
        self.d_chi2 = ctx.malloc(o * 8)
        self.set_active(np.ones(self.O, np.uint8))

    def set_active(self, active) -> None:
        if self.O:
            self.d_active.upload(np.ascontiguousarray(active, np.uint8).reshape(self.O))

    def normal_equations(self, pose12, huber_delta: float):
        """(H [6,6], b [6], chi2 [O]) at ``pose12`` with Huber weights (delta <= 0: none)."""
        c = self.ctx
        self.d_pose.upload(np.ascontiguousarray(pose12, np.float64).reshape(12))
        check(c.lib.slam_pose_normal_eq_f64(c.handle, self.d_pose.ptr, self.d_points.ptr, self.d_meas.ptr,
                                            self.d_active.ptr, self.O, self.fx, self.fy, self.cx, self.cy,
                                            float(huber_delta), self.d_H.ptr, self.d_b.ptr, self.d_chi2.ptr))
        H = self.d_H.download(np.float64, (6, 6))
        b = self.d_b.download(np.float64, (6,))
        chi2 = self.d_chi2.download(np.float64, (self.O,)) if self.O else np.zeros(0)
        return H, b, chi2

    def free(self) -> None:
        for b in (self.d_points, self.d_meas, self.d_active, self.d_pose, self.d_H, self.d_b, self.d_chi2):
            b.free()
