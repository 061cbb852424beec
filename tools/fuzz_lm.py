"""Randomised comparison of the one-launch pose-only LM (slam_pose_optimize_f64) with the oracle's independent LM
(oracle.pose_lm_np) on synthetic frames (development aid).    python tools/fuzz_lm.py [problems] [seed]

Reports how many problems agree (pose <= 1e-7, identical inlier sets) and prints the ones that do not: at a decision
boundary (rho ~ 0, chi2 ~ threshold) rounding may legitimately send the two implementations down different branches."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))
sys.path.insert(0, ROOT)
from backend import Backend  # noqa: E402
from oracle import oracle  # noqa: E402

FX, FY, CX, CY = 458.654, 457.296, 367.215, 248.375
problems = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
be = Backend()
agree, t0 = 0, time.time()
worst = 0.0
for it in range(problems):
    O = int(rng.choice([rng.integers(6, 30), rng.integers(30, 250), rng.integers(250, 700)]))
    from scipy.spatial.transform import Rotation
    T = np.eye(4)
    T[:3, :3] = Rotation.from_rotvec(rng.uniform(-0.2, 0.2, 3)).as_matrix()
    T[:3, 3] = rng.uniform(-0.5, 0.5, 3)
    X = np.c_[rng.uniform(-4, 4, (O, 2)), rng.uniform(5, 18, O)]
    pc = X @ T[:3, :3].T + T[:3, 3]
    meas = np.c_[FX * pc[:, 0] / pc[:, 2] + CX, FY * pc[:, 1] / pc[:, 2] + CY] + rng.normal(0, rng.uniform(0.1, 1.5), (O, 2))
    bad = rng.uniform(size=O) < rng.uniform(0, 0.35)
    meas[bad] += rng.uniform(15, 150, (int(bad.sum()), 2)) * rng.choice([-1, 1], (int(bad.sum()), 2))
    meas = meas.astype(np.int32).astype(np.float64)
    T0 = oracle.se3_exp_np(rng.normal(0, rng.uniform(0.005, 0.06), 6)) @ T
    Tr, inl, chi2, acc = oracle.pose_lm_np(T0, X, meas, FX, FY, CX, CY)
    got = be.optimize_pose(T0, X, meas, FX, FY, CX, CY, on_device=True)
    dp = float(np.abs(got.pose - Tr).max())
    same = dp <= 1e-7 and np.array_equal(got.inliers, inl)
    agree += same
    worst = max(worst, dp if same else 0.0)
    if not same:
        print(f"problem {it}: O={O} outliers={int(bad.sum())} |dpose|max={dp:.3e} inlier sets differ at {int((got.inliers != inl).sum())} edges, "
              f"accepted steps {got.iterations} vs {acc}, chi2 near the gate: {int((np.abs(chi2 - 5.991**2) < 1e-6 * 35.9).sum())}", flush=True)
print(f"{agree} of {problems} problems agree (pose <= 1e-7, identical inliers; worst agreeing |dpose| {worst:.2e}) in {time.time() - t0:.0f} s")
