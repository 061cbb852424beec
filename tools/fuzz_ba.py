"""Randomised comparison of the one-launch window bundle adjustment (slam_ba_optimize_f64 through
bundle_adjust_one_launch) with the plain-C statement of the same Schur-complement LM (oracle.ba_lm_c) on synthetic windows
(development aid).    python tools/fuzz_ba.py [windows] [seed]

Windows: 2..17 poses with 1..3 of them fixed (or more fixed than 16 free allow), 20..6000 points seen by 35..95 % of the
poses, observations in arbitrary order, pixel noise 0.1..1, with or without the Huber kernel, 1..8 LM steps; every tenth
window is a few poses with many points, so that the camera-block tasks are cut into slices.  Reports how many windows
agree (same number of accepted steps, final cost to 1e-9 relative, poses to 1e-8, points to 1e-7) and prints those that
do not: at a decision boundary (gain ratio ~ 0) rounding may legitimately send the two down different branches."""
import os
import sys
import time

import numpy as np
from scipy.spatial.transform import Rotation

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))
sys.path.insert(0, ROOT)
from slamhip.ba import bundle_adjust_one_launch  # noqa: E402
from slamhip.device import default_context  # noqa: E402
from oracle import oracle  # noqa: E402

FX, FY, CX, CY = 458.654, 457.296, 367.215, 248.375
windows = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = default_context()
agree, t0, worst_pose, worst_cost, sliced = 0, time.time(), 0.0, 0.0, 0
for it in range(windows):
    if it % 10 == 9:
        K, L = int(rng.integers(2, 5)), int(rng.integers(3000, 6000))
    else:
        K, L = int(rng.integers(2, 18)), int(rng.choice([rng.integers(20, 200), rng.integers(200, 1500)]))
    n_fixed = max(int(rng.integers(1, 4)), K - 16)
    n_fixed = min(n_fixed, K)
    fixed = tuple(sorted(rng.choice(K, n_fixed, replace=False).tolist()))
    T = np.tile(np.eye(4), (K, 1, 1))
    T[:, :3, :3] = Rotation.from_rotvec(rng.uniform(-0.15, 0.15, (K, 3))).as_matrix()
    T[:, :3, 3] = rng.uniform(-0.5, 0.5, (K, 3))
    X = np.c_[rng.uniform(-4, 4, (L, 2)), rng.uniform(6, 15, L)]
    op = np.repeat(np.arange(K), L).astype(np.int32); ol = np.tile(np.arange(L), K).astype(np.int32)
    keep = rng.uniform(size=K * L) < rng.uniform(0.35, 0.95)
    op, ol = op[keep], ol[keep]
    perm = rng.permutation(len(op))
    op, ol = op[perm], ol[perm]
    pc = np.einsum("oij,oj->oi", T[op, :3, :3], X[ol]) + T[op, :3, 3]
    meas = np.c_[FX * pc[:, 0] / pc[:, 2] + CX, FY * pc[:, 1] / pc[:, 2] + CY] + rng.normal(0, rng.uniform(0.1, 1.0), (len(op), 2))
    T0 = T.copy()
    for k in range(K):
        if k not in fixed:
            T0[k] = oracle.se3_exp_np(rng.normal(0, rng.uniform(0.002, 0.02), 6)) @ T[k]
    X0 = X + rng.normal(0, rng.uniform(0.01, 0.08), X.shape)
    delta = float(rng.choice([0.0, 0.0, rng.uniform(0.5, 3.0)]))
    iters = int(rng.integers(1, 9))
    got = bundle_adjust_one_launch(T0, X0, op, ol, meas, (FX, FY, CX, CY), iterations=iters, fixed_poses=fixed, huber_delta=delta, ctx=ctx)
    Tr, Xr, c0, c1, acc, _ = oracle.ba_lm_c(T0[:, :3, :4].reshape(K, 12), X0, op, ol, meas, FX, FY, CX, CY, iters, fixed, delta)
    dp, dx = float(np.abs(got.poses - Tr).max()), float(np.abs(got.points - Xr).max())
    dc = abs(got.chi2_final - c1) / max(c1, 1.0)
    same = got.iterations == acc and dc <= 1e-9 and dp <= 1e-8 and dx <= 1e-7
    ntask = K + (K - n_fixed) * (K - n_fixed + 1) // 2
    sliced += int((len(op) + 511) // 512 >= 2 * ntask)
    if same:
        agree += 1
        worst_pose, worst_cost = max(worst_pose, dp), max(worst_cost, dc)
    else:
        print(f"window {it}: K={K} L={L} O={len(op)} fixed={fixed} delta={delta:.2f} steps {got.iterations} vs {acc}, "
              f"cost {got.chi2_final:.9g} vs {c1:.9g}, pose diff {dp:.2e}, point diff {dx:.2e}", flush=True)
    if it % 50 == 49:
        print(f"  ... {it + 1} windows, {agree} agree, {time.time() - t0:.0f} s", flush=True)
print(f"{agree} / {windows} windows agree with the C statement (same accepted steps, cost to 1e-9, poses to 1e-8, points to 1e-7); "
      f"{sliced} of them ran with the camera-block tasks cut into slices; worst agreeing pose difference {worst_pose:.2e}, "
      f"cost difference {worst_cost:.2e} relative")
