"""Two contexts (two threads, two streams) running the one-launch window LM at the same time on one GPU: every launch
must complete (its grid barrier needs all of its workgroups resident while the other context's launch holds its own), and
every result must equal the first one bit for bit (development aid).    python tools/stress_ba.py [launches per thread] [K] [L]"""
import os
import sys
import threading
import time

import numpy as np
from scipy.spatial.transform import Rotation

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))
import slamhip  # noqa: E402
from slamhip.ba import bundle_adjust_one_launch  # noqa: E402
from slamhip.pose_opt import se3_exp  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
L = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
FX, FY, CX, CY = 458.654, 457.296, 367.215, 248.375
rng = np.random.default_rng(K)
T = np.tile(np.eye(4), (K, 1, 1))
T[:, :3, :3] = Rotation.from_rotvec(rng.uniform(-0.15, 0.15, (K, 3))).as_matrix()
T[:, :3, 3] = rng.uniform(-0.5, 0.5, (K, 3))
X = np.c_[rng.uniform(-4, 4, (L, 2)), rng.uniform(6, 15, L)]
op = np.repeat(np.arange(K), L).astype(np.int32); ol = np.tile(np.arange(L), K).astype(np.int32)
keep = rng.uniform(size=K * L) < 0.6
op, ol = op[keep], ol[keep]
pc = np.einsum("oij,oj->oi", T[op, :3, :3], X[ol]) + T[op, :3, 3]
meas = np.c_[FX * pc[:, 0] / pc[:, 2] + CX, FY * pc[:, 1] / pc[:, 2] + CY] + rng.normal(0, 0.2, (len(op), 2))
T0 = np.stack([T[0], T[1]] + [se3_exp(rng.normal(0, 0.01, 6)) @ T[k] for k in range(2, K)])
X0 = X + rng.normal(0, 0.05, X.shape)
out = {}


def work(name):
    ctx = slamhip.Context(0)
    first, bad, t0 = None, 0, time.time()
    for _ in range(N):
        r = bundle_adjust_one_launch(T0, X0, op, ol, meas, (FX, FY, CX, CY), iterations=5, fixed_poses=(0, 1), ctx=ctx)
        if first is None:
            first = r
        elif not (np.array_equal(r.poses, first.poses) and np.array_equal(r.points, first.points) and r.chi2_final == first.chi2_final):
            bad += 1
    out[name] = (first, bad, time.time() - t0)


threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
for t in threads:
    t.start()
for t in threads:
    t.join()
a, b = out[0], out[1]
same = np.array_equal(a[0].poses, b[0].poses) and a[0].chi2_final == b[0].chi2_final
print(f"K={K} L={L} O={len(op)}: 2 contexts x {N} launches at the same time: {a[1] + b[1]} results differ from their thread's first, "
      f"the two threads agree: {same}; {a[2]:.2f} s and {b[2]:.2f} s ({(a[2] + b[2]) / 2 / N * 1e3:.2f} ms per launch while sharing the device), "
      f"cost {a[0].chi2_initial:.1f} -> {a[0].chi2_final:.1f} in {a[0].iterations} steps")
