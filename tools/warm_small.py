"""Does a short launch run at the clock a long one reaches?  The 4096 x 4096 search (16 us) and the one-launch window LM
(0.3 ms) timed on an idle device, and again right behind 40 passes of the 64k x 64k search (development aid)."""
import ctypes
import os
import sys

import numpy as np
from scipy.spatial.transform import Rotation

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))
import slamhip  # noqa: E402
from slamhip.pose_opt import se3_exp  # noqa: E402

ctx = slamhip.default_context()
rng = np.random.default_rng(3)
big = 65536
dq = slamhip.DeviceDescriptors(ctx, rng.integers(0, 256, (big, 32), dtype=np.uint8))
dt = slamhip.DeviceDescriptors(ctx, rng.integers(0, 256, (big, 32), dtype=np.uint8))
tab = slamhip.Top2Table(ctx, big)


def search(n, reps):
    ctx.sync(); ctx.timer_start()
    for _ in range(reps):
        slamhip.knn2_device(ctx, dq.buf, n, dt.buf, n, tab.idx, tab.dist)
    return ctx.timer_stop() / reps * 1e3


# the reference's window for the one-launch LM (as tools/ba_time.py builds it)
FX, FY, CX, CY = 458.654, 457.296, 367.215, 248.375
K, L = 7, 1400
r = np.random.default_rng(K)
T = np.tile(np.eye(4), (K, 1, 1))
T[:, :3, :3] = Rotation.from_rotvec(r.uniform(-0.15, 0.15, (K, 3))).as_matrix()
T[:, :3, 3] = r.uniform(-0.5, 0.5, (K, 3))
X = np.c_[r.uniform(-4, 4, (L, 2)), r.uniform(6, 15, L)]
op = np.repeat(np.arange(K), L).astype(np.int32); ol = np.tile(np.arange(L), K).astype(np.int32)
keep = r.uniform(size=K * L) < 0.6
op, ol = op[keep], ol[keep]
pc = np.einsum("oij,oj->oi", T[op, :3, :3], X[ol]) + T[op, :3, 3]
meas = np.c_[FX * pc[:, 0] / pc[:, 2] + CX, FY * pc[:, 1] / pc[:, 2] + CY] + r.normal(0, 0.2, (len(op), 2))
T0 = np.stack([T[0], T[1]] + [se3_exp(r.normal(0, 0.01, 6)) @ T[k] for k in range(2, K)])
X0 = X + r.normal(0, 0.05, X.shape)
O = len(op)
pt_obs = np.argsort(ol, kind="stable").astype(np.int32); ps_obs = np.argsort(op, kind="stable").astype(np.int32)
pt_ptr = np.zeros(L + 1, np.int32); pt_ptr[1:] = np.cumsum(np.bincount(ol, minlength=L))
ps_ptr = np.zeros(K + 1, np.int32); ps_ptr[1:] = np.cumsum(np.bincount(op, minlength=K))
free = np.arange(2, K, dtype=np.int32)
d = [ctx.upload(a) for a in (op, ol, meas, pt_ptr, pt_obs, ps_ptr, ps_obs, free)]
sT = np.concatenate([T0[:, :3, :4].reshape(-1), np.zeros(K * 12)]); sX = np.concatenate([X0.reshape(-1), np.zeros(L * 3)])
dT, dX = ctx.upload(sT), ctx.upload(sX)
need = ctypes.c_uint64(0)
ctx.lib.slam_ba_optimize_workspace(K, L, O, ctypes.byref(need))
dW, dS = ctx.malloc(need.value), ctx.malloc(64)


def window_lm():
    dT.upload(sT); dX.upload(sX)
    ctx.sync(); ctx.timer_start()
    assert ctx.lib.slam_ba_optimize_f64(ctx.handle, K, L, O, *[b.ptr for b in d], len(free), FX, FY, CX, CY, 0.0, 5, dT.ptr, dX.ptr,
                                        dW.ptr, need.value, dS.ptr) == 0
    return ctx.timer_stop() * 1e3


search(4096, 20); window_lm()
print(f"idle device : 4096 x 4096 {search(4096, 200):6.1f} us per search; window LM (K=7, five steps) {min(window_lm() for _ in range(5)):6.1f} us")
search(big, 40)
a = search(4096, 200)
search(big, 40)
b = window_lm()
print(f"behind 40 passes of 64k x 64k: 4096 x 4096 {a:6.1f} us per search; window LM {b:6.1f} us")
