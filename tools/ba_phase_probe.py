"""Phase timeline of the one-launch window LM (development aid; needs tools/exp/libslamhip_bastamps.so from
tools/build_ba_stamps.sh): thread 0 of every workgroup stamps wall_clock64() before and after every grid barrier.

    python tools/ba_phase_probe.py [K L iterations]
"""
import ctypes
import os
import sys

import numpy as np
from scipy.spatial.transform import Rotation

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))
from slamhip import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "tools", "exp", "libslamhip_bastamps.so")
from slamhip.device import default_context  # noqa: E402
from slamhip.pose_opt import se3_exp  # noqa: E402

K, L, iters = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (7, 1400, 5)
FX, FY, CX, CY = 458.654, 457.296, 367.215, 248.375
ctx = default_context()
ctx.lib.slam_exp_set_ba_stamps.argtypes = [ctypes.c_void_p]
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
O = len(op)
pt_obs = np.argsort(ol, kind="stable").astype(np.int32); ps_obs = np.argsort(op, kind="stable").astype(np.int32)
pt_ptr = np.zeros(L + 1, np.int32); pt_ptr[1:] = np.cumsum(np.bincount(ol, minlength=L))
ps_ptr = np.zeros(K + 1, np.int32); ps_ptr[1:] = np.cumsum(np.bincount(op, minlength=K))
free = np.arange(2, K, dtype=np.int32)
d = [ctx.upload(a) for a in (op, ol, meas, pt_ptr, pt_obs, ps_ptr, ps_obs, free)]
state_T = np.concatenate([T0[:, :3, :4].reshape(-1), np.zeros(K * 12)]); state_X = np.concatenate([X0.reshape(-1), np.zeros(L * 3)])
dT, dX = ctx.upload(state_T), ctx.upload(state_X)
need = ctypes.c_uint64(0)
ctx.lib.slam_ba_optimize_workspace(K, L, O, ctypes.byref(need))
dW, dS = ctx.malloc(need.value), ctx.malloc(64)
stamps = ctx.malloc((128 * 256 + 2048) * 8)
ctx.lib.slam_memset(ctx.handle, stamps.ptr, 0, (128 * 256 + 2048) * 8)


def run():
    dT.upload(state_T); dX.upload(state_X)
    ctx.sync(); ctx.timer_start()
    assert ctx.lib.slam_ba_optimize_f64(ctx.handle, K, L, O, *[b.ptr for b in d], len(free), FX, FY, CX, CY, 0.0, iters, dT.ptr, dX.ptr,
                                        dW.ptr, need.value, dS.ptr) == 0
    return ctx.timer_stop()


for _ in range(4):
    run()
assert ctx.lib.slam_exp_set_ba_stamps(stamps.ptr) == 0
ms = run()
st = dS.download(np.float64, (8,))
G = int(st[7])
raw = stamps.download(np.uint64, (128 * 256 + 2048,)).astype(np.int64)
tr = raw[:128 * 256].reshape(128, 256)[:G]
n = int((tr[0, :200] > 0).sum())
t0 = tr[:, 0].min()
us = (tr - t0) * 0.01
print(f"K={K} L={L} O={O} iterations={iters}: {G} workgroups, event time {ms * 1e3:.1f} us, {int(st[2])} accepted / {int(st[3])} trials; "
      f"first stamp spread {us[:, 0].max():.1f} us, last stamp {us[:, n - 1].max():.1f} us")
# stamps: 0 = start; then pairs (before barrier b, after barrier b)
print("barrier  work before it (wg 0 / slowest wg / mean)   barrier itself (last arrival -> last release)   released at")
prev = us[:, 0]
nb = (n - 1) // 2
for b in range(nb):
    arrive, leave = us[:, 1 + 2 * b], us[:, 2 + 2 * b]
    work = arrive - prev
    print(f"  {b:3d}    {work[0]:7.2f} / {work.max():7.2f} (wg {int(work.argmax()):3d}) / {work.mean():7.2f}            {leave.max() - arrive.max():6.2f}"
          f"                               {leave.max():8.2f}")
    prev = leave
s2 = tr[0, 200:]
s2 = (s2[s2 > 0] - t0) * 0.01
print("workgroup 0 solve stamps (assembled, solved) per solve:", " ".join(f"{v:.2f}" for v in s2))
# rows: 0-1 the observation table, 2 the first pass (points + pose tasks), then per trial points | camera blocks | solve |
# step; averaged over the accepted trials after the first (rows 7..)
if nb >= 11 and (nb - 7) % 4 == 0:
    names = ("points", "camera blocks", "solve", "step")
    work = np.zeros(4); bar = np.zeros(4); cnt = 0
    for b0 in range(7, nb, 4):
        for x in range(4):
            b = b0 + x
            arrive, leave = us[:, 1 + 2 * b], us[:, 2 + 2 * b]
            before = us[:, 2 * b]
            work[x] += (arrive - before).max(); bar[x] += leave.max() - arrive.max()
        cnt += 1
    print("per iteration: " + ", ".join(f"{n} {w / cnt:.1f} (+{q / cnt:.1f} barrier)" for n, w, q in zip(names, work, bar)) +
          f"; sum {(work.sum() + bar.sum()) / cnt:.1f} us")
# inside workgroup 0's factorisation (first solve): after each panel's row solves, after its trailing update, after the
# forward and after the backward substitution
s3 = raw[128 * 256:]
s3 = (s3[s3 > 0] - t0) * 0.01
per = (6 * (K - 2)) // 6 * 2 + 2
if len(s3) >= per and len(s2) >= 2:
    first = s3[:per]
    print("first solve from 'assembled': " + " ".join(f"{v - s2[0]:.2f}" for v in first))
