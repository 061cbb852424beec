"""Where a small call's time goes: C-ABI primitives (malloc/free/upload/download) and the two small kernels."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))
from slamhip.device import default_context
from slamhip.pose_opt import se3_exp

ctx = default_context()
lib, h = ctx.lib, ctx.handle


def t(f, n=200):
    for _ in range(10):
        f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e6


host = np.zeros(6400, np.uint8)
buf = ctx.malloc(6400)
print(f"malloc+free 6400 B      {t(lambda: ctx.malloc(6400).free()):8.1f} us")
print(f"upload 6400 B           {t(lambda: buf.upload(host)):8.1f} us")
print(f"download 6400 B         {t(lambda: buf.download(np.uint8, (6400,))):8.1f} us")
print(f"sync (idle)             {t(ctx.sync):8.1f} us")

rng = np.random.default_rng(228)
X = np.c_[rng.uniform(-4, 4, (200, 2)), rng.uniform(6, 15, 200)]
pix = np.c_[458.654 * X[:, 0] / X[:, 2] + 367.215, 457.296 * X[:, 1] / X[:, 2] + 248.375] + rng.normal(0, 0.3, (200, 2))
T0 = (se3_exp([0.01, -0.01, 0.005, 0.05, -0.03, 0.04]) @ np.eye(4))[:3, :4].reshape(12)
d_in, d_pts, d_meas = ctx.upload(T0), ctx.upload(X), ctx.upload(pix)
d_out, d_inl, d_chi2, d_stats = ctx.malloc(96), ctx.malloc(200), ctx.malloc(1600), ctx.malloc(8)
def po():
    lib.slam_pose_optimize_f64(h, d_in.ptr, d_pts.ptr, d_meas.ptr, 200, 458.654, 457.296, 367.215, 248.375, 4, 10, 5.991**2, 1.0,
                               d_out.ptr, d_inl.ptr, d_chi2.ptr, d_stats.ptr)
po(); ctx.sync(); ctx.timer_start()
for _ in range(50):
    po()
print(f"pose_opt_kernel 200 edges {ctx.timer_stop() / 50 * 1e3:8.1f} us per launch (device time, back to back); "
      f"accepted steps {d_stats.download(np.int32, (2,))[1]}")

q = rng.integers(0, 256, (200, 32), dtype=np.uint8); tr = rng.integers(0, 256, (200, 32), dtype=np.uint8)
dq, dt = ctx.upload(q), ctx.upload(tr)
di, dd = ctx.malloc(1600), ctx.malloc(1600)
def bf():
    lib.slam_bf_knn2_u256(h, dq.ptr, 200, dt.ptr, 200, 0, di.ptr, dd.ptr)
bf(); ctx.sync(); ctx.timer_start()
for _ in range(200):
    bf()
print(f"slam_bf_knn2_u256 200x200 {ctx.timer_stop() / 200 * 1e3:8.1f} us per call (device time, back to back)")
print(f"  same, launch+sync from the host {t(lambda: (bf(), ctx.sync())):8.1f} us")
