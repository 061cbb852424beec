"""How long does the GPU take to reach its steady clock?  Per-step device time of the 64k x 64k search."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))
import slamhip
ctx = slamhip.default_context()
n = 65536
q = np.random.default_rng(228).integers(0, 256, (n, 32), dtype=np.uint8)
t = np.random.default_rng(229).integers(0, 256, (n, 32), dtype=np.uint8)
dq, dt = slamhip.DeviceDescriptors(ctx, q), slamhip.DeviceDescriptors(ctx, t)
tab = slamhip.Top2Table(ctx, n)
ctx.sync()
ts = []
for i in range(120):
    ctx.timer_start()
    slamhip.knn2_device(ctx, dq.buf, n, dt.buf, n, tab.idx, tab.dist)
    ts.append(ctx.timer_stop())
print("per-step ms:", " ".join(f"{x:.3f}" for x in ts[:40]))
print("steps 40-120 mean %.4f min %.4f max %.4f" % (np.mean(ts[40:]), np.min(ts[40:]), np.max(ts[40:])))
time.sleep(2.0)
ts = []
for i in range(10):
    ctx.timer_start()
    slamhip.knn2_device(ctx, dq.buf, n, dt.buf, n, tab.idx, tab.dist)
    ts.append(ctx.timer_stop())
print("after 2 s idle:", " ".join(f"{x:.3f}" for x in ts))
