"""Development aid: device time of a batch of equal searches (slam_bf_knn2_batch_u256) for several forced chunk lengths
(needs the experiment build tools/exp/libslamhip_expchunk.so, which reads SLAM_EXP_CHUNK)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))
import slamhip
from slamhip import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "exp", "libslamhip_expchunk.so")
ctx = slamhip.Context(0)
rng = np.random.default_rng(1)
cases = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(4096, 16), (4096, 4), (4096, 1), (1000, 8), (2048, 32), (8192, 4), (8192, 1)]
for n, B in cases:
    qs = [slamhip.DeviceDescriptors(ctx, rng.integers(0, 256, (n, 32), dtype=np.uint8)) for _ in range(B)]
    ts = [slamhip.DeviceDescriptors(ctx, rng.integers(0, 256, (n, 32), dtype=np.uint8)) for _ in range(B)]
    tabs = [slamhip.Top2Table(ctx, n) for _ in range(B)]
    bat = lambda: slamhip.knn2_device_batch(ctx, [(qs[i].buf, n, ts[i].buf, n, tabs[i].idx, tabs[i].dist) for i in range(B)])
    line = f"{B:2d} x {n}^2:"
    for chunk in (0, 64, 128, 160, 192, 256, 384, 512, 1024, 2048):
        if chunk > n:
            continue
        if chunk:
            os.environ["SLAM_EXP_CHUNK"] = str(chunk)
        else:
            os.environ.pop("SLAM_EXP_CHUNK", None)
        for _ in range(20):
            bat()
        ctx.sync()
        ctx.timer_start()
        for _ in range(100):
            bat()
        us = ctx.timer_stop() / 100 * 1e3
        line += f"  chunk={chunk or 'plan'}: {us:7.1f} us ({B * n * n / us * 1e6:.2e})"
    os.environ.pop("SLAM_EXP_CHUNK", None)
    print(line, ctx.plan_info(n, n), flush=True)
    for o in (*qs, *ts, *tabs):
        o.free()
