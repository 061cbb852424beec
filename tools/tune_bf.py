"""Sweep the top-2 kernel's tuning knobs on one GPU (development aid, not part of the product)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))
import slamhip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
m = int(sys.argv[2]) if len(sys.argv) > 2 else n
ctx = slamhip.default_context()
lib = ctx.lib
rng = np.random.default_rng(228)
q = rng.integers(0, 256, (n, 32), dtype=np.uint8)
t = np.random.default_rng(229).integers(0, 256, (m, 32), dtype=np.uint8)
dq, dt = slamhip.DeviceDescriptors(ctx, q), slamhip.DeviceDescriptors(ctx, t)
tab = slamhip.Top2Table(ctx, n)
ref = None
for R in (1, 2, 4, 8, 1):
    for bpc in (0, 32):
        ctx.set_tuning(R=R, blocks_per_cu=bpc)
        for _ in range(3):
            slamhip.knn2_device(ctx, dq.buf, n, dt.buf, m, tab.idx, tab.dist)
        ctx.sync()
        ctx.prof_enable(True)
        ctx.timer_start()
        K = 10
        for _ in range(K):
            slamhip.knn2_device(ctx, dq.buf, n, dt.buf, m, tab.idx, tab.dist)
        ms = ctx.timer_stop() / K
        cnt, kms = ctx.prof_read()
        ctx.prof_enable(False)
        idx, dist = tab.download()
        if ref is None:
            ref = (idx, dist)
        ok = np.array_equal(idx, ref[0]) and np.array_equal(dist, ref[1])
        pairs = n * m
        print(f"R={R} bpc={bpc:2d} total {ms:8.4f} ms kernel {kms / cnt:8.4f} ms  {pairs / (kms / cnt) / 1e6:8.1f} Gpairs/s "
              f"valu_frac(16.6 ops @2.4GHz)={pairs * 16.625 / (kms / cnt * 1e-3) / 7.864e13:.3f} same={ok}", flush=True)
ctx.set_tuning()
