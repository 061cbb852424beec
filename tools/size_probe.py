import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))
from slamhip.device import default_context
ctx = default_context(); lib, h = ctx.lib, ctx.handle
rng = np.random.default_rng(1)
SIZES = ((200, 200), (1000, 1000), (2000, 2000), (3000, 3000), (4096, 4096), (6000, 6000), (8192, 8192), (12000, 12000), (20000, 20000), (65536, 4096), (4096, 65536), (65536, 8192), (8192, 65536), (32768, 32768), (65536, 65536), (131072, 1048576), (1048576, 1048576))
if os.environ.get('PROBE_SMALL'):
    SIZES = SIZES[:12]
for n, m in SIZES:
    q = ctx.upload(rng.integers(0, 256, (n, 32), dtype=np.uint8)); t = ctx.upload(rng.integers(0, 256, (m, 32), dtype=np.uint8))
    oi, od = ctx.malloc(n * 8), ctx.malloc(n * 8)
    f = lambda: lib.slam_bf_knn2_u256(h, q.ptr, n, t.ptr, m, 0, oi.ptr, od.ptr)
    for _ in range(30 if n < 100000 else 1): f()
    ctx.sync(); ctx.timer_start()
    reps = 200 if n < 60000 else (50 if n < 100000 else 3)
    for _ in range(reps): f()
    print(f"{n}x{m}: {(ms := ctx.timer_stop() / reps) * 1e3:9.1f} us  {n * m / ms / 1e9:8.1f} Gpairs/ms-normalised = {n * m / (ms * 1e-3):.3e} pairs/s", flush=True)
    for b in (q, t, oi, od): b.free()
