"""Where the host-visible time of a frame-sized match() goes (development aid): the bare C call, the Python wrapper,
and the device time of the two kernels."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))
import slamhip
from slamhip import matching as m
from feature_matchers import BruteForceFeatureMatcher
ctx = slamhip.default_context(); lib, h = ctx.lib, ctx.handle
rng = np.random.default_rng(1)
for n in (200, 1000):
    q = rng.integers(0, 256, (n, 32), dtype=np.uint8); t = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    qi, ti, d = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.float32)
    cnt = ctypes.c_int64(0)
    args = (h, q.ctypes.data, n, t.ctypes.data, None, n, None, 0, 0.0, qi.ctypes.data, ti.ctypes.data, d.ctypes.data, ctypes.byref(cnt))
    def bare(): lib.slam_bf_match_host(*args)
    dt = ctx.upload(t); keep = ctx.malloc(n * 32)
    args2 = (h, q.ctypes.data, n, None, dt.ptr, n, keep.ptr, 0, 0.0, qi.ctypes.data, ti.ctypes.data, d.ctypes.data, ctypes.byref(cnt))
    def bare_cached(): lib.slam_bf_match_host(*args2)
    idx, dist = np.empty((n, 2), np.int32), np.empty((n, 2), np.int32)
    def bare_knn(): lib.slam_bf_knn2_u256_host(h, q.ctypes.data, n, t.ctypes.data, n, idx.ctypes.data, dist.ctypes.data)
    dq = ctx.upload(q); oi, od = ctx.malloc(n * 8), ctx.malloc(n * 8)
    def dev_only(): lib.slam_bf_knn2_u256(h, dq.ptr, n, dt.ptr, n, 0, oi.ptr, od.ptr); lib.slam_sync(h)
    def sync_only(): lib.slam_sync(h)
    bf = BruteForceFeatureMatcher(6)
    for f, name in ((sync_only, "slam_sync on an idle stream"), (dev_only, "slam_bf_knn2_u256 (device rows) + slam_sync"),
                    (bare_knn, "slam_bf_knn2_u256_host (C call only)"), (bare, "slam_bf_match_host, host train (C call only)"),
                    (bare_cached, "slam_bf_match_host, device train + keep (C call only)"),
                    (lambda: m.match_arrays(t, q), "match_arrays() (Python wrapper)"), (lambda: bf.match(t, q), "BruteForceFeatureMatcher.match()")):
        for _ in range(20): f()
        t0 = time.perf_counter()
        for _ in range(200): f()
        print(f"n={n:5d} {name:58s} {(time.perf_counter() - t0) / 200 * 1e6:7.1f} us")
    ctx.prof_enable(True)
    for _ in range(50): lib.slam_bf_knn2_u256(h, dq.ptr, n, dt.ptr, n, 0, oi.ptr, od.ptr)
    k, ms = ctx.prof_read(); ctx.prof_enable(False)
    print(f"n={n:5d} top-2 kernel between its two events: {ms / k * 1e3:.1f} us")
