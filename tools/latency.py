"""Host-visible latency of the drop-in match() at the reference's own size (200 x 200, slam.py:23)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))
from feature_matchers import BruteForceFeatureMatcher
import slamhip
bf = BruteForceFeatureMatcher(norm_type=6)
rng = np.random.default_rng(228)
for n in (200, 2000, 20000):
    src = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    qry = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    for f, name in ((lambda: bf.match(src, qry), "match() -> DMatch list"), (lambda: bf.match_arrays(src, qry), "match_arrays()"),
                    (lambda: bf.knn_match_arrays(qry, src, 2), "knn_match_arrays(k=2)")):
        for _ in range(5):
            f()
        t0 = time.perf_counter()
        for _ in range(50):
            f()
        print(f"n={n:6d} {name:28s} {(time.perf_counter() - t0) / 50 * 1e3:8.3f} ms per call")
# the tracking loop as it really is: a different number of features on every frame (ORB finds what it finds), fresh copies of
# both matrices per call (Frame.get_descriptors), the previous frame's rows recognised by the matcher's frame cache
frames = [rng.integers(0, 256, (int(k), 32), dtype=np.uint8) for k in rng.integers(150, 201, 80)]
for _ in range(2):
    for a, b in zip(frames[:-1], frames[1:]):
        bf.match(a.copy(), b.copy())
t0 = time.perf_counter()
for a, b in zip(frames[:-1], frames[1:]):
    bf.match(a.copy(), b.copy())
print(f"n=150..200 varying per frame, match(last.copy(), cur.copy()) -> MatchList  {(time.perf_counter() - t0) / 79 * 1e3:8.3f} ms per frame")
cq, ct = frames[0], frames[1]
for f, name in ((lambda: bf.cross_check_match(cq, ct), "cross_check_match (forward + reverse search)"),):
    for _ in range(5):
        f()
    t0 = time.perf_counter()
    for _ in range(50):
        f()
    print(f"n=   200 {name:40s} {(time.perf_counter() - t0) / 50 * 1e3:8.3f} ms per call")
big_q = rng.integers(0, 256, (65536, 32), dtype=np.uint8); big_t = rng.integers(0, 256, (65536, 32), dtype=np.uint8)
for _ in range(3):
    bf.knn_match_arrays(big_q, big_t, 2)
t0 = time.perf_counter()
for _ in range(20):
    bf.knn_match_arrays(big_q, big_t, 2)
dt = (time.perf_counter() - t0) / 20
print(f"n= 65536 knn_match_arrays(k=2) on host buffers (PCIe inclusive) {dt * 1e3:8.3f} ms per call = {65536 * 65536 / dt:.3e} pairs/s")
# BASELINE configs[1]: 4096 x 4096, knn = 2 + Lowe ratio test (0.75), one GPU - rows resident on the device (search + the
# selection kernel between two events) and through the host-buffer call (4096 rows a side: zero-copy over PCIe)
_r1 = np.random.default_rng(4096)                     # a generator of its own: the draws below stay what they were
c1q, c1t = _r1.integers(0, 256, (4096, 32), dtype=np.uint8), _r1.integers(0, 256, (4096, 32), dtype=np.uint8)
c1t[:40] = c1q[:40]                                   # some true matches, so that the ratio test keeps something
_ctx = slamhip.default_context()
_dq, _dt = slamhip.DeviceDescriptors(_ctx, c1q), slamhip.DeviceDescriptors(_ctx, c1t)
_tab, _keep = slamhip.Top2Table(_ctx, 4096), _ctx.malloc(4096)
import ctypes as _ct
_cnt, _mind = _ct.c_int64(0), _ct.c_int32(0)
def _dev():                                           # round 3: search + selection kernel + the count read back (one synchronisation)
    slamhip.knn2_device(_ctx, _dq.buf, 4096, _dt.buf, 4096, _tab.idx, _tab.dist)
    assert _ctx.lib.slam_bf_match_filter(_ctx.handle, _tab.idx.ptr, _tab.dist.ptr, 4096, 2, 0.75, _keep.ptr, _ct.byref(_cnt), _ct.byref(_mind)) == 0
def _fused():                                         # round 4: the search's decode makes the selection (ONE launch, one synchronisation)
    return slamhip.knn2_select_device(_ctx, _dq.buf, 4096, _dt.buf, 4096, _tab.idx, _tab.dist, _keep, 2, 0.75)
two_step_us = 0.0
for _f in (_dev, _fused):
    for _ in range(20):
        _f()
    t0 = time.perf_counter()
    for _ in range(200):
        _f()
    dev_us = (time.perf_counter() - t0) / 200 * 1e6
    two_step_us = two_step_us or dev_us
assert _fused() == _cnt.value
for _ in range(5):
    kept = slamhip.ratio_test_arrays(c1q, c1t, 0.75)
t0 = time.perf_counter()
for _ in range(50):
    kept = slamhip.ratio_test_arrays(c1q, c1t, 0.75)
print(f"n=  4096 knn=2 + ratio 0.75 (BASELINE configs[1]): {dev_us:6.1f} us per call with the rows resident (ONE launch: the search's decode makes the selection; count read back; "
      f"{two_step_us:.1f} us as search + selection kernel) = "
      f"{4096 * 4096 / dev_us * 1e6:.2e} pairs/s; ratio_test_arrays() on host buffers {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per call, "
      f"{len(kept[0])} matches kept")
for _o in (_dq, _dt, _tab, _keep):
    _o.free()
rm = slamhip.ResidentMatcher()
fr = [rng.integers(0, 256, (200, 32), dtype=np.uint8) for _ in range(60)]
rm.push(fr[0])
t0 = time.perf_counter()
for f in fr[1:]:
    rm.push(f)
print(f"ResidentMatcher.push n=200: {(time.perf_counter() - t0) / 59 * 1e3:.3f} ms per frame")

# pose-only refinement: the reference's per-frame _correct_current_pose (<= 200 edges)
from backend import Backend
from slamhip.pose_opt import se3_exp
be = Backend()
X = np.c_[rng.uniform(-4, 4, (200, 2)), rng.uniform(6, 15, 200)]
T = np.eye(4)
pix = np.c_[458.654 * X[:, 0] / X[:, 2] + 367.215, 457.296 * X[:, 1] / X[:, 2] + 248.375] + rng.normal(0, 0.3, (200, 2))
T0 = se3_exp([0.01, -0.01, 0.005, 0.05, -0.03, 0.04]) @ T
for on_device in (False, True):
    for _ in range(3):
        be.optimize_pose(T0, X, pix, 458.654, 457.296, 367.215, 248.375, on_device=on_device)
    t0 = time.perf_counter()
    for _ in range(20):
        r = be.optimize_pose(T0, X, pix, 458.654, 457.296, 367.215, 248.375, on_device=on_device)
    print(f"optimize_pose 200 edges, {'one launch on device' if on_device else 'host-driven LM      '}: {(time.perf_counter() - t0) / 20 * 1e3:7.3f} ms per call, {r.iterations} accepted steps")
# several frames refined in one launch (one workgroup per frame) against the same work one launch at a time
ps = [se3_exp(rng.normal(0, 0.01, 6)) @ T for _ in range(16)]
for _ in range(3):
    be.optimize_poses(np.stack(ps), [X] * 16, [pix] * 16, 458.654, 457.296, 367.215, 248.375)
t0 = time.perf_counter()
for _ in range(10):
    be.optimize_poses(np.stack(ps), [X] * 16, [pix] * 16, 458.654, 457.296, 367.215, 248.375)
print(f"optimize_poses 16 frames x 200 edges, one launch: {(time.perf_counter() - t0) / 10 * 1e3:7.3f} ms per call ({(time.perf_counter() - t0) / 160 * 1e3:.3f} ms per frame)")

# The pose refinement beside a plain C loop on ONE host core (oracle/pose_lm_oracle.c; test infrastructure, the CPU baseline of
# this path: the reference runs it as g2o C++ calling back into Python per edge, frontend.py:272-291) and the kernel's own time.
import ctypes
sys.path.insert(0, ROOT)
from oracle import oracle
ctx = be.ctx
lib = ctx.lib
for O in (50, 200, 256, 257, 1000):
    Xo = np.c_[rng.uniform(-4, 4, (O, 2)), rng.uniform(6, 15, O)]
    po = np.c_[458.654 * Xo[:, 0] / Xo[:, 2] + 367.215, 457.296 * Xo[:, 1] / Xo[:, 2] + 248.375] + rng.normal(0, 0.3, (O, 2))
    po[::9] += 70.0
    po = po.astype(np.int32).astype(np.float64)
    p12 = np.ascontiguousarray(T0[:3, :4].reshape(12))
    oracle.pose_lm_c(p12, Xo, po, 458.654, 457.296, 367.215, 248.375)
    t0 = time.perf_counter()
    for _ in range(50):
        Tc, inl_c, _, acc_c = oracle.pose_lm_c(p12, Xo, po, 458.654, 457.296, 367.215, 248.375)
    cpu_ms = (time.perf_counter() - t0) / 50 * 1e3
    d = [ctx.upload(p12), ctx.upload(Xo), ctx.upload(po), ctx.malloc(96), ctx.malloc(O), ctx.malloc(O * 8), ctx.malloc(8)]
    call = lambda: lib.slam_pose_optimize_f64(ctx.handle, d[0].ptr, d[1].ptr, d[2].ptr, O, 458.654, 457.296, 367.215, 248.375, 4, 10,
                                              5.991 ** 2, 1.0, d[3].ptr, d[4].ptr, d[5].ptr, d[6].ptr)
    for _ in range(5):
        assert call() == 0
    ctx.sync()
    ctx.timer_start()
    for _ in range(50):
        call()
    dev_ms = ctx.timer_stop() / 50
    got = d[3].download(np.float64, (12,))
    st = d[6].download(np.int32, (2,))
    same = bool(np.array_equal(d[4].download(np.uint8, (O,)).astype(bool), inl_c))
    print(f"pose refinement {O:5d} edges: device {dev_ms * 1e3:7.1f} us per launch ({st[1]} accepted steps) | plain C on one host core "
          f"{cpu_ms * 1e3:7.1f} us ({acc_c} steps) | max pose difference {np.abs(got - Tc).max():.1e}, same inlier set: {same}")
    for b in d:
        b.free()

# several frame-sized searches in ONE launch (slam_bf_knn2_batch_u256) against the same searches one launch at a time
for n, B in ((200, 2), (4096, 16), (1000, 8)):
    qs = [slamhip.DeviceDescriptors(ctx, rng.integers(0, 256, (n, 32), dtype=np.uint8)) for _ in range(B)]
    ts = [slamhip.DeviceDescriptors(ctx, rng.integers(0, 256, (n, 32), dtype=np.uint8)) for _ in range(B)]
    tabs = [slamhip.Top2Table(ctx, n) for _ in range(B)]
    one = lambda: [slamhip.knn2_device(ctx, qs[i].buf, n, ts[i].buf, n, tabs[i].idx, tabs[i].dist) for i in range(B)]
    bat = lambda: slamhip.knn2_device_batch(ctx, [(qs[i].buf, n, ts[i].buf, n, tabs[i].idx, tabs[i].dist) for i in range(B)])
    res = {}
    for name, f in (("one launch each", one), ("ONE launch", bat)):
        for _ in range(20):
            f()
        ctx.sync()
        ctx.timer_start()
        for _ in range(100):
            f()
        res[name] = ctx.timer_stop() / 100 * 1e3
    print(f"{B:2d} searches of {n} x {n}: {res['one launch each']:8.1f} us one launch each ({B * n * n / res['one launch each'] * 1e6:.2e} pairs/s), "
          f"{res['ONE launch']:8.1f} us in one launch ({B * n * n / res['ONE launch'] * 1e6:.2e} pairs/s)")
    for o in (*qs, *ts, *tabs):
        o.free()
