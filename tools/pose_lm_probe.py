"""Where the device time of slam_pose_optimize_f64 goes (development aid): launches with schedules that isolate the
evaluation (rounds x 2 evaluations, no LM iterations) from the trials (solve + exponential + evaluation).
    python tools/pose_lm_probe.py [edges ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))
import slamhip
from slamhip import _lib
if os.environ.get('SLAM_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['SLAM_LIB'])
from slamhip.pose_opt import se3_exp
ctx = slamhip.Context(0)
lib = ctx.lib
rng = np.random.default_rng(228)
for O in [int(a) for a in sys.argv[1:]] or [50, 200, 256, 257, 512]:
    X = np.c_[rng.uniform(-4, 4, (O, 2)), rng.uniform(6, 15, O)]
    pix = np.c_[458.654 * X[:, 0] / X[:, 2] + 367.215, 457.296 * X[:, 1] / X[:, 2] + 248.375] + rng.normal(0, 0.3, (O, 2))
    pix[::9] += 70.0
    T0 = se3_exp([0.01, -0.01, 0.005, 0.05, -0.03, 0.04])
    d = [ctx.upload(np.ascontiguousarray(T0[:3, :4].reshape(12))), ctx.upload(X), ctx.upload(pix), ctx.malloc(96), ctx.malloc(O), ctx.malloc(O * 8), ctx.malloc(8)]
    res = {}
    for rounds, iters in ((0, 0), (1, 0), (4, 0), (8, 0), (1, 10), (4, 10)):
        call = lambda: lib.slam_pose_optimize_f64(ctx.handle, d[0].ptr, d[1].ptr, d[2].ptr, O, 458.654, 457.296, 367.215, 248.375, rounds, iters,
                                                  5.991 ** 2, 1.0, d[3].ptr, d[4].ptr, d[5].ptr, d[6].ptr)
        for _ in range(5):
            assert call() == 0
        ctx.sync()
        ctx.timer_start()
        for _ in range(100):
            call()
        res[(rounds, iters)] = (ctx.timer_stop() / 100 * 1e3, int(d[6].download(np.int32, (2,))[1]))
    ev = (res[(8, 0)][0] - res[(4, 0)][0]) / 8
    full, steps = res[(4, 10)]
    print(f"{O} edges: empty launch {res[(0, 0)][0]:.1f} us; one evaluation {ev:.2f} us; full schedule {full:.1f} us with {steps} accepted steps "
          f"-> about {(full - res[(4, 0)][0]) / max(steps, 1):.2f} us per accepted step (solve + exponential + evaluation + bookkeeping); "
          f"rounds=1: {res[(1, 10)][0]:.1f} us / {res[(1, 10)][1]} steps")
    for b in d:
        b.free()
