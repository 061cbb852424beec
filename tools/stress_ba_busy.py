"""The SLAM_ERR_BUSY path on a real device (development aid): a kernel on ANOTHER STREAM of this process holds the LDS of every
compute unit (tools/ubench/liblds_hog.so) while one-launch window adjustments are asked for.  (The hog has to sit in the same
process: the queues of another process are time-sliced against ours, its waves - LDS included - are swapped out while ours
run, and the adjustment completes as if the device were idle; that was the first form of this tool.)  Their workgroups (58 KiB of LDS each) cannot
become resident, the launch gives up at its first grid barrier after the barrier's time limit (50 ms of wall clock), the host
call returns SLAM_ERR_BUSY, and bundle_adjust_auto redoes the window with the per-phase kernels, which need no co-residency.
Also prints what round 3's count-bounded wait (2^22 polls) amounted to.

    hipcc --offload-arch=gfx950 -O3 -o tools/ubench/lds_hog tools/ubench/lds_hog.hip
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/ubench/liblds_hog.so tools/ubench/lds_hog.hip && python tools/stress_ba_busy.py
"""
import ctypes
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))
import slamhip  # noqa: E402
from slamhip.ba import bundle_adjust_auto, bundle_adjust_device, bundle_adjust_one_launch  # noqa: E402

HOG = os.path.join(ROOT, "tools", "ubench", "lds_hog")
print(subprocess.run([HOG, "polls", str(1 << 22)], capture_output=True, text=True, timeout=60).stdout.strip(), flush=True)

K, L = 7, 1500
rng = np.random.default_rng(7)
T = np.tile(np.eye(4), (K, 1, 1)); T[:, :3, 3] = rng.uniform(-0.5, 0.5, (K, 3))
X = np.c_[rng.uniform(-4, 4, (L, 2)), rng.uniform(6, 15, L)]
op = np.repeat(np.arange(K), L).astype(np.int32); ol = np.tile(np.arange(L), K).astype(np.int32)
pc = X[ol] + T[op, :3, 3]
cam = (458.654, 457.296, 367.215, 248.375)
meas = np.c_[cam[0] * pc[:, 0] / pc[:, 2] + cam[2], cam[1] * pc[:, 1] / pc[:, 2] + cam[3]] + rng.normal(0, 0.2, (len(op), 2))
args = (T, X + rng.normal(0, 0.05, X.shape), op, ol, meas, cam)
ctx = slamhip.Context(0)
quiet = bundle_adjust_one_launch(*args, iterations=5, fixed_poses=(0, 1), ctx=ctx)
ref = bundle_adjust_device(*args, iterations=5, fixed_poses=(0, 1), ctx=ctx)
print(f"idle device: one launch {quiet.chi2_initial:.1f} -> {quiet.chi2_final:.1f}; per-phase kernels -> {ref.chi2_final:.1f}", flush=True)

hoglib = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", "liblds_hog.so"))
hoglib.lds_hog_start.argtypes = [ctypes.c_double]
t_hog = time.perf_counter()
cus = hoglib.lds_hog_start(400.0)
assert cus > 0, cus
print(f"hog: 3 x 48 KiB of LDS on each of {cus} compute units for 400 ms, on another stream of this process", flush=True)
time.sleep(0.02)
for i in range(3):
    t0 = time.perf_counter()
    try:
        bundle_adjust_one_launch(*args, iterations=5, fixed_poses=(0, 1), ctx=ctx)
        outcome = "completed"
    except slamhip.SlamHipBusy as exc:
        outcome = f"SlamHipBusy after {(time.perf_counter() - t0) * 1e3:.1f} ms: {exc}"
    print(f"one-launch attempt {i}: {outcome}", flush=True)
    t0 = time.perf_counter()
    seen = []
    r = bundle_adjust_auto(*args, iterations=5, fixed_poses=(0, 1), ctx=ctx, on_busy=seen.append)
    same = np.array_equal(r.poses, ref.poses) and np.array_equal(r.points, ref.points)
    print(f"bundle_adjust_auto  {i}: fell back: {bool(seen)}, {(time.perf_counter() - t0) * 1e3:.1f} ms in all, result equals the per-phase "
          f"kernels' on an idle device: {same if seen else 'n/a (one launch)'}", flush=True)
assert hoglib.lds_hog_wait() == 0
print(f"hog: done {(time.perf_counter() - t_hog) * 1e3:.0f} ms after its launch", flush=True)
after = bundle_adjust_one_launch(*args, iterations=5, fixed_poses=(0, 1), ctx=ctx)
print(f"after the hog: one launch again, identical to the idle run: {np.array_equal(after.poses, quiet.poses) and after.chi2_final == quiet.chi2_final}")
