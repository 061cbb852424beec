"""How often the top-2 kernel takes its update path, counted exactly by the experiment build libslamhip_count.so
(tools/build_exp.sh): per launch, the 16-row groups whose filter fired, the rows inside them that were really
updated, and the rows of the one-by-one tail path that fired.  Shares are of all wave-rows / wave-groups of the launch.

    python tools/fire_probe.py [NxM ...]      for the shipped plan, and with leaders / tail switched off
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))
import slamhip  # noqa: E402
from slamhip import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "tools", "exp", "libslamhip_count.so")
sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(8192, 65536), (65536, 65536), (20000, 20000), (4096, 4096)]
ctx = slamhip.Context(0)
lib = ctx.lib
lib.slam_exp_set_fire.argtypes = [ctypes.c_void_p]
cnt = ctx.malloc(64)
for n, m in sizes:
    q = slamhip.DeviceDescriptors(ctx, np.random.default_rng(228).integers(0, 256, (n, 32), dtype=np.uint8))
    t = slamhip.DeviceDescriptors(ctx, np.random.default_rng(229).integers(0, 256, (m, 32), dtype=np.uint8))
    tab = slamhip.Top2Table(ctx, n)
    # the counters sit in the C++ (LDS-tile) form of the scan; the SGPR-fed form runs the same filter with the same thresholds
    for label, kw in (("shipped plan (LDS-tile form forced)", {"feed": -1}), ("no leaders, no tail (round-1 plan)", {"lead_rows": -1, "tail": -1, "feed": -1})):
        ctx.set_tuning(**kw)
        for _ in range(3):
            slamhip.knn2_device(ctx, q.buf, n, t.buf, m, tab.idx, tab.dist)
        ctx.sync()
        cnt.upload(np.zeros(8, np.uint64))
        assert lib.slam_exp_set_fire(cnt.ptr) == 0
        reps = 5
        for _ in range(reps):
            slamhip.knn2_device(ctx, q.buf, n, t.buf, m, tab.idx, tab.dist)
        ctx.sync()
        assert lib.slam_exp_set_fire(None) == 0
        groups, rows, singles = (cnt.download(np.uint64, (8,))[:3] / reps).tolist()
        wave_rows = ((n + 63) // 64) * m
        p = ctx.plan_info(n, m)
        print(f"{n}x{m} {label}: plan {p}\n    update path taken by {100 * (rows + singles) / wave_rows:.2f} % of the wave-rows "
              f"({rows + singles:.0f} of {wave_rows}), {100 * groups / (wave_rows / 16):.1f} % of the 16-row groups", flush=True)
    ctx.set_tuning()
    for o in (tab, q, t):
        o.free()
