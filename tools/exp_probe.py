"""Experiments behind DESIGN.md §3 (development aid, not part of the product):

    SLAM_LIB=tools/exp/libslamhip_keepbound.so python tools/exp_probe.py keep  NxM ...
        an experiment build whose last arriver leaves the FINAL 2nd-best distance in bound[] instead of restoring it, so
        the next search of the same inputs starts every block with a perfect threshold: the ceiling of any seeding scheme.
    python tools/exp_probe.py pipe NxM ...
        the shipped library; searches issued alternately on two contexts (two streams, two merge states) so the tail of
        one launch overlaps the head of the next, against the same searches back to back on one context.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))
import slamhip  # noqa: E402
from slamhip import _lib  # noqa: E402

if os.environ.get("SLAM_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["SLAM_LIB"])
mode = sys.argv[1]
sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[2:]] or [(8192, 65536), (65536, 65536), (4096, 4096), (20000, 20000)]


def setup(ctx, n, m):
    q = slamhip.DeviceDescriptors(ctx, np.random.default_rng(228).integers(0, 256, (n, 32), dtype=np.uint8))
    t = slamhip.DeviceDescriptors(ctx, np.random.default_rng(229).integers(0, 256, (m, 32), dtype=np.uint8))
    tab = slamhip.Top2Table(ctx, n)
    return q, t, tab


def reps_for(n, m):
    return 200 if n * m < 1 << 28 else (100 if n * m < 1 << 31 else 40)


if mode == "keep":
    ctx = slamhip.Context(0)
    lib, h = ctx.lib, ctx.handle
    for n, m in sizes:
        q, t, tab = setup(ctx, n, m)
        f = lambda: lib.slam_bf_knn2_u256(h, q.buf.ptr, n, t.buf.ptr, m, 0, tab.idx.ptr, tab.dist.ptr)
        for bpc, queue in ((0, 1), (16, -1), (32, -1)):      # the queue plan; one block per chunk at 16 / 32 blocks per CU
            ctx.set_tuning(blocks_per_cu=bpc, lead_rows=-1 if queue < 0 else 0, queue=queue)
            lib.slam_bf_reset_state(h)
            f()
            ctx.sync()
            first = tab.download()
            reps = reps_for(n, m)
            for _ in range(24):
                f()
            ctx.sync()
            ctx.timer_start()
            for _ in range(reps):
                f()
            us = ctx.timer_stop() / reps * 1e3
            got = tab.download()
            same = np.array_equal(got[0], first[0]) and np.array_equal(got[1], first[1])
            print(f"keep  {n}x{m} queue={queue:2d} bpc={bpc:2d} {us:9.1f} us {n * m / us / 1e6:7.3f} Tpairs/s same={same}", flush=True)
        lib.slam_bf_reset_state(h)
        for o in (tab, q, t):
            o.free()
elif mode == "pipe":
    a, b = slamhip.Context(0), slamhip.Context(0)
    for n, m in sizes:
        sa, sb = setup(a, n, m), setup(b, n, m)
        fa = lambda: a.lib.slam_bf_knn2_u256(a.handle, sa[0].buf.ptr, n, sa[1].buf.ptr, m, 0, sa[2].idx.ptr, sa[2].dist.ptr)
        fb = lambda: b.lib.slam_bf_knn2_u256(b.handle, sb[0].buf.ptr, n, sb[1].buf.ptr, m, 0, sb[2].idx.ptr, sb[2].dist.ptr)
        reps = reps_for(n, m)
        for label, seq in (("one ctx ", (fa, fa)), ("two ctxs", (fa, fb))):
            for _ in range(24):
                seq[0](); seq[1]()
            a.sync(); b.sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                seq[0](); seq[1]()
            a.sync(); b.sync()
            us = (time.perf_counter() - t0) / (2 * reps) * 1e6
            print(f"pipe  {n}x{m} {label} {us:9.1f} us per search {n * m / us / 1e6:7.3f} Tpairs/s", flush=True)
        ra, rb = sa[2].download(), sb[2].download()
        print("      same tables on both contexts:", np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1]), flush=True)
        for o in (*sa, *sb):
            o.free()
