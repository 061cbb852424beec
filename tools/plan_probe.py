"""Device time of slam_bf_knn2_u256 for a sweep of launch plans (development aid).

    [PLANS="lead_rows,lead_chunk,tail,bpc;..."] python tools/plan_probe.py [NxM ...]

0 = the shipped choice of that knob, -1 = off.  Every variant's tables are compared with the first plan's."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))
import slamhip  # noqa: E402

ctx = slamhip.default_context()
lib, h = ctx.lib, ctx.handle
sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(8192, 65536), (65536, 65536), (4096, 4096), (20000, 20000)]
default = "-1,0,-1,0;0,0,0,0;-1,0,0,0;0,0,-1,0;0,0,8,0;0,0,32,0;4096,0,0,0;2048,0,0,0"
plans = [tuple(int(v) for v in p.split(",")) for p in os.environ.get("PLANS", default).split(";")]
for n, m in sizes:
    q = slamhip.DeviceDescriptors(ctx, np.random.default_rng(228).integers(0, 256, (n, 32), dtype=np.uint8))
    t = slamhip.DeviceDescriptors(ctx, np.random.default_rng(229).integers(0, 256, (m, 32), dtype=np.uint8))
    tab = slamhip.Top2Table(ctx, n)
    ref = None
    f = lambda: lib.slam_bf_knn2_u256(h, q.buf.ptr, n, t.buf.ptr, m, 0, tab.idx.ptr, tab.dist.ptr)
    for _ in range(40):                       # clock spin-up before the first plan is timed
        f()
    for lead_rows, lead_chunk, tail, bpc in plans:
        ctx.set_tuning(blocks_per_cu=bpc, lead_rows=lead_rows, lead_chunk=lead_chunk, tail=tail)
        p = ctx.plan_info(n, m)
        reps = 200 if n * m < 1 << 28 else (60 if n * m < 1 << 31 else 30)
        for _ in range(max(reps // 4, 24)):
            f()
        ctx.sync()
        ctx.timer_start()
        for _ in range(reps):
            f()
        us = ctx.timer_stop() / reps * 1e3
        got = tab.download()
        if ref is None:
            ref = got
        same = np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
        print(f"{n}x{m} lead={lead_rows:5d}/{lead_chunk:4d} tail={tail:3d} bpc={bpc:2d} plan(chunk={p['chunk']},S={p['chunks']},"
              f"lead={p['lead_rows']}/{p['lead_chunks']},tail={p['tail_chunks']}) {us:9.1f} us {n * m / us / 1e6:7.3f} Tpairs/s same={same}", flush=True)
    ctx.set_tuning()
    for o in (tab, q, t):
        o.free()
