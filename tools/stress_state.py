"""Soak test of the merge state across launches (development aid): searches of changing shape and changing data, back
to back on ONE context, every table compared with the CPU oracle.  The per-query merge state (best / bound / arrivals) is
restored by the last block of every query block inside the kernel, and since the second half of round 3 the arrival ticket
has no release fence in front of it: a bound or a key of search k that survived into search k + 1 would show up here as
a wrong neighbour.  Since round 4 the state itself is downloaded after every search as well (slam_bf_state_dirty: every
word of best / bound / arrivals / cursor at its idle value), so a leftover is caught even when the next search would have
hidden it.  Shapes are drawn from the regimes with queue plans, leaders, several dispatch rounds and a shrinking tail; every
third search forces the other kind of plan (queue on / off).

    python tools/stress_state.py [iterations] [seed]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))
sys.path.insert(0, ROOT)
import slamhip  # noqa: E402
from oracle import oracle  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = slamhip.Context(0)
pool = rng.integers(0, 256, (1 << 18, 32), dtype=np.uint8)
t0 = time.time()
pairs = 0
for it in range(iters):
    n = int(rng.choice([rng.integers(1, 3000), rng.integers(3000, 20000)]))
    m = int(rng.choice([rng.integers(256, 16384), rng.integers(16384, 140000)]))
    qo, to = int(rng.integers(0, (1 << 18) - n)), int(rng.integers(0, (1 << 18) - m))
    mask = rng.integers(0, 256, 32, dtype=np.uint8)
    q = pool[qo:qo + n] ^ mask                      # different data every time, cheaply
    t = pool[to:to + m] ^ rng.integers(0, 256, 32, dtype=np.uint8)
    if it % 3 == 0:                                  # planted near-duplicates: tight bounds, ties on the index
        rows = rng.integers(0, m, min(n, 64))
        t[rows] = q[rng.integers(0, n, len(rows))]
    ctx.set_tuning(queue=(0, 0, 1, 0, 0, -1)[it % 6], merge=(0, 1, -1, 0)[it % 4])
    gi, gd = slamhip.knn_match_arrays(q, t, 2, ctx=ctx)
    dirty = ctx.state_dirty()
    if dirty:
        print(f"STATE NOT IDLE after iteration {it}: {dirty} words, n={n} m={m} plan={ctx.plan_info(n, m)}", flush=True)
        sys.exit(1)
    ei, ed = oracle.bf_knn_c(q, t, 2, threads=os.cpu_count() or 1)
    pairs += n * m
    if not (np.array_equal(gi, ei) and np.array_equal(gd, ed)):
        bad = np.flatnonzero((gi != ei).any(1) | (gd != ed).any(1))
        print(f"MISMATCH at iteration {it}: n={n} m={m} plan={ctx.plan_info(n, m)} first bad queries {bad[:5]}", flush=True)
        sys.exit(1)
    if it % 50 == 49:
        print(f"{it + 1} searches ok ({time.time() - t0:.0f} s)", flush=True)
print(f"state soak ok: {iters} searches of changing shape and data on one context, merge state idle after every one, "
      f"{pairs:.3e} pairs, {time.time() - t0:.0f} s")
