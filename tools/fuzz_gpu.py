"""Randomised parity run of the top-2 search against the CPU oracle (development aid; the suite has a short version).

    python tools/fuzz_gpu.py [iterations] [seed]

Every iteration draws a shape, a launch plan (queries per lane, blocks per CU, leader rows / chunk, tail length), and a
data distribution (uniform bytes, low-entropy rows with mass ties, planted duplicates across chunk boundaries), runs
slam_bf_knn2_u256 on device-resident rows and compares both tables bit for bit with oracle.bf_knn_c."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))
sys.path.insert(0, ROOT)
import slamhip  # noqa: E402
from oracle import oracle  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = slamhip.default_context()
t_start = time.time()
worst = None
for it in range(iters):
    n = int(rng.choice([rng.integers(1, 70), rng.integers(1, 700), rng.integers(1, 6000)]))
    m = int(rng.choice([rng.integers(1, 70), rng.integers(1, 3000), rng.integers(1, 50000)]))
    kind = rng.integers(0, 4)
    if kind == 0:                                           # two byte values only: nearly everything ties
        q = rng.choice(np.array([0, 255], np.uint8), (n, 32))
        t = rng.choice(np.array([0, 255], np.uint8), (m, 32))
    elif kind == 1:                                         # a few distinct rows repeated
        base = rng.integers(0, 256, (8, 32), dtype=np.uint8)
        q, t = base[rng.integers(0, 8, n)], base[rng.integers(0, 8, m)]
    else:
        q = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        t = rng.integers(0, 256, (m, 32), dtype=np.uint8)
        if kind == 3 and m > 4:                             # planted copies of queries at random train rows, some twice
            rows = rng.integers(0, m, min(n, m // 2))
            t[rows] = q[rng.integers(0, n, len(rows))]
            t[rng.integers(0, m, len(rows) // 2 + 1)] = t[rows[: len(rows) // 2 + 1]]
    knobs = dict(R=int(rng.choice([0, 1, 1, 2, 4, 8])), blocks_per_cu=int(rng.choice([0, 0, 1, 4, 16, 64])),
                 lead_rows=int(rng.choice([0, 0, -1, 32, 256, 1024, 4096])), lead_chunk=int(rng.choice([0, 0, 32, 64, 512])),
                 tail=int(rng.choice([0, 0, -1, 1, 5, 64])), cold=int(rng.choice([0, 0, -1, 16, 48, 128, 1024])),
                 chunk=int(rng.choice([0, 0, 0, 32, 96, 256, 1024])))
    knobs["feed"] = int(rng.choice([0, -1, 1])) if knobs["R"] in (0, 1) else 0     # train rows through the LDS tile / SGPRs
    # queue plans (resident workers drawing chunks by ticket): forced on a third of the searches that may take one
    knobs["queue"] = int(rng.choice([0, 1, -1])) if knobs["R"] in (0, 1) and knobs["feed"] != -1 else int(rng.choice([0, -1]))
    knobs["merge"] = int(rng.choice([0, 1, -1]))          # how the workers of a queue plan exchange what they know
    ctx.set_tuning(**knobs)
    dq, dt = slamhip.DeviceDescriptors(ctx, q), slamhip.DeviceDescriptors(ctx, t)
    tab = slamhip.Top2Table(ctx, n)
    slamhip.knn2_device(ctx, dq.buf, n, dt.buf, m, tab.idx, tab.dist)
    gi, gd = tab.download()
    ei, ed = oracle.bf_knn_c(q, t, 2, threads=8)
    ok = np.array_equal(gi, ei) and np.array_equal(gd, ed) and ctx.state_dirty() == 0
    for o in (tab, dq, dt):
        o.free()
    if not ok:
        print(f"MISMATCH at iteration {it}: n={n} m={m} kind={kind} knobs={knobs} plan={ctx.plan_info(n, m)}", flush=True)
        bad = np.flatnonzero((gi != ei).any(1) | (gd != ed).any(1))[:5]
        for b in bad:
            print("  query", b, "got", gi[b], gd[b], "expected", ei[b], ed[b])
        ctx.set_tuning()
        sys.exit(1)
    if it % 200 == 199:
        print(f"{it + 1} iterations ok ({time.time() - t_start:.0f} s)", flush=True)
ctx.set_tuning()
print(f"fuzz ok: {iters} iterations, seed {seed}, {time.time() - t_start:.0f} s")

# ---- the per-frame host-buffer calls: match (with and without the frame cache), ratio, crossCheck ---------------------
t_start = time.time()
cache = slamhip.matching.FrameCache(ctx)
prev = None
for it in range(iters // 4):
    n = int(rng.choice([0, rng.integers(1, 300), rng.integers(1, 5000), rng.integers(4000, 4200)]))
    m = int(rng.choice([0, rng.integers(1, 300), rng.integers(1, 5000), rng.integers(4000, 4200)]))
    low = rng.integers(0, 3) == 0
    gen = (lambda k: rng.choice(np.array([0, 255, 15], np.uint8), (k, 32))) if low else (lambda k: rng.integers(0, 256, (k, 32), dtype=np.uint8))
    q = gen(n)
    t = prev if (prev is not None and rng.integers(0, 2)) else gen(m)      # half of the calls continue the frame sequence
    thr = [None, 0.0, 20.0, 64.0, 130.0, 400.0][rng.integers(0, 6)]
    got = slamhip.match_arrays(t.copy(), q, thr, cache=cache)
    exp = oracle.bf_match_c(t, q, thr, threads=4)
    ok = all(np.array_equal(a, b) for a, b in zip(got, exp))
    if ok and n and len(t) and it % 3 == 0:
        ei, ed = oracle.bf_knn_c(q, t, 2, threads=4)
        keep = oracle.bf_ratio_c(ei, ed, 0.8)
        rq, rt, rd = slamhip.ratio_test_arrays(q, t, 0.8, ctx=ctx)
        oi, od = oracle.bf_cross_check_c(q, t, threads=4)
        cq, ct, cd = slamhip.cross_check_arrays(q, t, ctx=ctx)
        ok = (np.array_equal(rq, np.flatnonzero(keep)) and np.array_equal(rt, ei[keep, 0]) and
              np.array_equal(cq, np.flatnonzero(oi >= 0)) and np.array_equal(ct, oi[oi >= 0]) and
              np.array_equal(cd, od[oi >= 0].astype(np.float32)))
    if not ok:
        print(f"HOST-CALL MISMATCH at iteration {it}: n={n} m={len(t)} thr={thr} low={low}", flush=True)
        sys.exit(1)
    prev = q if n else None
    if it % 500 == 499:
        print(f"{it + 1} host-call iterations ok ({time.time() - t_start:.0f} s, cache hits {cache.hits}/{cache.calls})", flush=True)
cache.free()
print(f"host-call fuzz ok: {iters // 4} iterations, cache hits {cache.hits}/{cache.calls}, {time.time() - t_start:.0f} s")

# ---- several searches in one launch (slam_bf_knn2_batch_u256): ragged batches against the oracle --------------------------
t_start = time.time()
for it in range(iters // 20):
    B = int(rng.integers(1, 9))
    pairs = []
    for _ in range(B):
        n = int(rng.choice([0, rng.integers(1, 70), rng.integers(1, 3000)]))
        m = int(rng.choice([0, rng.integers(1, 70), rng.integers(1, 9000)]))
        low = rng.integers(0, 3) == 0
        gen = (lambda k: rng.choice(np.array([0, 255, 15], np.uint8), (k, 32))) if low else (lambda k: rng.integers(0, 256, (k, 32), dtype=np.uint8))
        pairs.append((gen(n), gen(m)))
    got = slamhip.knn_match_arrays_batch(pairs, ctx=ctx)
    for (q, t), (gi, gd) in zip(pairs, got):
        if len(q) == 0:
            continue
        ei, ed = oracle.bf_knn_c(q, t, 2, threads=8)
        if not (np.array_equal(gi, ei) and np.array_equal(gd, ed)):
            print(f"BATCH MISMATCH at iteration {it}: shapes {[(len(a), len(b)) for a, b in pairs]}", flush=True)
            sys.exit(1)
    if it % 100 == 99:
        print(f"{it + 1} batches ok ({time.time() - t_start:.0f} s)", flush=True)
print(f"batch fuzz ok: {iters // 20} batches, {time.time() - t_start:.0f} s")
