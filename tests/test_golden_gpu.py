"""GPU: every committed matching golden (tests/golden/kat_*.json, hand-derived from the semantics the reference relies
on, feature_matchers.py:36-44 + OpenCV's BFMatcher rules) is fed to the HIP path through the C ABI, and crossCheck is
checked as its contract (mutual nearest neighbours) on random and tie-heavy inputs."""
import ctypes
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
INT_MAX = 2**31 - 1


def gold(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def d(rows):
    return np.array(rows, np.uint8).reshape(-1, 32)


def test_ladder_ties_short_train_through_hip(gpu_ctx):
    """knn_match_arrays (slam_bf_knn2_u256_host) and the device-pointer form (slam_bf_knn2_u256)."""
    import slamhip

    for name in ("kat_ladder.json", "kat_ties.json"):
        g = gold(name)
        idx, dist = slamhip.knn_match_arrays(d(g["query"]), d(g["train"]), 2, ctx=gpu_ctx)
        assert idx.tolist() == g["idx"] and dist.tolist() == g["dist"], name
        i1, d1 = slamhip.knn_match_arrays(d(g["query"]), d(g["train"]), 1, ctx=gpu_ctx)
        assert i1[:, 0].tolist() == [r[0] for r in g["idx"]] and d1[:, 0].tolist() == [r[0] for r in g["dist"]]
        # device-resident rows, every kernel variant (queries per lane) and grid split
        dq, dt = slamhip.DeviceDescriptors(gpu_ctx, d(g["query"])), slamhip.DeviceDescriptors(gpu_ctx, d(g["train"]))
        tab = slamhip.Top2Table(gpu_ctx, dq.rows)
        try:
            for R, bpc in ((0, 0), (1, 1), (2, 4), (4, 16), (8, 2)):
                gpu_ctx.set_tuning(R=R, blocks_per_cu=bpc, lead_rows=32 * (R % 3), tail=bpc)
                slamhip.knn2_device(gpu_ctx, dq.buf, dq.rows, dt.buf, dt.rows, tab.idx, tab.dist)
                idx, dist = tab.download()
                assert idx.tolist() == g["idx"] and dist.tolist() == g["dist"], (name, R, bpc)
        finally:
            gpu_ctx.set_tuning()
            for o in (tab, dq, dt):
                o.free()
    g = gold("kat_short_train.json")
    idx, dist = slamhip.knn_match_arrays(d(g["query"]), d(g["train_one"]), 2, ctx=gpu_ctx)
    assert idx.tolist() == g["idx_one"] and dist.tolist() == g["dist_one"]
    idx, dist = slamhip.knn_match_arrays(d(g["query"]), np.zeros((0, 32), np.uint8), 2, ctx=gpu_ctx)
    assert idx.tolist() == g["idx_none"] and dist.tolist() == g["dist_none"]


def test_reference_filter_through_hip(gpu_ctx):
    """BruteForceFeatureMatcher.match(source, query, dist_threshold), feature_matchers.py:36-44: drop-in and arrays."""
    import slamhip
    from feature_matchers import BruteForceFeatureMatcher

    g = gold("kat_filter.json")
    bf = BruteForceFeatureMatcher(norm_type=6)
    for case in g["cases"]:
        ms = bf.match(d(g["source"]), d(g["query"]), case["thr"])
        assert [m.queryIdx for m in ms] == case["queryIdx"], case
        assert [m.trainIdx for m in ms] == case["trainIdx"], case
        assert [m.distance for m in ms] == case["distance"], case
        assert all(m.imgIdx == 0 for m in ms)
        q, t, dist = slamhip.match_arrays(d(g["source"]), d(g["query"]), case["thr"], ctx=gpu_ctx)
        assert q.tolist() == case["queryIdx"] and t.tolist() == case["trainIdx"] and dist.tolist() == case["distance"]
    assert bf.match(np.zeros((0, 32), np.uint8), d(g["query"])) == []
    assert bf.match(d(g["source"]), np.zeros((0, 32), np.uint8)) == []
    # the same selection through the device-pointer entry point (slam_bf_knn2_u256 + slam_bf_match_filter)
    dq, dt = slamhip.DeviceDescriptors(gpu_ctx, d(g["query"])), slamhip.DeviceDescriptors(gpu_ctx, d(g["source"]))
    tab = slamhip.Top2Table(gpu_ctx, dq.rows)
    keep = gpu_ctx.malloc(dq.rows)
    slamhip.knn2_device(gpu_ctx, dq.buf, dq.rows, dt.buf, dt.rows, tab.idx, tab.dist)
    for case in g["cases"]:
        cnt, mind = ctypes.c_int64(-1), ctypes.c_int32(-1)
        mode = 1 if case["thr"] else 0
        assert gpu_ctx.lib.slam_bf_match_filter(gpu_ctx.handle, tab.idx.ptr, tab.dist.ptr, dq.rows, mode,
                                                float(case["thr"] or 0.0), keep.ptr, ctypes.byref(cnt),
                                                ctypes.byref(mind)) == 0
        k = keep.download(np.uint8, (dq.rows,)).astype(bool)
        assert np.flatnonzero(k).tolist() == case["queryIdx"] and cnt.value == len(case["queryIdx"]) and mind.value == 2
    for o in (keep, tab, dq, dt):
        o.free()


def test_ratio_through_hip(gpu_ctx):
    import slamhip

    g = gold("kat_ratio.json")
    for ratio, key in ((0.75, "keep_075"), (0.5, "keep_050")):
        q, t, dist = slamhip.ratio_test_arrays(d(g["query"]), d(g["train"]), ratio, ctx=gpu_ctx)
        assert q.tolist() == [i for i, k in enumerate(g[key]) if k]
        assert t.tolist() == [0] * len(q) and dist.tolist() == [float(2 + i) for i in q.tolist()]


def test_multi_image_through_hip(gpu_ctx):
    import slamhip

    g = gold("kat_multi_image.json")
    imgs = [d(i) for i in g["images"]]
    img, tr, dist = slamhip.knn_match_collection(d(g["query"]), imgs, 2, ctx=gpu_ctx)
    assert img.tolist() == g["img"] and tr.tolist() == g["train"] and dist.tolist() == g["dist"]
    db = slamhip.KeyframeDatabase(gpu_ctx, capacity_rows=2)
    for im in imgs:
        db.add(im)
    img, tr, dist = db.query(d(g["query"]), 2)
    assert img.tolist() == g["img"] and tr.tolist() == g["train"] and dist.tolist() == g["dist"]
    db.free()


def _cross_check_abi(ctx, q, t):
    """slam_bf_knn2_u256 forward + reverse, then slam_bf_cross_check, all on device pointers."""
    import slamhip

    n, m = q.shape[0], t.shape[0]
    dq, dt = slamhip.DeviceDescriptors(ctx, q), slamhip.DeviceDescriptors(ctx, t)
    fwd, rev = slamhip.Top2Table(ctx, n), slamhip.Top2Table(ctx, m)
    oi, od = ctx.malloc(max(n, 1) * 4), ctx.malloc(max(n, 1) * 4)
    try:
        slamhip.knn2_device(ctx, dq.buf, n, dt.buf, m, fwd.idx, fwd.dist)
        slamhip.knn2_device(ctx, dt.buf, m, dq.buf, n, rev.idx, rev.dist)
        cnt = ctypes.c_int64(-1)
        rc = ctx.lib.slam_bf_cross_check(ctx.handle, fwd.idx.ptr, fwd.dist.ptr, n, rev.idx.ptr, m, oi.ptr, od.ptr,
                                         ctypes.byref(cnt))
        assert rc == 0
        out_i, out_d = oi.download(np.int32, (n,)), od.download(np.int32, (n,))
        assert cnt.value == int((out_i >= 0).sum())
        return out_i, out_d
    finally:
        for o in (oi, od, fwd, rev, dq, dt):
            o.free()


def test_cross_check_goldens_through_hip(gpu_ctx):
    """cv2.BFMatcher(crossCheck=True).match: mutual nearest neighbours only (kat_cross_check.json), through the
    one-call host path (slam_bf_match_host mode 3), the drop-in method and the device-pointer entry point."""
    import slamhip
    from feature_matchers import BruteForceFeatureMatcher

    g = gold("kat_cross_check.json")
    cases = [(d(g["query"]), d(g["train"]), g["out_idx"], g["out_dist"]),
             (d(g["query2"]), d(g["train"]), g["out_idx2"], g["out_dist2"]),
             (d(g["query3"]), d(g["train3"]), g["out_idx3"], g["out_dist3"])]
    for q, t, ei, ed in cases:
        keep = [i for i, v in enumerate(ei) if v >= 0]
        qi, ti, dist = slamhip.cross_check_arrays(q, t, ctx=gpu_ctx)
        assert qi.tolist() == keep and ti.tolist() == [ei[i] for i in keep]
        assert dist.tolist() == [float(ed[i]) for i in keep]
        ms = BruteForceFeatureMatcher(norm_type=6).cross_check_match(q, t)
        assert [(m.queryIdx, m.trainIdx, m.distance) for m in ms] == [(i, ei[i], float(ed[i])) for i in keep]
        oi, od = _cross_check_abi(gpu_ctx, q, t)
        assert oi.tolist() == ei and od.tolist() == ed


def _mutual_pairs(q, t):
    """(i, j) iff j is the first nearest train row of i and i is the first nearest query row of j (numpy, no oracle)."""
    dm = np.bitwise_count(q[:, None, :] ^ t[None, :, :]).sum(-1, dtype=np.int32)
    fwd, rev = dm.argmin(1), dm.argmin(0)
    ok = rev[fwd] == np.arange(len(q))
    return np.where(ok, fwd, -1).astype(np.int32), np.where(ok, dm[np.arange(len(q)), fwd], INT_MAX).astype(np.int32)


@pytest.mark.parametrize("seed", range(6))
def test_cross_check_is_exactly_the_mutual_nearest_pairs(gpu_ctx, seed):
    """Property: every returned pair is a mutual nearest neighbour and every mutual pair is returned — random rows,
    planted copies, and low-entropy rows where ties on both sides are the rule."""
    import slamhip
    from oracle import oracle

    rng = np.random.default_rng(1000 + seed)
    n, m = int(rng.integers(1, 600)), int(rng.integers(1, 700))
    if seed % 2:
        q = rng.choice(np.array([0x00, 0xFF, 0x0F], np.uint8), (n, 2)).repeat(16, axis=1)
        t = rng.choice(np.array([0x00, 0xFF, 0x0F], np.uint8), (m, 2)).repeat(16, axis=1)
    else:
        q = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        t = rng.integers(0, 256, (m, 32), dtype=np.uint8)
        k = min(n, m) // 3
        t[:k] = q[rng.permutation(n)[:k]]
    ei, ed = _mutual_pairs(q, t)
    oi, od = _cross_check_abi(gpu_ctx, q, t)
    assert np.array_equal(oi, ei) and np.array_equal(od, ed)
    ci, cd = oracle.bf_cross_check_c(q, t)
    assert np.array_equal(oi, ci) and np.array_equal(od, cd)
    qi, ti, dist = slamhip.cross_check_arrays(q, t, ctx=gpu_ctx)
    assert np.array_equal(qi, np.flatnonzero(ei >= 0)) and np.array_equal(ti, ei[ei >= 0])
    assert np.array_equal(dist, ed[ei >= 0].astype(np.float32))
    assert len(set(ti.tolist())) == len(ti)          # a train row is paired at most once
