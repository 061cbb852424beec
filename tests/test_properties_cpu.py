"""CPU suite: property tests of the oracle (hypothesis) — the invariants the GPU tests later rely on."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import oracle


def descs(n_max):
    return st.integers(0, n_max).flatmap(
        lambda n: st.binary(min_size=32 * n, max_size=32 * n).map(lambda b: np.frombuffer(b, np.uint8).reshape(n, 32)))


@settings(max_examples=60, deadline=None, derandomize=True)
@given(descs(40), descs(60))
def test_knn_is_sorted_complete_and_matches_numpy(q, t):
    idx, dist = oracle.bf_knn_c(q, t, 2)
    nidx, ndist = oracle.bf_knn_np(q, t, 2)
    assert np.array_equal(idx, nidx) and np.array_equal(dist, ndist)
    have = idx >= 0
    assert (have.sum(1) == min(2, len(t))).all()
    if len(t) >= 2 and len(q):
        assert (dist[:, 0] <= dist[:, 1]).all()
        tie = dist[:, 0] == dist[:, 1]
        assert (idx[tie, 0] < idx[tie, 1]).all()
        d = oracle.hamming_matrix_np(q, t)
        assert np.array_equal(dist[:, 0], d.min(1))


@settings(max_examples=40, deadline=None, derandomize=True)
@given(descs(30), descs(50), st.integers(1, 7))
def test_train_sharding_then_merge_equals_monolithic(q, t, parts):
    """Top-2 over train shards merged by (dist, global idx) == top-2 over the whole set (what slam_bf_merge_top2 does)."""
    idx, dist = oracle.bf_knn_c(q, t, 2)
    bounds = np.linspace(0, len(t), parts + 1).astype(int)
    cand = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        i, d = oracle.bf_knn_c(q, t[a:b], 2)
        cand.append((np.where(i >= 0, i + a, -1), d))
    ci = np.concatenate([c[0] for c in cand], 1)
    cd = np.concatenate([c[1] for c in cand], 1)
    key = np.where(ci >= 0, cd.astype(np.int64) << 32 | ci.astype(np.int64), np.iinfo(np.int64).max)
    order = np.argsort(key, axis=1, kind="stable")[:, :2]
    mi = np.take_along_axis(ci, order, 1)
    md = np.take_along_axis(cd, order, 1)
    assert np.array_equal(mi, idx) and np.array_equal(md, dist)


@settings(max_examples=40, deadline=None, derandomize=True)
@given(descs(30), descs(40), st.one_of(st.none(), st.floats(0, 300)))
def test_match_filter_is_a_subset_with_strict_limit(src, qry, thr):
    q0, t0, d0 = oracle.bf_match_c(src, qry, None)
    q1, t1, d1 = oracle.bf_match_c(src, qry, thr)
    assert set(q1.tolist()) <= set(q0.tolist())
    if thr and len(d0):
        lim = max(2 * float(d0.min()), thr)
        # compare as Python does: DMatch.distance is a Python float (f64), so a denormal threshold still counts
        assert (d1.astype(np.float64) < lim).all() and len(d1) == int((d0.astype(np.float64) < lim).sum())
    else:
        assert np.array_equal(q0, q1)


def low_entropy_descs(n_max):
    """Rows drawn from a handful of byte patterns: equal distances (ties) on both sides are the rule."""
    return st.integers(0, n_max).flatmap(
        lambda n: st.lists(st.lists(st.sampled_from([0x00, 0xFF, 0x0F]), min_size=2, max_size=2), min_size=n, max_size=n)
        .map(lambda rows: np.array([r * 16 for r in rows], np.uint8).reshape(n, 32)))


def mutual_pairs(q, t):
    """The contract of BFMatcher(crossCheck=True), spelled out with explicit loops: (i, j) iff j is the first nearest
    train row of i and i is the first nearest query row of j."""
    d = oracle.hamming_matrix_np(q, t)
    pairs = {}
    for i in range(len(q)):
        j = min(range(len(t)), key=lambda c: (d[i, c], c))
        back = min(range(len(q)), key=lambda r: (d[r, j], r))
        if back == i:
            pairs[i] = (j, int(d[i, j]))
    return pairs


@settings(max_examples=80, deadline=None, derandomize=True)
@given(st.one_of(descs(25), low_entropy_descs(25)), st.one_of(descs(30), low_entropy_descs(30)))
def test_cross_check_returns_exactly_the_mutual_nearest_pairs(q, t):
    for cc in (oracle.bf_cross_check_c, oracle.bf_cross_check_np):
        oi, od = cc(q, t)
        got = {i: (int(oi[i]), int(od[i])) for i in range(len(q)) if oi[i] >= 0}
        exp = mutual_pairs(q, t) if len(q) and len(t) else {}
        assert got == exp
        assert (od[oi < 0] == 2**31 - 1).all()
        assert len(set(v[0] for v in got.values())) == len(got)      # a train row is paired at most once
