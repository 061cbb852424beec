"""GPU: size-independent properties of the top-2 search at BASELINE.json's full sizes (65536 x 65536), where the CPU
oracle is too slow to check every row: invariances that any exact Hamming top-2 with OpenCV's (distance, index) order
must satisfy, each comparing two runs of the HIP path with each other, plus oracle checks on samples."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N = M = 65536


def _search(ctx, q, t):
    import slamhip

    dq, dt = slamhip.DeviceDescriptors(ctx, q), slamhip.DeviceDescriptors(ctx, t)
    tab = slamhip.Top2Table(ctx, q.shape[0])
    try:
        slamhip.knn2_device(ctx, dq.buf, q.shape[0], dt.buf, t.shape[0], tab.idx, tab.dist)
        return tab.download()
    finally:
        for o in (tab, dq, dt):
            o.free()


@pytest.fixture(scope="module")
def base(gpu_ctx):
    rng = np.random.default_rng(228)
    q = rng.integers(0, 256, (N, 32), dtype=np.uint8)
    t = np.random.default_rng(229).integers(0, 256, (M, 32), dtype=np.uint8)
    return q, t, _search(gpu_ctx, q, t)


def test_xor_mask_invariance(gpu_ctx, base):
    """popcount((q ^ m) ^ (t ^ m)) = popcount(q ^ t): flipping the same bits everywhere changes nothing at all."""
    q, t, (idx, dist) = base
    mask = np.random.default_rng(1).integers(0, 256, 32, dtype=np.uint8)
    i2, d2 = _search(gpu_ctx, q ^ mask, t ^ mask)
    assert np.array_equal(i2, idx) and np.array_equal(d2, dist)


def test_query_permutation_equivariance(gpu_ctx, base):
    """Queries are independent: permuting them permutes the rows of the tables and nothing else (different lanes,
    waves, blocks and leader / tail chunks see each query)."""
    q, t, (idx, dist) = base
    perm = np.random.default_rng(2).permutation(N)
    i2, d2 = _search(gpu_ctx, q[perm], t)
    assert np.array_equal(i2, idx[perm]) and np.array_equal(d2, dist[perm])


def test_train_permutation_and_tie_rule(gpu_ctx, base):
    """Permuting the train rows keeps every distance; indices map through the permutation wherever the order is not
    decided by a tie, and where it is, the lower (new) index comes first."""
    q, t, (idx, dist) = base
    perm = np.random.default_rng(3).permutation(M)             # new row j holds old row perm[j]
    inv = np.empty(M, np.int64)
    inv[perm] = np.arange(M)
    i2, d2 = _search(gpu_ctx, q, t[perm])
    assert np.array_equal(d2, dist)
    strict = dist[:, 0] < dist[:, 1]
    assert np.array_equal(i2[strict, 0], inv[idx[strict, 0]].astype(np.int32))
    tie = ~strict
    assert tie.any() and (i2[tie, 0] < i2[tie, 1]).all()
    # the reported rows really are at the reported distances
    for col in (0, 1):
        assert np.array_equal(np.bitwise_count(q ^ t[perm][i2[:, col]]).sum(1), d2[:, col])


def test_far_rows_change_nothing_and_near_rows_win(gpu_ctx, base):
    """Appending rows that are farther from every query than its 2nd neighbour leaves the tables alone; planting exact
    copies of some queries at the END of the train set puts them first at distance 0 (they beat earlier rows on
    distance, not on index)."""
    q, t, (idx, dist) = base
    # rows that agree with nobody on the last 64 bits: queries and train rows end in 8 zero bytes, the far rows in 8
    # 0xFF bytes, so a far row is at 64 + (a 192-bit random distance, ~96) >= 110 from every query, while the 2nd
    # neighbours among 65536 rows of 192 random bits sit below 80
    qz, tz = q.copy(), t.copy()
    qz[:, 24:] = 0
    tz[:, 24:] = 0
    iz, dz = _search(gpu_ctx, qz, tz)
    assert int(dz[:, 1].max()) < 100
    far = np.random.default_rng(7).integers(0, 256, (4096, 32), dtype=np.uint8)
    far[:, 24:] = 0xFF
    i2, d2 = _search(gpu_ctx, qz, np.concatenate([tz[:M // 2], far, tz[M // 2:]]))   # in the middle: indices behind them shift
    back = np.where(i2 >= M // 2, i2 - 4096, i2)
    assert (i2 < M // 2).__or__(i2 >= M // 2 + 4096).all()
    assert np.array_equal(back, iz) and np.array_equal(d2, dz)
    sel = np.random.default_rng(4).choice(N, 3000, replace=False)
    i3, d3 = _search(gpu_ctx, q, np.concatenate([t, q[sel]]))
    assert (d3[sel, 0] == 0).all() and np.array_equal(i3[sel, 0], (M + np.arange(3000)).astype(np.int32))
    assert (d3[sel, 1] <= dist[sel, 0]).all() and (i3[sel, 1] == idx[sel, 0]).mean() > 0.9   # old best is now 2nd (a copy of
    rest = np.setdiff1d(np.arange(N), sel)                                                  # another query is a random row)
    assert (d3[rest] <= dist[rest]).all() and (i3[rest] != idx[rest]).any(1).mean() < 0.2


def test_sampled_rows_against_the_oracle(gpu_ctx, base):
    from oracle import oracle

    q, t, (idx, dist) = base
    sel = np.random.default_rng(5).choice(N, 768, replace=False)
    ridx, rdist = oracle.bf_knn_c(q[sel], t, 2, threads=8)
    assert np.array_equal(idx[sel], ridx) and np.array_equal(dist[sel], rdist)


def test_strided_readonly_and_oddly_typed_inputs(gpu_ctx):
    """What callers may hand over besides fresh contiguous arrays: row-strided and column-sliced views, Fortran order,
    read-only arrays, lists of rows (Frame.get_descriptors stacks a Python list, primitives.py:200-205), and wrong
    dtypes / widths, which must be refused before the FFI call."""
    import slamhip
    from oracle import oracle

    rng = np.random.default_rng(11)
    big_q = rng.integers(0, 256, (600, 40), dtype=np.uint8)
    big_t = rng.integers(0, 256, (900, 32), dtype=np.uint8)
    q = big_q[::2, 4:36]                                       # every other row, columns 4..35: neither contiguous
    t = np.asfortranarray(big_t[::-1])                         # reversed rows, column-major
    t.setflags(write=False)
    ei, ed = oracle.bf_knn_c(np.ascontiguousarray(q), np.ascontiguousarray(t), 2, threads=4)
    gi, gd = slamhip.knn_match_arrays(q, t, 2, ctx=gpu_ctx)
    assert np.array_equal(gi, ei) and np.array_equal(gd, ed)
    mq, mt, md = slamhip.match_arrays([row for row in t], [row for row in q], 70.0, ctx=gpu_ctx)   # lists of rows
    eq, et, edd = oracle.bf_match_c(np.ascontiguousarray(t), np.ascontiguousarray(q), 70.0)
    assert np.array_equal(mq, eq) and np.array_equal(mt, et) and np.array_equal(md, edd)
    for bad in (q.astype(np.int32), q.astype(np.float32), np.zeros((5, 31), np.uint8), np.zeros((5, 32, 1), np.uint8)):
        with pytest.raises(ValueError):
            slamhip.knn_match_arrays(bad, t, 2, ctx=gpu_ctx)
        with pytest.raises(ValueError):
            slamhip.match_arrays(t, bad, ctx=gpu_ctx)


def test_ratio_filter_and_cross_check_properties_4k(gpu_ctx):
    """BASELINE configs[1] (4096 x 4096, knn=2 + ratio 0.75): the device-side selections against the oracle on every
    row, and crossCheck's symmetry: swapping the roles of query and train returns the same pairs, transposed."""
    import slamhip
    from oracle import oracle

    rng = np.random.default_rng(6)
    t = rng.integers(0, 256, (4096, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (4096, 32), dtype=np.uint8)
    near = rng.choice(4096, 800, replace=False)
    q[near] = t[rng.permutation(4096)[:800]]
    q[near, 5] ^= 0x21                                          # planted near-duplicates: the ratio test keeps them
    ei, ed = oracle.bf_knn_c(q, t, 2, threads=8)
    keep = oracle.bf_ratio_c(ei, ed, 0.75)
    rq, rt, rd = slamhip.ratio_test_arrays(q, t, 0.75, ctx=gpu_ctx)
    assert keep.sum() >= 800 and np.array_equal(rq, np.flatnonzero(keep)) and np.array_equal(rt, ei[keep, 0])
    assert np.array_equal(rd, ed[keep, 0].astype(np.float32))
    for thr in (None, 20.0, 64.0, 300.0):
        got = slamhip.match_arrays(t, q, thr, ctx=gpu_ctx)
        exp = oracle.bf_match_c(t, q, thr, threads=8)
        assert all(np.array_equal(a, b) for a, b in zip(got, exp)), thr
    cq, ct, cd = slamhip.cross_check_arrays(q, t, ctx=gpu_ctx)
    sq, st, sd = slamhip.cross_check_arrays(t, q, ctx=gpu_ctx)   # roles swapped
    a = sorted(zip(cq.tolist(), ct.tolist(), cd.tolist()))
    b = sorted(zip(st.tolist(), sq.tolist(), sd.tolist()))
    assert a == b and len(a) >= 800
    oi, od = oracle.bf_cross_check_c(q, t, threads=8)
    assert np.array_equal(cq, np.flatnonzero(oi >= 0)) and np.array_equal(ct, oi[oi >= 0])
