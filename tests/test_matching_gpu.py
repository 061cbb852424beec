"""GPU parity: HIP top-2 / filters through the C ABI vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rand(n, seed):
    return np.random.default_rng(seed).integers(0, 256, (n, 32), dtype=np.uint8)


@pytest.mark.parametrize("n,m", [(1, 1), (1, 2), (2, 1), (63, 65), (64, 64), (65, 63), (200, 200), (257, 511),
                                 (1000, 3), (3, 1000), (513, 2049), (4096, 4096)])
def test_knn2_matches_oracle(gpu_ctx, n, m):
    import slamhip
    from oracle import oracle

    q, t = _rand(n, 228), _rand(m, 229)
    idx, dist = slamhip.knn_match_arrays(q, t, 2)
    ridx, rdist = oracle.bf_knn_c(q, t, 2, threads=8)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(dist, rdist)


@pytest.mark.parametrize("R", [1, 2, 4, 8])
@pytest.mark.parametrize("bpc", [1, 4, 16])
def test_every_kernel_variant(gpu_ctx, R, bpc):
    """Force each queries-per-lane variant and grid split; all must agree with the oracle."""
    import slamhip
    from oracle import oracle

    lib = slamhip.load()
    q, t = _rand(1500, 1), _rand(2300, 2)
    t[100:140] = t[7]          # 41-way tie on one row
    q[3] = t[7]
    try:
        gpu_ctx.set_tuning(R=R, blocks_per_cu=bpc)
        idx, dist = slamhip.knn_match_arrays(q, t, 2)
    finally:
        gpu_ctx.set_tuning()
    ridx, rdist = oracle.bf_knn_c(q, t, 2, threads=8)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    assert idx[3].tolist() == [7, 100] and dist[3].tolist() == [0, 0]


def test_all_ties_and_extremes(gpu_ctx):
    import slamhip
    from oracle import oracle

    q = np.zeros((130, 32), np.uint8)
    t = np.zeros((700, 32), np.uint8)             # every distance 0: indices must be 0 and 1
    idx, dist = slamhip.knn_match_arrays(q, t, 2)
    assert (idx == [0, 1]).all() and (dist == 0).all()
    t[:] = 0xFF                                    # every distance 256
    idx, dist = slamhip.knn_match_arrays(q, t, 2)
    assert (idx == [0, 1]).all() and (dist == 256).all()
    # descending distances: every new row improves -> update path taken on every step
    t = np.zeros((257, 32), np.uint8)
    bits = np.unpackbits(t, axis=1)
    for i in range(257):
        bits[i, : 256 - i] = 1
    t = np.packbits(bits, axis=1)
    idx, dist = slamhip.knn_match_arrays(q, t, 2)
    ridx, rdist = oracle.bf_knn_c(q, t, 2)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    assert idx[0].tolist() == [256, 255] and dist[0].tolist() == [0, 1]


def test_empty_inputs(gpu_ctx):
    import slamhip

    q = _rand(5, 1)
    idx, dist = slamhip.knn_match_arrays(q, np.zeros((0, 32), np.uint8), 2)
    assert (idx == -1).all() and (dist == 2**31 - 1).all()
    idx, dist = slamhip.knn_match_arrays(np.zeros((0, 32), np.uint8), q, 2)
    assert idx.shape == (0, 2)
    # Frame.get_descriptors() of an empty frame: float64 (0,) (primitives.py:200-205)
    assert slamhip.match_arrays(np.array([]), q)[0].shape == (0,)
    assert slamhip.match_arrays(q, np.array([]))[0].shape == (0,)


def test_m1_gives_single_neighbour(gpu_ctx):
    import slamhip

    idx, dist = slamhip.knn_match_arrays(_rand(10, 3), _rand(1, 4), 2)
    assert (idx[:, 0] == 0).all() and (idx[:, 1] == -1).all() and (dist[:, 1] == 2**31 - 1).all()


@pytest.mark.parametrize("thr", [None, 0.0, 30.0, 90.0, 120.0, 400.0])
def test_match_dropin_vs_oracle(gpu_ctx, thr):
    """BruteForceFeatureMatcher.match(source, query, dist_threshold) == feature_matchers.py:36-44."""
    from feature_matchers import BruteForceFeatureMatcher
    from oracle import oracle

    src, qry = _rand(300, 10), _rand(210, 11)
    qry[5] = src[17]
    qry[6] = src[17] ^ np.uint8(1)
    ms = BruteForceFeatureMatcher(norm_type=6).match(src, qry, thr)
    oq, ot, od = oracle.bf_match_c(src, qry, thr)
    assert [m.queryIdx for m in ms] == oq.tolist()
    assert [m.trainIdx for m in ms] == ot.tolist()
    assert [m.distance for m in ms] == od.tolist()
    assert all(m.imgIdx == 0 for m in ms)


def test_ratio_and_cross_check(gpu_ctx):
    import slamhip
    from oracle import oracle

    t = _rand(900, 20)
    q = _rand(800, 21)
    # planted near-duplicates so the ratio test keeps something
    rng = np.random.default_rng(5)
    for i in range(0, 800, 7):
        row = t[rng.integers(0, 900)].copy()
        row[rng.integers(0, 32)] ^= np.uint8(1 << rng.integers(0, 8))
        q[i] = row
    qi, ti, d = slamhip.ratio_test_arrays(q, t, 0.75)
    ridx, rdist = oracle.bf_knn_c(q, t, 2)
    keep = oracle.bf_ratio_c(ridx, rdist, 0.75)
    assert keep.sum() > 50
    assert np.array_equal(qi, np.nonzero(keep)[0]) and np.array_equal(ti, ridx[keep, 0])
    assert np.array_equal(d, rdist[keep, 0].astype(np.float32))
    qi, ti, d = slamhip.cross_check_arrays(q, t)
    oi, od = oracle.bf_cross_check_c(q, t)
    k = oi >= 0
    assert np.array_equal(qi, np.nonzero(k)[0]) and np.array_equal(ti, oi[k]) and np.array_equal(d, od[k].astype(np.float32))


def test_collection_matches_opencv_multi_image_order(gpu_ctx):
    import slamhip
    from oracle import oracle

    imgs = [_rand(r, 100 + i) for i, r in enumerate([300, 1, 0, 257, 512])]
    q = _rand(333, 99)
    imgs[3][4] = imgs[0][9]     # same row in two images: lower imgIdx wins
    q[0] = imgs[0][9]
    img, tr, dist = slamhip.knn_match_collection(q, imgs, 2)
    rimg, rtr, rdist = oracle.bf_knn_multi_c(q, imgs, 2)
    assert np.array_equal(img, rimg) and np.array_equal(tr, rtr) and np.array_equal(dist, rdist)
    assert img[0].tolist() == [0, 3] and tr[0].tolist() == [9, 4]


def test_full_size_properties_64k(gpu_ctx):
    """BASELINE size 65536 x 65536: size-independent properties instead of the (slow) oracle."""
    import slamhip

    ctx = gpu_ctx
    q, t = _rand(65536, 228), _rand(65536, 229)
    perm = np.random.default_rng(7).permutation(65536)
    q[:4096] = t[perm[:4096]]                       # planted exact copies: 1-NN must be (copy, 0)
    dq, dt = slamhip.DeviceDescriptors(ctx, q), slamhip.DeviceDescriptors(ctx, t)
    tab = slamhip.Top2Table(ctx, 65536)
    slamhip.knn2_device(ctx, dq.buf, 65536, dt.buf, 65536, tab.idx, tab.dist)
    idx, dist = tab.download()
    assert (dist[:4096, 0] == 0).all() and np.array_equal(idx[:4096, 0], perm[:4096].astype(np.int32))
    assert (dist[:, 0] <= dist[:, 1]).all()
    tie = dist[:, 0] == dist[:, 1]
    assert (idx[tie, 0] < idx[tie, 1]).all()        # (distance, index) order
    # reported distances are the true distances of the reported rows
    for col in (0, 1):
        d = np.bitwise_count(q ^ t[idx[:, col]]).sum(1)
        assert np.array_equal(d, dist[:, col])
    # a random slice of queries checked exhaustively against the oracle
    from oracle import oracle
    sel = np.random.default_rng(8).choice(65536, 512, replace=False)
    ridx, rdist = oracle.bf_knn_c(q[sel], t, 2, threads=8)
    assert np.array_equal(idx[sel], ridx) and np.array_equal(dist[sel], rdist)
    # sharded (query halves, train halves + merge) == monolithic
    lib = ctx.lib
    parts_i, parts_d = ctx.malloc(2 * 65536 * 8), ctx.malloc(2 * 65536 * 8)
    for g in range(2):
        slamhip.knn2_device(ctx, dq.buf, 65536, dt.rows_view(g * 32768, (g + 1) * 32768), 32768,
                            parts_i.view(g * 65536 * 8), parts_d.view(g * 65536 * 8), train_base=g * 32768)
    out = slamhip.Top2Table(ctx, 65536)
    assert lib.slam_bf_merge_top2(ctx.handle, parts_i.ptr, parts_d.ptr, 2, 65536, out.idx.ptr, out.dist.ptr) == 0
    midx, mdist = out.download()
    assert np.array_equal(midx, idx) and np.array_equal(mdist, dist)
    for b in (parts_i, parts_d):
        b.free()
    for o in (out, tab, dq, dt):
        o.free()


def test_loop_closure_all_to_all_1m(gpu_ctx):
    """BASELINE configs[3]: 512 keyframes x 2048 descriptors matched all-to-all (1,048,576 x 1,048,576 pairs grid)."""
    import slamhip
    from oracle import oracle

    kf, per = 512, 2048
    rng = np.random.default_rng(228)
    allrows = rng.integers(0, 256, (kf * per, 32), dtype=np.uint8)
    images = [allrows[i * per:(i + 1) * per] for i in range(kf)]
    img, tr, dist = slamhip.knn_match_collection(allrows, images, 2)
    n = kf * per
    # every descriptor finds itself first (distance 0, its own keyframe / row), then its true nearest other row
    assert (dist[:, 0] == 0).all()
    assert np.array_equal(img[:, 0], np.repeat(np.arange(kf), per)) and np.array_equal(tr[:, 0], np.tile(np.arange(per), kf))
    assert (dist[:, 1] > 0).all() and (dist[:, 1] < 110).all()
    sel = rng.choice(n, 96, replace=False)
    rimg, rtr, rdist = oracle.bf_knn_multi_c(allrows[sel], images, 2, threads=8)
    assert np.array_equal(img[sel], rimg) and np.array_equal(tr[sel], rtr) and np.array_equal(dist[sel], rdist)
    # query-sharded (8 shards of 64 keyframes, what 8 GPUs would each compute) == monolithic
    ctx = gpu_ctx
    dall = slamhip.DeviceDescriptors(ctx, allrows)
    tab = slamhip.Top2Table(ctx, n // 8)
    for g in (0, 5):
        a, b = g * (n // 8), (g + 1) * (n // 8)
        slamhip.knn2_device(ctx, dall.rows_view(a, b), b - a, dall.buf, n, tab.idx, tab.dist)
        sidx, sdist = tab.download()
        simg, str_ = slamhip.split_image_index(sidx, [per] * kf)
        assert np.array_equal(simg, img[a:b]) and np.array_equal(str_, tr[a:b]) and np.array_equal(sdist, dist[a:b])
    tab.free()
    dall.free()


def test_split_index_on_the_device_equals_the_host_statement(gpu_ctx):
    """slam_bf_split_index: global train row -> (imgIdx, trainIdx) for ragged collections with empty images, "no
    neighbour" entries and indices past the end; equal to slamhip.split_image_index and to OpenCV's imgIdx << 18 form."""
    import slamhip

    ctx, lib = gpu_ctx, gpu_ctx.lib
    rng = np.random.default_rng(11)
    for rows in ([2048] * 512, [5, 0, 0, 17, 1, 0, 300, 2], [1], [0, 0, 9]):
        off = np.concatenate([[0], np.cumsum(rows)]).astype(np.int32)
        total = int(off[-1])
        g = rng.integers(0, total, 5000).astype(np.int32)
        g[::97] = -1
        g[:len(off) - 1] = off[:-1].clip(0, total - 1)          # first rows of the images (an empty image shares its row with the next)
        g[-1] = total - 1
        d_g, d_off = ctx.upload(g), ctx.upload(off)
        d_img, d_loc = ctx.malloc(g.size * 4), ctx.malloc(g.size * 4)
        assert lib.slam_bf_split_index(ctx.handle, d_g.ptr, g.size, d_off.ptr, len(rows), d_img.ptr, d_loc.ptr) == 0
        img, loc = d_img.download(np.int32, g.shape), d_loc.download(np.int32, g.shape)
        himg, hloc = slamhip.split_image_index(g, rows)
        assert np.array_equal(img, himg) and np.array_equal(loc, hloc), rows
        ok = g >= 0
        assert (off[img[ok]] + loc[ok] == g[ok]).all() and (loc[ok] < np.asarray(rows)[img[ok]]).all()
        assert (img[~ok] == -1).all() and (loc[~ok] == -1).all()
        for b in (d_g, d_off, d_img, d_loc):
            b.free()
    assert lib.slam_bf_split_index(ctx.handle, None, 5, None, 1, None, None) == -1
    assert lib.slam_bf_split_index(ctx.handle, None, 0, None, 1, None, None) == 0
    assert lib.slam_bf_split_index(ctx.handle, None, 5, None, 9000, None, None) == -1     # more images than imgIdx << 18 allows


def test_resident_matcher_matches_frame_to_frame(gpu_ctx):
    """Descriptors stay on the device between frames; results equal the stateless path (and the oracle)."""
    import slamhip
    from oracle import oracle

    rm = slamhip.ResidentMatcher(gpu_ctx)
    frames = [_rand(n, 300 + i) for i, n in enumerate([200, 180, 0, 150, 200])]   # slam.py:23: 200 features per frame
    assert rm.push(frames[0]) is None
    for prev, cur in zip(frames[:-1], frames[1:]):
        got = rm.push(cur, 70.0)
        exp = oracle.bf_match_c(prev, cur, 70.0)
        assert all(np.array_equal(a, b) for a, b in zip(got, exp))
    rm.reset()


def test_random_shapes_fuzz(gpu_ctx):
    """Many odd (N, M) shapes incl. tile / wave / group boundaries, low-entropy rows (heavy ties), one context."""
    import slamhip
    from oracle import oracle

    rng = np.random.default_rng(2026)
    shapes = [(int(rng.integers(1, 700)), int(rng.integers(1, 1300))) for _ in range(25)]
    shapes += [(255, 257), (256, 256), (257, 255), (64, 16), (64, 17), (1, 4097), (4097, 1), (300, 768), (513, 15)]
    for n, m in shapes:
        if rng.uniform() < 0.3:      # few distinct values per word -> many equal distances
            q = rng.integers(0, 2, (n, 32), dtype=np.uint8) * 255
            t = rng.integers(0, 2, (m, 32), dtype=np.uint8) * 255
        else:
            q = rng.integers(0, 256, (n, 32), dtype=np.uint8)
            t = rng.integers(0, 256, (m, 32), dtype=np.uint8)
        idx, dist = slamhip.knn_match_arrays(q, t, 2)
        ridx, rdist = oracle.bf_knn_c(q, t, 2, threads=8)
        assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist), (n, m)


def test_repeated_calls_keep_the_merge_state_clean(gpu_ctx):
    """The kernel restores best/bound/arrivals itself: alternate sizes and tunings on one context, results must not drift."""
    import slamhip
    from oracle import oracle

    lib = slamhip.load()
    q, t = _rand(3000, 41), _rand(5000, 42)
    ridx, rdist = oracle.bf_knn_c(q, t, 2, threads=8)
    try:
        for rep in range(3):
            for R, bpc in ((1, 0), (2, 8), (1, 64), (4, 2), (8, 1)):
                gpu_ctx.set_tuning(R=R, blocks_per_cu=bpc)
                idx, dist = slamhip.knn_match_arrays(q, t, 2)
                assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist), (rep, R, bpc)
                small = slamhip.knn_match_arrays(q[:100], t[:77], 2)
                assert np.array_equal(small[0], oracle.bf_knn_c(q[:100], t[:77], 2)[0])
    finally:
        gpu_ctx.set_tuning()


def test_train_set_larger_than_one_key_range(gpu_ctx):
    """M > 2^23 rows: the library loops passes over the train set and merges them by (dist, idx)."""
    import slamhip
    from oracle import oracle

    m = (1 << 23) + 777
    rng = np.random.default_rng(11)
    t = rng.integers(0, 256, (m, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (130, 32), dtype=np.uint8)
    q[0] = t[m - 1]                  # exact copy in the second pass
    q[1] = t[5]
    t[(1 << 23) + 9] = t[5]          # duplicate across the pass boundary: the lower index must come first
    idx, dist = slamhip.knn_match_arrays(q, t, 2)
    ridx, rdist = oracle.bf_knn_c(q, t, 2, threads=16)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    assert idx[0, 0] == m - 1 and dist[0, 0] == 0
    assert idx[1].tolist() == [5, (1 << 23) + 9] and dist[1].tolist() == [0, 0]


@pytest.mark.parametrize("n,m", [(4096, 4096), (4097, 100), (100, 4097), (16384, 300), (16385, 300), (1, 4096), (255, 33)])
def test_host_call_path_boundaries(gpu_ctx, n, m):
    """match_arrays / ratio_test_arrays / knn_match_arrays across the internal switch points of the host-buffer
    calls (zero-copy up to 4096 rows a side, one-workgroup filter up to 16384 queries, sub-tile train chunks)."""
    import slamhip
    from oracle import oracle

    rng = np.random.default_rng(n * 31 + m)
    t = rng.integers(0, 256, (m, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    dup = rng.choice(n, max(1, n // 50), replace=False)           # planted near-duplicates so filters keep something
    q[dup] = t[rng.integers(0, m, len(dup))]
    q[dup, 0] ^= 0x11
    ei, ed = oracle.bf_knn_c(q, t, 2, threads=8)
    gi, gd = slamhip.knn_match_arrays(q, t, 2, ctx=gpu_ctx)
    assert np.array_equal(gi, ei) and np.array_equal(gd, ed)
    for thr in (None, 30.0, 400.0):
        got = slamhip.match_arrays(t, q, thr, ctx=gpu_ctx)
        exp = oracle.bf_match_c(t, q, thr, threads=8)
        assert all(np.array_equal(a, b) for a, b in zip(got, exp)), thr
    keep = oracle.bf_ratio_c(ei, ed, 0.75)
    rq, rt, rd = slamhip.ratio_test_arrays(q, t, 0.75, ctx=gpu_ctx)
    assert np.array_equal(rq, np.flatnonzero(keep)) and np.array_equal(rt, ei[keep, 0])
    assert np.array_equal(rd, ed[keep, 0].astype(np.float32))


def test_against_cv2_when_available(gpu_ctx):
    """Confirmation against the real reference arithmetic (last row): runs only on a host where
    opencv-python happens to be installed (it is absent from the build container and from the GPU image used in
    round 1, where this test is skipped and parity stays 'unpinned')."""
    cv2 = pytest.importorskip("cv2")
    import slamhip

    rng = np.random.default_rng(228)
    t = rng.integers(0, 256, (700, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (500, 32), dtype=np.uint8)
    q[:100] = t[:100]
    q[:100, 3] ^= 0x05
    t[650:] = t[600:650]                                          # duplicate train rows: tie order matters
    knn = cv2.BFMatcher(cv2.NORM_HAMMING).knnMatch(q, t, k=2)
    gi, gd = slamhip.knn_match_arrays(q, t, 2, ctx=gpu_ctx)
    assert [[m.trainIdx for m in row] for row in knn] == gi.tolist()
    assert [[int(m.distance) for m in row] for row in knn] == gd.tolist()
    one = cv2.BFMatcher(cv2.NORM_HAMMING).match(q, t)
    mq, mt, md = slamhip.match_arrays(t, q, None, ctx=gpu_ctx)
    assert [m.queryIdx for m in one] == mq.tolist() and [m.trainIdx for m in one] == mt.tolist()
    assert [m.distance for m in one] == md.tolist()
    cross = cv2.BFMatcher(cv2.NORM_HAMMING, crossCheck=True).match(q, t)
    cq, ct, cd = slamhip.cross_check_arrays(q, t, ctx=gpu_ctx)
    assert [(m.queryIdx, m.trainIdx, m.distance) for m in cross] == list(zip(cq.tolist(), ct.tolist(), cd.tolist()))
    bf = cv2.BFMatcher(cv2.NORM_HAMMING)
    imgs = [t[:300], t[300:450], t[450:]]
    bf.add(imgs)
    multi = bf.knnMatch(q, k=2)
    img, loc, dist = slamhip.knn_match_collection(q, imgs, 2, ctx=gpu_ctx)
    assert [[(m.imgIdx, m.trainIdx, int(m.distance)) for m in row] for row in multi] == \
        [[(int(img[i, k]), int(loc[i, k]), int(dist[i, k])) for k in range(2)] for i in range(len(q))]


def test_keyframe_database_matches_collection_semantics(gpu_ctx):
    """KeyframeDatabase (descriptors of all keyframes resident in HBM, growing) == knnMatch against the same
    images as a collection (oracle: OpenCV's imgIdx<<18|trainIdx order), across a capacity growth and with an
    empty keyframe in between."""
    import slamhip
    from oracle import oracle

    rng = np.random.default_rng(314)
    db = slamhip.KeyframeDatabase(gpu_ctx, capacity_rows=256)
    imgs = []
    for rows in (200, 0, 300, 1000, 37):
        kf = rng.integers(0, 256, (rows, 32), dtype=np.uint8)
        if rows and imgs and imgs[0].shape[0]:
            kf[: min(rows, 20)] = imgs[0][: min(rows, 20)]        # repeated places: ties across keyframes
        assert db.add(kf) == len(imgs)
        imgs.append(kf)
        q = rng.integers(0, 256, (150, 32), dtype=np.uint8)
        q[:20] = imgs[0][:20]
        img, loc, dist = db.query(q, 2)
        ei, el, ed = oracle.bf_knn_multi_c(q, imgs, 2)
        assert np.array_equal(img, ei) and np.array_equal(loc, el) and np.array_equal(dist, ed)
    assert db.total == 1537 and db.rows == [200, 0, 300, 1000, 37]
    i1, l1, d1 = db.query(q, 1)
    assert np.array_equal(i1, ei[:, :1]) and np.array_equal(l1, el[:, :1]) and np.array_equal(d1, ed[:, :1])
    e = db.query(np.zeros((0, 32), np.uint8), 2)
    assert all(a.shape == (0, 2) for a in e)
    db.free()


def test_batched_searches_equal_single_launches_and_the_oracle(gpu_ctx):
    """slam_bf_knn2_batch_u256: B independent (query, train) searches in ONE launch - ragged sizes from one row to
    several tiles, an empty query side, an empty train side, ties across chunk boundaries, a train_base - each table
    bit-identical to its own slam_bf_knn2_u256 launch and to the oracle."""
    import slamhip
    from oracle import oracle

    ctx = gpu_ctx
    rng = np.random.default_rng(31)
    shapes = [(200, 200), (1, 1), (4096, 4096), (777, 3), (5, 9000), (0, 50), (60, 0), (300, 1500), (2500, 2500), (64, 16500)]
    pairs = []
    for n, m in shapes:
        q = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        t = rng.integers(0, 256, (m, 32), dtype=np.uint8)
        if n > 4 and m > 600:
            t[m - 1] = t[3]
            t[m // 2] = t[3]
            q[2] = t[3]                       # a three-way tie at distance 0 spread over the chunks
        pairs.append((q, t))
    got = slamhip.knn_match_arrays_batch(pairs, ctx=ctx)
    assert len(got) == len(shapes)
    for (q, t), (idx, dist), (n, m) in zip(pairs, got, shapes):
        assert idx.shape == (n, 2) and dist.shape == (n, 2)
        if n == 0:
            continue
        sidx, sdist = slamhip.knn_match_arrays(q, t, 2, ctx=ctx)
        assert np.array_equal(idx, sidx) and np.array_equal(dist, sdist), (n, m)
        ridx, rdist = oracle.bf_knn_c(q, t, 2, threads=8)
        assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist), (n, m)
    # device-pointer form with a train_base, twice in a row (the merge state must come back clean), next to single launches
    q, t = pairs[2]
    dq, dt = slamhip.DeviceDescriptors(ctx, q), slamhip.DeviceDescriptors(ctx, t)
    tabs = [slamhip.Top2Table(ctx, 4096) for _ in range(3)]
    for _ in range(2):
        slamhip.knn2_device_batch(ctx, [(dq.buf, 4096, dt.buf, 4096, tabs[0].idx, tabs[0].dist, 1000),
                                        (dt.buf, 4096, dq.buf, 4096, tabs[1].idx, tabs[1].dist)])
        slamhip.knn2_device(ctx, dq.buf, 4096, dt.buf, 4096, tabs[2].idx, tabs[2].dist, 1000)
        a, b = tabs[0].download(), tabs[2].download()
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[0].min() >= 1000
        ridx, rdist = oracle.bf_knn_c(t, q, 2, threads=8)
        r = tabs[1].download()
        assert np.array_equal(r[0], ridx) and np.array_equal(r[1], rdist)
    for o in (*tabs, dq, dt):
        o.free()
    # limits and misuse are reported, not launched
    import ctypes
    from slamhip._lib import BfSearch
    lib = ctx.lib
    assert lib.slam_bf_knn2_batch_u256(ctx.handle, 0, None) == 0
    assert lib.slam_bf_knn2_batch_u256(ctx.handle, 33, (BfSearch * 33)()) == -1
    assert lib.slam_bf_knn2_batch_u256(ctx.handle, 1, None) == -1
    bad = (BfSearch * 1)(BfSearch(None, 5, None, 5, 0, None, None))
    assert lib.slam_bf_knn2_batch_u256(ctx.handle, 1, bad) == -1
    try:
        ctx.set_tuning(R=2)
        with pytest.raises(slamhip.SlamHipError):
            slamhip.knn_match_arrays_batch(pairs[:1], ctx=ctx)
    finally:
        ctx.set_tuning()
    # sixteen 4096 x 4096 searches (BASELINE configs[1]) in one launch, all against the oracle on sampled rows
    many = [(rng.integers(0, 256, (4096, 32), dtype=np.uint8), rng.integers(0, 256, (4096, 32), dtype=np.uint8)) for _ in range(16)]
    out = slamhip.knn_match_arrays_batch(many, ctx=ctx)
    for (q, t), (idx, dist) in zip(many, out):
        sel = rng.choice(4096, 64, replace=False)
        ridx, rdist = oracle.bf_knn_c(q[sel], t, 2, threads=8)
        assert np.array_equal(idx[sel], ridx) and np.array_equal(dist[sel], rdist)


@pytest.mark.parametrize("n,m", [(1, 1), (70, 1), (300, 2), (4096, 4096), (1000, 20000), (777, 70000), (8192, 65536)])
def test_selection_fused_into_the_search_equals_the_two_step_way(gpu_ctx, n, m):
    """slam_bf_knn2_select_u256: the search's own decode flags the queries (mode 0: has a neighbour; mode 2: Lowe ratio, BASELINE
    configs[1]) - same tables, same flags and the same count as search + slam_bf_match_filter and as the oracle, under every
    kind of plan (one-round grids, one block per chunk, queue plans with both exchanges, several queries per lane)."""
    import ctypes

    import slamhip
    from oracle import oracle

    ctx = gpu_ctx
    rng = np.random.default_rng(n + m)
    q = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (m, 32), dtype=np.uint8)
    k = min(n, m, 50)
    t[rng.choice(m, k, replace=False)] = q[rng.choice(n, k, replace=False)]        # true matches: the ratio test keeps something
    ridx, rdist = oracle.bf_knn_c(q, t, 2, threads=8)
    dq, dt = slamhip.DeviceDescriptors(ctx, q), slamhip.DeviceDescriptors(ctx, t)
    tab, keep, keep2 = slamhip.Top2Table(ctx, n), ctx.malloc(n + 64), ctx.malloc(n + 64)
    try:
        for knobs in (dict(), dict(queue=-1), dict(queue=1, merge=1), dict(queue=1, merge=-1), dict(R=2), dict(R=8), dict(feed=-1)):
            ctx.set_tuning(**knobs)
            for mode, param in ((0, 0.0), (2, 0.75), (2, 0.9), (2, 1.0), (2, 0.0)):
                got = slamhip.knn2_select_device(ctx, dq.buf, n, dt.buf, m, tab.idx, tab.dist, keep, mode, param)
                idx, dist = tab.download()
                assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist), (knobs, mode)
                flags = keep.download(np.uint8, (n,))
                want = oracle.bf_ratio_c(ridx, rdist, param).astype(np.uint8) if mode == 2 else (ridx[:, 0] >= 0).astype(np.uint8)
                assert np.array_equal(flags, want) and got == int(want.sum()), (knobs, mode, param, got, int(want.sum()))
                cnt, mind = ctypes.c_int64(0), ctypes.c_int32(0)          # the two-step way on the same tables
                assert ctx.lib.slam_bf_match_filter(ctx.handle, tab.idx.ptr, tab.dist.ptr, n, mode, param, keep2.ptr, ctypes.byref(cnt),
                                                    ctypes.byref(mind)) == 0
                assert cnt.value == got and np.array_equal(keep2.download(np.uint8, (n,)), flags)
            assert ctx.state_dirty() == 0
    finally:
        ctx.set_tuning()
        for o in (tab, dq, dt, keep, keep2):
            o.free()
    # no train rows: nothing is kept; a bad mode is refused
    e = slamhip.DeviceDescriptors(ctx, np.zeros((0, 32), np.uint8))
    dq = slamhip.DeviceDescriptors(ctx, q)
    tab, keep = slamhip.Top2Table(ctx, n), ctx.malloc(n + 64)
    try:
        assert slamhip.knn2_select_device(ctx, dq.buf, n, e.buf, 0, tab.idx, tab.dist, keep, 2, 0.75) == 0
        assert not keep.download(np.uint8, (n,)).any()
        with pytest.raises(slamhip.SlamHipError):
            slamhip.knn2_select_device(ctx, dq.buf, n, dq.buf, n, tab.idx, tab.dist, keep, 1, 30.0)
    finally:
        for o in (tab, dq, e, keep):
            o.free()


@pytest.mark.parametrize("n,m", [(16384, 600000), (16385, 3000), (200, 200)])
def test_polled_completion_long_searches_and_its_limits(gpu_ctx, n, m):
    """A search with a count is waited for by polling the completion words its last arrivers store into pinned memory
    (slam_wait_done) when it has at most 64 query blocks.  16384 x 600000 has exactly 64 and runs longer than the 2 ms the
    poll spins for (it falls back to the stream); 16385 rows are 65 query blocks (never polled); 200 x 200 three hundred times
    in a row crosses the every-256th-call synchronisation.  Same count and flags as the two-step way, sampled rows against the
    oracle, the merge state idle."""
    import ctypes

    import slamhip
    from oracle import oracle

    ctx = gpu_ctx
    rng = np.random.default_rng(n * 7 + m)
    q = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (m, 32), dtype=np.uint8)
    k = min(n, m, 100)
    t[rng.choice(m, k, replace=False)] = q[rng.choice(n, k, replace=False)]
    dq, dt = slamhip.DeviceDescriptors(ctx, q), slamhip.DeviceDescriptors(ctx, t)
    tab, keep, keep2 = slamhip.Top2Table(ctx, n), ctx.malloc(n + 64), ctx.malloc(n + 64)
    try:
        for rep in range(300 if n == 200 else 2):
            got = slamhip.knn2_select_device(ctx, dq.buf, n, dt.buf, m, tab.idx, tab.dist, keep, 2, 0.8)
            if rep % 100 == 0 or n != 200:
                idx, dist = tab.download()
                flags = keep.download(np.uint8, (n,))
                cnt, mind = ctypes.c_int64(0), ctypes.c_int32(0)
                assert ctx.lib.slam_bf_match_filter(ctx.handle, tab.idx.ptr, tab.dist.ptr, n, 2, 0.8, keep2.ptr, ctypes.byref(cnt),
                                                    ctypes.byref(mind)) == 0
                assert cnt.value == got == int(flags.sum()) and np.array_equal(keep2.download(np.uint8, (n,)), flags)
                sel = rng.choice(n, min(n, 128), replace=False)
                ridx, rdist = oracle.bf_knn_c(q[sel], t, 2, threads=8)
                assert np.array_equal(idx[sel], ridx) and np.array_equal(dist[sel], rdist)
                assert got >= 1
        assert ctx.state_dirty() == 0
    finally:
        for o in (tab, dq, dt, keep, keep2):
            o.free()
