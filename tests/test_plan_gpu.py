"""GPU: the launch plan of the top-2 search (leader chunks that publish exact bounds, shrinking chunks at the end of
the grid, queries per lane, train rows through an LDS tile or through SGPRs, the unfiltered start of a cold chunk, the
chunk length of small train sets) must not change a single bit of the result."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _search(ctx, dq, n, dt, m, tab):
    import slamhip

    slamhip.knn2_device(ctx, dq.buf, n, dt.buf, m, tab.idx, tab.dist)
    return tab.download()


@pytest.mark.parametrize("n,m", [(300, 1500), (1000, 5000), (257, 769), (64, 16384), (5000, 20000)])
def test_every_plan_shape_is_bit_identical(gpu_ctx, n, m):
    import slamhip
    from oracle import oracle

    ctx = gpu_ctx
    rng = np.random.default_rng(n * 7 + m)
    q = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (m, 32), dtype=np.uint8)
    # ties across the leader / rest boundary: the same row before and after it, exact copies of queries on both sides
    t[m - 3] = t[5]
    t[m // 2] = t[5]
    q[0] = t[5]
    q[1] = t[m - 1]
    ridx, rdist = oracle.bf_knn_c(q, t, 2, threads=8)
    assert ridx[0].tolist() == [5, m // 2] and rdist[0].tolist() == [0, 0]
    dq, dt = slamhip.DeviceDescriptors(ctx, q), slamhip.DeviceDescriptors(ctx, t)
    tab = slamhip.Top2Table(ctx, n)
    try:
        for R in (1, 2, 4, 8):
            for lead_rows, lead_chunk in ((-1, 0), (256, 0), (512, 64), (1024, 32), (0, 0)):
                for bpc, tail in ((0, 0), (8, 5), (0, 64), (32, -1)):
                    for feed in ((-1, 1) if R == 1 else (0,)):     # R = 1: both ways of feeding the train rows
                        for cold in (0, -1, 16, 1024):             # shipped / no unfiltered start / one group / whole chunks
                            ctx.set_tuning(R=R, blocks_per_cu=bpc, lead_rows=lead_rows, lead_chunk=lead_chunk, tail=tail, feed=feed,
                                           cold=cold)
                            idx, dist = _search(ctx, dq, n, dt, m, tab)
                            assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist), (R, lead_rows, lead_chunk, bpc, tail, feed, cold)
        assert ctx.state_dirty() == 0
        # queue plans (round 4): resident workers whose waves draw the chunks by ticket - any uniform chunk, shortest chunk
        # and unfiltered start; with and without the shrinking end of the queue
        # and both ways of exchanging what the workers know (a per-query bound / merging their pairs into the result slot)
        for chunk in (0, 32, 64, 512):
            for tail in (0, -1, 16, 48):
                for cold in (0, -1, 16, 1024):
                    for merge in (1, -1):
                        ctx.set_tuning(queue=1, chunk=chunk, tail=tail, cold=cold, merge=merge)
                        idx, dist = _search(ctx, dq, n, dt, m, tab)
                        assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist), ("queue", chunk, tail, cold, merge)
                        assert ctx.state_dirty() == 0, ("queue", chunk, tail, cold, merge)
    finally:
        ctx.set_tuning()
        for o in (tab, dq, dt):
            o.free()


@pytest.mark.parametrize("n,m", [(200, 200), (300, 1500), (4100, 4000), (70, 16383)])
def test_small_train_sets_any_chunk_length_and_cold_start(gpu_ctx, n, m):
    """Train sets below the leader regime: forced chunk lengths (one-round grids run without a boundary table and
    without any bound exchange when the chunk fits into the unfiltered start) against the oracle, ties included."""
    import slamhip
    from oracle import oracle

    ctx = gpu_ctx
    rng = np.random.default_rng(n + 3 * m)
    q = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (m, 32), dtype=np.uint8)
    t[m - 1] = t[0]                      # the same row in the first and in the last chunk
    t[m // 2 + 1] = t[m // 2]            # and next to each other
    q[0] = t[0]
    q[1] = t[m // 2]
    ridx, rdist = oracle.bf_knn_c(q, t, 2, threads=8)
    assert ridx[0].tolist() == [0, m - 1] and ridx[1].tolist() == [m // 2, m // 2 + 1]
    dq, dt = slamhip.DeviceDescriptors(ctx, q), slamhip.DeviceDescriptors(ctx, t)
    tab = slamhip.Top2Table(ctx, n)
    try:
        for chunk in (0, 32, 96, 128, 160, 512, 4096):
            for cold in (0, -1, 16, 64, 4096):
                for tail in (0, -1, 3):
                    for feed in (-1, 1):
                        ctx.set_tuning(chunk=chunk, cold=cold, tail=tail, feed=feed)
                        idx, dist = _search(ctx, dq, n, dt, m, tab)
                        assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist), (chunk, cold, tail, feed)
    finally:
        ctx.set_tuning()
        for o in (tab, dq, dt):
            o.free()


def test_plan_tables_and_degenerate_inputs(gpu_ctx):
    """Shipped plans: leaders from 16384 train rows up, none for frame-sized inputs; forced plans cover the train set
    exactly; all-equal rows (every distance ties) and train sets whose best rows all sit in / behind the leader rows."""
    import slamhip
    from oracle import oracle

    ctx = gpu_ctx
    try:
        ctx.set_tuning(queue=-1)
        for m, lead in ((200, 0), (16383, 0), (16384, 2048), (65536, 8192), (1 << 20, 8192)):
            p = ctx.plan_info(8192, m)
            assert p["lead_rows"] == lead and (p["lead_chunks"] > 0) == (lead > 0), (m, p)
        ctx.set_tuning(lead_rows=1000, lead_chunk=96, tail=9, blocks_per_cu=16, queue=-1)
        p = ctx.plan_info(5000, 30000)
        assert p["lead_rows"] == 992 and p["lead_chunks"] == 11 and p["tail_chunks"] == 9 and p["chunks"] > 20
    finally:
        ctx.set_tuning()
    # the feed follows the chunk length: SGPRs from 512 rows per chunk up and for chunks that fit the unfiltered start (rows
    # in device memory, which is what plan_info assumes; rows in the pinned staging block of a host call take the LDS tile),
    # the LDS tile in between
    assert ctx.plan_info(65536, 65536)["sgpr_feed"] == 1 and ctx.plan_info(1 << 20, 1 << 20)["sgpr_feed"] == 1
    assert ctx.plan_info(65536, 4096)["sgpr_feed"] == 1 and ctx.plan_info(8192, 65536)["sgpr_feed"] == 1
    assert ctx.plan_info(4096, 4096)["sgpr_feed"] == 1 and ctx.plan_info(200, 200)["sgpr_feed"] == 1
    assert ctx.plan_info(8192, 8192)["sgpr_feed"] == 1 and ctx.plan_info(16000, 16000)["sgpr_feed"] == 1
    # small train sets: one block per CU up to 128 rows a chunk, about 8 sqrt(that) beyond; no tail up to two blocks per CU
    cus = ctx.plan_info(200, 200)["cus"]
    if cus == 256:
        for (n_, m_), chunk in (((200, 200), 32), ((2000, 2000), 64), ((4096, 4096), 128), ((8192, 8192), 256), ((65536, 4096), 512)):
            p = ctx.plan_info(n_, m_)
            assert p["chunk"] == chunk and p["cold_rows"] == (max(128, chunk) if chunk <= 384 else 128) and p["lead_rows"] == 0, (n_, m_, p)
        assert ctx.plan_info(4096, 4096)["tail_chunks"] == 0 and ctx.plan_info(12000, 12000)["tail_chunks"] > 0
    n, m = 700, 20000
    rng = np.random.default_rng(3)
    q = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    for case in ("zeros", "front", "back"):
        if case == "zeros":
            t = np.zeros((m, 32), np.uint8)
        else:
            t = rng.integers(0, 256, (m, 32), dtype=np.uint8)
            rows = np.arange(n) if case == "front" else m - 1 - np.arange(n)
            t[rows] = q                                         # every query has an exact copy in / behind the leader rows
        ridx, rdist = oracle.bf_knn_c(q, t, 2, threads=8)
        for wait in (0, 1):
            for feed, queue in ((-1, 0), (1, -1), (1, 1)):
                try:
                    ctx.set_tuning(lead_rows=-wait, tail=7 * wait, feed=feed, queue=queue)
                    idx, dist = slamhip.knn_match_arrays(q, t, 2, ctx=ctx)
                finally:
                    ctx.set_tuning()
                assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist), (case, wait, feed, queue)
                assert ctx.state_dirty() == 0


def test_reset_state_between_searches(gpu_ctx):
    """slam_bf_reset_state puts best/bound/arrivals/led back: searches before and after it agree with the oracle."""
    import slamhip
    from oracle import oracle

    ctx, lib = gpu_ctx, gpu_ctx.lib
    q = np.random.default_rng(1).integers(0, 256, (900, 32), dtype=np.uint8)
    t = np.random.default_rng(2).integers(0, 256, (17000, 32), dtype=np.uint8)
    ridx, rdist = oracle.bf_knn_c(q, t, 2, threads=8)
    for _ in range(2):
        idx, dist = slamhip.knn_match_arrays(q, t, 2, ctx=ctx)
        assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
        assert ctx.state_dirty() == 0                     # every search leaves best / bound / arrivals / cursors idle by itself
        assert lib.slam_bf_reset_state(ctx.handle) == 0
        assert ctx.state_dirty() == 0


def test_queue_plan_is_what_large_searches_run_and_it_leaves_the_state_idle(gpu_ctx):
    """The shipped plan of a query shard (8192 x 65536: the per-rank problem of an 8-GPU run) is a queue of resident workers;
    it is bit-identical to the one-block-per-chunk plan and to the oracle on sampled rows, planted ties across chunk
    boundaries included, and no key, bound, ticket or cursor survives the launch (asserted on the state itself,
    not inferred from later results)."""
    import slamhip
    from oracle import oracle

    ctx = gpu_ctx
    n, m = 8192, 65536
    p, tbl = slamhip.plan_describe(n, m, num_cu=ctx.plan_info(n, m)["cus"])
    assert p["workers"] >= 2 and p["workers"] * p["qblocks"] <= p["cus"] * p["resident"]
    rng = np.random.default_rng(2026)
    q = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (m, 32), dtype=np.uint8)
    for i, b in enumerate(tbl[1:40]):                      # copies of query i on both sides of chunk boundaries: (0, b - 1) and (0, b)
        t[b - 1] = q[2 * i]
        t[b] = q[2 * i + 1]
        t[m - 1 - i] = q[2 * i]                            # and once more in the short chunks at the end of the queue
    dq, dt = slamhip.DeviceDescriptors(ctx, q), slamhip.DeviceDescriptors(ctx, t)
    tab = slamhip.Top2Table(ctx, n)
    try:
        got = {}
        for queue, merge in ((0, 0), (-1, 0), (1, 1), (1, -1)):
            ctx.set_tuning(queue=queue, merge=merge)
            got[queue, merge] = _search(ctx, dq, n, dt, m, tab)
            assert ctx.state_dirty() == 0, (queue, merge)
        for key in got:
            assert np.array_equal(got[0, 0][0], got[key][0]) and np.array_equal(got[0, 0][1], got[key][1]), key
        got = {0: got[0, 0]}
        rows = np.r_[np.arange(80), rng.choice(n, 300, replace=False)]
        ridx, rdist = oracle.bf_knn_c(q[rows], t, 2, threads=8)
        assert np.array_equal(got[0][0][rows], ridx) and np.array_equal(got[0][1][rows], rdist)
        assert got[0][0][0].tolist() == [tbl[1] - 1, m - 1] and got[0][1][0].tolist() == [0, 0]
    finally:
        ctx.set_tuning()
        for o in (tab, dq, dt):
            o.free()
