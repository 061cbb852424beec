"""CPU suite: the launch planner of the top-2 search as a pure function (slam_bf_plan_describe needs no device) - the
chunk boundary table covers the train rows exactly under every knob setting, and the shipped rules are the measured ones
(leaders from 16384 rows up, one block per CU up to 128 rows a chunk and about 8 sqrt(that) beyond,
no tail and no table for grids of up to two blocks per CU, no bound exchange when every chunk fits its unfiltered start,
SGPR feed for long chunks and for chunks that are unfiltered throughout but never for rows in pinned host memory)."""
import itertools

import numpy as np
import pytest


@pytest.fixture(scope="module")
def plan(built):
    import slamhip

    return slamhip.plan_describe


def _check_table(p, tbl, m):
    assert len(tbl) == p["chunks"] + 1 and tbl[0] == 0 and tbl[-1] == m, (p, tbl[:4], tbl[-4:])
    steps = np.diff(tbl)
    assert (steps > 0).all(), "every chunk holds at least one row"
    if p["lead_chunks"]:
        assert tbl[p["lead_chunks"]] == p["lead_rows"]
    if p["tail_chunks"]:
        tail = steps[-p["tail_chunks"]:]
        assert (np.diff(tail[:-1]) <= 0).all(), "the tail shrinks"           # (the very last chunk takes the remainder)
    if p["table_free"]:
        assert p["lead_chunks"] == 0 and p["tail_chunks"] == 0
        assert (steps[:-1] == p["chunk"]).all() and steps[-1] <= p["chunk"], "a table-free plan is by * chunk"
    if p["workers"]:                                                             # a queue plan
        assert p["lead_chunks"] == 0 and not p["table_free"] and not p["bound_free"] and p["sgpr_feed"] == 1 and p["R"] == 1
        assert 1 <= p["workers"] <= p["chunks"]
        assert p["workers"] == 1 or p["workers"] * p["qblocks"] <= p["cus"] * p["resident"], "the workers are resident all at once"
        assert (steps[:-1] % 16 == 0).all(), "only the last chunk of a queue may hold a ragged group"
        assert steps.max() <= max(p["chunk"], 64)
    if p["bound_free"]:
        assert steps.max() <= p["cold_rows"] and p["lead_chunks"] == 0
        assert p["qblocks"] * p["chunks"] <= p["cus"] * 8, "only grids that are resident all at once go without bounds"


def test_tables_cover_the_train_rows_under_any_knobs(plan):
    rng = np.random.default_rng(5)
    shapes = [(1, 1), (200, 200), (1, 70000), (257, 769), (4096, 4096), (5000, 16383), (5000, 16384), (8192, 65536),
              (65536, 65536), (70, 1 << 20), (1 << 20, 1 << 20), (3, (1 << 23))]
    shapes += [(int(rng.integers(1, 9000)), int(rng.integers(1, 300000))) for _ in range(40)]
    for n, m in shapes:
        for cus in (256, 304, 8):
            p, tbl = plan(n, m, num_cu=cus)
            _check_table(p, tbl, m)
            assert p["R"] == 1 and p["qblocks"] == (n + 255) // 256 and p["cus"] == cus
    knob_sets = itertools.product((0, 2), (0, 1, 16, 64), (0, -1, 256, 4096), (0, 64), (0, -1, 5, 64), (0, -1), (0, -1, 16, 1024), (0, 96, 512),
                                  (0, -1))
    for (R, bpc, lead, lchunk, tail, feed, cold, chunk, queue), (n, m) in zip(knob_sets, itertools.cycle(shapes[1:20])):
        p, tbl = plan(n, m, R=R, blocks_per_cu=bpc, lead_rows=lead, lead_chunk=lchunk, tail=tail, feed=feed, cold=cold, chunk=chunk,
                      queue=queue)
        _check_table(p, tbl, m)
        assert queue == 0 or p["workers"] == 0
        assert p["R"] == (R or 1) and (p["sgpr_feed"] == 0 or p["R"] == 1)
        if cold == -1:
            assert p["cold_rows"] == 0 and not p["bound_free"]
        if lead == -1:
            assert p["lead_chunks"] == 0
        if tail == -1:
            assert p["tail_chunks"] == 0
    # forced queue plans: any shape, any chunk / shortest-chunk / cold setting
    for (chunk, tail, cold), (n, m) in zip(itertools.product((0, 32, 512, 4096), (0, -1, 16, 100), (0, -1, 64)), itertools.cycle(shapes)):
        p, tbl = plan(n, m, queue=1, chunk=chunk, tail=tail, cold=cold)
        _check_table(p, tbl, m)
        assert p["workers"] >= 1 and (tail != -1 or p["tail_chunks"] == 0)


def test_queue_plans_on_a_256_cu_device(plan):
    """Round 4: train sets from 16384 rows up whose query blocks each get at least two resident workers with at least 1024
    rows apiece run as a QUEUE - workers x query blocks fill the chip once, the chunks of the table are drawn by ticket."""
    for (n, m), workers, chunk in (((8192, 65536), 48, 256), ((16384, 65536), 24, 256), ((32768, 65536), 12, 256),
                                   ((65536, 65536), 6, 256), ((131072, 65536), 3, 1024), ((32768, 1 << 18), 12, 1024), ((20000, 20000), 19, 256),
                                   ((65536, 1 << 17), 6, 1024), ((3000, 200000), 128, 256), ((2048, 1 << 18), 192, 256)):   # (1024-row chunks from 16384 rows a worker)
        p, tbl = plan(n, m)
        assert p["workers"] == workers and p["chunk"] == chunk and p["resident"] == 6 and p["lead_rows"] == 0, (n, m, p)
        steps = np.diff(tbl)
        assert steps[0] == chunk and steps[-2] == 64 and p["tail_chunks"] > p["workers"]            # the queue ends on short chunks
        assert len(tbl) <= 4096, "the table fits one slot of the ring (no stream synchronisation when the shape changes)"
        assert p["merge"] == (1 if 12 <= workers <= 96 else 0)        # many (not very many) workers per query: they exchange by merging
        assert plan(n, m, merge=1)[0]["merge"] == 1 and plan(n, m, merge=-1)[0]["merge"] == 0
    # not a queue: more query blocks than half the resident slots, workers that would fill less than 96 % of the chip, few rows per worker, small train sets, batches, host rows
    for (n, m), kw in (((200000, 65536), {}), ((1 << 20, 1 << 20), {}), ((100, 20000), {}), ((5000, 30000), {}), ((8192, 16000), {}), ((4096, 65536), {}), ((120000, 65536), {}), ((50000, 20000), {}), ((131072, 1 << 20), {}), ((1000, 400000), {}),
                       ((8192, 65536), dict(qb_all=64)), ((8192, 65536), dict(rows_on_host=True)), ((8192, 65536), dict(queue=-1)),
                       ((8192, 65536), dict(feed=-1)), ((8192, 65536), dict(R=2))):
        assert plan(n, m, **kw)[0]["workers"] == 0, (n, m, kw)


def test_shipped_rules_on_a_256_cu_device(plan):
    """The one-block-per-chunk plans (every train set below 16384 rows; above, what queue=-1 or a shape that does not qualify
    for a queue gets)."""
    # leaders: M / 8 up to 8192 rows from 16384 train rows up, none below
    for m, lead in ((200, 0), (16383, 0), (16384, 2048), (65536, 8192), (1 << 20, 8192)):
        p, _ = plan(8192, m, queue=-1)
        assert p["lead_rows"] == lead and (p["lead_chunks"] > 0) == (lead > 0), (m, p)
    # small train sets: one block per CU up to 128 rows a chunk, about 8 sqrt(that) beyond, at most 512
    for (n, m), chunk in (((200, 200), 32), ((1000, 1000), 32), ((2000, 2000), 64), ((4096, 4096), 128), ((8192, 8192), 256),
                          ((12000, 12000), 384), ((65536, 4096), 512)):
        p, _ = plan(n, m)
        # the unfiltered start is 128 rows - and the whole chunk for chunks of up to 384 rows
        assert p["chunk"] == chunk and p["cold_rows"] == (max(128, chunk) if chunk <= 384 else 128) and p["lead_rows"] == 0, (n, m, p)
    # sixteen 4096 x 4096 searches in one launch plan their chunks for the whole grid
    assert plan(4096, 4096, qb_all=256)[0]["chunk"] == 512
    # up to two blocks per CU: uniform chunks, no table; with every chunk inside the unfiltered start also no bounds
    for n, m in ((200, 200), (1000, 1000), (3000, 3000), (4096, 4096)):
        p, _ = plan(n, m)
        assert p["table_free"] and p["bound_free"] and p["tail_chunks"] == 0, (n, m, p)
    p, _ = plan(12000, 12000)
    assert not p["table_free"] and p["bound_free"] and p["tail_chunks"] > 0 and p["cold_rows"] == 384
    p, _ = plan(16000, 16000)
    assert not p["table_free"] and not p["bound_free"] and p["cold_rows"] == 128
    p, _ = plan(65536, 65536, queue=-1)
    assert not p["table_free"] and not p["bound_free"] and p["chunk"] == 4096 and p["lead_chunks"] == 1 and p["chunks"] == 19   # 16 blocks per CU
    # the feed: SGPRs wherever the rows lie in device memory (since round 4 also on the leader regime's one-tile chunks); the LDS
    # tile for rows in pinned host memory unless the chunks are long, and for several queries per lane
    assert plan(65536, 65536)[0]["sgpr_feed"] == 1 and plan(8192, 65536)[0]["sgpr_feed"] == 1
    assert plan(4096, 4096)[0]["sgpr_feed"] == 1 and plan(200, 200)[0]["sgpr_feed"] == 1
    assert plan(8192, 8192)[0]["sgpr_feed"] == 1 and plan(12000, 12000)[0]["sgpr_feed"] == 1
    assert plan(5000, 20000, chunk=256, lead_rows=-1)[0]["sgpr_feed"] == 1 and plan(200, 20000)[0]["sgpr_feed"] == 1
    assert plan(2048, 40000)[0]["sgpr_feed"] == 1 and plan(5000, 20000, feed=-1)[0]["sgpr_feed"] == 0 and plan(5000, 20000, R=2)[0]["sgpr_feed"] == 0
    assert plan(3000, 20000, rows_on_host=True)[0]["sgpr_feed"] == 0                # rows in pinned host memory: the LDS tile
    assert plan(4096, 4096, rows_on_host=True)[0]["sgpr_feed"] == 0 and plan(200, 200, rows_on_host=True)[0]["sgpr_feed"] == 0
    # no more query blocks than CUs: 16 blocks per CU and up to 32 shrinking chunks at the end (the 1/8 shard of the headline
    # grid); more query blocks than CUs: 32 blocks per CU
    p, _ = plan(8192, 65536, queue=-1)
    assert p["chunk"] == 512 and p["qblocks"] == 32 and p["tail_chunks"] == 32 and p["lead_chunks"] == 8
    assert plan(131072, 65536, queue=-1)[0]["chunk"] == 4096 and plan(1 << 20, 1 << 20)[0]["chunk"] == 131072   # no chunk longer than that


def test_bad_arguments_are_refused(plan):
    import slamhip

    for kw in (dict(R=3), dict(lead_chunk=48), dict(cold=24), dict(chunk=100), dict(feed=2), dict(R=2, feed=1), dict(queue=2),
               dict(queue=1, R=2), dict(queue=1, feed=-1)):
        with pytest.raises(slamhip.SlamHipError):
            plan(1000, 1000, **kw)
    with pytest.raises(slamhip.SlamHipError):
        plan(0, 10)
    with pytest.raises(slamhip.SlamHipError):
        plan(10, (1 << 23) + 1)
    with pytest.raises(TypeError):
        plan(10, 10, nonsense=1)
