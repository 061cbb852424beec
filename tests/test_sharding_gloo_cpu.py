"""CPU suite: the N>1 path's sharding / gather logic over gloo, world_size 2 (per-shard search = the oracle)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, m, out):
    sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from oracle import oracle
    from slamhip.dist import ShardPlan

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q = np.random.default_rng(228).integers(0, 256, (n, 32), dtype=np.uint8)
    t = np.random.default_rng(229).integers(0, 256, (m, 32), dtype=np.uint8)
    plan = ShardPlan(n, world)
    a, b = plan.rows(rank)
    idx, dst = oracle.bf_knn_c(q[a:b], t, 2)           # what the GPU kernel computes for this rank's rows
    per = plan.rows_per_rank
    slot_i = torch.full((per, 2), -1, dtype=torch.int32)
    slot_d = torch.full((per, 2), 2**31 - 1, dtype=torch.int32)
    slot_i[: b - a] = torch.from_numpy(idx)
    slot_d[: b - a] = torch.from_numpy(dst)
    gi = [torch.empty_like(slot_i) for _ in range(world)]
    gd = [torch.empty_like(slot_d) for _ in range(world)]
    dist.all_gather(gi, slot_i)                           # stands in for the RCCL all-gather of equal-sized slots
    dist.all_gather(gd, slot_d)
    full_i = torch.cat(gi)[:n].numpy()
    full_d = torch.cat(gd)[:n].numpy()
    ri, rd = oracle.bf_knn_c(q, t, 2)
    ok = np.array_equal(full_i, ri) and np.array_equal(full_d, rd)
    # the unique-id broadcast used to bootstrap RCCL
    box = [bytes(range(128)) if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    ok = ok and box[0] == bytes(range(128))
    dist.barrier()
    dist.destroy_process_group()
    out.put((rank, bool(ok)))


@pytest.mark.parametrize("n,m", [(1000, 777), (7, 50), (1, 3)])
def test_query_sharded_gather_equals_monolithic(n, m):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29500 + (os.getpid() + n) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, m, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def test_shard_plan_edges():
    from slamhip.dist import ShardPlan, gather_rows_host

    p = ShardPlan(10, 4)
    assert p.rows_per_rank == 3 and [p.rows(r) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    p = ShardPlan(2, 8)
    assert [p.rows(r) for r in range(8)] == [(0, 1), (1, 2)] + [(2, 2)] * 6
    assert ShardPlan(0, 8).rows_per_rank == 0
    shards = [np.full((b - a, 2), r, np.int32) for r, (a, b) in enumerate(ShardPlan(10, 4).rows(r) for r in range(4))]
    full = gather_rows_host(ShardPlan(10, 4), shards)
    assert full.shape == (10, 2) and full[:, 0].tolist() == [0, 0, 0, 1, 1, 1, 2, 2, 2, 3]


def test_call_with_deadline_turns_a_call_that_never_returns_into_an_error():
    """What bench.py wraps the RCCL set-up and its first collective round in."""
    import time

    from slamhip.dist import DeadlineExceeded, call_with_deadline

    assert call_with_deadline(lambda: 5, None, "x") == 5 and call_with_deadline(lambda: 6, 1.0, "x") == 6
    with pytest.raises(ZeroDivisionError):
        call_with_deadline(lambda: 1 / 0, 1.0, "x")
    t0 = time.monotonic()
    with pytest.raises(DeadlineExceeded, match="the sleeping call did not return within"):
        call_with_deadline(lambda: time.sleep(30), 0.3, "the sleeping call")
    assert time.monotonic() - t0 < 5
