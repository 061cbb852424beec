"""Worker for test_comm_gpu.py: one rank of a sharded search (started by slamhip.launch.spawn_ranks; every rank on GPU 0).

Not collected by pytest (leading underscore).  argv[1] = "query" | "train": which partitioning to run.  The ranks
gather through the peer-copy tier (RCCL refuses ranks that share a GPU) and compare with the oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "slam-experiments_amd"), ROOT):
    sys.path.insert(0, p)


def main() -> int:
    import slamhip
    from oracle import oracle
    from slamhip.dist import ShardedMatcher, TrainShardedMatcher
    from slamhip.launch import Rendezvous, from_env

    mode = sys.argv[1]
    rank, _, world, name = from_env()
    rz = Rendezvous(rank, world, name)
    ctx = slamhip.Context(0)
    allgather_obj = rz.allgather

    def barrier():
        ctx.sync()
        rz.barrier()

    rng = np.random.default_rng(99)
    n, m = {"query": (3001, 7000), "train": (500, 9001), "train1": (1, 700)}[mode]   # ragged on purpose; train1: 8-byte slots
    query = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    train = rng.integers(0, 256, (m, 32), dtype=np.uint8)
    train[m // 2:m // 2 + 50] = train[:50]                      # duplicates across the shard boundary: tie rule
    query[:min(n, 50)] = train[:min(n, 50)]
    if mode == "query":
        sm = ShardedMatcher(ctx, rank, world, query, train, collective=None)
    else:
        per = (m + world - 1) // world
        a, b = min(rank * per, m), min((rank + 1) * per, m)
        sm = TrainShardedMatcher(ctx, rank, world, query, train[a:b], a, collective=None)
    ok = sm.enable_p2p(allgather_obj, barrier)
    for _ in range(3):                                           # several passes: buffers are reused
        sm.step()
    barrier()
    idx, d = sm.result()
    ei, ed = oracle.bf_knn_c(query, train, 2, threads=4)
    good = bool(ok and sm.collective == "p2p" and np.array_equal(idx, ei) and np.array_equal(d, ed))
    try:
        sm.free()                                                # peers are mapped: freeing without the barrier must refuse
        good = False
    except RuntimeError:
        pass
    sm.free(barrier)                                             # unmap -> barrier -> free
    ctx.close()
    verdicts = allgather_obj(good)
    rz.close()
    if rank == 0:
        print("SHARD_WORKER_OK" if all(verdicts) else f"SHARD_WORKER_FAILED {verdicts}")
    return 0 if all(verdicts) else 1


if __name__ == "__main__":
    sys.exit(main())
