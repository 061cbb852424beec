"""GPU: the RCCL plumbing that a 1-GPU box can exercise (single-rank communicator, launcher path)."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_single_rank_communicator_allgather_and_broadcast(gpu_ctx):
    """dlopen(librccl) + ncclGetUniqueId + ncclCommInitRank(1 rank) + in-place all-gather / broadcast on the ctx stream."""
    import slamhip
    from slamhip._lib import check
    from slamhip.dist import ShardedMatcher, init_comm

    ctx = slamhip.Context(0)                     # own context: the communicator belongs to it
    try:
        init_comm(ctx, 0, 1, lambda ident: ident)
        q = np.random.default_rng(1).integers(0, 256, (1000, 32), dtype=np.uint8)
        t = np.random.default_rng(2).integers(0, 256, (3000, 32), dtype=np.uint8)
        sm = ShardedMatcher(ctx, 0, 1, q, t)
        sm.world = 2                                   # force the collective path of step() on the 1-rank communicator
        sm.gathered.append(ctx.malloc(sm.slot_bytes))
        sm.my_idx.append(sm.gathered[1].view(0, sm.per * 8))
        sm.my_dist.append(sm.gathered[1].view(sm.per * 8, sm.per * 8))
        for _ in range(5):                             # both buffers, wait_buffer / overlapped gather on the second stream
            sm.step()
        sm.world = 1
        idx, dist = sm.result()
        ridx, rdist = slamhip.knn_match_arrays(q, t, 2)
        assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
        check(ctx.lib.slam_comm_allgather(ctx.handle, sm.my_idx[0].ptr, sm.gathered[0].ptr, sm.slot_bytes))
        check(ctx.lib.slam_comm_broadcast(ctx.handle, sm.gathered[0].ptr, sm.slot_bytes, 0))
        ctx.sync()
        sm.free()
        # calling twice is an error, not a crash
        buf = ctypes.create_string_buffer(128)
        assert ctx.lib.slam_comm_init(ctx.handle, 1, 0, buf) == -5
        check(ctx.lib.slam_comm_destroy(ctx.handle))
        assert ctx.lib.slam_comm_allgather(ctx.handle, None, None, 8) == -5   # not initialised any more
    finally:
        ctx.close()


def test_bench_under_the_driver_launcher_one_rank(built):
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` (the driver's launch line shape)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3",
           "--warmup", "1", "--no-cpu-baseline", "--no-reproj"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 1 and rec["parity_spot_check"] is True and rec["value"] > 1e11
    assert rec["roofline"]["kernel_ms"] > 0 and rec["unit"] == "pairs/s"


@pytest.mark.parametrize("force,expect", [("", ("rccl", "xgmi-p2p-copies")), ("gloo", ("gloo-host-fallback",))])
def test_two_ranks_on_one_gpu_take_the_loud_fallbacks(built, force, expect):
    """Two ranks sharing GPU 0: RCCL refuses duplicate devices, so this exercises the rendezvous, the failure
    handling and the next tiers of bench.py end to end - the peer-copy all-gather through HIP IPC mappings (a real
    device-to-device gather between two processes), and with SLAM_BENCH_COLLECTIVE=gloo the host gather below it.
    Either way the sharded result must equal the oracle spot check."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", SLAM_BENCH_SINGLE_DEVICE="1", SLAM_BENCH_COLLECTIVE=force)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29633" if force else "29635", os.path.join(ROOT, "bench.py"), "--gpus", "2",
           "--steps", "3", "--warmup", "1"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["parity_spot_check"] is True
    assert rec["config"]["collective"] in expect, (rec["config"]["collective"], out.stderr[-2000:])
    assert rec["cpu_baseline"] is None


@pytest.mark.parametrize("mode,port", [("query", "29641"), ("train", "29643")])
def test_sharded_matchers_between_processes(built, mode, port):
    """Three ranks on GPU 0 under torch.distributed.run: the query-sharded and the train-sharded matcher of
    slamhip.dist gather through IPC peer copies (ragged shards, several passes) and must equal the oracle."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr",
           "127.0.0.1", "--master-port", port, os.path.join(ROOT, "tests", "_shard_worker.py"), mode]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0 and "SHARD_WORKER_OK" in out.stdout, (out.stdout[-500:], out.stderr[-2500:])
