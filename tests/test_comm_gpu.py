"""GPU: the RCCL plumbing that a 1-GPU box can exercise (single-rank communicator, launcher path)."""
import ctypes
import json
import os
import subprocess
import time
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


def _bench_line(out):
    assert out.returncode == 0, (out.stdout[-500:], out.stderr[-3000:])
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_under_the_driver_launcher_one_rank(built):
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` (the driver's launch line shape)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3",
           "--warmup", "1", "--no-cpu-baseline", "--no-reproj", "--no-pipelined"]
    rec = _bench_line(subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT))
    assert rec["n_gpus"] == 1 and rec["parity_spot_check"] is True and rec["value"] > 1e11
    assert rec["roofline"]["kernel_ms"] > 0 and rec["unit"] == "pairs/s"


@pytest.mark.parametrize("force,expect", [("", ("rccl", "xgmi-p2p-copies")), ("host", ("host-fallback",))])
def test_self_contained_launch_two_ranks_on_one_gpu(built, force, expect):
    """`python bench.py --gpus 2`, nothing else: bench.py starts its own rank processes (slamhip.launch, no torch).
    Two ranks sharing GPU 0: RCCL refuses duplicate devices, so this exercises the rendezvous, the failure handling
    and the next tiers end to end - the peer-copy all-gather through HIP IPC mappings (a real device-to-device gather
    between two processes), and with SLAM_BENCH_COLLECTIVE=host the host gather below it.  Either way the sharded
    result must equal the oracle spot check."""
    env = dict(os.environ, SLAM_BENCH_SINGLE_DEVICE="1", SLAM_BENCH_COLLECTIVE=force)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SLAM_RDZV"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    rec = _bench_line(out)
    assert rec["n_gpus"] == 2 and rec["parity_spot_check"] is True
    assert rec["config"]["collective"] in expect, (rec["config"]["collective"], out.stderr[-2000:])
    assert rec["cpu_baseline"] is None and "N=1" in rec["cpu_baseline_note"]


def test_n_gt_1_line_says_which_collective_ran_and_why(built):
    """the N > 1 line must be readable from the record alone - which tier carried the gather and why
    the tiers above it did not (the exception text of the RCCL / peer-mapping set-up), the RCCL version, the shard
    kernel's roofline and the per-rank kernel times."""
    env = dict(os.environ, SLAM_BENCH_SINGLE_DEVICE="1", SLAM_BENCH_COLLECTIVE="host")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SLAM_RDZV"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    rec = _bench_line(subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT))
    cfg, roof = rec["config"], rec["roofline"]
    assert cfg["collective"] == "host-fallback" and "SLAM_BENCH_COLLECTIVE=host" in cfg["collective_fallback_reason"]
    assert isinstance(cfg["rccl_version"], int) and cfg["rccl_version"] > 20000
    assert roof["kernel"] == "bf_top2_kernel" and roof["achieved"] > 0 and 0 < roof["frac"] < 1
    assert roof["algorithmic_bytes_per_launch"] == 32 * (32768 + 65536) + 16 * 32768          # the SHARD's bytes
    assert 0 < roof["kernel_ms_per_rank"]["min"] <= roof["kernel_ms_per_rank"]["max"] == roof["kernel_ms"]
    assert 0 < roof["valu_int"]["frac_of_32lane_peak"] <= 1 and 0 < roof["valu_int"]["frac_of_issue_floor"] <= 1
    assert cfg["launch_plan"]["qblocks"] == 128


def test_a_communicator_set_up_that_never_returns_becomes_a_labelled_fallback(built):
    """The RCCL set-up runs under a deadline: rank 1 "never returns" from its ncclCommInitRank (simulated), rank 0's real
    one waits for it in vain - both give up after SLAM_BENCH_DEADLINE seconds, the line says so, the run finishes on the
    peer-copy tier with a fresh context and the processes leave (os._exit) although helper threads are still stuck."""
    env = dict(os.environ, SLAM_BENCH_SINGLE_DEVICE="1", SLAM_BENCH_FAKE_HANG="init", SLAM_BENCH_DEADLINE="6")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SLAM_RDZV", "SLAM_BENCH_COLLECTIVE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    t0 = time.monotonic()
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert time.monotonic() - t0 < 300, "the run must not wait for the stuck calls"
    rec = _bench_line(out)
    assert rec["n_gpus"] == 2 and rec["parity_spot_check"] is True
    assert rec["config"]["collective"] in ("xgmi-p2p-copies", "host-fallback"), rec["config"]
    assert "did not return within" in rec["config"]["collective_fallback_reason"], rec["config"]["collective_fallback_reason"]


@pytest.mark.parametrize("ranks,force", [(1, ""), (3, "p2p")])
def test_loop_closure_workload(built, ranks, force):
    """BASELINE configs[3] end to end: `bench.py --workload loop-closure` - 512 keyframes x 2048 rows all-to-all, query
    keyframes sharded over the ranks, top-2 rows gathered (three processes on the one GPU: the peer-copy tier, a real
    device-to-device gather between processes) and decoded on the device to (imgIdx, trainIdx, distance); the spot check
    is the oracle's multi-image search (imgIdx << 18 | trainIdx, OpenCV matchers.cpp) on 256 sampled queries."""
    env = dict(os.environ, SLAM_BENCH_SINGLE_DEVICE="1", SLAM_BENCH_COLLECTIVE=force)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SLAM_RDZV"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--workload", "loop-closure", "--steps", "2",
           "--warmup", "0", "--no-cpu-baseline"]
    rec = _bench_line(subprocess.run(cmd, capture_output=True, text=True, timeout=1100, env=env, cwd=ROOT))
    assert rec["n_gpus"] == ranks and rec["parity_spot_check"] is True
    assert rec["config"]["n_query"] == rec["config"]["n_train"] == 512 * 2048 and "loop closure" in rec["config"]["workload"]
    assert rec["config"]["collective"] == ("none" if ranks == 1 else "xgmi-p2p-copies")
    assert rec["roofline"]["kernel"] == "bf_top2_kernel" and rec["roofline"]["achieved"] > 0
    assert rec["config"]["launch_plan"]["sgpr_feed"] == 1
    if ranks == 1:
        assert rec["value"] > 2.5e12          # round 2 measured 2.91e12 pairs/s here


def test_driver_launch_line_two_ranks_needs_no_torch_in_the_ranks(built):
    """The driver's N > 1 line (`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`): the ranks read
    RANK / WORLD_SIZE / MASTER_PORT from the environment and rendezvous through slamhip.launch; torch stays the
    launcher's business (a sitecustomize hook makes `import torch` inside a rank fail the run)."""
    hook = os.path.join(ROOT, "gpurun_out", "_no_torch_hook")
    os.makedirs(hook, exist_ok=True)
    with open(os.path.join(hook, "sitecustomize.py"), "w") as f:
        f.write("import os, sys\n"
                "if os.environ.get('RANK') is not None and sys.argv and sys.argv[0].endswith('bench.py'):\n"
                "    import importlib.abc\n"
                "    class _Deny(importlib.abc.MetaPathFinder):\n"
                "        def find_spec(self, name, path, target=None):\n"
                "            if name == 'torch' or name.startswith('torch.'):\n"
                "                raise ImportError('torch imported inside a bench.py rank')\n"
                "    sys.meta_path.insert(0, _Deny())\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", SLAM_BENCH_SINGLE_DEVICE="1",
               PYTHONPATH=hook + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29637", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
    rec = _bench_line(subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT))
    assert rec["n_gpus"] == 2 and rec["parity_spot_check"] is True


@pytest.mark.parametrize("mode", ["query", "train", "train1"])
def test_sharded_matchers_between_processes(built, mode):
    """Three ranks on GPU 0 (slamhip.launch.spawn_ranks): the query-sharded and the train-sharded matcher of slamhip.dist
    gather through IPC peer copies (ragged shards, several passes, a one-query train-sharded case with 8-byte slots)
    and must equal the oracle; freeing peer-mapped buffers without the barrier is refused."""
    from slamhip.launch import spawn_ranks  # noqa: F401  (the child interpreter below does the spawning)

    code = ("import sys; sys.path.insert(0, %r); from slamhip.launch import spawn_ranks; "
            "sys.exit(spawn_ranks(%r, [%r], 3, timeout=500))") % (
                os.path.join(ROOT, "slam-experiments_amd"), os.path.join(ROOT, "tests", "_shard_worker.py"), mode)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0 and "SHARD_WORKER_OK" in out.stdout, (out.stdout[-500:], out.stderr[-2500:])
