"""CPU suite: the torch-free launcher (slamhip.launch) - rendezvous collectives between real processes, failure paths
that must end in an error instead of a hang, and the environment conventions of both launch styles."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))
WORKER = os.path.join(ROOT, "tests", "_rdzv_worker.py")


def _run(mode, world, extra=None):
    code = ("import sys; sys.path.insert(0, %r); from slamhip.launch import spawn_ranks; "
            "sys.exit(spawn_ranks(%r, [%r], %d, %r, timeout=120))") % (
                os.path.join(ROOT, "slam-experiments_amd"), WORKER, mode, world, extra or {})
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=180)
    return out, time.monotonic() - t0


def test_collectives_between_processes():
    for world in (2, 5):
        out, _ = _run("ok", world)
        assert out.returncode == 0 and "RDZV_OK" in out.stdout, (out.stdout, out.stderr[-2000:])


def test_a_rank_that_leaves_or_dies_becomes_an_error_not_a_hang():
    for mode in ("skip", "die"):
        out, dt = _run(mode, 3, {"RDZV_TIMEOUT": "5"})
        assert out.returncode != 0 and dt < 60, (mode, out.returncode, dt)
        assert "RDZV_ERROR rank 0" in out.stdout, (mode, out.stdout, out.stderr[-2000:])


def test_identity_from_both_launch_styles():
    from slamhip.launch import Rendezvous, from_env

    assert from_env({}) == (0, 0, 1, None)
    assert from_env({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})[3] is None
    # python -m torch.distributed.run sets these; every rank derives the same name without importing torch
    env = {"WORLD_SIZE": "8", "RANK": "3", "LOCAL_RANK": "3", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29500",
           "TORCHELASTIC_RUN_ID": "none"}
    assert from_env(env) == (3, 3, 8, "127.0.0.1-29500-none")
    assert from_env(dict(env, SLAM_RDZV="spawn-1-2"))[3] == "spawn-1-2"
    # a single-rank "group" needs no server
    rz = Rendezvous(0, 1, "unused")
    assert rz.allgather(5) == [5] and rz.bcast(7) == 7
    rz.barrier()
    rz.close()


def test_launcher_module_imports_no_framework():
    code = ("import sys; sys.path.insert(0, %r); import slamhip.launch, slamhip.dist; "
            "assert 'torch' not in sys.modules and 'mpi4py' not in sys.modules") % os.path.join(ROOT, "slam-experiments_amd")
    assert subprocess.run([sys.executable, "-c", code]).returncode == 0
