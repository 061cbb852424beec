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


def test_spin_barrier_orders_ranks_and_a_dead_rank_is_an_error():
    """The shared-memory barrier bench.py brackets its timed region with: correct under stragglers, cheap, bounded."""
    out, _ = _run("spin", 4)
    assert out.returncode == 0 and "RDZV_OK" in out.stdout, (out.stdout, out.stderr[-2000:])
    us = float(out.stdout.split("SPIN_US")[1].split()[0])
    assert us < 5000.0, us                               # a few microseconds with a core per rank (the socket barrier: 170-630 us); a box with
                                                         # fewer cores than ranks deschedules spinners - only a runaway is an error here
    out, dt = _run("spin-die", 3, {"RDZV_TIMEOUT": "4"})
    assert out.returncode != 0 and dt < 60, (out.returncode, dt)
    assert "RDZV_ERROR rank 0" in out.stdout and "[1]" in out.stdout, (out.stdout, out.stderr[-2000:])


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
    rz.spin_barrier()
    rz.close()


def test_launcher_module_imports_no_framework():
    code = ("import sys; sys.path.insert(0, %r); import slamhip.launch, slamhip.dist; "
            "assert 'torch' not in sys.modules and 'mpi4py' not in sys.modules") % os.path.join(ROOT, "slam-experiments_amd")
    assert subprocess.run([sys.executable, "-c", code]).returncode == 0


def test_wire_codec_round_trips_plain_objects_and_refuses_the_rest():
    import numpy as np
    import pytest
    from slamhip import launch

    objs = [None, True, False, 0, -(2 ** 62), 1.5, "text", b"\x00\xff" * 64, [1, (2.0, "x"), {"k": b"v"}],
            {"rank": 3, "blob": bytes(1000)}, (np.arange(12, dtype=np.int32).reshape(3, 4), np.zeros(0, np.uint8))]
    for o in objs:
        r = launch.loads(launch.dumps(o))
        if isinstance(o, tuple) and isinstance(o[0], np.ndarray):
            assert all(a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b) for a, b in zip(o, r))
        else:
            assert r == o and type(r) is type(o)
    for bad in (object(), {1, 2}, np.array([object()], dtype=object), lambda: 0):
        with pytest.raises(TypeError):
            launch.dumps(bad)
    import pickle
    for junk in (pickle.dumps({"a": 1}), b"", b"Zjunk", launch.dumps([1, 2])[:-3], launch.dumps(1) + b"x"):
        with pytest.raises((ValueError, struct_error(), IndexError)):
            launch.loads(junk)


def struct_error():
    import struct
    return struct.error


def test_server_turns_away_strangers_and_still_admits_the_real_ranks():
    """A local process that knows the socket name but not the token (or claims a bad rank) cannot join, cannot make
    the server unpickle anything, and does not stop the real ranks (launch.py)."""
    import pickle
    import socket
    import threading
    from slamhip import launch

    name = f"test-auth-{os.getpid()}-{time.monotonic_ns()}"
    token = os.urandom(32)
    out = {}

    def rank(r):
        rz = launch.Rendezvous(r, 2, name, timeout=20, token=token)
        out[r] = rz.allgather(("hello", r))
        rz.close()

    t0 = threading.Thread(target=rank, args=(0,))
    t0.start()
    time.sleep(0.3)                                   # server is up, waiting for rank 1
    # 1) a pickle bomb where the hello should be
    c = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    c.connect(launch._address(name))
    c.sendall(pickle.dumps(os.system) + b"\0" * 64)
    # 2) right layout, wrong token
    d = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    d.connect(launch._address(name))
    d.sendall(launch._HELLO.pack(launch._MAGIC, 2, 1, b"\x01" * 32))
    # 3) right token, impossible rank / duplicate rank
    for bad_rank in (7, -1, 0):
        e = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        e.connect(launch._address(name))
        e.sendall(launch._HELLO.pack(launch._MAGIC, 2, bad_rank, token))
        e.settimeout(5)
        assert e.recv(4) == b""                       # closed without an acknowledgement
        e.close()
    for s in (c, d):
        s.settimeout(5)
        try:
            assert s.recv(4) == b""                   # closed ...
        except ConnectionResetError:
            pass                                      # ... or reset (the server left part of the junk unread)
        s.close()
    t1 = threading.Thread(target=rank, args=(1,))
    t1.start()
    t0.join(30)
    t1.join(30)
    assert out == {0: [("hello", 0), ("hello", 1)], 1: [("hello", 0), ("hello", 1)]}


def test_external_launcher_token_file_is_private_and_removed():
    """Under torch.distributed.run no token is handed down: rank 0 publishes one in a 0600 file, the others read it
    only if it is a private regular file of the same user."""
    import stat
    import threading
    import pytest
    from slamhip import launch

    name = f"test-file-{os.getpid()}-{time.monotonic_ns()}"
    env_had = os.environ.pop("SLAM_RDZV_TOKEN", None)
    try:
        res = {}

        def rank(r):
            rz = launch.Rendezvous(r, 2, name, timeout=20)
            if r == 0:
                st = os.stat(launch._token_path(name))
                res["mode"] = stat.S_IMODE(st.st_mode)
            res[r] = rz.bcast(b"id" * 64 if r == 0 else None)
            rz.barrier()
            rz.close()

        ts = [threading.Thread(target=rank, args=(r,)) for r in (0, 1)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(30)
        assert res[0] == res[1] == b"id" * 64 and res["mode"] == 0o600
        assert not os.path.exists(launch._token_path(name))
        # a token file readable by others is refused
        other = f"test-perm-{os.getpid()}-{time.monotonic_ns()}"
        with open(launch._token_path(other), "wb") as f:
            f.write(b"\x02" * 32)
        os.chmod(launch._token_path(other), 0o644)
        try:
            with pytest.raises(launch.RendezvousError):
                launch._read_token(other, 0.2)
        finally:
            os.unlink(launch._token_path(other))
    finally:
        if env_had is not None:
            os.environ["SLAM_RDZV_TOKEN"] = env_had


def test_a_stale_token_file_of_a_crashed_run_does_not_lock_a_rank_out():
    """Under an external launcher a rank may read the token file before rank 0 has replaced the leftover of an earlier,
    crashed run of the same name: it is turned away once, reads the file again and joins."""
    import threading
    from slamhip import launch

    name = f"test-stale-{os.getpid()}-{time.monotonic_ns()}"
    env_had = os.environ.pop("SLAM_RDZV_TOKEN", None)
    path = launch._token_path(name)
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
    os.write(fd, b"\x5a" * 32)                              # the leftover
    os.close(fd)
    res = {}

    def rank(r, delay):
        time.sleep(delay)
        rz = launch.Rendezvous(r, 2, name, timeout=20)
        res[r] = rz.allgather(r)
        rz.close()

    try:
        ts = [threading.Thread(target=rank, args=(1, 0.0)), threading.Thread(target=rank, args=(0, 0.5))]   # rank 1 looks first
        for t in ts:
            t.start()
        for t in ts:
            t.join(40)
        assert res == {0: [0, 1], 1: [0, 1]}
        assert not os.path.exists(path)
    finally:
        if env_had is not None:
            os.environ["SLAM_RDZV_TOKEN"] = env_had
        if os.path.exists(path):
            os.unlink(path)
