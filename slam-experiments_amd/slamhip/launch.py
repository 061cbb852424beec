"""Process plumbing for multi-GPU runs, standard library only (no torch, no MPI).

The reference is a single process (``slam.py:22-35``); the sharded searches of ``slamhip.dist`` need three things
from a launcher: the ranks' identities, a way to ship small objects between them (RCCL's 128-byte id, HIP IPC
handles, verdicts, timings) and a barrier.  This module provides them:

* ``Rendezvous``: rank 0 serves an abstract-namespace Unix socket; every rank (rank 0 included) connects to it and
  all collectives are one primitive, ``allgather(obj)``: each rank sends one pickled object, the server answers
  with the list of all of them.  ``bcast`` and ``barrier`` are built on it.  Every wait is bounded: a peer that
  died or took another code path turns into a ``RendezvousError`` naming the missing ranks, not a hang.
* ``spawn_ranks``: start ``world`` copies of a script as child processes (RANK / LOCAL_RANK / WORLD_SIZE /
  SLAM_RDZV in their environment) from a parent that never touches the GPU, forward rank 0's stdout, return the
  worst exit code.
* ``from_env``: the identity of this process, whether it was started by ``spawn_ranks`` or by
  ``python -m torch.distributed.run`` (whose RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT variables are read; torch
  itself is not imported).
"""
from __future__ import annotations

import os
import pickle
import socket
import struct
import subprocess
import sys
import threading
import time
from typing import List, Optional, Sequence, Tuple


class RendezvousError(RuntimeError):
    pass


def _address(name: str) -> str:
    return "\0slamhip-" + name          # abstract namespace: no file to clean up, private to this network namespace


def _send(sock: socket.socket, obj) -> None:
    blob = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)      # our own objects between our own processes
    sock.sendall(struct.pack("<Q", len(blob)) + blob)


def _recv(sock: socket.socket, timeout: float):
    """One framed object, or EOFError if the peer closed, or socket.timeout."""
    sock.settimeout(timeout)
    buf = bytearray()
    need = 8
    header = True
    while True:
        while len(buf) < need:
            chunk = sock.recv(min(1 << 20, need - len(buf)))
            if not chunk:
                raise EOFError("peer closed the rendezvous connection")
            buf += chunk
        if header:
            need, header = struct.unpack("<Q", bytes(buf))[0], False
            buf = bytearray()
            if need == 0:
                return None
        else:
            return pickle.loads(bytes(buf))


class _Server(threading.Thread):
    """Rank 0's side: accept `world` connections, then serve rounds of all-gather until every rank has said goodbye."""

    def __init__(self, name: str, world: int, timeout: float):
        super().__init__(daemon=True)
        self.world, self.timeout = world, timeout
        self.sock = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        self.sock.bind(_address(name))
        self.sock.listen(world)
        self.error: Optional[str] = None

    def run(self) -> None:
        conns = {}
        try:
            self.sock.settimeout(self.timeout)
            while len(conns) < self.world:
                c, _ = self.sock.accept()
                conns[_recv(c, self.timeout)] = c
            while conns:
                items, gone = {}, []
                deadline = time.monotonic() + self.timeout
                for r, c in sorted(conns.items()):
                    try:
                        kind, payload = _recv(c, max(0.01, deadline - time.monotonic()))
                    except socket.timeout:
                        raise RendezvousError(f"rank {r} did not reach the collective within {self.timeout:.0f} s "
                                              f"(arrived: {sorted(items)})") from None
                    except EOFError:
                        raise RendezvousError(f"rank {r} went away (arrived: {sorted(items)})") from None
                    if kind == "bye":
                        gone.append(r)
                    else:
                        items[r] = payload
                if gone and items:
                    raise RendezvousError(f"ranks {gone} left while ranks {sorted(items)} were in a collective")
                for r in gone:
                    conns.pop(r).close()
                if items:
                    out = ("ok", [items[r] for r in sorted(items)])
                    for c in conns.values():
                        _send(c, out)
        except Exception as exc:   # noqa: BLE001 - tell whoever is still listening, then stop serving
            self.error = f"{type(exc).__name__}: {exc}"
            for c in conns.values():
                try:
                    _send(c, ("error", self.error))
                except OSError:
                    pass
        finally:
            for c in conns.values():
                c.close()
            self.sock.close()


class Rendezvous:
    """One per process.  ``name`` must be the same on all ranks of a run and unique among concurrent runs."""

    def __init__(self, rank: int, world: int, name: str, timeout: float = 180.0):
        self.rank, self.world, self.timeout = rank, world, timeout
        self._server = None
        self._conn: Optional[socket.socket] = None
        if world <= 1:
            return
        if rank == 0:
            self._server = _Server(name, world, timeout)
            self._server.start()
        deadline = time.monotonic() + timeout
        while True:
            c = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            try:
                c.connect(_address(name))
                self._conn = c
                break
            except (ConnectionRefusedError, FileNotFoundError):
                c.close()
                if time.monotonic() > deadline:
                    raise RendezvousError(f"rank {rank}: no rendezvous server '{name}' after {timeout:.0f} s") from None
                time.sleep(0.02)
        _send(self._conn, rank)

    def allgather(self, obj) -> list:
        """[obj of rank 0, obj of rank 1, ...] on every rank.  Every rank must make the same sequence of calls."""
        if self._conn is None:
            return [obj]
        try:
            _send(self._conn, ("item", obj))
            kind, payload = _recv(self._conn, self.timeout + 5.0)
        except socket.timeout:
            raise RendezvousError(f"rank {self.rank}: no answer from the rendezvous server within {self.timeout:.0f} s") from None
        except (EOFError, OSError) as exc:
            raise RendezvousError(f"rank {self.rank}: the rendezvous server went away ({exc})") from None
        if kind != "ok":
            raise RendezvousError(f"rank {self.rank}: {payload}")
        return payload

    def bcast(self, obj, src: int = 0):
        return self.allgather(obj if self.rank == src else None)[src]

    def barrier(self) -> None:
        self.allgather(None)

    def close(self) -> None:
        if self._conn is not None:
            try:
                _send(self._conn, ("bye", None))
            except OSError:
                pass
            self._conn.close()
            self._conn = None
        if self._server is not None:
            self._server.join(self.timeout)
            self._server = None


def from_env(env=os.environ) -> Tuple[int, int, int, Optional[str]]:
    """(rank, local_rank, world, rendezvous name) of this process; name is None when it is not part of a group."""
    world = int(env.get("WORLD_SIZE", "1"))
    rank, local = int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", env.get("RANK", "0")))
    if world <= 1:
        return 0, local, 1, None
    name = env.get("SLAM_RDZV")
    if not name:   # started by torch.distributed.run (the driver's launch line): derive a name all ranks agree on
        name = f"{env.get('MASTER_ADDR', '127.0.0.1')}-{env.get('MASTER_PORT', '0')}-{env.get('TORCHELASTIC_RUN_ID', 'none')}"
    return rank, local, world, name


def spawn_ranks(script: str, args: Sequence[str], world: int, env_extra: Optional[dict] = None,
                timeout: Optional[float] = None) -> int:
    """Run ``python script args...`` as ``world`` rank processes and wait for them.  The caller must not have
    touched the GPU (children are separate processes started with a fresh interpreter, not forks of GPU state)."""
    name = f"spawn-{os.getpid()}-{time.monotonic_ns()}"
    procs: List[subprocess.Popen] = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), SLAM_RDZV=name,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.update(env_extra or {})
        procs.append(subprocess.Popen([sys.executable, script, *args], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = b""
    deadline = None if timeout is None else time.monotonic() + timeout
    rc = 0
    try:
        out0, _ = procs[0].communicate(timeout=timeout)
        for p in procs:
            left = None if deadline is None else max(1.0, deadline - time.monotonic())
            rc = max(rc, abs(p.wait(timeout=left)))
    except subprocess.TimeoutExpired:
        rc = 124
    finally:
        for p in procs:                     # exactly the processes started here, never by pattern
            if p.poll() is None:
                p.kill()
                p.wait()
    sys.stdout.write(out0.decode("utf-8", "replace"))
    sys.stdout.flush()
    return rc
