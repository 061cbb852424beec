"""Process plumbing for multi-GPU runs, standard library only (no torch, no MPI).

The reference is a single process (``slam.py:22-35``); the sharded searches of ``slamhip.dist`` need three things
from a launcher: the ranks' identities, a way to ship small objects between them (RCCL's 128-byte id, HIP IPC
handles, verdicts, timings) and a barrier.  This module provides them:

* ``Rendezvous``: rank 0 serves an abstract-namespace Unix socket; every rank (rank 0 included) connects to it and
  all collectives are one primitive, ``allgather(obj)``: each rank sends one object, the server answers with the
  list of all of them.  ``bcast`` and ``barrier`` are built on it.  Every wait is bounded: a peer that died or took
  another code path turns into a ``RendezvousError`` naming the missing ranks, not a hang.
  An abstract socket has no file permissions, so the server trusts nobody by default: a peer must run under the
  same uid (``SO_PEERCRED``), open with a fixed-layout hello (magic, world, rank, a 32-byte token) and present the
  run's random token (``spawn_ranks`` hands it down in the environment; under ``torch.distributed.run`` rank 0
  leaves it in a 0600 file that only the same user can read); ranks outside ``[0, world)`` and duplicates are
  turned away.  Clients check the server's uid the same way.  Nothing on the wire is ever unpickled: objects
  travel in a small tagged encoding (None / bool / int / float / str / bytes / list / tuple / dict / numpy array).
* ``spawn_ranks``: start ``world`` copies of a script as child processes (RANK / LOCAL_RANK / WORLD_SIZE /
  SLAM_RDZV in their environment) from a parent that never touches the GPU, forward rank 0's stdout, return the
  worst exit code.
* ``from_env``: the identity of this process, whether it was started by ``spawn_ranks`` or by
  ``python -m torch.distributed.run`` (whose RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT variables are read; torch
  itself is not imported).
"""
from __future__ import annotations

import hmac
import os
import secrets
import socket
import stat
import struct
import subprocess
import sys
import tempfile
import threading
import time
from typing import List, Optional, Sequence, Tuple


_SLOT = struct.Struct("<q")


class RendezvousError(RuntimeError):
    pass


def _address(name: str) -> str:
    return "\0slamhip-" + name          # abstract namespace: no file to clean up, private to this network namespace


_MAGIC = b"SLRZ"
_HELLO = struct.Struct("<4sIi32s")            # magic, world, rank, token
_MAX_FRAME = 1 << 31


def _enc(obj, out: bytearray) -> None:
    """Tagged binary encoding of the plain objects the ranks exchange (no code travels, nothing is unpickled)."""
    import numpy as np

    if obj is None:
        out += b"N"
    elif obj is True or obj is False:
        out += b"T" if obj else b"F"
    elif isinstance(obj, (int, np.integer)):
        out += b"i" + struct.pack("<q", int(obj))
    elif isinstance(obj, (float, np.floating)):
        out += b"d" + struct.pack("<d", float(obj))
    elif isinstance(obj, str):
        b = obj.encode("utf-8")
        out += b"s" + struct.pack("<Q", len(b)) + b
    elif isinstance(obj, (bytes, bytearray, memoryview)):
        b = bytes(obj)
        out += b"b" + struct.pack("<Q", len(b)) + b
    elif isinstance(obj, (list, tuple)):
        out += (b"l" if isinstance(obj, list) else b"t") + struct.pack("<Q", len(obj))
        for x in obj:
            _enc(x, out)
    elif isinstance(obj, dict):
        out += b"m" + struct.pack("<Q", len(obj))
        for k, v in obj.items():
            _enc(k, out)
            _enc(v, out)
    else:
        raise TypeError(f"{type(obj).__name__} cannot cross the rendezvous")


def _dec(buf: memoryview, pos: int):
    tag = bytes(buf[pos:pos + 1])
    pos += 1
    if tag == b"N":
        return None, pos
    if tag in (b"T", b"F"):
        return tag == b"T", pos
    if tag == b"i":
        return struct.unpack_from("<q", buf, pos)[0], pos + 8
    if tag == b"d":
        return struct.unpack_from("<d", buf, pos)[0], pos + 8
    if tag in (b"s", b"b"):
        n = struct.unpack_from("<Q", buf, pos)[0]
        pos += 8
        if n > len(buf) - pos:
            raise ValueError("truncated rendezvous frame")
        raw = bytes(buf[pos:pos + n])
        return (raw.decode("utf-8") if tag == b"s" else raw), pos + n
    if tag in (b"l", b"t"):
        n = struct.unpack_from("<Q", buf, pos)[0]
        pos += 8
        items = []
        for _ in range(n):
            if pos >= len(buf):
                raise ValueError("truncated rendezvous frame")
            x, pos = _dec(buf, pos)
            items.append(x)
        return (items if tag == b"l" else tuple(items)), pos
    if tag == b"m":
        n = struct.unpack_from("<Q", buf, pos)[0]
        pos += 8
        d = {}
        for _ in range(n):
            k, pos = _dec(buf, pos)
            v, pos = _dec(buf, pos)
            d[k] = v
        return d, pos
    raise ValueError(f"unknown tag {tag!r} in a rendezvous frame")


def dumps(obj) -> bytes:
    """Encode ``obj`` for the wire.  numpy arrays are wrapped as {"__nd__": [dtype, shape, bytes]}."""
    import numpy as np

    def wrap(x):
        if isinstance(x, np.ndarray):
            a = np.ascontiguousarray(x)
            if a.dtype.hasobject:
                raise TypeError("object arrays cannot cross the rendezvous")
            return {"__nd__": [a.dtype.str, list(a.shape), a.tobytes()]}
        if isinstance(x, list):
            return [wrap(v) for v in x]
        if isinstance(x, tuple):
            return tuple(wrap(v) for v in x)
        if isinstance(x, dict):
            return {k: wrap(v) for k, v in x.items()}
        return x

    out = bytearray()
    _enc(wrap(obj), out)
    return bytes(out)


def loads(blob: bytes):
    import numpy as np

    def unwrap(x):
        if isinstance(x, dict):
            if set(x) == {"__nd__"}:
                dt, shape, raw = x["__nd__"]
                dtype = np.dtype(dt)
                if dtype.hasobject:
                    raise ValueError("object arrays cannot cross the rendezvous")
                return np.frombuffer(raw, dtype=dtype).reshape(shape).copy()
            return {k: unwrap(v) for k, v in x.items()}
        if isinstance(x, list):
            return [unwrap(v) for v in x]
        if isinstance(x, tuple):
            return tuple(unwrap(v) for v in x)
        return x

    obj, pos = _dec(memoryview(blob), 0)
    if pos != len(blob):
        raise ValueError("trailing bytes in a rendezvous frame")
    return unwrap(obj)


def _send(sock: socket.socket, obj) -> None:
    blob = dumps(obj)
    sock.sendall(struct.pack("<Q", len(blob)) + blob)


def _recv_exact(sock: socket.socket, need: int) -> bytes:
    buf = bytearray()
    while len(buf) < need:
        chunk = sock.recv(min(1 << 20, need - len(buf)))
        if not chunk:
            raise EOFError("peer closed the rendezvous connection")
        buf += chunk
    return bytes(buf)


def _recv(sock: socket.socket, timeout: float):
    """One framed object, or EOFError if the peer closed, or socket.timeout."""
    sock.settimeout(timeout)
    need = struct.unpack("<Q", _recv_exact(sock, 8))[0]
    if need > _MAX_FRAME:
        raise ValueError(f"rendezvous frame of {need} bytes refused")
    return loads(_recv_exact(sock, need))


def _peer_uid(sock: socket.socket) -> int:
    """uid of the process at the other end of a Unix socket (SO_PEERCRED: pid, uid, gid)."""
    cred = sock.getsockopt(socket.SOL_SOCKET, socket.SO_PEERCRED, struct.calcsize("3i"))
    return struct.unpack("3i", cred)[1]


def _token_path(name: str) -> str:
    safe = "".join(ch if ch.isalnum() or ch in "-_." else "_" for ch in name)
    return os.path.join(os.environ.get("XDG_RUNTIME_DIR") or tempfile.gettempdir(), f"slamhip-rdzv-{os.getuid()}-{safe}.token")


def _publish_token(name: str) -> Tuple[bytes, str]:
    """Rank 0 under an external launcher: a fresh random token in a file only this user can read."""
    path = _token_path(name)
    try:
        os.unlink(path)                      # a stale file of an earlier run of ours (someone else's cannot be removed
    except FileNotFoundError:                # from a sticky directory, and then O_EXCL below fails loudly)
        pass
    token = secrets.token_bytes(32)
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
    with os.fdopen(fd, "wb") as f:
        f.write(token)
    return token, path


def _read_token(name: str, timeout: float) -> bytes:
    path = _token_path(name)
    deadline = time.monotonic() + timeout
    while True:
        try:
            fd = os.open(path, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
            try:
                st = os.fstat(fd)
                if st.st_uid != os.getuid() or stat.S_IMODE(st.st_mode) & 0o077 or not stat.S_ISREG(st.st_mode):
                    raise RendezvousError(f"{path} is not a private file of uid {os.getuid()}: refusing its token")
                tok = os.read(fd, 64)
            finally:
                os.close(fd)
            if len(tok) == 32:
                return tok
        except FileNotFoundError:
            pass
        if time.monotonic() > deadline:
            raise RendezvousError(f"no rendezvous token at {path} after {timeout:.0f} s")
        time.sleep(0.02)


class _Server(threading.Thread):
    """Rank 0's side: accept `world` authenticated connections, then serve rounds of all-gather until every rank has
    said goodbye.  A connection that fails the checks (foreign uid, wrong magic / world / token, rank out of range or
    already taken) is closed and ignored: it cannot join, and it cannot stop the real ranks from joining."""

    def __init__(self, name: str, world: int, timeout: float, token: bytes):
        super().__init__(daemon=True)
        self.world, self.timeout, self.token = world, timeout, token
        self.sock = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        try:
            self.sock.bind(_address(name))
        except OSError as exc:
            self.sock.close()
            raise RendezvousError(f"rendezvous name '{name}' is already taken on this host ({exc})") from None
        self.sock.listen(max(world, 8))
        self.error: Optional[str] = None
        self.rejected: List[str] = []

    def _admit(self, c: socket.socket, conns: dict) -> None:
        try:
            if _peer_uid(c) != os.getuid():
                raise ValueError(f"peer runs under uid {_peer_uid(c)}")
            c.settimeout(min(self.timeout, 10.0))
            magic, world, rank, token = _HELLO.unpack(_recv_exact(c, _HELLO.size))
            if magic != _MAGIC or world != self.world:
                raise ValueError("not a hello of this run")
            if not hmac.compare_digest(token, self.token):
                raise ValueError("wrong token")
            if not 0 <= rank < self.world or rank in conns:
                raise ValueError(f"rank {rank} is out of range or already connected")
            c.sendall(_MAGIC)                         # admitted
            conns[rank] = c
        except (ValueError, OSError, EOFError, struct.error) as exc:
            self.rejected.append(str(exc))
            c.close()

    def run(self) -> None:
        conns = {}
        try:
            deadline = time.monotonic() + self.timeout
            while len(conns) < self.world:
                self.sock.settimeout(max(0.01, deadline - time.monotonic()))
                try:
                    c, _ = self.sock.accept()
                except socket.timeout:
                    missing = sorted(set(range(self.world)) - set(conns))
                    raise RendezvousError(f"ranks {missing} did not join within {self.timeout:.0f} s"
                                          + (f" (turned away: {self.rejected})" if self.rejected else "")) from None
                self._admit(c, conns)
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

    def __init__(self, rank: int, world: int, name: str, timeout: float = 180.0, token: Optional[bytes] = None):
        self.rank, self.world, self.timeout = rank, world, timeout
        self._server = None
        self._conn: Optional[socket.socket] = None
        self._token_file: Optional[str] = None
        if world <= 1:
            return
        if not 0 <= rank < world:
            raise RendezvousError(f"rank {rank} outside [0, {world})")
        # the run's shared secret: handed down by spawn_ranks (SLAM_RDZV_TOKEN), given by the caller, or - under an
        # external launcher such as torch.distributed.run - published by rank 0 in a file private to this user
        if token is None and os.environ.get("SLAM_RDZV_TOKEN"):
            token = bytes.fromhex(os.environ["SLAM_RDZV_TOKEN"])
        from_file = False
        if token is None:
            if rank == 0:
                token, self._token_file = _publish_token(name)
            else:
                token, from_file = _read_token(name, timeout), True
        if len(token) != 32:
            raise RendezvousError("the rendezvous token must be 32 bytes")
        if rank == 0:
            self._server = _Server(name, world, timeout, token)
            self._server.start()
        deadline = time.monotonic() + timeout
        while True:
            c = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            try:
                c.connect(_address(name))
            except (ConnectionRefusedError, FileNotFoundError):
                c.close()
                if time.monotonic() > deadline:
                    raise RendezvousError(f"rank {rank}: no rendezvous server '{name}' after {timeout:.0f} s") from None
                time.sleep(0.02)
                continue
            if _peer_uid(c) != os.getuid():
                c.close()
                raise RendezvousError(f"rank {rank}: the rendezvous server '{name}' belongs to another user")
            try:
                c.sendall(_HELLO.pack(_MAGIC, world, rank, token))
                c.settimeout(max(0.05, deadline - time.monotonic()))
                if _recv_exact(c, 4) != _MAGIC:
                    raise EOFError("bad acknowledgement")
                self._conn = c
                return
            except (EOFError, OSError) as exc:
                c.close()
                # A token read from the file may be a leftover of an earlier run that crashed under the same name (rank 0
                # replaces the file when it starts, possibly after this rank looked): read it again and retry until the
                # deadline.  A token that was handed down (environment, argument) is final.
                if not from_file or time.monotonic() > deadline:
                    raise RendezvousError(f"rank {rank}: the rendezvous server turned this rank away ({exc})") from None
                time.sleep(0.05)
                token = _read_token(name, max(0.05, deadline - time.monotonic()))

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

    # ---- a barrier for timing brackets ---------------------------------------------------------------------------------
    # barrier() is a round trip through rank 0's socket per rank, served one after the other: 0.17 ms with two ranks,
    # 0.63 ms with eight (measured, 8 host cores).  Inside a timed bracket of twenty 0.16 ms steps that is 16 % of the
    # measurement.  spin_barrier() keeps one 64-byte slot per rank in a shared-memory file: a rank writes its barrier
    # count into its own slot and polls the others' - a few microseconds.  The file is created by rank 0 (0600, random
    # name, O_EXCL), its path travels over the authenticated socket, every rank checks owner and mode before mapping, and
    # rank 0 unlinks it once everybody has it mapped.  A rank that never arrives is a RendezvousError after `timeout`
    # seconds, naming it; where /dev/shm cannot be used every rank falls back to barrier().
    def _spin_setup(self) -> None:
        import mmap
        import secrets
        import stat

        path = None
        if self.rank == 0:
            try:
                path = f"/dev/shm/slamhip-bar-{os.getpid()}-{secrets.token_hex(8)}"
                fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)
                os.ftruncate(fd, 64 * self.world)
                os.close(fd)
            except OSError:
                path = None
        path = self.bcast(path)
        mm = None
        if path is not None:
            try:
                fd = os.open(path, os.O_RDWR | getattr(os, "O_NOFOLLOW", 0))
                try:
                    st = os.fstat(fd)
                    if st.st_uid != os.getuid() or not stat.S_ISREG(st.st_mode) or st.st_mode & 0o077 or st.st_size != 64 * self.world:
                        raise OSError("not this run's barrier file")
                    mm = mmap.mmap(fd, 64 * self.world)
                finally:
                    os.close(fd)
            except (OSError, ValueError):
                mm = None
        ok = all(self.allgather(mm is not None))
        if self.rank == 0 and path is not None:
            try:
                os.unlink(path)                     # the mappings stay valid
            except OSError:
                pass
        if not ok and mm is not None:
            mm.close()
            mm = None
        self._spin_mm, self._spin_gen = mm, 0
        self._spin_ready = True

    def spin_barrier(self) -> None:
        """Low-latency barrier over shared memory (see above); every rank must call it in the same sequence."""
        if self._conn is None:
            return
        if not getattr(self, "_spin_ready", False):
            self._spin_setup()
        mm = self._spin_mm
        if mm is None:
            return self.barrier()
        self._spin_gen += 1
        gen = self._spin_gen
        _SLOT.pack_into(mm, 64 * self.rank, gen)
        start = time.monotonic()
        for r in range(self.world):
            spins = 0
            while _SLOT.unpack_from(mm, 64 * r)[0] < gen:
                spins += 1
                if spins & 0x3FF == 0:
                    waited = time.monotonic() - start
                    if waited > self.timeout:
                        missing = [q for q in range(self.world) if _SLOT.unpack_from(mm, 64 * q)[0] < gen]
                        raise RendezvousError(f"rank {self.rank}: ranks {missing} did not reach the barrier within {self.timeout:.0f} s")
                    if waited > 0.002:
                        time.sleep(0.0001)          # somebody is far behind: stop burning a core

    def close(self) -> None:
        mm = getattr(self, "_spin_mm", None)
        if mm is not None:
            mm.close()
            self._spin_mm = None
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
        if self._token_file:
            try:
                os.unlink(self._token_file)
            except OSError:
                pass
            self._token_file = None


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
    token = secrets.token_hex(32)           # the run's shared secret: only these children get it
    procs: List[subprocess.Popen] = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), SLAM_RDZV=name, SLAM_RDZV_TOKEN=token,
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
