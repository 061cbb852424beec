"""Query-sharded multi-GPU matching: one process per GPU, RCCL all-gather of the per-shard top-2 rows.

The reference is single-process (``slam.py:22-35``); this is the sharding of
SURVEY.md §8e.  Every (query, train) pair is independent, so rank g searches
query rows ``[g*per, (g+1)*per)`` against the full (replicated) train set and
writes its rows straight into its slot of the gathered table; one
``ncclAllGather`` over xGMI then gives every rank the complete ``[N,2]``
result, bit-identical to the single-GPU run (no merge step is involved).

The rendezvous (who is rank 0, how the 128-byte RCCL id travels) is the
launcher's business: ``init_comm`` takes a ``bcast(bytes|None) -> bytes``
callable, e.g. ``slamhip.launch.Rendezvous.bcast`` (standard library only).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Callable, Optional, Tuple

import numpy as np

from ._lib import COMM_ID_BYTES, DESC_BYTES, P2P_HANDLE_BYTES, check
from .device import Context, DeviceBuffer
from .matching import DeviceDescriptors, as_descriptors, knn2_device


@dataclass(frozen=True)
class ShardPlan:
    """Equal-sized query shards (RCCL all-gather needs equal counts; the tail shard is padded)."""

    n_query: int
    world: int

    @property
    def rows_per_rank(self) -> int:
        return (self.n_query + self.world - 1) // self.world if self.world > 0 else 0

    def rows(self, rank: int) -> Tuple[int, int]:
        """[start, stop) of the real query rows owned by ``rank`` (may be empty for tail ranks)."""
        per = self.rows_per_rank
        start = min(rank * per, self.n_query)
        return start, min(start + per, self.n_query)

    @property
    def padded_rows(self) -> int:
        return self.rows_per_rank * self.world


class DeadlineExceeded(RuntimeError):
    """A native call did not return in time; the thread that made it is still inside it."""


def call_with_deadline(fn: Callable[[], object], seconds: Optional[float], what: str):
    """``fn()`` - in a helper thread when ``seconds`` is given, so that a native call that never returns (a communicator
    bootstrap waiting for a rank that will not come, a collective on a fabric that does not answer) becomes a
    ``DeadlineExceeded`` instead of a hung process.  ctypes drops the GIL inside the call and every entry point of the
    library selects its device itself, so the calling thread does not matter.  After a ``DeadlineExceeded`` the helper thread
    is still inside the call: whatever it holds must be abandoned, not reused or destroyed, and the process should leave
    with ``os._exit`` once it has said what happened."""
    if seconds is None:
        return fn()
    import threading

    box = {}

    def run():
        try:
            box["value"] = fn()
        except BaseException as exc:   # noqa: BLE001 - handed to the caller below
            box["error"] = exc

    t = threading.Thread(target=run, name=f"deadline:{what}", daemon=True)
    t.start()
    t.join(seconds)
    if t.is_alive():
        raise DeadlineExceeded(f"{what} did not return within {seconds:.0f} s")
    if "error" in box:
        raise box["error"]
    return box.get("value")


def init_comm(ctx: Context, rank: int, world: int, bcast: Callable[[Optional[bytes]], bytes],
              deadline: Optional[float] = None) -> None:
    """Create the RCCL communicator of ``ctx``; ``bcast`` ships rank 0's unique id to everybody.  ``deadline``: seconds
    each of the two native calls may take (``call_with_deadline``); the ``bcast`` between them runs on the calling thread,
    so the launcher's socket is never used from two threads."""
    ident, failure = b"", None
    if rank == 0:
        try:
            buf = ctypes.create_string_buffer(COMM_ID_BYTES)
            call_with_deadline(lambda: check(ctx.lib.slam_comm_unique_id(buf)), deadline, "ncclGetUniqueId")
            ident = buf.raw
        except Exception as exc:   # noqa: BLE001 - the other ranks are waiting in bcast: tell them instead of hanging them
            failure = exc
    ident = bcast(ident)
    if failure is not None:
        raise failure
    if not ident:
        raise RuntimeError("rank 0 could not create an RCCL unique id")
    if not isinstance(ident, (bytes, bytearray)) or len(ident) != COMM_ID_BYTES:
        raise ValueError("bcast must return the 128-byte id produced on rank 0")
    ident_buf = ctypes.create_string_buffer(bytes(ident), COMM_ID_BYTES)
    call_with_deadline(lambda: check(ctx.lib.slam_comm_init(ctx.handle, world, rank, ident_buf)), deadline, "ncclCommInitRank")


class PeerMap:
    """The peers' gathered buffers mapped into this process through HIP IPC (``slam_p2p_*``): the direct
    all-gather over xGMI used when no RCCL communicator can be created."""

    def __init__(self, ctx: Context, rank: int, world: int, buffers, allgather_obj: Callable[[object], list]):
        """``buffers``: this rank's gathered buffers (base allocations); ``allgather_obj(x) -> [x of rank 0, ...]``
        is the launcher's object all-gather (e.g. ``slamhip.launch.Rendezvous.allgather``)."""
        self.ctx, self.rank, self.world = ctx, rank, world
        self.ptrs, self.tables = [], []     # [buffer][rank] -> device pointer (0 for this rank); and as ctypes arrays
        mine, failure = [], None
        try:
            for b in buffers:
                h = ctypes.create_string_buffer(P2P_HANDLE_BYTES)
                check(ctx.lib.slam_p2p_export(ctx.handle, b.ptr, h))
                mine.append(h.raw)
        except Exception as exc:   # noqa: BLE001 - still take part in the exchange below, or the peers would hang
            mine, failure = None, exc
        everyone = allgather_obj(mine)
        if failure is not None:
            raise failure
        if any(x is None for x in everyone):
            raise RuntimeError("a peer could not export its gathered buffers")
        try:
            for bi in range(len(buffers)):
                row = []
                self.ptrs.append(row)
                for r in range(world):
                    if r == rank:
                        row.append(0)
                        continue
                    p = ctypes.c_void_p()
                    check(ctx.lib.slam_p2p_open(ctx.handle, ctypes.create_string_buffer(everyone[r][bi], P2P_HANDLE_BYTES),
                                                ctypes.byref(p)))
                    row.append(p.value or 0)
                self.tables.append((ctypes.c_void_p * world)(*row))
        except Exception:
            self.close()
            raise

    def push(self, send: DeviceBuffer, bytes_per_rank: int, buffer_id: int) -> None:
        c = self.ctx
        check(c.lib.slam_p2p_allgather_overlapped(c.handle, send.ptr, bytes_per_rank, self.rank, self.tables[buffer_id],
                                                  self.world, buffer_id))

    def close(self) -> None:
        for row in self.ptrs:
            for p in row:
                if p:
                    check(self.ctx.lib.slam_p2p_close(self.ctx.handle, p))
        self.ptrs, self.tables = [], []


class _Gathering:
    """What both sharded matchers share: gathered buffers with one slot per rank, and the peer-copy tier.
    Subclasses set ``ctx, rank, world, gathered`` (device buffers), ``_slots`` (this rank's slot in each of them),
    ``slot_bytes``, ``collective`` and ``peers``."""

    def _gather(self, b: int) -> None:
        """All-gather this rank's slot of gathered buffer ``b`` (asynchronous, second stream)."""
        ctx = self.ctx
        if self.collective == "rccl":
            check(ctx.lib.slam_comm_allgather_overlapped(ctx.handle, self._slots[b].ptr, self.gathered[b].ptr,
                                                         self.slot_bytes, b))
        elif self.collective == "p2p":
            self.peers.push(self._slots[b], self.slot_bytes, b)

    def enable_p2p(self, allgather_obj: Callable[[object], list], barrier: Callable[[], None]) -> bool:
        """Map the peers' gathered buffers and prove the mapping with a round of marker writes; True on every rank
        or False on every rank (``allgather_obj`` carries the verdicts).  On success ``collective`` becomes "p2p";
        ``result()`` is then only complete after the launcher's barrier (all ranks synced, then a process barrier)."""
        ok = True
        try:
            self.peers = PeerMap(self.ctx, self.rank, self.world, self.gathered, allgather_obj)
        except Exception as exc:   # noqa: BLE001 - any rank may fail to map; the verdict is agreed below
            self.peers, ok = None, False
            self.p2p_error = str(exc)
        verdicts = allgather_obj(ok)
        if not all(verdicts):
            if self.peers is not None:
                self.peers.close()
                self.peers = None
            return False
        # marker round: every rank writes rank+1 into the first bytes of its slot everywhere, then checks all slots
        nmark = min(16, self.slot_bytes)                         # a train-sharded slot is 8 bytes when there is one query
        mark = np.full(nmark, self.rank + 1, np.uint8)
        for b in range(len(self.gathered)):
            self._slots[b].upload(mark)
            self.peers.push(self._slots[b], self.slot_bytes, b)    # the whole slot: the offset is rank * slot_bytes
        barrier()
        good = True
        for g in self.gathered:
            raw = g.download(np.uint8, (self.world, self.slot_bytes))
            good = good and all((raw[r, :nmark] == r + 1).all() for r in range(self.world))
        for g in self.gathered:                                  # back to "no match" everywhere before real passes
            check(self.ctx.lib.slam_memset(self.ctx.handle, g.ptr, 0xFF, g.nbytes))
        self.ctx.sync()
        verdicts = allgather_obj(bool(good))
        barrier()
        if not all(verdicts):
            self.peers.close()
            self.peers = None
            return False
        self.collective = "p2p"
        return True


    def close_peers(self) -> None:
        """Unmap the peers' buffers (first step of the teardown of the peer-copy tier)."""
        if self.peers is not None:
            self.peers.close()
            self.peers = None

    def _release(self, barrier: Optional[Callable[[], None]], buffers) -> None:
        """Teardown in the order HIP IPC requires: every rank unmaps what it imported, then ALL ranks meet at the
        launcher's barrier, and only then does anybody free the allocations it exported - freeing an exported
        buffer while a peer still has it mapped (or has a copy into it in flight) is undefined behaviour."""
        self.ctx.sync()
        if self.peers is not None:
            if barrier is None:
                raise RuntimeError("this matcher's buffers are mapped by peer processes: call free(barrier) with the "
                                   "launcher's barrier (all ranks synced, then a process barrier)")
            self.close_peers()
            barrier()
        for b in buffers:
            b.free()


class ShardedMatcher(_Gathering):
    """knn=2 search of a query set sharded over ``world`` GPUs against a replicated train set.

    ``collective``: "rccl" (communicator made by ``init_comm``), "p2p" (peer copies over HIP IPC, enabled with
    ``enable_p2p``) or None (no gather: every rank keeps only its own slot)."""

    def __init__(self, ctx: Context, rank: int, world: int, query, train, collective: Optional[str] = "rccl",
                 broadcast_train: bool = False, image_rows=None):
        """``broadcast_train``: with an RCCL communicator, only rank 0 uploads the train rows and the other ranks
        receive them over the fabric (``slam_comm_broadcast``: 32 MiB for the 512 x 2048 loop-closure set) instead of
        every rank pushing its own copy through PCIe.  ``image_rows``: the train set is a collection of images
        (``BFMatcher.add``), ``image_rows[i]`` rows each; ``result_images()`` then reports (imgIdx, trainIdx, dist)."""
        self.ctx, self.rank, self.world = ctx, rank, world
        self.collective = collective if world > 1 else None
        self.peers: Optional[PeerMap] = None
        query, train = as_descriptors(query), as_descriptors(train)
        self.plan = ShardPlan(query.shape[0], world)
        self.n_train = train.shape[0]
        a, b = self.plan.rows(rank)
        self.n_local = b - a
        self.d_query = DeviceDescriptors(ctx, query[a:b])
        self.train_replication = "per-rank upload"
        if broadcast_train and self.collective == "rccl" and self.n_train:
            self.d_train = DeviceDescriptors(ctx, train if rank == 0 else None, rows=self.n_train)
            check(ctx.lib.slam_comm_broadcast(ctx.handle, self.d_train.buf.ptr, self.n_train * DESC_BYTES, 0))
            self.train_replication = "rccl broadcast from rank 0"
        else:
            self.d_train = DeviceDescriptors(ctx, train)   # replicated: 2 MiB at 64k rows
        self.image_rows = None
        self._img_out = None
        if image_rows is not None:
            rows = [int(r) for r in image_rows]
            if sum(rows) != self.n_train or any(r < 0 or r >= (1 << 18) for r in rows) or not 1 <= len(rows) <= 8191:
                raise ValueError("image_rows must cover the train rows: 1..8191 images of fewer than 2^18 rows each")
            self.image_rows = rows
            self._offsets = ctx.upload(np.concatenate([[0], np.cumsum(rows)]).astype(np.int32))
        per = max(self.plan.rows_per_rank, 1)
        self.per = per
        # A gathered buffer has one slot per rank; a slot = that rank's idx rows [per,2] followed by its
        # dist rows [per,2] (int32), so ONE all-gather per pass moves both tables (16*per bytes per rank:
        # 128 KiB at 64k queries / 8 GPUs - latency-bound, so fewer collectives matter more than bytes).
        # Two such buffers alternate between passes: the gather of pass i runs on the ctx's second stream
        # while the search of pass i+1 fills the other buffer.
        self.slot_bytes = per * 16
        self.gathered = [ctx.malloc(self.slot_bytes * world) for _ in range(2 if world > 1 else 1)]
        off = rank * self.slot_bytes
        self.my_idx, self.my_dist = [], []
        for g in self.gathered:
            # padded tail rows of short shards must read as "no match" after the gather, not garbage
            check(ctx.lib.slam_memset(ctx.handle, g.ptr, 0xFF, g.nbytes))
            self.my_idx.append(g.view(off, per * 8))
            self.my_dist.append(g.view(off + per * 8, per * 8))
        self._slots = self.my_idx            # a slot starts with its idx rows
        self.passes = 0
        self.last = 0

    def step(self) -> None:
        """One pass: local search into this rank's slot, then the all-gather of that buffer (all asynchronous)."""
        ctx = self.ctx
        b = self.passes % len(self.gathered)
        if self.collective:
            check(ctx.lib.slam_comm_wait_buffer(ctx.handle, b))      # the gather that last used this buffer is done
        if self.n_local:
            knn2_device(ctx, self.d_query.buf, self.n_local, self.d_train.buf, self.n_train, self.my_idx[b],
                        self.my_dist[b])
        self._gather(b)
        self.last = b
        self.passes += 1

    def result(self) -> Tuple[np.ndarray, np.ndarray]:
        """The complete [N,2] tables of the most recent pass (waits for both streams)."""
        n, per = self.plan.n_query, self.per
        if self.plan.padded_rows == 0:
            return np.zeros((0, 2), np.int32), np.zeros((0, 2), np.int32)
        self.ctx.sync()
        raw = self.gathered[self.last].download(np.int32, (self.world, 2, per, 2))   # [rank][idx|dist][row][k]
        idx = raw[:, 0].reshape(self.world * per, 2)[:n]
        dist = raw[:, 1].reshape(self.world * per, 2)[:n]
        return np.ascontiguousarray(idx), np.ascontiguousarray(dist)

    def decode_images(self) -> None:
        """Turn the global train rows of the most recent pass's gathered table into (imgIdx, trainIdx) on the device
        (``slam_bf_split_index``; asynchronous).  With peer copies call it after the launcher's barrier."""
        if self.image_rows is None:
            raise ValueError("this matcher was built without image_rows")
        ctx, per = self.ctx, self.per
        if self._img_out is None:
            self._img_out = (ctx.malloc(self.world * per * 8), ctx.malloc(self.world * per * 8))
        if self.collective:
            check(ctx.lib.slam_comm_wait_buffer(ctx.handle, self.last))   # main stream: behind the gather of this buffer
        g = self.gathered[self.last]
        for r in range(self.world):                                       # a slot = idx rows [per,2], then dist rows [per,2]
            check(ctx.lib.slam_bf_split_index(ctx.handle, g.ptr + r * self.slot_bytes, per * 2, self._offsets.ptr,
                                              len(self.image_rows), self._img_out[0].ptr + r * per * 8,
                                              self._img_out[1].ptr + r * per * 8))

    def result_images(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(imgIdx, trainIdx, dist) int32 [N,2] of the most recent pass, in OpenCV's multi-image form."""
        n, per = self.plan.n_query, self.per
        self.decode_images()
        _, dist = self.result()
        img = self._img_out[0].download(np.int32, (self.world * per, 2))[:n]
        local = self._img_out[1].download(np.int32, (self.world * per, 2))[:n]
        return np.ascontiguousarray(img), np.ascontiguousarray(local), dist

    def free(self, barrier: Optional[Callable[[], None]] = None) -> None:
        """``barrier`` is required when the peer-copy tier is active (see ``_release``)."""
        self._release(barrier, self.gathered)
        self.d_query.free()
        self.d_train.free()
        if self.image_rows is not None:
            self._offsets.free()
        if self._img_out is not None:
            for b in self._img_out:
                b.free()
            self._img_out = None


class TrainShardedMatcher(_Gathering):
    """The other partitioning of SURVEY.md §8e, for train sets much larger than the query set (or too large for one
    GPU): rank g searches ALL query rows against its own slice of the train rows, reporting global train indices
    (``train_base``); the per-rank top-2 tables are all-gathered (8*N bytes per rank and table) and every rank merges
    them by (distance, index) with ``slam_bf_merge_top2`` - the order that keeps OpenCV's lowest-index tie rule
    across shards, so the result is bit-identical to the single-GPU search."""

    def __init__(self, ctx: Context, rank: int, world: int, query, train_shard, train_base: int,
                 collective: Optional[str] = "rccl"):
        self.ctx, self.rank, self.world = ctx, rank, world
        self.collective = collective if world > 1 else None
        self.peers: Optional[PeerMap] = None
        query, train_shard = as_descriptors(query), as_descriptors(train_shard)
        self.n, self.m_local, self.train_base = query.shape[0], train_shard.shape[0], int(train_base)
        self.d_query = DeviceDescriptors(ctx, query)
        self.d_train = DeviceDescriptors(ctx, train_shard)
        self.slot_bytes = max(self.n, 1) * 8
        self.gathered = [ctx.malloc(self.slot_bytes * world) for _ in range(2)]   # [0]: idx tables, [1]: dist tables
        for g in self.gathered:
            check(ctx.lib.slam_memset(ctx.handle, g.ptr, 0xFF, g.nbytes))         # idx -1 = "no neighbour" until filled
        self._slots = [g.view(rank * self.slot_bytes, self.slot_bytes) for g in self.gathered]
        self.out_idx, self.out_dist = ctx.malloc(self.slot_bytes), ctx.malloc(self.slot_bytes)

    def step(self) -> None:
        """Local search over this rank's train slice, then the gathers of both tables (all asynchronous)."""
        ctx = self.ctx
        if self.collective:
            for b in (0, 1):
                check(ctx.lib.slam_comm_wait_buffer(ctx.handle, b))
        if self.n:
            knn2_device(ctx, self.d_query.buf, self.n, self.d_train.buf, self.m_local, self._slots[0], self._slots[1],
                        self.train_base)
        for b in (0, 1):
            self._gather(b)

    def result(self) -> Tuple[np.ndarray, np.ndarray]:
        """Merged [N,2] tables.  With peer copies call this after the launcher's barrier (see ``enable_p2p``)."""
        if self.n == 0:
            return np.zeros((0, 2), np.int32), np.zeros((0, 2), np.int32)
        c = self.ctx
        c.sync()
        check(c.lib.slam_bf_merge_top2(c.handle, self.gathered[0].ptr, self.gathered[1].ptr, self.world, self.n,
                                       self.out_idx.ptr, self.out_dist.ptr))
        return self.out_idx.download(np.int32, (self.n, 2)), self.out_dist.download(np.int32, (self.n, 2))

    def free(self, barrier: Optional[Callable[[], None]] = None) -> None:
        """``barrier`` is required when the peer-copy tier is active (see ``_release``)."""
        self._release(barrier, (*self.gathered, self.out_idx, self.out_dist))
        self.d_query.free()
        self.d_train.free()


def gather_rows_host(plan: ShardPlan, shards) -> np.ndarray:
    """Host-side statement of what the all-gather produces: concatenate equal-sized (padded) shards, trim."""
    per = plan.rows_per_rank
    out = []
    for g, s in enumerate(shards):
        s = np.asarray(s)
        pad = per - s.shape[0]
        if pad:
            s = np.concatenate([s, np.full((pad,) + s.shape[1:], -1, s.dtype)], 0)
        out.append(s)
    return np.concatenate(out, 0)[: plan.n_query] if out else np.zeros((0, 2), np.int32)
