"""Query-sharded multi-GPU matching: one process per GPU, RCCL all-gather of the per-shard top-2 rows.

The reference is single-process (``slam.py:22-35``); this is the sharding of
SURVEY.md §8e.  Every (query, train) pair is independent, so rank g searches
query rows ``[g*per, (g+1)*per)`` against the full (replicated) train set and
writes its rows straight into its slot of the gathered table; one
``ncclAllGather`` over xGMI then gives every rank the complete ``[N,2]``
result, bit-identical to the single-GPU run (no merge step is involved).

The rendezvous (who is rank 0, how the 128-byte RCCL id travels) is the
launcher's business: ``init_comm`` takes a ``bcast(bytes|None) -> bytes``
callable, e.g. built on ``torch.distributed`` (gloo) object broadcast.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Callable, Optional, Tuple

import numpy as np

from ._lib import COMM_ID_BYTES, DESC_BYTES, check
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


def init_comm(ctx: Context, rank: int, world: int, bcast: Callable[[Optional[bytes]], bytes]) -> None:
    """Create the RCCL communicator of ``ctx``; ``bcast`` ships rank 0's unique id to everybody."""
    ident, failure = b"", None
    if rank == 0:
        try:
            buf = ctypes.create_string_buffer(COMM_ID_BYTES)
            check(ctx.lib.slam_comm_unique_id(buf))
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
    check(ctx.lib.slam_comm_init(ctx.handle, world, rank, ctypes.create_string_buffer(bytes(ident), COMM_ID_BYTES)))


class ShardedMatcher:
    """knn=2 search of a query set sharded over ``world`` GPUs against a replicated train set."""

    def __init__(self, ctx: Context, rank: int, world: int, query, train):
        self.ctx, self.rank, self.world = ctx, rank, world
        query, train = as_descriptors(query), as_descriptors(train)
        self.plan = ShardPlan(query.shape[0], world)
        self.n_train = train.shape[0]
        a, b = self.plan.rows(rank)
        self.n_local = b - a
        self.d_query = DeviceDescriptors(ctx, query[a:b])
        self.d_train = DeviceDescriptors(ctx, train)       # replicated: 2 MiB at 64k rows
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
        self.passes = 0
        self.last = 0

    def step(self) -> None:
        """One pass: local search into this rank's slot, then the all-gather of that buffer (all asynchronous)."""
        ctx = self.ctx
        b = self.passes % len(self.gathered)
        if self.world > 1:
            check(ctx.lib.slam_comm_wait_buffer(ctx.handle, b))      # the gather that last used this buffer is done
        if self.n_local:
            knn2_device(ctx, self.d_query.buf, self.n_local, self.d_train.buf, self.n_train, self.my_idx[b],
                        self.my_dist[b])
        if self.world > 1:
            check(ctx.lib.slam_comm_allgather_overlapped(ctx.handle, self.my_idx[b].ptr, self.gathered[b].ptr,
                                                         self.slot_bytes, b))
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

    def free(self) -> None:
        self.ctx.sync()
        for g in self.gathered:
            g.free()
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
