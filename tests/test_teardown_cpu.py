"""CPU suite: teardown order of the peer-copy (HIP IPC) tier of slamhip.dist, on a recording stand-in for the library.

HIP leaves freeing an exported allocation undefined while a peer still has it mapped, so a sharded matcher must
unmap what it imported, meet all ranks at the launcher's barrier, and only then free what it exported."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))


class _RecorderLib:
    """Every libslamhip call succeeds and is written down."""

    def __init__(self, log):
        self._log = log

    def __getattr__(self, name):
        def call(*args):
            self._log.append(name)
            return 0
        return call


class _Buf:
    def __init__(self, log, nbytes, name="buf"):
        self._log, self.nbytes, self.ptr, self.name = log, nbytes, 0x1000, name

    def view(self, off, n=None):
        return _Buf(self._log, self.nbytes - off if n is None else n, self.name + "-view")

    def upload(self, host):
        return self

    def download(self, dtype, shape):
        return np.zeros(shape, dtype)

    def free(self):
        self._log.append("free:" + self.name)


class _Ctx:
    def __init__(self, log):
        self.log = log
        self.lib = _RecorderLib(log)
        self.handle = 1

    def malloc(self, nbytes):
        return _Buf(self.log, nbytes)

    def sync(self):
        self.log.append("sync")


class _Peers:
    def __init__(self, log):
        self._log = log

    def close(self):
        self._log.append("close_peers")


@pytest.mark.parametrize("kind", ["query", "train"])
def test_close_then_barrier_then_free(kind, monkeypatch):
    from slamhip import dist

    log = []
    ctx = _Ctx(log)
    monkeypatch.setattr(dist, "DeviceDescriptors", lambda c, rows: _Buf(log, 32 * max(len(rows), 1), "descriptors"))
    q = np.zeros((10, 32), np.uint8)
    t = np.zeros((20, 32), np.uint8)
    sm = dist.ShardedMatcher(ctx, 0, 2, q, t, collective=None) if kind == "query" else \
        dist.TrainShardedMatcher(ctx, 0, 2, q, t, 0, collective=None)
    sm.peers = _Peers(log)                           # as after a successful enable_p2p
    sm.collective = "p2p"
    del log[:]
    with pytest.raises(RuntimeError, match="barrier"):
        sm.free()                                    # peers mapped and no barrier: refused, nothing freed
    assert not any(x.startswith("free:") for x in log) and "close_peers" not in log
    del log[:]
    sm.free(lambda: log.append("barrier"))
    order = [x for x in log if x in ("close_peers", "barrier") or x.startswith("free:")]
    assert order[0] == "close_peers" and order[1] == "barrier" and all(x.startswith("free:") for x in order[2:])
    assert len(order) >= 4 and log.index("sync") < log.index("close_peers")
    # without peers no barrier is needed
    sm2 = dist.ShardedMatcher(ctx, 0, 1, q, t)
    del log[:]
    sm2.free()
    assert "barrier" not in log and any(x.startswith("free:") for x in log)


def test_step_sequence_of_the_sharded_matchers(monkeypatch):
    """What one pass issues, in order, on the recording library: wait for the gather that last used the buffer, search
    into this rank's slot, gather that buffer on the second stream; two buffers alternate between passes (the gather of
    pass i overlaps the search of pass i+1).  The train-sharded matcher gathers both tables and merges them by
    (distance, index) when the result is read."""
    from slamhip import dist

    log = []
    ctx = _Ctx(log)
    monkeypatch.setattr(dist, "DeviceDescriptors", lambda c, rows: type("D", (), {"buf": _Buf(log, 32 * max(len(rows), 1), "descriptors"), "free": lambda self: None})())
    q, t = np.zeros((10, 32), np.uint8), np.zeros((20, 32), np.uint8)
    sm = dist.ShardedMatcher(ctx, 1, 4, q, t, collective="rccl")
    assert sm.n_local == 3 and sm.per == 3 and sm.slot_bytes == 48 and len(sm.gathered) == 2     # rows 3..5 of 10 on rank 1 of 4
    del log[:]
    for _ in range(3):
        sm.step()
    calls = [x for x in log if x.startswith("slam_")]
    assert calls == ["slam_comm_wait_buffer", "slam_bf_knn2_u256", "slam_comm_allgather_overlapped"] * 3
    assert sm.passes == 3 and sm.last == 0                                                       # buffers 0, 1, 0
    # a rank whose shard is empty (more ranks than rows) still takes part in every gather
    sm_empty = dist.ShardedMatcher(ctx, 3, 4, q[:2], t, collective="rccl")
    assert sm_empty.n_local == 0
    del log[:]
    sm_empty.step()
    assert [x for x in log if x.startswith("slam_")] == ["slam_comm_wait_buffer", "slam_comm_allgather_overlapped"]
    # train-sharded: one search with global indices, two gathers (idx table, dist table), merge on read
    ts = dist.TrainShardedMatcher(ctx, 2, 4, q, t[:5], 10, collective="rccl")
    del log[:]
    ts.step()
    assert [x for x in log if x.startswith("slam_")] == ["slam_comm_wait_buffer"] * 2 + ["slam_bf_knn2_u256"] + ["slam_comm_allgather_overlapped"] * 2
    del log[:]
    ts.result()
    assert "slam_bf_merge_top2" in log and log.index("sync") < log.index("slam_bf_merge_top2")
    # a single rank issues no collective at all
    one = dist.ShardedMatcher(ctx, 0, 1, q, t)
    del log[:]
    one.step()
    assert [x for x in log if x.startswith("slam_")] == ["slam_bf_knn2_u256"]
