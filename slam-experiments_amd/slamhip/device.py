"""Device context and buffers on top of the C ABI (one Context = one GPU + one stream)."""
from __future__ import annotations

import ctypes
import threading
from typing import Optional, Tuple

import numpy as np

from . import _lib
from ._lib import check


class DeviceBuffer:
    """A device allocation (or a view into one) owned by a Context."""

    __slots__ = ("ctx", "ptr", "nbytes", "_owner")

    def __init__(self, ctx: "Context", ptr: int, nbytes: int, owner: Optional["DeviceBuffer"] = None):
        self.ctx = ctx
        self.ptr = ptr
        self.nbytes = nbytes
        self._owner = owner

    def view(self, offset: int, nbytes: Optional[int] = None) -> "DeviceBuffer":
        nbytes = self.nbytes - offset if nbytes is None else nbytes
        if offset < 0 or nbytes < 0 or offset + nbytes > self.nbytes:
            raise ValueError("view out of range")
        return DeviceBuffer(self.ctx, self.ptr + offset, nbytes, self._owner or self)

    def upload(self, host: np.ndarray) -> "DeviceBuffer":
        host = np.ascontiguousarray(host)
        if host.nbytes > self.nbytes:
            raise ValueError(f"upload of {host.nbytes} B into a {self.nbytes} B buffer")
        check(self.ctx.lib.slam_upload(self.ctx.handle, self.ptr, host.ctypes.data, host.nbytes))
        return self

    def download(self, dtype, shape) -> np.ndarray:
        out = np.empty(shape, dtype)
        if out.nbytes > self.nbytes:
            raise ValueError(f"download of {out.nbytes} B from a {self.nbytes} B buffer")
        check(self.ctx.lib.slam_download(self.ctx.handle, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self) -> None:
        if self._owner is None and self.ptr:
            check(self.ctx.lib.slam_free(self.ctx.handle, self.ptr))
            self.ptr = 0


class Context:
    """One HIP device + stream.  Raises if libslamhip or a gfx950 GPU is missing."""

    def __init__(self, device: int = 0):
        self.lib = _lib.load()
        h = ctypes.c_void_p()
        check(self.lib.slam_ctx_create(device, ctypes.byref(h)))
        self.handle = h
        self.device = device

    def close(self) -> None:
        if self.handle:
            check(self.lib.slam_ctx_destroy(self.handle))
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def malloc(self, nbytes: int) -> DeviceBuffer:
        p = ctypes.c_void_p()
        check(self.lib.slam_malloc(self.handle, int(nbytes), ctypes.byref(p)))
        return DeviceBuffer(self, p.value or 0, int(nbytes))

    def upload(self, host: np.ndarray) -> DeviceBuffer:
        host = np.ascontiguousarray(host)
        return self.malloc(host.nbytes).upload(host)

    def set_tuning(self, R: int = 0, blocks_per_cu: int = 0, lead_rows: int = 0, lead_chunk: int = 0,
                   tail: int = 0, feed: int = 0, cold: int = 0, chunk: int = 0, queue: int = 0, merge: int = 0) -> None:
        """Experiment knobs of the top-2 search on this context (``slam_bf_set_tuning``); no arguments = shipped plan."""
        knobs = (ctypes.c_int32 * 10)(R, blocks_per_cu, lead_rows, lead_chunk, tail, feed, cold, chunk, queue, merge)
        check(self.lib.slam_bf_set_tuning(self.handle, knobs, 10))

    def state_dirty(self) -> int:
        """Words of the search's merge state that are not idle once the stream has drained (``slam_bf_state_dirty``):
        0 after every completed search."""
        n = ctypes.c_int64(-1)
        check(self.lib.slam_bf_state_dirty(self.handle, ctypes.byref(n)))
        return n.value

    def plan_info(self, n: int, m: int) -> dict:
        """The launch plan the top-2 search would use for n x m (``slam_bf_plan_info``)."""
        p = (ctypes.c_int32 * 10)()
        check(self.lib.slam_bf_plan_info(self.handle, n, m, p))
        return dict(zip(("R", "qblocks", "chunk", "chunks", "lead_rows", "lead_chunks", "tail_chunks", "cus", "sgpr_feed", "cold_rows"), p))

    def sync(self) -> None:
        check(self.lib.slam_sync(self.handle))

    def timer_start(self) -> None:
        check(self.lib.slam_timer_start(self.handle))

    def timer_stop(self) -> float:
        ms = ctypes.c_float(0)
        check(self.lib.slam_timer_stop(self.handle, ctypes.byref(ms)))
        return ms.value

    def prof_enable(self, on: bool = True) -> None:
        check(self.lib.slam_prof_enable(self.handle, int(on)))

    def prof_read(self) -> Tuple[int, float]:
        n = ctypes.c_int64(0)
        ms = ctypes.c_double(0)
        check(self.lib.slam_prof_read(self.handle, ctypes.byref(n), ctypes.byref(ms)))
        return n.value, ms.value


_default_lock = threading.Lock()
_default_ctx: Optional[Context] = None


def default_context() -> Context:
    """Process-wide context on device 0 (what the drop-in matcher uses)."""
    global _default_ctx
    with _default_lock:
        if _default_ctx is None:
            _default_ctx = Context(0)
        return _default_ctx


PLAN_KNOBS = ("R", "blocks_per_cu", "lead_rows", "lead_chunk", "tail", "feed", "cold", "chunk", "queue", "merge")


def plan_describe(n: int, m: int, num_cu: int = 256, qb_all: int = 0, rows_on_host: bool = False, **knobs):
    """The launch plan of the top-2 search for n x m on a device with ``num_cu`` CUs, WITHOUT a device
    (``slam_bf_plan_describe``): (plan dict, chunk boundary table).  ``knobs``: as ``Context.set_tuning``."""
    unknown = set(knobs) - set(PLAN_KNOBS)
    if unknown:
        raise TypeError(f"unknown knobs {sorted(unknown)}")
    lib = _lib.load()
    k = (ctypes.c_int32 * len(PLAN_KNOBS))(*(int(knobs.get(name, 0)) for name in PLAN_KNOBS))
    plan = (ctypes.c_int32 * 14)()
    cap = 1 << 16
    tbl = (ctypes.c_int32 * cap)()
    length = ctypes.c_int64(0)
    check(lib.slam_bf_plan_describe(num_cu, k, len(PLAN_KNOBS), n, m, qb_all, int(rows_on_host), plan, tbl, cap, ctypes.byref(length)))
    names = ("R", "qblocks", "chunk", "chunks", "lead_rows", "lead_chunks", "tail_chunks", "cus", "sgpr_feed", "cold_rows",
             "table_free", "bound_free", "workers", "resident")
    out = dict(zip(names, plan))
    out["merge"], out["resident"] = out["resident"] >> 8, out["resident"] & 255
    return out, list(tbl[:min(length.value, cap)])
