"""ctypes binding of libslamhip.so (the C ABI declared in include/slamhip.h).

The product path has no CPU fallback: if the shared library is missing or no
gfx950 device is visible, every entry point raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint8, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libslamhip.so")

SLAM_OK = 0
NO_MATCH_IDX = -1
NO_MATCH_DIST = 2**31 - 1
DESC_BYTES = 32
COMM_ID_BYTES = 128
P2P_HANDLE_BYTES = 64


class SlamHipError(RuntimeError):
    """A libslamhip call returned a negative status."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libslamhip error {code}: {message}")
        self.code = code


class SlamHipBusy(SlamHipError):
    """SLAM_ERR_BUSY (-6): a resource condition, not a caller error - a kernel that needs all its workgroups resident at once
    could not get them (other work holds compute units).  Inputs and outputs are untouched; retry, or use the form without
    a residency requirement."""


SLAM_ERR_INVALID, SLAM_ERR_BUSY = -1, -6


# name -> (restype, argtypes); must list every SLAM_API symbol of include/slamhip.h
SIGNATURES = {
    "slam_last_error": (c_char_p, []),
    "slam_version": (c_char_p, []),
    "slam_device_count": (c_int, [POINTER(c_int)]),
    "slam_ctx_create": (c_int, [c_int, POINTER(c_void_p)]),
    "slam_ctx_destroy": (c_int, [c_void_p]),
    "slam_ctx_device": (c_int, [c_void_p, POINTER(c_int)]),
    "slam_sync": (c_int, [c_void_p]),
    "slam_malloc": (c_int, [c_void_p, c_uint64, POINTER(c_void_p)]),
    "slam_free": (c_int, [c_void_p, c_void_p]),
    "slam_memset": (c_int, [c_void_p, c_void_p, c_int, c_uint64]),
    "slam_copy": (c_int, [c_void_p, c_void_p, c_void_p, c_uint64]),
    "slam_upload": (c_int, [c_void_p, c_void_p, c_void_p, c_uint64]),
    "slam_download": (c_int, [c_void_p, c_void_p, c_void_p, c_uint64]),
    "slam_timer_start": (c_int, [c_void_p]),
    "slam_timer_stop": (c_int, [c_void_p, POINTER(c_float)]),
    "slam_prof_enable": (c_int, [c_void_p, c_int]),
    "slam_prof_read": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_double)]),
    "slam_bf_knn2_u256": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "slam_bf_knn2_batch_u256": (c_int, [c_void_p, c_int64, c_void_p]),
    "slam_bf_knn2_select_u256": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int, c_double,
                                         c_void_p, POINTER(c_int64)]),
    "slam_bf_merge_top2": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "slam_bf_knn2_u256_host": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    "slam_bf_set_tuning": (c_int, [c_void_p, POINTER(c_int32), c_int]),
    "slam_bf_plan_info": (c_int, [c_void_p, c_int64, c_int64, POINTER(c_int32)]),
    "slam_bf_plan_describe": (c_int, [c_int, POINTER(c_int32), c_int, c_int64, c_int64, c_int64, c_int, POINTER(c_int32),
                                      POINTER(c_int32), c_int64, POINTER(c_int64)]),
    "slam_bf_reset_state": (c_int, [c_void_p]),
    "slam_bf_state_dirty": (c_int, [c_void_p, POINTER(c_int64)]),
    "slam_bf_match_filter": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_double, c_void_p,
                                     POINTER(c_int64), POINTER(c_int32)]),
    "slam_bf_cross_check": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                    POINTER(c_int64)]),
    "slam_bf_split_index": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    "slam_reproj_rj_f64": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                   c_int64, c_double, c_double, c_double, c_double, c_void_p, c_void_p, c_void_p]),
    "slam_pose_normal_eq_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_double,
                                        c_double, c_double, c_double, c_double, c_void_p, c_void_p, c_void_p]),
    "slam_pose_optimize_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_double, c_double, c_double,
                                       c_double, c_int, c_int, c_double, c_double, c_void_p, c_void_p, c_void_p,
                                       c_void_p]),
    "slam_pose_optimize_batch_f64": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_double,
                                             c_double, c_double, c_double, c_int, c_int, c_double, c_double, c_void_p,
                                             c_void_p, c_void_p, c_void_p]),
    "slam_bf_match_host": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_double,
                                   c_void_p, c_void_p, c_void_p, c_void_p]),
    "slam_index_errors": (c_int, [c_void_p, POINTER(c_int64)]),
    "slam_io_counters": (c_int, [c_void_p, POINTER(c_uint64), POINTER(c_uint64)]),
    "slam_pose_optimize_host_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_double, c_double,
                                            c_double, c_double, c_int, c_int, c_double, c_double, c_void_p, c_void_p,
                                            c_void_p, c_void_p]),
    "slam_ba_reduce_f64": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                   c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_double,
                                   c_double, c_double, c_double, c_double, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "slam_ba_cost_f64": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double,
                                 c_double, c_double, c_double, c_double, c_void_p]),
    "slam_ba_backsub_f64": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p]),
    "slam_ba_optimize_workspace": (c_int, [c_int64, c_int64, c_int64, POINTER(c_uint64)]),
    "slam_ba_optimize_host_f64": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_double, c_double, c_double, c_double, c_double, c_int, c_void_p, c_void_p,
                                          c_void_p]),
    "slam_ba_optimize_f64": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_int64, c_double, c_double, c_double, c_double, c_double, c_int, c_void_p,
                                     c_void_p, c_void_p, c_uint64, c_void_p]),
    "slam_comm_version": (c_int, [POINTER(c_int)]),
    "slam_comm_unique_id": (c_int, [c_void_p]),
    "slam_comm_init": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "slam_comm_destroy": (c_int, [c_void_p]),
    "slam_comm_allgather": (c_int, [c_void_p, c_void_p, c_void_p, c_uint64]),
    "slam_comm_broadcast": (c_int, [c_void_p, c_void_p, c_uint64, c_int]),
    "slam_comm_allgather_overlapped": (c_int, [c_void_p, c_void_p, c_void_p, c_uint64, c_int]),
    "slam_comm_wait_buffer": (c_int, [c_void_p, c_int]),
    "slam_p2p_export": (c_int, [c_void_p, c_void_p, c_void_p]),
    "slam_p2p_open": (c_int, [c_void_p, c_void_p, POINTER(c_void_p)]),
    "slam_p2p_close": (c_int, [c_void_p, c_void_p]),
    "slam_p2p_allgather_overlapped": (c_int, [c_void_p, c_void_p, c_uint64, c_int, c_void_p, c_int, c_int]),
}

class BfSearch(ctypes.Structure):
    """``slam_bf_search`` of include/slamhip.h: one entry of a batched search."""

    _fields_ = [("d_query", c_void_p), ("N", c_int64), ("d_train", c_void_p), ("M", c_int64), ("train_base", c_int64),
                ("d_idx", c_void_p), ("d_dist", c_void_p)]


BF_BATCH_MAX = 32
_lib = None


def load() -> ctypes.CDLL:
    """Load libslamhip.so and attach the signatures; raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C slam-experiments_amd/csrc). There is no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and this table disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def addr(a) -> int:
    """Address of a C-contiguous numpy array: 0.46 us through the buffer protocol against 1.3-1.4 us for ``a.ctypes.data`` or
    ``a.__array_interface__`` (a frame-sized ``match()`` is a 5 us kernel; the per-frame host calls pass up to nine pointers)."""
    try:
        return ctypes.addressof(ctypes.c_char.from_buffer(a))
    except (TypeError, ValueError):              # a read-only or empty array has no writable buffer to borrow
        return a.__array_interface__["data"][0]


def check(rc: int) -> None:
    if rc != SLAM_OK:
        msg = load().slam_last_error()
        raise (SlamHipBusy if rc == SLAM_ERR_BUSY else SlamHipError)(rc, msg.decode("utf-8", "replace") if msg else "unknown error")


def device_count() -> int:
    n = c_int(0)
    rc = load().slam_device_count(ctypes.byref(n))
    return n.value if rc == SLAM_OK else 0
