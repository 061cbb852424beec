"""Brute-force Hamming matching of 256-bit ORB descriptors on MI355X.

Array-level API under the drop-in ``feature_matchers.BruteForceFeatureMatcher``.
Semantics are those of ``cv2.BFMatcher(NORM_HAMMING)`` as the reference uses it
(``feature_matchers.py:33-44``): per query row the nearest train rows ordered by
(distance asc, train index asc).  Everything here runs on the GPU through
``libslamhip.so``; there is no CPU path.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence, Tuple

import numpy as np

from ._lib import DESC_BYTES, NO_MATCH_DIST, NO_MATCH_IDX, addr, check
from .device import Context, DeviceBuffer, default_context

NORM_HAMMING = 6           # cv2.NORM_HAMMING, the only norm the reference constructs (slam.py:24)
IMGIDX_SHIFT = 18          # OpenCV's per-image row limit for multi-image train sets
MODE_ALL, MODE_MIN_DIST, MODE_RATIO, MODE_CROSS = 0, 1, 2, 3


def as_descriptors(a) -> np.ndarray:
    """Coerce a descriptor matrix to C-contiguous (n,32) uint8.

    Accepts what ``Frame.get_descriptors`` produces (``primitives.py:200-205``),
    including the n == 0 case, which numpy turns into a float64 ``(0,)`` array."""
    a = np.asarray(a)
    if a.size == 0:
        return np.zeros((0, DESC_BYTES), np.uint8)
    if a.dtype != np.uint8:
        raise ValueError(f"descriptors must be uint8, got {a.dtype}")
    if a.ndim != 2 or a.shape[1] != DESC_BYTES:
        raise ValueError(f"descriptors must have shape (n, {DESC_BYTES}), got {a.shape}")
    return np.ascontiguousarray(a)


class DeviceDescriptors:
    """A descriptor matrix resident in HBM: (n,32) uint8, row-major, 16-B aligned."""

    def __init__(self, ctx: Context, host: Optional[np.ndarray] = None, rows: Optional[int] = None):
        self.ctx = ctx
        if host is not None:
            host = as_descriptors(host)
            rows = host.shape[0]
        self.rows = int(rows or 0)
        self.buf = ctx.malloc(max(self.rows, 1) * DESC_BYTES)
        if host is not None and self.rows:
            self.buf.upload(host)

    def rows_view(self, start: int, stop: int) -> DeviceBuffer:
        return self.buf.view(start * DESC_BYTES, (stop - start) * DESC_BYTES)

    def free(self) -> None:
        self.buf.free()


class Top2Table:
    """Device tables idx/dist int32 [rows,2] written by the top-2 kernel."""

    def __init__(self, ctx: Context, rows: int):
        self.ctx = ctx
        self.rows = int(rows)
        self.idx = ctx.malloc(max(self.rows, 1) * 8)
        self.dist = ctx.malloc(max(self.rows, 1) * 8)

    def download(self) -> Tuple[np.ndarray, np.ndarray]:
        if self.rows == 0:
            return np.zeros((0, 2), np.int32), np.zeros((0, 2), np.int32)
        return self.idx.download(np.int32, (self.rows, 2)), self.dist.download(np.int32, (self.rows, 2))

    def free(self) -> None:
        self.idx.free()
        self.dist.free()


def knn2_device(ctx: Context, query: DeviceBuffer, n: int, train: DeviceBuffer, m: int, out_idx: DeviceBuffer,
                out_dist: DeviceBuffer, train_base: int = 0) -> None:
    """Launch the top-2 search on device-resident rows (asynchronous on the ctx stream)."""
    check(ctx.lib.slam_bf_knn2_u256(ctx.handle, query.ptr, n, train.ptr, m, train_base, out_idx.ptr, out_dist.ptr))


def knn_match_arrays(query, train, k: int = 2, ctx: Optional[Context] = None) -> Tuple[np.ndarray, np.ndarray]:
    """``knnMatch(query, train, k)`` as arrays: (idx, dist) int32 [N,k], k in {1,2}.

    Missing neighbours (M < k) are (-1, INT32_MAX)."""
    if k not in (1, 2):
        raise ValueError("k must be 1 or 2")
    q, t = as_descriptors(query), as_descriptors(train)
    ctx = ctx or default_context()
    n, m = q.shape[0], t.shape[0]
    idx = np.empty((n, 2), np.int32)
    dist = np.empty((n, 2), np.int32)
    if n:
        check(ctx.lib.slam_bf_knn2_u256_host(ctx.handle, addr(q), n, addr(t) if m else None, m, addr(idx), addr(dist)))
    return np.ascontiguousarray(idx[:, :k]), np.ascontiguousarray(dist[:, :k])


class _MatchOutputs:
    """Result buffers of ``slam_bf_match_host`` kept between calls (grown on demand) with their addresses: a frame loop
    allocates nothing per call and looks no pointer up twice."""

    __slots__ = ("qi", "ti", "dist", "cnt", "p_qi", "p_ti", "p_dist", "p_cnt", "rows")

    def __init__(self):
        self.rows = 0
        self.cnt = ctypes.c_int64(0)
        self.p_cnt = ctypes.byref(self.cnt)
        self.reserve(1)                                  # (an empty query side still gets valid pointers and empty slices)

    def reserve(self, n: int) -> None:
        if n > self.rows:
            self.rows = max(256, n + (n >> 2))
            self.qi, self.ti, self.dist = np.empty(self.rows, np.int32), np.empty(self.rows, np.int32), np.empty(self.rows, np.float32)
            self.p_qi, self.p_ti, self.p_dist = addr(self.qi), addr(self.ti), addr(self.dist)


def _match_host(ctx: Context, q: np.ndarray, t: Optional[np.ndarray], d_train: Optional[DeviceBuffer], m: int,
                keep_query: Optional[DeviceBuffer], mode: int, param: float, out: Optional[_MatchOutputs] = None):
    """One ``slam_bf_match_host`` call: upload, top-2 search, selection, download, one wait (frame-sized: zero-copy, completion polled)."""
    n = q.shape[0]
    out = out or _MatchOutputs()
    out.reserve(n)
    check(ctx.lib.slam_bf_match_host(ctx.handle, addr(q) if n else None, n,
                                     addr(t) if t is not None and m else None,
                                     d_train.ptr if d_train is not None and m else None, m,
                                     keep_query.ptr if keep_query is not None else None, mode, float(param),
                                     out.p_qi, out.p_ti, out.p_dist, out.p_cnt))
    c = out.cnt.value
    return out.qi[:c].copy(), out.ti[:c].copy(), out.dist[:c].copy()      # (the buffers are reused by the next call)


class FrameCache:
    """The query rows of a matcher's last call, kept in HBM (SURVEY.md §8f row f2 for unchanged ``slam.py`` users).

    ``Frontend._match_features`` (``frontend.py:181-187``) calls ``match(desc_last, desc_cur)`` on every frame with
    two fresh numpy copies (``Frame.get_descriptors``, ``primitives.py:200-205``), so the train side of frame k is
    byte for byte the query side of frame k-1.  The cache keeps that matrix on the device (and its bytes on the host,
    to recognise it: comparing 6.4 KB costs less than sending them); a call whose source rows equal it passes the
    device copy as ``d_train`` and uploads only the new query rows, straight into the buffer the next call will search.
    Two buffers alternate, grown on demand, so a steady stream of frames allocates nothing."""

    MAX_ROWS = 1 << 16      # beyond this the comparison costs as much as the upload: such calls bypass the cache

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self._bufs = [None, None]
        self._cur = 0                       # index of the buffer holding the last query rows
        self._host: Optional[bytes] = None  # the remembered rows as bytes (one memcmp recognises them)
        self._host_shape = None
        self.out = _MatchOutputs()          # result buffers reused from call to call
        self.hits = 0                       # calls whose train side was served from the device
        self.calls = 0

    def _buffer(self, which: int, rows: int) -> DeviceBuffer:
        need = max(rows, 1) * DESC_BYTES
        b = self._bufs[which]
        if b is None or b.nbytes < need:
            if b is not None:
                b.free()
            b = self._bufs[which] = self.ctx.malloc(max(need, 256 * DESC_BYTES) * 2)
        return b

    def lookup(self, t: np.ndarray) -> Optional[DeviceBuffer]:
        """The device copy of ``t`` if it is the matrix remembered from the last call."""
        if self._host is None or self._host_shape != t.shape or t.tobytes() != self._host:
            return None
        return self._bufs[self._cur]

    def next_buffer(self, rows: int) -> DeviceBuffer:
        return self._buffer(self._cur ^ 1, rows)

    def remember(self, q: np.ndarray) -> None:
        self._cur ^= 1
        self._host, self._host_shape = q.tobytes(), q.shape

    def forget(self) -> None:
        self._host = None

    def free(self) -> None:
        for b in self._bufs:
            if b is not None:
                b.free()
        self._bufs, self._host = [None, None], None


def match_arrays(source, query, dist_threshold: Optional[float] = None, ctx: Optional[Context] = None,
                 cache: Optional[FrameCache] = None):
    """``BruteForceFeatureMatcher.match`` as arrays (``feature_matchers.py:36-44``).

    Note the reference's argument order: first the train ("source", last frame)
    descriptors, then the query (current frame).  Returns (queryIdx, trainIdx,
    distance float32), one entry per match, ascending queryIdx; the
    ``dist_threshold`` filter keeps ``distance < max(2*min_dist, dist_threshold)``.
    With a ``cache`` (see ``FrameCache``) a source matrix equal to the previous call's
    query matrix is not uploaded again."""
    q, t = as_descriptors(query), as_descriptors(source)
    ctx = ctx or (cache.ctx if cache is not None else default_context())
    mode = MODE_MIN_DIST if dist_threshold else MODE_ALL   # `if dist_threshold and ...` (feature_matchers.py:41)
    param = float(dist_threshold or 0.0)
    n, m = q.shape[0], t.shape[0]
    if cache is None or n > FrameCache.MAX_ROWS or m > FrameCache.MAX_ROWS:
        return _match_host(ctx, q, t, None, m, None, mode, param)
    cache.calls += 1
    d_train = cache.lookup(t) if m else None
    keep = cache.next_buffer(n) if n else None
    try:
        out = _match_host(ctx, q, None if d_train is not None else t, d_train, m, keep, mode, param, cache.out)
    except Exception:
        cache.forget()
        raise
    if d_train is not None:
        cache.hits += 1
    if n:
        cache.remember(q)
    else:
        cache.forget()
    return out


def knn2_select_device(ctx: Context, d_query: DeviceBuffer, n: int, d_train: DeviceBuffer, m: int, d_idx: DeviceBuffer,
                       d_dist: DeviceBuffer, d_keep: DeviceBuffer, mode: int = MODE_RATIO, param: float = 0.75,
                       train_base: int = 0) -> int:
    """Top-2 search + selection in ONE launch on device-resident rows (``slam_bf_knn2_select_u256``): ``mode`` 0 keeps every
    query that has a neighbour, 2 is the Lowe ratio test ``dist0 < param * dist1`` (BASELINE configs[1]: knn = 2 + ratio).
    The tables go to ``d_idx`` / ``d_dist`` as ``knn2_device`` leaves them, one flag per query to ``d_keep`` (uint8 [n]);
    returns how many were kept (one wait: the completion words up to 16384 queries, else the stream)."""
    cnt = ctypes.c_int64(0)
    check(ctx.lib.slam_bf_knn2_select_u256(ctx.handle, d_query.ptr if n else None, n, d_train.ptr if m else None, m, train_base,
                                           d_idx.ptr, d_dist.ptr, mode, float(param), d_keep.ptr, ctypes.byref(cnt)))
    return cnt.value


def knn2_device_batch(ctx: Context, searches) -> None:
    """Several independent searches in ONE launch (``slam_bf_knn2_batch_u256``): ``searches`` is a sequence of
    ``(d_query, n_query, d_train, n_train, d_idx, d_dist[, train_base])`` with device buffers, at most 32 entries.
    Each table comes out bit-identical to ``knn2_device`` on the same arguments."""
    from ._lib import BF_BATCH_MAX, BfSearch

    searches = list(searches)
    if len(searches) > BF_BATCH_MAX:
        raise ValueError(f"at most {BF_BATCH_MAX} searches per launch")
    arr = (BfSearch * max(len(searches), 1))()
    for i, sr in enumerate(searches):
        dq, n, dt, m, di, dd = sr[:6]
        arr[i] = BfSearch(dq.ptr if n else None, n, dt.ptr if m else None, m, sr[6] if len(sr) > 6 else 0, di.ptr, dd.ptr)
    check(ctx.lib.slam_bf_knn2_batch_u256(ctx.handle, len(searches), arr))


def knn_match_arrays_batch(pairs, ctx: Optional[Context] = None):
    """``knn_match_arrays(query, train, 2)`` for several independent (query, train) pairs with ONE kernel launch:
    candidate verifications, or the two directions of a cross check.  Returns a list of (idx [N,2], dist [N,2])."""
    ctx = ctx or default_context()
    prepared = [(as_descriptors(q), as_descriptors(t)) for q, t in pairs]
    bufs, searches = [], []
    try:
        for q, t in prepared:
            dq, dt = DeviceDescriptors(ctx, q), DeviceDescriptors(ctx, t)
            tab = Top2Table(ctx, q.shape[0])
            bufs.append((dq, dt, tab))
            searches.append((dq.buf, q.shape[0], dt.buf, t.shape[0], tab.idx, tab.dist))
        knn2_device_batch(ctx, searches)
        return [tab.download() if q.shape[0] else (np.zeros((0, 2), np.int32), np.zeros((0, 2), np.int32))
                for (q, _), (_, _, tab) in zip(prepared, bufs)]
    finally:
        for dq, dt, tab in bufs:
            for o in (tab, dq, dt):
                o.free()


def ratio_test_arrays(query, train, ratio: float = 0.75, ctx: Optional[Context] = None):
    """knn=2 + Lowe ratio test: (queryIdx, trainIdx, distance) of queries with d0 < ratio * d1."""
    q, t = as_descriptors(query), as_descriptors(train)
    ctx = ctx or default_context()
    return _match_host(ctx, q, t, None, t.shape[0], None, MODE_RATIO, ratio)


def cross_check_arrays(query, train, ctx: Optional[Context] = None):
    """``cv2.BFMatcher(NORM_HAMMING, crossCheck=True).match(query, train)`` as arrays: the pairs (q, t) where t is
    q's nearest train row and q is t's nearest query row (ties to the lowest index on both sides, as OpenCV 4.x's
    ``batchDistance`` crosscheck branch decides them).  One library call: forward search, reverse search and the
    selection all stay on the device (``slam_bf_match_host`` mode 3)."""
    q, t = as_descriptors(query), as_descriptors(train)
    ctx = ctx or default_context()
    return _match_host(ctx, q, t, None, t.shape[0], None, MODE_CROSS, 0.0)


def split_image_index(global_idx: np.ndarray, image_rows: Sequence[int]) -> Tuple[np.ndarray, np.ndarray]:
    """Global train row -> (imgIdx, trainIdx) for a concatenated multi-image train set."""
    offsets = np.concatenate([[0], np.cumsum(np.asarray(image_rows, np.int64))])
    g = np.asarray(global_idx, np.int64)
    img = np.searchsorted(offsets, g, side="right") - 1
    local = g - offsets[np.clip(img, 0, len(offsets) - 1)]
    none = g < 0
    return np.where(none, -1, img).astype(np.int32), np.where(none, -1, local).astype(np.int32)


def knn_match_collection(query, train_images: Sequence[np.ndarray], k: int = 2, ctx: Optional[Context] = None):
    """knnMatch against a collection of train images (``BFMatcher.add``): loop-closure layout.

    Returns (imgIdx, trainIdx, dist) int32 [N,k]; order (dist, imgIdx, trainIdx) as OpenCV."""
    imgs = [as_descriptors(t) for t in train_images]
    rows = [t.shape[0] for t in imgs]
    if any(r >= (1 << IMGIDX_SHIFT) for r in rows):
        raise ValueError("each train image must have fewer than 2^18 rows (OpenCV IMGIDX_ONE)")
    cat = np.concatenate(imgs, 0) if imgs else np.zeros((0, DESC_BYTES), np.uint8)
    idx, dist = knn_match_arrays(query, cat, k, ctx)
    img, local = split_image_index(idx, rows)
    return img, local, dist


class ResidentMatcher:
    """Frame-to-frame matching with descriptors kept in HBM.

    ``Frontend._match_features`` (``frontend.py:181-187``) matches the last frame against the current one
    on every frame, re-stacking both descriptor matrices from per-feature rows each time
    (``primitives.py:200-205``).  Here the train side of frame k is the query side of frame k-1:
    each frame's descriptors are uploaded once and stay on the device for the next call."""

    def __init__(self, ctx: Optional[Context] = None):
        self.ctx = ctx or default_context()
        self._bufs: list = [None, None]        # two device buffers used alternately: the last frame's rows and the current one's
        self._which = 0                        # the buffer that holds the last frame
        self._rows: Optional[int] = None       # rows of the last frame (None: nothing pushed yet)

    def reset(self) -> None:
        for i, b in enumerate(self._bufs):
            if b is not None:
                b.free()
            self._bufs[i] = None
        self._rows = None

    def push(self, descriptors, dist_threshold: Optional[float] = None):
        """Match ``descriptors`` (current frame, query) against the previously pushed frame (train).

        Returns (queryIdx, trainIdx, distance) like ``match_arrays``, or None for the first frame.
        One call into the library per frame: the rows go up once, into the buffer the next frame will
        search as its train side.  The two buffers are kept from frame to frame and only grow (an allocation and a free per
        frame cost more than the search: 0.033 -> 0.021 ms per frame at 200 features)."""
        q = as_descriptors(descriptors)
        cur = self._which ^ 1
        need = max(q.shape[0], 1) * DESC_BYTES
        b = self._bufs[cur]
        if b is None or b.nbytes < need:
            if b is not None:
                b.free()
            b = self._bufs[cur] = self.ctx.malloc(max(need, 256 * DESC_BYTES) * 2)
        first = self._rows is None
        mode = MODE_MIN_DIST if dist_threshold else MODE_ALL
        out = _match_host(self.ctx, q, None, None if first else self._bufs[self._which], 0 if first else self._rows,
                          b if q.shape[0] else None, mode, float(dist_threshold or 0.0))
        self._which, self._rows = cur, q.shape[0]
        return None if first else out


class KeyframeDatabase:
    """Loop-closure layout kept in HBM (BASELINE.json configs[3]): the descriptors of every keyframe added so far
    live in one growing device buffer, and a query frame is searched against all of them in one launch.

    The result has OpenCV's multi-image form (``BFMatcher.add`` + ``knnMatch``): per query and neighbour the
    keyframe (``imgIdx``), the row inside it (``trainIdx``) and the distance, ordered by (distance, imgIdx, trainIdx).
    Only the new keyframe's rows cross PCIe on ``add``; queries upload the query rows and download the [N,k] tables."""

    def __init__(self, ctx: Optional[Context] = None, capacity_rows: int = 1 << 16):
        self.ctx = ctx or default_context()
        self._cap = max(int(capacity_rows), 1)
        self._buf = self.ctx.malloc(self._cap * DESC_BYTES)
        self.rows: list = []            # rows per keyframe, in insertion order
        self.total = 0
        self._qbuf: Optional[DeviceBuffer] = None   # query rows + result tables, reused from query to query
        self._qcap = 0

    def add(self, descriptors) -> int:
        """Append one keyframe's descriptors; returns its index (the ``imgIdx`` later results refer to)."""
        d = as_descriptors(descriptors)
        n = d.shape[0]
        if n >= (1 << IMGIDX_SHIFT):
            raise ValueError("a keyframe must have fewer than 2^18 rows (OpenCV IMGIDX_ONE)")
        if self.total + n > self._cap:
            cap = self._cap
            while cap < self.total + n:
                cap *= 2
            grown = self.ctx.malloc(cap * DESC_BYTES)
            if self.total:                                   # device-to-device through the library's copy primitive
                check(self.ctx.lib.slam_copy(self.ctx.handle, grown.ptr, self._buf.ptr, self.total * DESC_BYTES))
            self._buf.free()
            self._buf, self._cap = grown, cap
        if n:
            self._buf.view(self.total * DESC_BYTES, n * DESC_BYTES).upload(d)
        self.rows.append(n)
        self.total += n
        return len(self.rows) - 1

    def query(self, descriptors, k: int = 2):
        """(imgIdx, trainIdx, dist) int32 [N,k] of the query rows against every keyframe added so far."""
        if k not in (1, 2):
            raise ValueError("k must be 1 or 2")
        q = as_descriptors(descriptors)
        n = q.shape[0]
        if n == 0:
            z = np.zeros((0, k), np.int32)
            return z, z.copy(), z.copy()
        if n > self._qcap:                                   # query rows + both tables in one buffer, kept between queries
            if self._qbuf is not None:
                self._qbuf.free()
            self._qcap = max(2 * n, 1024)
            self._qbuf = self.ctx.malloc(self._qcap * (DESC_BYTES + 16))
        dq = self._qbuf.view(0, n * DESC_BYTES).upload(q)
        d_idx = self._qbuf.view(self._qcap * DESC_BYTES, n * 8)
        d_dist = self._qbuf.view(self._qcap * (DESC_BYTES + 8), n * 8)
        knn2_device(self.ctx, dq, n, self._buf, self.total, d_idx, d_dist)
        idx, dist = d_idx.download(np.int32, (n, 2)), d_dist.download(np.int32, (n, 2))
        img, local = split_image_index(idx, self.rows)
        return (np.ascontiguousarray(img[:, :k]), np.ascontiguousarray(local[:, :k]), np.ascontiguousarray(dist[:, :k]))

    def free(self) -> None:
        self._buf.free()
        if self._qbuf is not None:
            self._qbuf.free()
            self._qbuf, self._qcap = None, 0
        self.rows, self.total = [], 0
