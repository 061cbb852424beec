"""Drop-in for the reference's ``feature_matchers.py`` backed by MI355X HIP kernels.

Put this directory ahead of the reference on ``sys.path`` and ``slam.py`` /
``frontend.py`` pick it up unchanged: same module name, same classes, same
signatures (``feature_matchers.py:9-14,33-38``), same result order.  The
brute-force search runs in ``libslamhip.so`` (``slamhip.matching``); there is no
CPU fallback — without the library or a gfx950 GPU, ``match`` raises.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from collections.abc import Sequence as _SequenceABC
from typing import Optional, Sequence

import threading
from itertools import repeat

import numpy as np

from slamhip import matching as _m

try:  # real cv2.DMatch objects when OpenCV is present (drawMatches needs them, feature_matchers.py:28)
    from cv2 import DMatch as _CvDMatch
except ImportError:  # pragma: no cover - cv2 is absent from the build image
    _CvDMatch = None


class DMatch:
    """Stand-in for ``cv2.DMatch`` when OpenCV is not installed: the four public fields, nothing else."""

    __slots__ = ("queryIdx", "trainIdx", "imgIdx", "distance")

    def __init__(self, queryIdx: int = -1, trainIdx: int = -1, imgIdx: int = -1, distance: float = float("inf")):
        self.queryIdx = queryIdx
        self.trainIdx = trainIdx
        self.imgIdx = imgIdx
        self.distance = distance

    def __repr__(self) -> str:
        return f"DMatch(queryIdx={self.queryIdx}, trainIdx={self.trainIdx}, imgIdx={self.imgIdx}, distance={self.distance})"


def _make_dmatch(q: int, t: int, d: float, img: int = 0):
    if _CvDMatch is not None:
        return _CvDMatch(int(q), int(t), int(img), float(d))
    return DMatch(int(q), int(t), int(img), float(d))


def _to_dmatches(qi: np.ndarray, ti: np.ndarray, dist: np.ndarray) -> list:
    # .tolist() already yields Python ints / floats; map() keeps the per-object cost to the constructor call
    cls = _CvDMatch if _CvDMatch is not None else DMatch
    return list(map(cls, qi.tolist(), ti.tolist(), repeat(0, len(qi)), dist.tolist()))


class MatchList(_SequenceABC):
    """What ``match`` returns: a read-only sequence of ``DMatch`` over the arrays the GPU produced, building the
    objects only when something looks at them.  The reference's consumers use ``len()`` (the ``< 5`` tracking-loss
    gates, ``frontend.py:116,163``), iteration and indexing with ``.queryIdx/.trainIdx/.distance``
    (``frontend.py:174-177,194,205-207``; ``utils.py:14-16,42-44``; ``feature_matchers.py:42-43``): ``len`` costs
    nothing, the first iteration materialises the whole list once (one constructor call per match), and vectorised
    callers can read ``queryIdx`` / ``trainIdx`` / ``distance`` as arrays without creating any object.
    ``cv2.drawMatches`` wants a real list of ``cv2.DMatch``: ``list(matches)`` (``draw_matches`` does that)."""

    __slots__ = ("queryIdx", "trainIdx", "distance", "_objs")

    def __init__(self, qi: np.ndarray, ti: np.ndarray, dist: np.ndarray):
        self.queryIdx, self.trainIdx, self.distance = qi, ti, dist
        self._objs = None

    def _all(self) -> list:
        if self._objs is None:
            self._objs = _to_dmatches(self.queryIdx, self.trainIdx, self.distance)
        return self._objs

    def __len__(self) -> int:
        return len(self.queryIdx)

    def __getitem__(self, i):
        if self._objs is None and isinstance(i, (int, np.integer)):
            return _make_dmatch(self.queryIdx[i], self.trainIdx[i], self.distance[i])
        return self._all()[i]

    def __iter__(self):
        return iter(self._all())

    def __eq__(self, other):
        return self._all() == (other._all() if isinstance(other, MatchList) else other)

    def __repr__(self) -> str:
        return f"MatchList({len(self)} matches)"


class FeatureMatcher(ABC):
    @abstractmethod
    def match(self, source_descriptors: np.ndarray, query_descriptors: np.ndarray) -> Sequence[DMatch]:
        raise NotImplementedError

    @classmethod
    def draw_matches(cls, source_img, source_keypoints, query_img, query_keypoints, matches) -> None:
        """Source img is on the right and Query img is on the left (needs OpenCV for the drawing itself)."""
        try:
            from cv2 import drawMatches, imshow
        except ImportError as exc:
            raise ImportError("draw_matches needs OpenCV (cv2), which is not installed") from exc
        matches_img = drawMatches(query_img, query_keypoints, source_img, source_keypoints, list(matches), None)
        imshow("Matches", matches_img)


class BruteForceFeatureMatcher(FeatureMatcher):
    """``cv2.BFMatcher(normType=NORM_HAMMING)`` replacement for 256-bit ORB descriptors."""

    def __init__(self, norm_type: int):
        if int(norm_type) != _m.NORM_HAMMING:
            raise NotImplementedError(
                f"norm_type {norm_type}: only cv2.NORM_HAMMING (6) is implemented (the reference constructs no other, slam.py:24)"
            )
        self.norm_type = int(norm_type)
        self._cache = None          # slamhip FrameCache, made on the first match(): last frame's rows stay in HBM
        self._lock = threading.Lock()

    def match(self, source_descriptors: np.ndarray, query_descriptors: np.ndarray,
              dist_threshold: Optional[float] = None) -> Sequence[DMatch]:
        """``bf.match(query, source)`` + the optional min-distance filter (``feature_matchers.py:36-44``).

        The tracking loop calls this with (last frame, current frame) on every frame (``frontend.py:181-187``): the
        current frame's rows stay on the device, so when they come back as the next call's ``source_descriptors``
        they are recognised and not uploaded again."""
        with self._lock:
            if self._cache is None:
                self._cache = _m.FrameCache(_m.default_context())
            qi, ti, dist = _m.match_arrays(source_descriptors, query_descriptors, dist_threshold, cache=self._cache)
        return MatchList(qi, ti, dist)

    def close(self) -> None:
        """Release the device buffers that hold the last frame's descriptors (optional: they are two small
        allocations that otherwise live as long as the process-wide context)."""
        with self._lock:
            if self._cache is not None:
                self._cache.free()
                self._cache = None

    # ---- array-level extensions (not in the reference) -----------------------
    def match_arrays(self, source_descriptors, query_descriptors, dist_threshold: Optional[float] = None):
        return _m.match_arrays(source_descriptors, query_descriptors, dist_threshold)

    def knn_match_arrays(self, query_descriptors, train_descriptors, k: int = 2):
        return _m.knn_match_arrays(query_descriptors, train_descriptors, k)

    def knn_match(self, query_descriptors, train_descriptors, k: int = 2) -> list:
        """``bf.knnMatch(query, train, k)``: one list of up to k DMatch per query."""
        idx, dist = _m.knn_match_arrays(query_descriptors, train_descriptors, k)
        return [[_make_dmatch(q, t, d) for t, d in zip(irow, drow) if t >= 0]
                for q, (irow, drow) in enumerate(zip(idx.tolist(), dist.tolist()))]

    def ratio_test(self, query_descriptors, train_descriptors, ratio: float = 0.75) -> list:
        return _to_dmatches(*_m.ratio_test_arrays(query_descriptors, train_descriptors, ratio))

    def cross_check_match(self, query_descriptors, train_descriptors) -> list:
        return _to_dmatches(*_m.cross_check_arrays(query_descriptors, train_descriptors))
