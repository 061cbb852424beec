"""Drop-in for the reference's ``feature_matchers.py`` backed by MI355X HIP kernels.

Put this directory ahead of the reference on ``sys.path`` and ``slam.py`` /
``frontend.py`` pick it up unchanged: same module name, same classes, same
signatures (``feature_matchers.py:9-14,33-38``), same result order.  The
brute-force search runs in ``libslamhip.so`` (``slamhip.matching``); there is no
CPU fallback — without the library or a gfx950 GPU, ``match`` raises.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Optional, Sequence

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
        matches_img = drawMatches(query_img, query_keypoints, source_img, source_keypoints, matches, None)
        imshow("Matches", matches_img)


class BruteForceFeatureMatcher(FeatureMatcher):
    """``cv2.BFMatcher(normType=NORM_HAMMING)`` replacement for 256-bit ORB descriptors."""

    def __init__(self, norm_type: int):
        if int(norm_type) != _m.NORM_HAMMING:
            raise NotImplementedError(
                f"norm_type {norm_type}: only cv2.NORM_HAMMING (6) is implemented (the reference constructs no other, slam.py:24)"
            )
        self.norm_type = int(norm_type)

    def match(self, source_descriptors: np.ndarray, query_descriptors: np.ndarray,
              dist_threshold: Optional[float] = None) -> Sequence[DMatch]:
        qi, ti, dist = _m.match_arrays(source_descriptors, query_descriptors, dist_threshold)
        return _to_dmatches(qi, ti, dist)

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
