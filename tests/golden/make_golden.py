"""Writes tests/golden/kat_*.json: hand-derived known-answer vectors for the hot path.

The reference ships no tests or fixtures and cv2 / g2o cannot be imported here
(SURVEY.md §8c), so these vectors are NOT captured from the reference: every expected
value below follows in closed form from the documented semantics (OpenCV BFMatcher
ordering rules; feature_matchers.py:41-43; frontend.py:272-291) and is written out by
hand in this script.  No oracle or product code is imported.
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
INT_MAX = 2**31 - 1


def prefix_ones(i):
    """32-byte descriptor whose first i bits (MSB-first within bytes) are 1."""
    bits = np.zeros(256, np.uint8)
    bits[:i] = 1
    return np.packbits(bits).tolist()


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1)


# 1. prefix-ones ladder: train row i has popcount i, so d(zeros, row i) = i and d(ones, row i) = 256 - i.
train = [prefix_ones(i) for i in range(257)]
dump("kat_ladder.json", {
    "doc": "train row i = first i bits set (i = 0..256); queries: zeros, ones, prefix_ones(100)",
    "train": train,
    "query": [prefix_ones(0), prefix_ones(256), prefix_ones(100)],
    # zeros: rows 0,1 at 0,1.  ones: rows 256,255 at 0,1.  prefix(100): row 100 at 0, then rows 99 and 101 tie at 1 -> lowest index 99
    "idx": [[0, 1], [256, 255], [100, 99]],
    "dist": [[0, 1], [0, 1], [0, 1]],
})

# 2. duplicates: identical train rows -> ties go to the lowest train index, 2nd neighbour is the next index.
row_a, row_b = prefix_ones(7), prefix_ones(200)
dump("kat_ties.json", {
    "doc": "train = [b, a, a, b, a]; query a -> (1,2) at 0; query b -> (0,3) at 0; query zeros -> a rows at 7: (1,2)",
    "train": [row_b, row_a, row_a, row_b, row_a],
    "query": [row_a, row_b, prefix_ones(0)],
    "idx": [[1, 2], [0, 3], [1, 2]],
    "dist": [[0, 0], [0, 0], [7, 7]],
})

# 3. k=2 with a single train row, and with none: missing neighbours are (-1, INT_MAX); match() returns nothing for M = 0.
dump("kat_short_train.json", {
    "doc": "M=1: one neighbour only; M=0: no neighbours, bf.match returns []",
    "query": [prefix_ones(3), prefix_ones(250)],
    "train_one": [prefix_ones(5)],
    "idx_one": [[0, -1], [0, -1]],
    "dist_one": [[2, INT_MAX], [245, INT_MAX]],
    "idx_none": [[-1, -1], [-1, -1]],
    "dist_none": [[INT_MAX, INT_MAX], [INT_MAX, INT_MAX]],
})

# 4. the reference's filter (feature_matchers.py:41-43): keep distance < max(2*min_dist, dist_threshold), strict.
#    source rows: popcounts 0, 10, 20, 40.  queries: prefix(2) -> nearest row 0 at 2; prefix(14) -> row 10 at 4;
#    prefix(28) -> row 20 at 8 (row 40 is at 12); prefix(36) -> row 40 at 4 (row 20 at 16).
#    1-NN distances [2, 4, 8, 4], min_dist 2.
#      thr 8.0 -> limit max(4, 8) = 8 -> 8 is NOT < 8 -> queries 0,1,3 stay
#      thr 3.0 -> limit max(4, 3) = 4 -> only query 0 (4 is not < 4)
#      thr 9.0 -> limit 9 -> all four;  thr None / 0.0 -> no filter (falsy), all four
dump("kat_filter.json", {
    "doc": "BruteForceFeatureMatcher.match(source, query, dist_threshold)",
    "source": [prefix_ones(0), prefix_ones(10), prefix_ones(20), prefix_ones(40)],
    "query": [prefix_ones(2), prefix_ones(14), prefix_ones(28), prefix_ones(36)],
    "cases": [
        {"thr": None, "queryIdx": [0, 1, 2, 3], "trainIdx": [0, 1, 2, 3], "distance": [2.0, 4.0, 8.0, 4.0]},
        {"thr": 0.0, "queryIdx": [0, 1, 2, 3], "trainIdx": [0, 1, 2, 3], "distance": [2.0, 4.0, 8.0, 4.0]},
        {"thr": 8.0, "queryIdx": [0, 1, 3], "trainIdx": [0, 1, 3], "distance": [2.0, 4.0, 4.0]},
        {"thr": 3.0, "queryIdx": [0], "trainIdx": [0], "distance": [2.0]},
        {"thr": 9.0, "queryIdx": [0, 1, 2, 3], "trainIdx": [0, 1, 2, 3], "distance": [2.0, 4.0, 8.0, 4.0]},
    ],
})

# 5. crossCheck (OpenCV 4.x batch_distance.cpp crosscheck branch, K = 1; documented contract of
#    BFMatcher(crossCheck=True): "(i, j) such that for i-th query descriptor the j-th descriptor in the matcher's
#    collection is the nearest and vice versa").  The branch computes the reverse table tidx (train row -> nearest
#    query), the forward table sidx (query -> nearest train row), scatters ascending train rows with strict "<",
#    and finally clears every query i with tidx[sidx[i]] != i.
#    Case 1: queries q0 = prefix(10), q1 = prefix(20); train t0 = prefix(12), t1 = prefix(11), t2 = prefix(30).
#      d(q0, .) = (2, 1, 20) -> sidx[q0] = t1;   d(q1, .) = (8, 9, 10) -> sidx[q1] = t0.
#      d(t0, .) = (2, 8) -> tidx[t0] = q0; d(t1, .) = (1, 9) -> tidx[t1] = q0; d(t2, .) = (20, 10) -> tidx[t2] = q1.
#      scatter: q0 <- t0 (2), then t1 (1 < 2) -> (t1, 1); q1 <- t2 (10).
#      forward check: tidx[sidx[q0] = t1] = q0 -> kept (t1, 1); tidx[sidx[q1] = t0] = q0 != q1 -> cleared.
#      Result [1, -1].  (Round 1 had [1, 2] here: the scatter alone, without the forward pass - a non-mutual pair.)
#    Case 2: query prefix(100) inserted as q1: d(q1, .) = (88, 89, 70) -> sidx = t2, tidx[t2] = q2 (10 < 70) -> cleared;
#      q2 = prefix(20) as q1 of case 1 -> cleared.  Result [1, -1, -1].
#    Case 3, ties on both sides: queries q0 = q1 = prefix(8), q2 = prefix(40); train t0 = t1 = prefix(8), t2 = prefix(41),
#      t3 = prefix(39).  sidx = (t0, t0, t2) [q2: d = (32, 32, 1, 1) -> lowest index t2]; tidx = (q0, q0, q2, q2).
#      q0: tidx[t0] = q0 kept (t0, 0).  q1: tidx[t0] = q0 != q1 cleared (its duplicate took the row).
#      q2: tidx[t2] = q2 kept (t2, 1) - not t3, although t3 is equally near and also points at q2.
#      Result idx [0, -1, 2], dist [0, -, 1].
dump("kat_cross_check.json", {
    "doc": "cv2.BFMatcher(NORM_HAMMING, crossCheck=True).match(query, train): mutual nearest neighbours only",
    "query": [prefix_ones(10), prefix_ones(20)],
    "train": [prefix_ones(12), prefix_ones(11), prefix_ones(30)],
    "out_idx": [1, -1], "out_dist": [1, INT_MAX],
    "query2": [prefix_ones(10), prefix_ones(100), prefix_ones(20)],
    "out_idx2": [1, -1, -1], "out_dist2": [1, INT_MAX, INT_MAX],
    "query3": [prefix_ones(8), prefix_ones(8), prefix_ones(40)],
    "train3": [prefix_ones(8), prefix_ones(8), prefix_ones(41), prefix_ones(39)],
    "out_idx3": [0, -1, 2], "out_dist3": [0, INT_MAX, 1],
})

# 6. Lowe ratio (d0 < ratio * d1, strict): train popcounts 0 and 8; query prefix(2): d = (2, 6): 2 < 0.75*6 = 4.5 keep;
#    query prefix(3): d = (3, 5): 3 < 3.75 keep; query prefix(4): d = (4, 4): 4 < 3 no; ratio 0.5: 2 < 3 keep, 3 < 2.5 no.
dump("kat_ratio.json", {
    "doc": "knn=2 + Lowe ratio test",
    "train": [prefix_ones(0), prefix_ones(8)],
    "query": [prefix_ones(2), prefix_ones(3), prefix_ones(4)],
    "keep_075": [True, True, False], "keep_050": [True, False, False],
})

# 7. multi-image train set (BFMatcher.add): order (dist, imgIdx, trainIdx).
#    images: A = [prefix(50), prefix(9)], B = [], C = [prefix(9), prefix(10)].  query prefix(9):
#    exact copies at (A,1) and (C,0) -> imgIdx 0 first; query prefix(10): (C,1) at 0, then (A,1) and (C,0) tie at 1 -> (A,1).
dump("kat_multi_image.json", {
    "doc": "knnMatch(k=2) against a collection of train images",
    "images": [[prefix_ones(50), prefix_ones(9)], [], [prefix_ones(9), prefix_ones(10)]],
    "query": [prefix_ones(9), prefix_ones(10)],
    "img": [[0, 2], [2, 0]], "train": [[1, 0], [1, 1]], "dist": [[0, 0], [0, 1]],
})

# 8. residual / Jacobian (frontend.py:272-291), EuRoC intrinsics (config/orb.yaml:1).
fx, fy, cx, cy = 458.654, 457.296, 367.215, 248.375
#  a) identity pose, point (0,0,1): projection = (cx, cy); e = meas - (cx, cy);
#     X = Y = 0, Z = 1 -> Zinv = 1/(1 + 1e-18) = 1.0 exactly in f64:
#     J = [[0, -fx, 0, -fx, 0, 0], [fy, 0, 0, 0, -fy, 0]]
#  b) identity rotation, t = (0,0,1), point (1, 2, 1): p_c = (1, 2, 2):
#     proj = (fx/2 + cx, fy + cy); Zinv = 0.5, Zinv2 = 0.25:
#     row0 = [fx*1*2*0.25, -fx - fx*0.25, fx*2*0.5, -fx*0.5, 0, fx*0.25] = [fx/2, -1.25 fx, fx, -fx/2, 0, fx/4]
#     row1 = [fy + fy*4*0.25, -fy*2*0.25, -fy*0.5, 0, -fy*0.5, fy*2*0.25] = [2 fy, -fy/2, -fy/2, 0, -fy/2, fy/2]
#     J_point = -A R with A = [[fx/2, 0, -fx/4], [0, fy/2, -fy/2]], R = I
dump("kat_reproj.json", {
    "doc": "EdgeProjectionPoseOnly.compute_error / linearize_oplus at hand-computed points",
    "intrinsics": [fx, fy, cx, cy],
    "poses12": [[1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 1]],
    "points": [[0, 0, 1], [1, 2, 1]],
    "obs_pose": [0, 1], "obs_point": [0, 1],
    "meas": [[370.0, 250.0], [600.0, 700.0]],
    "e": [[370.0 - cx, 250.0 - cy], [600.0 - (fx / 2 + cx), 700.0 - (fy + cy)]],
    "Jpose": [[[0, -fx, 0, -fx, 0, 0], [fy, 0, 0, 0, -fy, 0]],
              [[fx / 2, -1.25 * fx, fx, -fx / 2, 0, fx / 4], [2 * fy, -fy / 2, -fy / 2, 0, -fy / 2, fy / 2]]],
    "Jpoint": [[[-fx, 0, 0], [0, -fy, 0]], [[-fx / 2, 0, fx / 4], [0, -fy / 2, fy / 2]]],
})
print("golden vectors written to", HERE)
