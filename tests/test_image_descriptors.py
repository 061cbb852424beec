"""Real descriptor statistics: 600 + 600 binary descriptors of the two images that ship with the reference (its main.py
demo pair), made by tests/golden/make_image_descriptors.py (an ORB-like extractor in numpy; cv2 is absent, so these are
inputs, not OpenCV outputs).  Unlike uniform random bytes they have correlated bits, near-duplicates and dozens of tied
nearest distances.  CPU: the two oracle restatements against each other on them.  GPU: every matching entry point of the
HIP path against the oracle, bit-exact, in the argument order the frontend uses (last frame = source, current = query)."""
import os

import numpy as np
import pytest

from oracle import oracle

FIX = os.path.join(os.path.dirname(__file__), "golden", "image_descriptors.npz")


@pytest.fixture(scope="module")
def frames():
    z = np.load(FIX)                      # plain arrays, allow_pickle stays False
    return z["desc1"], z["desc2"], z["kp1"], z["kp2"]


def test_fixture_has_the_structure_the_tests_rely_on(frames):
    d1, d2, kp1, kp2 = frames
    assert d1.shape == (600, 32) and d2.shape == (600, 32) and d1.dtype == np.uint8 and kp1.shape == (600, 2)
    dist = oracle.hamming_matrix_np(d2, d1)
    s = np.sort(dist, 1)
    assert (s[:, 0] == s[:, 1]).sum() >= 20            # rows whose nearest distance is tied: the index rule decides
    assert dist.min() < 10 and np.median(s[:, 0]) < 60  # real correspondences, far below the 128 of random rows


def test_oracle_restatements_agree_on_real_descriptors(frames):
    d1, d2, _, _ = frames
    for k in (1, 2, 3):
        a, b = oracle.bf_knn_c(d2, d1, k, threads=4), oracle.bf_knn_np(d2, d1, k)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for thr in (None, 10.0, 30.0, 64.0):
        a, b = oracle.bf_match_c(d1, d2, thr), oracle.bf_match_np(d1, d2, thr)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), thr
    a, b = oracle.bf_cross_check_c(d2, d1), oracle.bf_cross_check_np(d2, d1)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and (a[0] >= 0).sum() > 200


@pytest.mark.gpu
def test_hip_path_on_real_descriptors(gpu_ctx, frames):
    import slamhip
    from feature_matchers import BruteForceFeatureMatcher

    d1, d2, kp1, kp2 = frames
    ei, ed = oracle.bf_knn_c(d2, d1, 2, threads=4)
    gi, gd = slamhip.knn_match_arrays(d2, d1, 2, ctx=gpu_ctx)
    assert np.array_equal(gi, ei) and np.array_equal(gd, ed)
    bf = BruteForceFeatureMatcher(norm_type=6)
    for thr in (None, 10.0, 30.0, 64.0):
        ms = bf.match(d1, d2, thr)                              # (source = frame 1, query = frame 2), frontend.py:187
        oq, ot, od = oracle.bf_match_c(d1, d2, thr)
        assert [m.queryIdx for m in ms] == oq.tolist() and [m.trainIdx for m in ms] == ot.tolist()
        assert [m.distance for m in ms] == od.tolist()
    keep = oracle.bf_ratio_c(ei, ed, 0.75)
    rq, rt, rd = slamhip.ratio_test_arrays(d2, d1, 0.75, ctx=gpu_ctx)
    assert np.array_equal(rq, np.flatnonzero(keep)) and np.array_equal(rt, ei[keep, 0]) and keep.sum() > 100
    oi, od = oracle.bf_cross_check_c(d2, d1)
    cq, ct, cd = slamhip.cross_check_arrays(d2, d1, ctx=gpu_ctx)
    assert np.array_equal(cq, np.flatnonzero(oi >= 0)) and np.array_equal(ct, oi[oi >= 0])
    assert np.array_equal(cd, od[oi >= 0].astype(np.float32))
    # what the matches are for: the mutual pairs agree on one image motion (the two frames show the same desk)
    shift = kp2[cq] - kp1[ct]
    med = np.median(shift, 0)
    assert (np.abs(shift - med).max(1) < 60).mean() > 0.6
    # each image against itself: every row finds itself first at distance 0 (ties with exact duplicates go to the lower row)
    si, sd = slamhip.knn_match_arrays(d1, d1, 2, ctx=gpu_ctx)
    ri, rd2 = oracle.bf_knn_c(d1, d1, 2)
    assert np.array_equal(si, ri) and np.array_equal(sd, rd2) and (sd[:, 0] == 0).all()
    # both frames as a two-image collection (loop-closure layout): order (dist, imgIdx, trainIdx)
    img, loc, dist = slamhip.knn_match_collection(d2, [d1, d2], 2, ctx=gpu_ctx)
    eimg, eloc, edist = oracle.bf_knn_multi_c(d2, [d1, d2], 2)
    assert np.array_equal(img, eimg) and np.array_equal(loc, eloc) and np.array_equal(dist, edist)
    assert (img[:, 0] == 1).all() and (dist[:, 0] == 0).all()
