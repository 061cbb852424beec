"""GPU: the drop-in modules used the way the reference's tracking loop uses them (frontend.py:143-187, 298-393),
on a synthetic scene (cv2 / g2o are absent, so images and ORB are replaced by known 3-D points + descriptors)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
FX, FY, CX, CY = 458.654, 457.296, 367.215, 248.375


def test_tracking_loop_pattern(gpu_ctx):
    from backend import Backend
    from feature_matchers import BruteForceFeatureMatcher
    from slamhip.pose_opt import se3_exp

    rng = np.random.default_rng(228)
    n_feat = 200                                              # OrbFeatureDetector(n_features=200), slam.py:23
    world = np.c_[rng.uniform(-4, 4, (n_feat, 2)), rng.uniform(6, 15, n_feat)]
    desc_world = rng.integers(0, 256, (n_feat, 32), dtype=np.uint8)

    def observe(T):
        """One 'frame': every landmark seen, descriptor = the landmark's with a few flipped bits, shuffled order."""
        order = rng.permutation(n_feat)
        d = desc_world[order].copy()
        flips = rng.integers(0, 256, (n_feat, 6))
        for i in range(n_feat):
            for b in flips[i]:
                d[i, b // 8] ^= np.uint8(1 << (b % 8))
        pc = world[order] @ T[:3, :3].T + T[:3, 3]
        px = np.c_[FX * pc[:, 0] / pc[:, 2] + CX, FY * pc[:, 1] / pc[:, 2] + CY] + rng.normal(0, 0.3, (n_feat, 2))
        return order, d, px.astype(np.int32).astype(np.float64)   # Feature.position is int-truncated (primitives.py:110-112)

    matcher = BruteForceFeatureMatcher(norm_type=6)           # cv2.NORM_HAMMING, slam.py:24
    backend = Backend()
    T_true = np.eye(4)
    last_order, last_desc, _ = observe(T_true)
    last_map_point = last_order.copy()                        # feature i of the last frame -> landmark id
    pose = np.eye(4)

    def pcie_bytes():
        import ctypes

        up, down = ctypes.c_uint64(0), ctypes.c_uint64(0)
        assert gpu_ctx.lib.slam_io_counters(gpu_ctx.handle, ctypes.byref(up), ctypes.byref(down)) == 0
        return up.value

    import slamhip

    for frame in range(1, 9):
        T_true = se3_exp([0.004, -0.003, 0.002, 0.03, 0.01, 0.02]) @ T_true
        order, desc, px = observe(T_true)
        before = pcie_bytes()
        # fresh copies on every call, as Frame.get_descriptors makes them (primitives.py:200-205)
        matches = matcher.match(last_desc.copy(), desc.copy())   # (source = last frame, query = current), frontend.py:187
        sent = pcie_bytes() - before
        # f2: the last frame's rows are already on the device from the previous call: ONE descriptor matrix per frame
        assert sent == (2 if frame == 1 else 1) * n_feat * 32, (frame, sent)
        assert len(matches) == n_feat
        sq, st, sd = slamhip.match_arrays(last_desc, desc, ctx=gpu_ctx)           # stateless path: both matrices go up
        assert np.array_equal(matches.queryIdx, sq) and np.array_equal(matches.trainIdx, st) and np.array_equal(matches.distance, sd)
        cur_map_point = np.full(n_feat, -1)
        for m in matches:                                     # frontend.py:174-177
            cur_map_point[m.queryIdx] = last_map_point[m.trainIdx]
        assert (cur_map_point == order).mean() > 0.97         # 6 flipped bits of 256: nearly every match is the right landmark
        have = cur_map_point >= 0
        res = backend.optimize_pose(pose, world[cur_map_point[have]], px[have], FX, FY, CX, CY)   # frontend.py:146
        pose = res.pose
        d = pose @ np.linalg.inv(T_true)
        assert np.linalg.norm(d[:3, 3]) < 0.03 and res.n_inliers >= 0.9 * have.sum()
        last_desc, last_map_point = desc, cur_map_point
    assert matcher._cache.hits == 7 and matcher._cache.calls == 8
    # a source matrix that is NOT the previous query (tracking was re-initialised from another keyframe,
    # frontend.py:223-229) is simply uploaded; an empty frame in between is handled too
    other = rng.integers(0, 256, (150, 32), dtype=np.uint8)
    before = pcie_bytes()
    m1 = matcher.match(other, desc)
    assert pcie_bytes() - before == (150 + n_feat) * 32 and len(m1) == n_feat
    assert len(matcher.match(desc, np.array([]))) == 0                                  # Frame with no features: (0,) float64
    m2 = matcher.match(np.array([]), desc)
    assert len(m2) == 0
    before = pcie_bytes()
    m3 = matcher.match(desc, other, 64.0)
    assert pcie_bytes() - before == 150 * 32                                           # desc was the last query again
    eq, et, ed = slamhip.match_arrays(desc, other, 64.0, ctx=gpu_ctx)
    assert np.array_equal(m3.queryIdx, eq) and np.array_equal(m3.trainIdx, et) and np.array_equal(m3.distance, ed)
