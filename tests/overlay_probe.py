"""Build-container probe: does the overlay install import and run under the REAL reference files?

Run as ``python tests/overlay_probe.py /root/reference`` (by ``tests/test_reference_overlay_cpu.py``, in a child process
so that the stub modules below never leak into the test session).  It puts ``slam-experiments_amd/`` and then the
reference directory on ``sys.path`` (INTEGRATION.md §2's install), imports the reference's own ``frontend`` and ``slam``
modules, and drives the reference's own classes (``Frontend``, ``OrbSLAM``, ``Frame``, ``Feature``, ``MapPoint``,
``Map``; ``/root/reference/frontend.py``, ``slam.py``, ``primitives.py``, ``backend.py``) against the overlay's
``BruteForceFeatureMatcher`` and ``Backend``.  Prints one JSON object of observations; the test asserts on it.

What the stubs are and are not: ``cv2``, ``g2o`` and ``jaxlie`` are not installed in this image, so the reference cannot
be imported as is.  The three modules installed in ``sys.modules`` here carry only the NAMES the reference binds at import
time plus inert data carriers (``KeyPoint.pt``, a 4x4-matrix ``SE3``); every computational entry point raises
``StubCalled``.  They are import-wiring stand-ins, NOT an oracle: no result of this probe pins the arithmetic of the hot
path (parity stays "unpinned"), it pins names, argument order, attribute and ``__slots__`` contracts.

The GPU is replaced by a host "device" that keeps the C ABI's contract for ``slam_bf_match_host`` (raw pointers, the
``d_train`` / ``keep_query`` residency arguments) and answers with the CPU oracle, so ``slamhip.matching`` and the
``FrameCache`` run for real.  Nothing from the reference is copied, and this file does nothing without the reference
directory (absent on the GPU box)."""
from __future__ import annotations

import ctypes
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "slam-experiments_amd")


class StubCalled(RuntimeError):
    pass


def _refuse(name):
    def f(*a, **k):
        raise StubCalled(f"{name} is a name-only stub")
    f.__name__ = name.rsplit(".", 1)[-1]
    return f


# ---------------------------------------------------------------------------------------------------------------------
# stubs: names the reference binds at import (feature_matchers.py:6, feature_detectors.py:5, frontend.py:9-12,253,262,
# primitives.py:7-8, utils.py:3-5, slam.py:5-7) and the data carriers its classes store
# ---------------------------------------------------------------------------------------------------------------------
def install_stubs(script):
    cv2 = types.ModuleType("cv2")

    class KeyPoint:
        def __init__(self, x=0.0, y=0.0, size=31.0):
            self.pt, self.size = (float(x), float(y)), size

    class DMatch:                                   # same constructor order as cv2.DMatch(queryIdx, trainIdx, imgIdx, distance)
        def __init__(self, queryIdx=-1, trainIdx=-1, imgIdx=-1, distance=float("inf")):
            self.queryIdx, self.trainIdx, self.imgIdx, self.distance = queryIdx, trainIdx, imgIdx, distance

    class _ScriptedOrb:                             # hands out the probe's synthetic keypoints/descriptors frame by frame
        def __init__(self, nfeatures):
            self.nfeatures, self.calls = nfeatures, []

        def detect(self, img, mask=None):
            return self.detectAndCompute(img, mask)[0]

        def detectAndCompute(self, img, mask=None):
            kps, desc = script[len(self.calls)]
            self.calls.append((img.shape, None if mask is None else (mask.shape, mask.dtype.str)))
            return kps, desc

    class ORB:
        last = None

        @staticmethod
        def create(nfeatures=500):
            ORB.last = _ScriptedOrb(nfeatures)
            return ORB.last

    rectangles = []

    def rectangle(mask, pt1, pt2, color, thickness):   # records the call, paints nothing
        rectangles.append((tuple(int(v) for v in pt1), tuple(int(v) for v in pt2), color, thickness))
        return mask

    cv2.KeyPoint, cv2.DMatch, cv2.ORB = KeyPoint, DMatch, ORB
    cv2.NORM_HAMMING, cv2.FILLED = 6, -1
    cv2.rectangle = rectangle
    cv2._rectangles = rectangles
    for name in ("BFMatcher", "drawMatches", "imshow", "findEssentialMat", "recoverPose", "triangulatePoints"):
        setattr(cv2, name, _refuse("cv2." + name))

    g2o = types.ModuleType("g2o")
    for name in ("EdgeSE3ProjectXYZOnlyPose", "VariableVectorXEdge"):      # base classes frontend.py:253,262 subclasses
        setattr(g2o, name, type(name, (), {"__init__": _refuse("g2o." + name)}))
    for name in ("OptimizationAlgorithmLevenberg", "BlockSolverSE3", "LinearSolverDenseSE3", "SparseOptimizer", "VertexSE3",
                 "Isometry3d", "RobustKernelHuber"):
        setattr(g2o, name, _refuse("g2o." + name))

    jaxlie = types.ModuleType("jaxlie")

    class SE3:                                      # a 4x4 matrix behind jaxlie.SE3's method names; no Lie algebra
        def __init__(self, T):
            self._T = np.array(T, np.float64).reshape(4, 4)

        @classmethod
        def identity(cls):
            return cls(np.eye(4))

        @classmethod
        def from_matrix(cls, T):
            return cls(T)

        def as_matrix(self):
            return self._T.copy()

        def inverse(self):
            return SE3(np.linalg.inv(self._T))

        def __matmul__(self, other):
            if isinstance(other, SE3):
                return SE3(self._T @ other._T)
            p = np.asarray(other, np.float64)
            return self._T[:3, :3] @ p + self._T[:3, 3]

        log = _refuse("jaxlie.SE3.log")

    jaxlie.SE3 = SE3
    jaxlie.SO3 = type("SO3", (), {"from_matrix": _refuse("jaxlie.SO3.from_matrix")})
    sys.modules.update(cv2=cv2, g2o=g2o, jaxlie=jaxlie)
    return cv2, g2o, jaxlie


# ---------------------------------------------------------------------------------------------------------------------
# a host "device": the slam_bf_match_host contract (include/slamhip.h) answered by the CPU oracle
# ---------------------------------------------------------------------------------------------------------------------
class HostDeviceLib:
    def __init__(self):
        self.calls = []
        self.mem = {}

    def slam_bf_match_host(self, handle, q_ptr, n, t_ptr, d_train, m, keep, mode, param, qi_ptr, ti_ptr, dist_ptr, cnt):
        from oracle import oracle

        def rows(ptr, count):
            ptr = getattr(ptr, "value", ptr)
            if not count:
                return np.zeros((0, 32), np.uint8)
            return np.ctypeslib.as_array((ctypes.c_uint8 * (count * 32)).from_address(ptr)).reshape(count, 32).copy()

        q = rows(q_ptr, n)
        t = rows(d_train if d_train else t_ptr, m)
        self.calls.append(dict(n=n, m=m, train_from_device=bool(d_train), uploaded_rows=n + (0 if d_train else m), mode=mode, param=param))
        if keep and n:
            ctypes.memmove(keep, q.ctypes.data, q.nbytes)
        oq, ot, od = oracle.bf_match_c(t, q, param if mode == 1 else None)
        assert mode in (0, 1)
        c = len(oq)
        for ptr, arr, ty in ((qi_ptr, oq, np.int32), (ti_ptr, ot, np.int32), (dist_ptr, od, np.float32)):
            a = np.ascontiguousarray(arr, ty)
            if c:
                ctypes.memmove(ptr, a.ctypes.data, a.nbytes)
        cnt._obj.value = c
        return 0

    def slam_malloc(self, handle, nbytes, out):
        buf = ctypes.create_string_buffer(nbytes)
        addr = ctypes.addressof(buf)
        self.mem[addr] = buf
        out._obj.value = addr
        return 0

    def slam_free(self, handle, ptr):
        self.mem.pop(getattr(ptr, "value", ptr))
        return 0

    def slam_last_error(self):
        return b"host device"


def main(ref_dir: str) -> dict:
    rng = np.random.default_rng(228)                 # the reference's own seed, main.py:65
    n_feat = 40

    def synth(seed_rows=None, flips=6):
        """(keypoints, descriptors): fresh random rows, or noisy permuted copies of ``seed_rows``."""
        if seed_rows is None:
            desc = rng.integers(0, 256, (n_feat, 32), dtype=np.uint8)
        else:
            perm = rng.permutation(len(seed_rows))
            bits = np.unpackbits(seed_rows[perm], axis=1)
            for r in bits:
                r[rng.choice(256, flips, replace=False)] ^= 1
            desc = np.packbits(bits, axis=1)
        kps = [sys.modules["cv2"].KeyPoint(20 + 13.7 * i, 15 + 9.3 * i) for i in range(len(desc))]
        return kps, desc

    script = []
    cv2, g2o, jaxlie = install_stubs(script)
    first = synth()
    script += [first, synth(first[1]), synth(first[1]), synth(first[1])]

    sys.path[:0] = [PKG, ref_dir, ROOT]              # the overlay first, then the reference
    out = {}
    import slam                                       # the reference's slam.py:1-13 import chain, unchanged
    import frontend
    import backend
    import feature_matchers
    import primitives

    def origin(obj):
        return os.path.dirname(os.path.abspath(sys.modules[obj.__module__].__file__))

    out["origins"] = {
        "slam": os.path.dirname(os.path.abspath(slam.__file__)),
        "frontend": os.path.dirname(os.path.abspath(frontend.__file__)),
        "primitives": os.path.dirname(os.path.abspath(primitives.__file__)),
        "slam.BruteForceFeatureMatcher": origin(slam.BruteForceFeatureMatcher),
        "slam.FeatureMatcher": origin(slam.FeatureMatcher),
        "frontend.FeatureMatcher": origin(frontend.FeatureMatcher),
        "slam.Backend": origin(slam.Backend),
        "frontend.Backend": origin(frontend.Backend),
        "slam.Map": os.path.dirname(os.path.abspath(slam.Map.insert_keyframe.__code__.co_filename)),
    }
    out["same_objects"] = bool(slam.Backend is backend.Backend and frontend.Backend is backend.Backend
                               and slam.Map is backend.Map and frontend.Map is backend.Map
                               and slam.BruteForceFeatureMatcher is feature_matchers.BruteForceFeatureMatcher)
    out["map_constants"] = [slam.Map.NUM_ACTIVE_KEYFRAMES, slam.Map.MIN_DIST_THRESHOLD]
    out["map_slots"] = list(slam.Map.__slots__)
    out["backend_zero_arg"] = isinstance(slam.Backend(), backend.Backend)

    # ---- the host "device" under slamhip.matching ------------------------------------------------------------------
    from slamhip import _lib as L
    from slamhip import device as D
    from slamhip import matching as M

    lib = HostDeviceLib()
    L._lib = lib

    class HostContext(D.Context):
        def __init__(self):
            self.lib, self.handle, self.device = lib, ctypes.c_void_p(1), 0

    hctx = HostContext()
    M.default_context = lambda: hctx
    D.default_context = lambda: hctx

    # ---- OrbSLAM as slam.py:22-28 builds it: reference Frontend + overlay matcher -------------------------------
    camera = primitives.Camera(458.654, 457.296, 367.215, 248.375)          # config/orb.yaml intrinsics
    system = slam.OrbSLAM(camera)
    fe = system.frontend
    out["matcher_in_frontend"] = type(fe._feature_matcher).__module__ + "." + type(fe._feature_matcher).__qualname__
    out["matcher_is_FeatureMatcher"] = isinstance(fe._feature_matcher, frontend.FeatureMatcher)
    out["orb_nfeatures"] = cv2.ORB.last.nfeatures

    img = np.zeros((480, 752), np.uint8)
    system.process(img, 0.0)                         # first frame: _init -> detect, becomes the last frame (frontend.py:108-112)
    out["first_frame_features"] = len(system.frontend.get_last_frame().features)
    last = fe.get_last_frame()

    # second frame through the reference's own _detect_features + _match_features (frontend.py:181-187,231-251)
    cur = primitives.Frame.create_frame(img, 0.05)
    fe._current_frame = cur
    fe._detect_features(True)
    matches = fe._match_features()
    from oracle import oracle

    oq, ot, od = oracle.bf_match_c(last.get_descriptors(), cur.get_descriptors())
    out["match_type"] = type(matches).__name__
    out["match_len"] = len(matches)
    out["match_equals_oracle"] = bool([m.queryIdx for m in matches] == oq.tolist() and [m.trainIdx for m in matches] == ot.tolist()
                                      and [m.distance for m in matches] == od.tolist())
    out["match_elem_type"] = type(matches[0]).__module__ + "." + type(matches[0]).__name__
    out["match_imgIdx"] = sorted({m.imgIdx for m in matches})

    # tracking step: the reference's _track_current_frame (frontend.py:156-179) propagates map points through the matches
    last.set_pose(jaxlie.SE3.identity())
    mps = {}
    for i in range(0, n_feat, 2):                    # every other feature of the last frame has a landmark
        mp = primitives.MapPoint(id=np.uint64(i), position=np.array([0.1 * i, -0.05 * i, 4.0 + i]))
        last.features[i].map_point = mp
        mps[i] = mp
    fe._last_frame = cur                             # what add_frame does after a TRACKING step (frontend.py:99-101) ...
    fe._last_frame = last                            # ... but here frame 2 is matched against the frame that owns the landmarks
    nxt = primitives.Frame.create_frame(img, 0.10)
    fe._current_frame = nxt
    before_rect = len(cv2._rectangles)
    fe._track_current_frame()
    oq, ot, od = oracle.bf_match_c(last.get_descriptors(), nxt.get_descriptors())
    want = {int(q): mps[int(t)] for q, t in zip(oq, ot) if int(t) in mps}
    got = {i: f.map_point for i, f in enumerate(nxt.features) if f.map_point is not None}
    out["propagated"] = len(got)
    out["propagation_equals_oracle"] = bool(got.keys() == want.keys() and all(got[k] is want[k] for k in got))
    out["mask_rectangles"] = len(cv2._rectangles) - before_rect
    out["status_after_track"] = fe.get_status().name

    # frame k's query rows are frame k+1's source rows: second call on consecutive frames is served from the device
    fe._last_frame, fe._current_frame = nxt, primitives.Frame.create_frame(img, 0.15)
    fe._detect_features(True)
    fe._match_features()
    out["device_calls"] = lib.calls
    out["cache_hits"] = fe._feature_matcher._cache.hits

    # empty frame: Frame.get_descriptors gives a float64 (0,) array (primitives.py:200-205) -> no matches, no error
    empty = primitives.Frame.create_frame(img, 0.2)
    fe._last_frame, fe._current_frame = nxt, empty
    out["empty_desc"] = [str(empty.get_descriptors().dtype), list(empty.get_descriptors().shape)]
    out["empty_matches"] = len(fe._match_features())

    # dist_threshold filter through the reference's call shape match(source, query, dist_threshold) (feature_matchers.py:36-44)
    fm = fe._feature_matcher.match(last.get_descriptors(), cur.get_descriptors(), 30.0)
    fq, ft, fd = oracle.bf_match_c(last.get_descriptors(), cur.get_descriptors(), 30.0)
    out["filtered_equals_oracle"] = bool([m.queryIdx for m in fm] == fq.tolist() and [m.trainIdx for m in fm] == ft.tolist())
    out["filtered_len"] = [len(fm), len(matches)]

    # ---- Backend glue on the reference's real containers (primitives.py:92-205, backend.py:10-53) ----------------
    from slamhip.ba import BAResult
    from slamhip.pose_opt import PoseOptResult

    def T(x):
        Mx = np.eye(4)
        Mx[0, 3] = x
        return Mx

    seen = {}

    def fake_optimize(self, poses, points, op, ol, meas, fx, fy, cx, cy, iterations=10, fixed_poses=(0,), huber_delta=0.0,
                      on_device=True):
        seen.update(poses=poses.copy(), points=points.copy(), op=op.tolist(), ol=ol.tolist(), meas=meas.tolist(),
                    fixed=list(fixed_poses), dtypes=[str(poses.dtype), str(points.dtype), str(op.dtype), str(meas.dtype)])
        return BAResult(poses=poses + 100.0, points=points + 0.5, chi2_initial=2.0, chi2_final=1.0, iterations=1)

    def fake_pose(self, pose, points, pixels, fx, fy, cx, cy, rounds=4, iterations=10, on_device=True):
        seen.update(pp=points.tolist(), px=pixels.tolist(), p0=pose.tolist())
        return PoseOptResult(pose=T(42), inliers=np.array([True, False, True]), chi2=np.zeros(3), n_inliers=2, iterations=3)

    backend.Backend.optimize = fake_optimize
    backend.Backend.optimize_pose = fake_pose

    KP, Feature, MapPoint, Frame = cv2.KeyPoint, primitives.Feature, primitives.MapPoint, primitives.Frame
    m = slam.Map()
    frames = []
    for x in (7.0, 3.0, 5.0):                        # keyframe ids are handed out by make_keyframe in this order
        f = Frame.create_frame(img, x)
        f.set_pose(jaxlie.SE3(T(x)))
        frames.append(f)
    stale = Frame.create_frame(img, 1.0)
    stale.set_pose(jaxlie.SE3(T(1.0)))
    a, b, c, d = (MapPoint(id=np.uint64(100 + i), position=np.array(p, float)) for i, p in
                  enumerate(([1, 1, 1], [2, 2, 2], [3, 3, 3], [4, 4, 4])))

    def feat(frame, px, mp):
        ft = Feature(frame, KP(px[0] + 0.75, px[1] + 0.25), rng.integers(0, 256, 32, dtype=np.uint8))   # .position truncates
        ft.map_point = mp
        frame.features.append(ft)
        return ft

    feat(frames[0], (10, 11), a); feat(frames[1], (12, 13), a); feat(stale, (0, 0), a)
    feat(frames[2], (20, 21), b)                                               # one view in the window: skipped
    feat(frames[1], (30, 31), c); feat(frames[1], (32, 33), c)                 # two features of one frame: skipped
    feat(frames[2], (40, 41), d); feat(frames[1], (42, 43), d); feat(frames[0], (44, 45), d)
    stale.make_keyframe()                            # adds its observations to the landmarks, never enters the window
    for f in frames:
        f.make_keyframe()                            # Frame.make_keyframe registers the observations (primitives.py:191-198)
        m.insert_keyframe(f)
    for mp in (a, b, c, d):
        m.insert_map_point(mp)
    kf_ids = [int(f.keyframe_id) for f in frames]

    def getter_outcome(map_):
        """Map.get_active_keyframes deep-copies (backend.py:49-50) objects that hold a threading.Lock (primitives.py:182):
        record what that does, it is why optimize_map reads ``_active_keyframes`` itself."""
        try:
            return "copy" if map_.get_active_keyframes()[frames[0].keyframe_id] is not frames[0] else "same object"
        except TypeError as exc:
            return f"TypeError: {exc}"

    res = backend.Backend().optimize_map(m, 1.0, 1.0, 0.0, 0.0, n_fixed=1)
    order = np.argsort(kf_ids)                       # optimize_map sorts the window by keyframe_id
    out["ba"] = dict(
        iterations=res.iterations, fixed=seen["fixed"], dtypes=seen["dtypes"],
        pose_x=[float(P[0, 3]) for P in seen["poses"]], want_pose_x=[[7.0, 3.0, 5.0][i] for i in order],
        points=seen["points"].tolist(), op=seen["op"], ol=seen["ol"], meas=seen["meas"],
        pose_types=[type(f.pose).__name__ for f in frames],
        pose_after=[float(f.pose.as_matrix()[0, 3]) for f in frames],
        a=a.position.tolist(), b=b.position.tolist(), c=c.position.tolist(), d=d.position.tolist(),
        active_keyframes=len(m._active_keyframes), getter=getter_outcome(m),
    )

    fr = Frame.create_frame(img, 9.0)
    fr.set_pose(jaxlie.SE3(T(0.0)))
    feat(fr, (1, 2), a); feat(fr, (3, 4), None); feat(fr, (5, 6), d); feat(fr, (7, 8), b)
    fr.features[1].is_outlier = True
    n_in = backend.Backend().correct_frame_pose(fr, 1.0, 1.0, 0.0, 0.0)
    out["pose_only"] = dict(
        inliers=n_in, px=seen["px"], pp=seen["pp"], pose_type=type(fr.pose).__name__, pose_x=float(fr.pose.as_matrix()[0, 3]),
        map_points=[None if f.map_point is None else int(f.map_point.id) for f in fr.features],
        outlier_flags=[bool(f.is_outlier) for f in fr.features],
        position_dtype=str(fr.features[0].position.dtype),
        no_landmarks=backend.Backend().correct_frame_pose(Frame.create_frame(img, 10.0), 1.0, 1.0, 0.0, 0.0),
    )
    return out


if __name__ == "__main__":
    print("OVERLAY_PROBE " + json.dumps(main(sys.argv[1])))
