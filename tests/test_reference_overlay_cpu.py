"""Build-container only: the overlay install against the REAL reference files.

``tests/overlay_probe.py`` runs in a child process with ``slam-experiments_amd/`` ahead of ``/root/reference`` on
``sys.path`` and name-only stubs for the three wheels the image lacks (cv2, g2o, jaxlie); this file asserts on what it
observed.  Skipped wherever the reference is absent (the GPU box): nothing of the reference is copied or shipped.
The stubs compute nothing, so this pins NAMES and CONTRACTS (imports, argument order, attributes, ``__slots__``,
write-back), not arithmetic: parity of the hot path stays "unpinned" (DESIGN.md)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "slam-experiments_amd")
REFERENCE = os.environ.get("SLAM_REFERENCE_DIR", "/root/reference")

pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REFERENCE, "slam.py")),
                                reason="the reference checkout is only present in the build container")


@pytest.fixture(scope="module")
def probe(built):
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "overlay_probe.py"), REFERENCE], capture_output=True,
                       text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("OVERLAY_PROBE ")]
    assert line, r.stdout[-2000:]
    return json.loads(line[-1][len("OVERLAY_PROBE "):])


def test_reference_imports_resolve_to_the_overlay(probe):
    """slam.py:9-13 and frontend.py:14-17 (`from backend import Backend, Map`, `from feature_matchers import ...`) work
    unchanged: matcher and Backend come from the overlay, Map (class attributes NUM_ACTIVE_KEYFRAMES / MIN_DIST_THRESHOLD,
    backend.py:10-12) and everything else from the reference."""
    o = probe["origins"]
    for name in ("slam", "frontend", "primitives", "slam.Map"):
        assert o[name] == REFERENCE, (name, o[name])
    for name in ("slam.BruteForceFeatureMatcher", "slam.FeatureMatcher", "frontend.FeatureMatcher", "slam.Backend", "frontend.Backend"):
        assert o[name] == PKG, (name, o[name])
    assert probe["same_objects"] and probe["backend_zero_arg"]
    assert probe["map_constants"] == [7, 0.2]
    assert probe["map_slots"] == ["_landmarks", "_keyframes", "_active_landmarks", "_active_keyframes", "_current_frame"]


def test_reference_frontend_runs_on_the_overlay_matcher(probe):
    """OrbSLAM.__init__ (slam.py:22-28) builds the overlay's matcher with cv2.NORM_HAMMING and hands it to the reference's
    Frontend; _match_features / _track_current_frame (frontend.py:156-187) consume what match() returns."""
    assert probe["matcher_in_frontend"] == "feature_matchers.BruteForceFeatureMatcher" and probe["matcher_is_FeatureMatcher"]
    assert probe["orb_nfeatures"] == 200 and probe["first_frame_features"] == 40
    assert probe["match_type"] == "MatchList" and probe["match_len"] == 40 and probe["match_equals_oracle"]
    assert probe["match_elem_type"].endswith(".DMatch") and probe["match_imgIdx"] == [0]     # cv2.DMatch when cv2 imports
    assert probe["propagated"] == 20 and probe["propagation_equals_oracle"]                  # map points follow trainIdx->queryIdx
    assert probe["mask_rectangles"] == 40
    assert probe["filtered_equals_oracle"] and probe["filtered_len"][0] <= probe["filtered_len"][1]
    assert probe["empty_desc"] == ["float64", [0]] and probe["empty_matches"] == 0           # primitives.py:200-205 with n = 0


def test_consecutive_frames_reuse_the_resident_rows(probe):
    """frontend.py:181-187 with real Frame.get_descriptors copies: frame k's query rows are frame k+1's source rows and
    are not sent again."""
    calls = probe["device_calls"]
    assert [c["train_from_device"] for c in calls] == [False, False, True, False, False]
    assert [c["uploaded_rows"] for c in calls] == [80, 80, 40, 40, 80]
    assert [c["mode"] for c in calls] == [0, 0, 0, 0, 1] and calls[-1]["param"] == 30.0
    assert probe["cache_hits"] == 1


def test_backend_glue_on_the_reference_containers(probe):
    """Backend.optimize_map / correct_frame_pose on real Frame / Feature / MapPoint / Map objects
    (primitives.py:92-205, backend.py:10-53): what is read (Feature.position int32 pixels, MapPoint.position,
    Map._active_keyframes by keyframe_id) and what is written back (Frame.set_pose with the pose's own class,
    MapPoint.set_position, Feature.map_point / is_outlier)."""
    ba = probe["ba"]
    assert ba["iterations"] == 1 and ba["fixed"] == [0] and ba["dtypes"] == ["float64", "float64", "int32", "float64"]
    assert ba["pose_x"] == ba["want_pose_x"] == [7.0, 3.0, 5.0]
    assert ba["points"] == [[1, 1, 1], [4, 4, 4]]                       # landmarks a and d; b (one view) and c (two in one frame) skipped
    assert ba["op"] == [0, 1, 0, 1, 2] and ba["ol"] == [0, 0, 1, 1, 1]
    assert ba["meas"] == [[10, 11], [12, 13], [44, 45], [42, 43], [40, 41]]          # KeyPoint.pt truncated to int32
    assert ba["pose_types"] == ["SE3", "SE3", "SE3"] and ba["pose_after"] == [7.0, 103.0, 105.0]   # the fixed one keeps its pose
    assert ba["a"] == [1.5] * 3 and ba["d"] == [4.5] * 3 and ba["b"] == [2.0] * 3 and ba["c"] == [3.0] * 3
    assert ba["active_keyframes"] == 3
    # the public getter cannot be used: it deep-copies Frames that hold a Lock (backend.py:49-50, primitives.py:182)
    assert ba["getter"].startswith("TypeError") or ba["getter"] == "copy"
    po = probe["pose_only"]
    assert po["inliers"] == 2 and po["px"] == [[1, 2], [5, 6], [7, 8]] and po["position_dtype"] == "int32"
    assert po["pp"] == [[1.5] * 3, [4.5] * 3, [2.0] * 3]
    assert po["pose_type"] == "SE3" and po["pose_x"] == 42.0
    assert po["map_points"] == [100, None, None, 101]                   # the outlier edge (second of three) loses its landmark
    assert po["outlier_flags"] == [False, True, False, False]           # flags cleared on edges only (frontend.py:388-391)
    assert po["no_landmarks"] == 0


def _class_members(tree):
    """{class name: names it offers} for every ClassDef in a syntax tree: ``__slots__`` entries, ``self.x`` assigned in
    ``__init__``, methods and properties.  Later definitions of the same name are merged (the tests re-declare them)."""
    import ast

    out = {}
    for node in ast.walk(tree):
        if not isinstance(node, ast.ClassDef):
            continue
        names = out.setdefault(node.name, set())
        for item in node.body:
            if isinstance(item, (ast.FunctionDef, ast.AsyncFunctionDef)):
                if item.name != "__init__":
                    names.add(item.name)
                else:
                    for sub in ast.walk(item):
                        if isinstance(sub, ast.Attribute) and isinstance(sub.value, ast.Name) and sub.value.id == "self" \
                                and isinstance(sub.ctx, ast.Store):
                            names.add(sub.attr)
            elif isinstance(item, ast.Assign) and any(isinstance(t, ast.Name) and t.id == "__slots__" for t in item.targets):
                names.update(ast.literal_eval(item.value))
    return out


def test_hand_made_stand_ins_match_the_reference_classes():
    """The duck-typed Frame / MapPoint / Feature / Map stand-ins that the GPU tests and the host tests declare by hand
    (they must run where the reference is absent) may only use names the reference's classes really have
    (primitives.py:92-205, backend.py:10-53).  Compared on the syntax trees; nothing is imported."""
    import ast

    real = {}
    for fname in ("primitives.py", "backend.py"):
        with open(os.path.join(REFERENCE, fname)) as f:
            for k, v in _class_members(ast.parse(f.read())).items():
                real.setdefault(k, set()).update(v)
    checked = 0
    for fname in ("test_optimize_gpu.py", "test_abi_and_host_cpu.py", "test_frontend_pattern_gpu.py"):
        with open(os.path.join(ROOT, "tests", fname)) as f:
            mine = _class_members(ast.parse(f.read()))
        for cls in ("Frame", "MapPoint", "Feature", "Map"):
            if cls in mine:
                extra = mine[cls] - real[cls]
                assert not extra, f"{fname}: stand-in {cls} uses {sorted(extra)}, which the reference's {cls} does not have"
                checked += 1
    assert checked >= 6
