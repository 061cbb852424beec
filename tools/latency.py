"""Host-visible latency of the drop-in match() at the reference's own size (200 x 200, slam.py:23)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))
from feature_matchers import BruteForceFeatureMatcher
import slamhip
bf = BruteForceFeatureMatcher(norm_type=6)
rng = np.random.default_rng(228)
for n in (200, 2000, 20000):
    src = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    qry = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    for f, name in ((lambda: bf.match(src, qry), "match() -> DMatch list"), (lambda: bf.match_arrays(src, qry), "match_arrays()"),
                    (lambda: bf.knn_match_arrays(qry, src, 2), "knn_match_arrays(k=2)")):
        for _ in range(5):
            f()
        t0 = time.perf_counter()
        for _ in range(50):
            f()
        print(f"n={n:6d} {name:28s} {(time.perf_counter() - t0) / 50 * 1e3:8.3f} ms per call")
rm = slamhip.ResidentMatcher()
fr = [rng.integers(0, 256, (200, 32), dtype=np.uint8) for _ in range(60)]
rm.push(fr[0])
t0 = time.perf_counter()
for f in fr[1:]:
    rm.push(f)
print(f"ResidentMatcher.push n=200: {(time.perf_counter() - t0) / 59 * 1e3:.3f} ms per frame")
