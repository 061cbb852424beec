"""Run the top-2 search NxM `reps` times on device-resident rows (a target for rocprofv3 passes on sizes other than
the bench's; development aid).    python tools/run_search.py NxM [reps] [knob=value,...]   (knobs of Context.set_tuning)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))
import slamhip  # noqa: E402
from slamhip import _lib  # noqa: E402

if os.environ.get("SLAM_LIB"):                            # another build of the library (tools/exp/...)
    _lib.LIB_PATH = os.path.abspath(os.environ["SLAM_LIB"])

n, m = (int(v) for v in sys.argv[1].split("x"))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ctx = slamhip.default_context()
if len(sys.argv) > 3 and sys.argv[3]:
    ctx.set_tuning(**{k: int(v) for k, v in (kv.split("=") for kv in sys.argv[3].split(","))})
q = slamhip.DeviceDescriptors(ctx, np.random.default_rng(228).integers(0, 256, (n, 32), dtype=np.uint8))
t = slamhip.DeviceDescriptors(ctx, np.random.default_rng(229).integers(0, 256, (m, 32), dtype=np.uint8))
tab = slamhip.Top2Table(ctx, n)
for _ in range(reps):
    slamhip.knn2_device(ctx, q.buf, n, t.buf, m, tab.idx, tab.dist)
ctx.sync()
print(n, m, reps, ctx.plan_info(n, m))
