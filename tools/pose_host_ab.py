"""The pose-only refinement on host buffers (slam_pose_optimize_host_f64), microseconds per call at 50 / 200 / 500 edges, for the
shipped library or another build (SLAM_LIB=tools/exp/libslamhip_X.so).  Development aid.    python tools/pose_host_ab.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))
from slamhip import _lib
if os.environ.get("SLAM_LIB"): _lib.LIB_PATH=os.path.abspath(os.environ["SLAM_LIB"])
import slamhip
from slamhip.pose_opt import se3_exp, optimize_pose_only_device as optimize_pose_device
rng=np.random.default_rng(3)
for O in (50,200,500):
    X=np.c_[rng.uniform(-4,4,(O,2)),rng.uniform(6,15,O)]
    pix=np.c_[458.654*X[:,0]/X[:,2]+367.215,457.296*X[:,1]/X[:,2]+248.375]+rng.normal(0,0.3,(O,2))
    T0=se3_exp([0.01,-0.01,0.005,0.05,-0.03,0.04])
    f=lambda: optimize_pose_device(T0,X,pix,(458.654,457.296,367.215,248.375))
    for _ in range(50): r=f()
    ts=[]
    for rep in range(5):
        t0=time.perf_counter()
        for _ in range(200): r=f()
        ts.append((time.perf_counter()-t0)/200*1e6)
    print(os.environ.get("SLAM_LIB","shipped"), O, "edges: %.1f us per call (min of 5 x 200; median %.1f)"%(min(ts), sorted(ts)[2]), r.iterations, flush=True)
