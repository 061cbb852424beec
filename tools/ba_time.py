"""Time one LM trial of the windowed BA: host numpy Schur (bundle_adjust) vs device reduction (bundle_adjust_device)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "slam-experiments_amd"))
from slamhip.device import default_context
from slamhip.ba import bundle_adjust, bundle_adjust_device, SchurProblem
from slamhip.pose_opt import se3_exp
from scipy.spatial.transform import Rotation

FX, FY, CX, CY = 458.654, 457.296, 367.215, 248.375
ctx = default_context()
for K, L in ((7, 1400), (16, 5000), (32, 20000)):
    rng = np.random.default_rng(K)
    T = np.tile(np.eye(4), (K, 1, 1))
    T[:, :3, :3] = Rotation.from_rotvec(rng.uniform(-0.15, 0.15, (K, 3))).as_matrix()
    T[:, :3, 3] = rng.uniform(-0.5, 0.5, (K, 3))
    X = np.c_[rng.uniform(-4, 4, (L, 2)), rng.uniform(6, 15, L)]
    op = np.repeat(np.arange(K), L).astype(np.int32); ol = np.tile(np.arange(L), K).astype(np.int32)
    keep = rng.uniform(size=K * L) < 0.6
    op, ol = op[keep], ol[keep]
    pc = np.einsum("oij,oj->oi", T[op, :3, :3], X[ol]) + T[op, :3, 3]
    meas = np.c_[FX * pc[:, 0] / pc[:, 2] + CX, FY * pc[:, 1] / pc[:, 2] + CY] + rng.normal(0, 0.2, (len(op), 2))
    T0 = np.stack([T[0], T[1]] + [se3_exp(rng.normal(0, 0.01, 6)) @ T[k] for k in range(2, K)])
    X0 = X + rng.normal(0, 0.05, X.shape)
    out = {}
    for name, fn in (("host", bundle_adjust), ("device", bundle_adjust_device)):
        if name == "host" and L > 5000:
            continue
        fn(T0, X0, op, ol, meas, (FX, FY, CX, CY), iterations=1, fixed_poses=(0, 1), ctx=ctx)
        t = time.perf_counter()
        r = fn(T0, X0, op, ol, meas, (FX, FY, CX, CY), iterations=5, fixed_poses=(0, 1), ctx=ctx)
        out[name] = (time.perf_counter() - t, r.iterations, r.chi2_final)
    sp = SchurProblem(ctx, K, L, op, ol, meas, (FX, FY, CX, CY))
    p12 = T0[:, :3, :4].reshape(K, 12)
    sp.reduce(p12, X0, 0.0, 1.0)
    lib, h = ctx.lib, ctx.handle
    ctx.sync(); ctx.timer_start()
    for _ in range(20):
        lib.slam_ba_reduce_f64(h, sp.d_poses.ptr, K, sp.d_points.ptr, L, sp.d_op.ptr, sp.d_ol.ptr, sp.d_meas.ptr, sp.O,
                               sp.d_pt_ptr.ptr, sp.d_pt_obs.ptr, sp.d_ps_ptr.ptr, sp.d_ps_obs.ptr, sp.d_lookup.ptr,
                               FX, FY, CX, CY, 0.0, 1.0, sp.d_rec.ptr, sp.d_E.ptr, sp.d_bl.ptr, sp.d_Hpp.ptr, sp.d_bp.ptr,
                               sp.d_ybl.ptr, sp.d_cost.ptr, sp.d_W.ptr, sp.d_hll.ptr)
    ms = ctx.timer_stop() / 20
    sp.free()
    print(f"K={K} L={L} O={len(op)}: reduce kernels {ms*1e3:.1f} us/call; " +
          "; ".join(f"{n}: {v[0]*1e3:.1f} ms for {v[1]} accepted steps (chi2 {v[2]:.1f})" for n, v in out.items()), flush=True)
