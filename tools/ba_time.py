"""Windowed BA: five LM steps end to end in its three forms - numpy Schur on GPU residuals (bundle_adjust), the
host-driven device reduction (bundle_adjust_device: slam_ba_reduce_f64 + numpy solve per trial) and the whole loop in ONE
launch (bundle_adjust_one_launch: slam_ba_optimize_f64) - plus the reduction kernels alone; then one sparse window at the
scale of BASELINE configs[4] (200 poses x 50 000 points, ~10^6 observations) through the host-driven device form."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "slam-experiments_amd"))
from slamhip.device import default_context
from slamhip.ba import bundle_adjust, bundle_adjust_device, bundle_adjust_one_launch, SchurProblem
from slamhip.pose_opt import se3_exp
from scipy.spatial.transform import Rotation
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import oracle

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
    for name, fn in (("host", bundle_adjust), ("device", bundle_adjust_device), ("one launch", bundle_adjust_one_launch)):
        if name == "host" and L > 5000:
            continue
        if name == "one launch" and K - 2 > 16:
            continue
        fn(T0, X0, op, ol, meas, (FX, FY, CX, CY), iterations=1, fixed_poses=(0, 1), ctx=ctx)
        t = time.perf_counter()
        r = fn(T0, X0, op, ol, meas, (FX, FY, CX, CY), iterations=5, fixed_poses=(0, 1), ctx=ctx)
        out[name] = (time.perf_counter() - t, r.iterations, r.chi2_final)
    # the same five steps in plain C on ONE host core (oracle/ba_lm_oracle.c: test infrastructure, the CPU baseline of this path)
    if L <= 5000:
        p0 = np.ascontiguousarray(T0[:, :3, :4]).reshape(K, 12)
        oracle.ba_lm_c(p0, X0, op, ol, meas, FX, FY, CX, CY, 1, (0, 1), 0.0)
        t = time.perf_counter()
        rc = oracle.ba_lm_c(p0, X0, op, ol, meas, FX, FY, CX, CY, 5, (0, 1), 0.0)
        out["plain C on one host core"] = (time.perf_counter() - t, rc[4], rc[3])
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
          "; ".join(f"{n}: {v[0]*1e3:.2f} ms for {v[1]} accepted steps (chi2 {v[2]:.1f})" for n, v in out.items()), flush=True)

# the kernel of the one-launch form alone (device time, everything already resident), at the reference's window
import ctypes
K, L = 7, 1400
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
O = len(op)
pt_obs = np.argsort(ol, kind="stable").astype(np.int32); ps_obs = np.argsort(op, kind="stable").astype(np.int32)
pt_ptr = np.zeros(L + 1, np.int32); pt_ptr[1:] = np.cumsum(np.bincount(ol, minlength=L))
ps_ptr = np.zeros(K + 1, np.int32); ps_ptr[1:] = np.cumsum(np.bincount(op, minlength=K))
free = np.arange(2, K, dtype=np.int32)
d = [ctx.upload(a) for a in (op, ol, meas, pt_ptr, pt_obs, ps_ptr, ps_obs, free)]
state_T = np.concatenate([T0[:, :3, :4].reshape(-1), np.zeros(K * 12)]); state_X = np.concatenate([X0.reshape(-1), np.zeros(L * 3)])
dT, dX = ctx.upload(state_T), ctx.upload(state_X)
need = ctypes.c_uint64(0)
ctx.lib.slam_ba_optimize_workspace(K, L, O, ctypes.byref(need))
dW, dS = ctx.malloc(need.value), ctx.malloc(64)
for iters in (0, 1, 5, 10):
    times = []
    for rep in range(6):
        dT.upload(state_T); dX.upload(state_X)
        ctx.sync(); ctx.timer_start()
        assert ctx.lib.slam_ba_optimize_f64(ctx.handle, K, L, O, *[b.ptr for b in d], len(free), FX, FY, CX, CY, 0.0, iters, dT.ptr, dX.ptr,
                                            dW.ptr, need.value, dS.ptr) == 0
        times.append(ctx.timer_stop())
    st = dS.download(np.float64, (8,))
    print(f"slam_ba_optimize_f64 K={K} L={L} O={O} iterations={iters}: {min(times[1:]) * 1e3:.1f} us device time, {int(st[2])} accepted steps, "
          f"{int(st[3])} trials, cost {st[0]:.1f} -> {st[1]:.1f}", flush=True)

# BASELINE configs[4] scale, sparse: 200 poses x 50 000 points, ~20 observations per point (host-driven device form; 1194 x 1194 host solve)
K, L = 200, 50000
rng = np.random.default_rng(228)
T = np.tile(np.eye(4), (K, 1, 1))
T[:, :3, :3] = Rotation.from_rotvec(rng.uniform(-0.1, 0.1, (K, 3))).as_matrix()
T[:, :3, 3] = rng.uniform(-0.5, 0.5, (K, 3))
X = np.c_[rng.uniform(-6, 6, (L, 2)), rng.uniform(8, 20, L)]
ol = np.repeat(np.arange(L), 20).astype(np.int32)
op = np.concatenate([rng.choice(K, 20, replace=False) for _ in range(L)]).astype(np.int32)
pc = np.einsum("oij,oj->oi", T[op, :3, :3], X[ol]) + T[op, :3, 3]
meas = np.c_[FX * pc[:, 0] / pc[:, 2] + CX, FY * pc[:, 1] / pc[:, 2] + CY] + rng.normal(0, 0.3, (len(op), 2))
T0 = np.stack([T[0]] + [se3_exp(rng.normal(0, 0.005, 6)) @ T[k] for k in range(1, K)])
X0 = X + rng.normal(0, 0.03, X.shape)
t = time.perf_counter()
r = bundle_adjust_device(T0, X0, op, ol, meas, (FX, FY, CX, CY), iterations=3, fixed_poses=(0,), ctx=ctx)
dt = time.perf_counter() - t
print(f"sparse window K={K} L={L} O={len(op)}: host-driven device form, {r.iterations} accepted steps in {dt:.2f} s "
      f"(cost {r.chi2_initial:.0f} -> {r.chi2_final:.0f})", flush=True)
