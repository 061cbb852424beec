"""GPU: the C ABI reports misuse through status codes + slam_last_error(), never by crashing."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_status_codes_and_messages(gpu_ctx):
    import slamhip
    from slamhip._lib import SlamHipError, check

    lib, ctx = gpu_ctx.lib, gpu_ctx
    h = ctypes.c_void_p()
    assert lib.slam_ctx_create(99, ctypes.byref(h)) == -1 and b"out of range" in lib.slam_last_error()
    assert lib.slam_ctx_create(0, None) == -1
    q = ctx.upload(np.zeros((8, 32), np.uint8))
    out = ctx.malloc(8 * 8)
    # misaligned descriptor pointer
    assert lib.slam_bf_knn2_u256(ctx.handle, q.ptr + 4, 4, q.ptr, 8, 0, out.ptr, out.ptr) == -1
    assert b"16-byte aligned" in lib.slam_last_error()
    # negative sizes, null outputs, oversized train_base
    assert lib.slam_bf_knn2_u256(ctx.handle, q.ptr, -1, q.ptr, 8, 0, out.ptr, out.ptr) == -1
    assert lib.slam_bf_knn2_u256(ctx.handle, q.ptr, 8, q.ptr, 8, 0, None, out.ptr) == -1
    assert lib.slam_bf_knn2_u256(ctx.handle, q.ptr, 8, q.ptr, 8, 2**31 - 4, out.ptr, out.ptr) == -1
    assert lib.slam_bf_match_filter(ctx.handle, out.ptr, out.ptr, 8, 7, 0.0, out.ptr, None, None) == -1
    knobs = lambda *v: (ctypes.c_int32 * len(v))(*v)
    assert lib.slam_bf_set_tuning(ctx.handle, knobs(3), 1) == -1                       # R not in {1, 2, 4, 8}
    assert lib.slam_bf_set_tuning(None, knobs(1), 1) == -1 and lib.slam_bf_set_tuning(ctx.handle, knobs(1, 0, -2), 3) == -1
    assert lib.slam_bf_set_tuning(ctx.handle, knobs(1, 0, 64, 48), 4) == -1            # leader chunk not a multiple of 32
    assert lib.slam_bf_set_tuning(ctx.handle, None, 3) == -1 and lib.slam_bf_set_tuning(ctx.handle, knobs(*[0] * 11), 11) == -1
    assert lib.slam_bf_set_tuning(ctx.handle, knobs(2, 0, 0, 0, 0, 1), 6) == -1        # the SGPR feed holds one query per lane
    assert lib.slam_bf_set_tuning(ctx.handle, knobs(1, 0, 0, 0, 0, 2), 6) == -1        # feed not in {-1, 0, 1}
    assert lib.slam_bf_set_tuning(ctx.handle, None, 0) == 0                            # reset to the shipped plan
    plan = (ctypes.c_int32 * 10)()
    assert lib.slam_bf_plan_info(ctx.handle, 65536, 65536, plan) == 0
    assert plan[0] == 1 and plan[1] == 256 and plan[4] % 32 == 0 and plan[3] >= 8 and plan[8] == 1   # long chunks: SGPR feed
    assert lib.slam_bf_plan_info(ctx.handle, 200, 200, plan) == 0 and plan[4] == 0 and plan[9] == 128  # frame-sized: no leaders, unfiltered start
    assert lib.slam_bf_plan_info(ctx.handle, 0, 5, plan) == -1
    assert lib.slam_bf_reset_state(ctx.handle) == 0 and lib.slam_bf_reset_state(None) == -1
    # freeing a pointer the context does not own
    assert lib.slam_free(ctx.handle, 0x1000) == -1 and b"not owned" in lib.slam_last_error()
    with pytest.raises(SlamHipError):
        check(lib.slam_free(ctx.handle, 0x1000))
    # N == 0 is a no-op, not an error
    assert lib.slam_bf_knn2_u256(ctx.handle, None, 0, None, 0, 0, None, None) == 0
    q.free(); out.free()
    # python-side validation happens before the FFI call
    with pytest.raises(ValueError):
        slamhip.knn_match_arrays(np.zeros((2, 32), np.uint8), np.zeros((2, 32), np.uint8), k=3)


def test_two_contexts_are_independent(gpu_ctx):
    import slamhip
    from oracle import oracle

    a, b = slamhip.Context(0), slamhip.Context(0)
    try:
        q = np.random.default_rng(1).integers(0, 256, (500, 32), dtype=np.uint8)
        t = np.random.default_rng(2).integers(0, 256, (900, 32), dtype=np.uint8)
        ra = slamhip.knn_match_arrays(q, t, 2, ctx=a)
        rb = slamhip.knn_match_arrays(t, q, 2, ctx=b)
        assert np.array_equal(ra[0], oracle.bf_knn_c(q, t, 2)[0]) and np.array_equal(rb[0], oracle.bf_knn_c(t, q, 2)[0])
    finally:
        a.close(); b.close()


def test_host_buffer_entry_points(gpu_ctx):
    """slam_bf_match_host / slam_bf_knn2_u256_host / slam_pose_optimize_host_f64: argument checks, the empty
    cases, and agreement with the device-pointer entry points they wrap."""
    from oracle import oracle

    lib, ctx = gpu_ctx.lib, gpu_ctx
    rng = np.random.default_rng(5)
    q = rng.integers(0, 256, (37, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (53, 32), dtype=np.uint8)
    qi, ti, di = np.empty(37, np.int32), np.empty(37, np.int32), np.empty(37, np.float32)
    cnt = ctypes.c_int64(-1)
    dt = ctx.upload(t)
    args = (qi.ctypes.data, ti.ctypes.data, di.ctypes.data, ctypes.byref(cnt))
    # both a host and a device train pointer, a bad mode, a missing count pointer, missing train rows
    assert lib.slam_bf_match_host(ctx.handle, q.ctypes.data, 37, t.ctypes.data, dt.ptr, 53, None, 0, 0.0, *args) == -1
    assert b"not both" in lib.slam_last_error()
    assert lib.slam_bf_match_host(ctx.handle, q.ctypes.data, 37, t.ctypes.data, None, 53, None, 4, 0.0, *args) == -1
    assert lib.slam_bf_match_host(ctx.handle, q.ctypes.data, 37, t.ctypes.data, None, 53, None, 0, 0.0, *args[:3], None) == -1
    assert lib.slam_bf_match_host(ctx.handle, q.ctypes.data, 37, None, None, 53, None, 0, 0.0, *args) == -1
    # empty sides: no matches, not an error
    assert lib.slam_bf_match_host(ctx.handle, None, 0, t.ctypes.data, None, 53, None, 0, 0.0, *args) == 0 and cnt.value == 0
    assert lib.slam_bf_match_host(ctx.handle, q.ctypes.data, 37, None, None, 0, None, 1, 30.0, *args) == 0 and cnt.value == 0
    # host train, device train and the kept query rows give the same matches as the oracle
    exp_i, exp_d = oracle.bf_knn_c(q, t, 2)
    oq, ot, od = oracle.bf_match_c(t, q, 64.0)          # the reference filter, restated in C
    dkeep = ctx.malloc(37 * 32)
    for h_train, d_train in ((t.ctypes.data, None), (None, dt.ptr)):
        assert lib.slam_bf_match_host(ctx.handle, q.ctypes.data, 37, h_train, d_train, 53, dkeep.ptr, 0, 0.0, *args) == 0
        assert cnt.value == 37 and np.array_equal(qi, np.arange(37)) and np.array_equal(ti, exp_i[:, 0])
        assert np.array_equal(di, exp_d[:, 0].astype(np.float32))
        assert np.array_equal(dkeep.download(np.uint8, (37, 32)), q)
        assert lib.slam_bf_match_host(ctx.handle, q.ctypes.data, 37, h_train, d_train, 53, None, 1, 64.0, *args) == 0
        lim = max(2.0 * exp_d[:, 0].min(), 64.0)
        sel = exp_d[:, 0] < lim
        c = cnt.value
        assert c == sel.sum() and np.array_equal(qi[:c], np.flatnonzero(sel)) and np.array_equal(ti[:c], exp_i[sel, 0])
        assert np.array_equal(qi[:c], oq) and np.array_equal(ti[:c], ot) and np.array_equal(di[:c], od)
    idx, dist = np.empty((37, 2), np.int32), np.empty((37, 2), np.int32)
    assert lib.slam_bf_knn2_u256_host(ctx.handle, q.ctypes.data, 37, t.ctypes.data, 53, idx.ctypes.data, dist.ctypes.data) == 0
    assert np.array_equal(idx, exp_i) and np.array_equal(dist, exp_d)
    assert lib.slam_bf_knn2_u256_host(ctx.handle, q.ctypes.data, 37, None, 0, idx.ctypes.data, dist.ctypes.data) == 0
    assert (idx == -1).all() and (dist == 2**31 - 1).all()
    assert lib.slam_bf_knn2_u256_host(ctx.handle, q.ctypes.data, 37, None, 5, idx.ctypes.data, dist.ctypes.data) == -1
    # a larger call after a small one grows the arena; results stay right
    q2 = rng.integers(0, 256, (5000, 32), dtype=np.uint8)
    i2, d2 = np.empty((5000, 2), np.int32), np.empty((5000, 2), np.int32)
    assert lib.slam_bf_knn2_u256_host(ctx.handle, q2.ctypes.data, 5000, t.ctypes.data, 53, i2.ctypes.data, d2.ctypes.data) == 0
    e2i, e2d = oracle.bf_knn_c(q2, t, 2)
    assert np.array_equal(i2, e2i) and np.array_equal(d2, e2d)
    dt.free(); dkeep.free()
    # pose optimisation on host buffers: null pointers are rejected, O == 0 returns the input pose
    pose = np.eye(4)[:3].reshape(12).copy()
    out, stats = np.empty(12), np.zeros(2, np.int32)
    assert lib.slam_pose_optimize_host_f64(ctx.handle, pose.ctypes.data, None, None, 5, 1.0, 1.0, 0.0, 0.0, 4, 10, 35.9, 1.0,
                                           out.ctypes.data, None, None, stats.ctypes.data) == -1
    assert lib.slam_pose_optimize_host_f64(ctx.handle, pose.ctypes.data, None, None, 0, 1.0, 1.0, 0.0, 0.0, 4, 10, 35.9, 1.0,
                                           out.ctypes.data, None, None, stats.ctypes.data) == 0
    assert np.array_equal(out, pose) and stats[0] == 0


def test_host_calls_from_two_threads_share_a_context(gpu_ctx):
    """ctypes releases the GIL during a call; the frontend thread and a backend thread of the reference
    (slam.py:26-35) may use the default context at once.  Host-buffer calls serialise on the context."""
    import threading

    import slamhip
    from oracle import oracle

    rng = np.random.default_rng(9)
    sets = [(rng.integers(0, 256, (150 + 37 * i, 32), dtype=np.uint8), rng.integers(0, 256, (90 + 11 * i, 32), dtype=np.uint8))
            for i in range(4)]
    expect = [oracle.bf_match_c(t, q, 40.0) for q, t in sets]
    errors = []

    def worker(k):
        try:
            for it in range(60):
                q, t = sets[(k + it) % 4]
                got = slamhip.match_arrays(t, q, 40.0, ctx=gpu_ctx)
                exp = expect[(k + it) % 4]
                if not all(np.array_equal(a, b) for a, b in zip(got, exp)):
                    errors.append((k, it))
        except Exception as exc:   # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]


@pytest.mark.parametrize("n,m", [(37, 53), (300, 120), (5000, 4500), (1, 9), (64, 1)])
def test_cross_check_in_one_call(gpu_ctx, n, m):
    """slam_bf_match_host mode 3 (crossCheck) against the oracle (OpenCV 4.x crosscheck branch: mutual nearest neighbours), host and device train."""
    from oracle import oracle

    lib, ctx = gpu_ctx.lib, gpu_ctx
    rng = np.random.default_rng(n + 7 * m)
    q = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (m, 32), dtype=np.uint8)
    t[: min(m, n) // 2] = q[: min(m, n) // 2]                    # mutual nearest neighbours exist
    oi, od = oracle.bf_cross_check_c(q, t)
    eq, et, ed = np.flatnonzero(oi >= 0), oi[oi >= 0], od[oi >= 0]
    qi, ti, di = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.float32)
    cnt = ctypes.c_int64(-1)
    dt = ctx.upload(t)
    for h_train, d_train in ((t.ctypes.data, None), (None, dt.ptr)):
        assert lib.slam_bf_match_host(ctx.handle, q.ctypes.data, n, h_train, d_train, m, None, 3, 0.0, qi.ctypes.data,
                                      ti.ctypes.data, di.ctypes.data, ctypes.byref(cnt)) == 0
        c = cnt.value
        assert c == len(eq) and np.array_equal(qi[:c], eq) and np.array_equal(ti[:c], et)
        assert np.array_equal(di[:c], np.asarray(ed, np.float32))
    dt.free()


def test_out_of_range_indices_are_reported_not_dereferenced(gpu_ctx):
    """slam_reproj_rj_f64 with pose / point indices outside [0, K) / [0, L): no fault, NaN outputs for exactly those
    observations, and slam_index_errors counts them (then clears)."""
    lib, ctx = gpu_ctx.lib, gpu_ctx
    K, L, O = 3, 10, 700
    rng = np.random.default_rng(5)
    poses = np.tile(np.eye(4)[:3, :4].reshape(12), (K, 1))
    points = np.c_[rng.uniform(-1, 1, (L, 2)), rng.uniform(4, 9, L)]
    op = rng.integers(0, K, O).astype(np.int32)
    ol = rng.integers(0, L, O).astype(np.int32)
    bad = np.array([0, 5, 64, 255, 256, 699])
    op[bad[:3]] = [K, -1, 2**31 - 1]
    ol[bad[3:]] = [L, -7, 1 << 30]
    meas = rng.uniform(0, 700, (O, 2))
    d = [ctx.upload(a) for a in (poses, points, op, ol, meas)]
    e, Jp, Jq = ctx.malloc(O * 16), ctx.malloc(O * 96), ctx.malloc(O * 48)
    cnt = ctypes.c_int64(-1)
    assert lib.slam_index_errors(ctx.handle, ctypes.byref(cnt)) == 0          # clear whatever earlier tests left
    for with_point in (True, False):
        assert lib.slam_reproj_rj_f64(ctx.handle, d[0].ptr, K, d[1].ptr, L, d[2].ptr, d[3].ptr, d[4].ptr, O, 458.0, 457.0,
                                      367.0, 248.0, e.ptr, Jp.ptr, Jq.ptr if with_point else None) == 0
        he = e.download(np.float64, (O, 2))
        isbad = np.zeros(O, bool)
        isbad[bad] = True
        assert np.isnan(he[isbad]).all() and np.isfinite(he[~isbad]).all()
        assert lib.slam_index_errors(ctx.handle, ctypes.byref(cnt)) == 0 and cnt.value == len(bad)
        assert lib.slam_index_errors(ctx.handle, ctypes.byref(cnt)) == 0 and cnt.value == 0
    for b in (*d, e, Jp, Jq):
        b.free()


def test_batch_pose_refinement_with_a_bad_offsets_table_is_reported_not_dereferenced(gpu_ctx):
    """slam_pose_optimize_batch_f64 with a table that is not ascending / leaves [0, O_total]: no access outside the
    arrays (the frames shrink to what lies inside), the per-context counter says how many frames were affected, and
    the frames of a correct table next to them are refined as usual."""
    import ctypes

    ctx, lib = gpu_ctx, gpu_ctx.lib
    rng = np.random.default_rng(5)
    O = 300
    X = np.c_[rng.uniform(-4, 4, (O, 2)), rng.uniform(6, 15, O)]
    meas = np.c_[458.654 * X[:, 0] / X[:, 2] + 367.215, 457.296 * X[:, 1] / X[:, 2] + 248.375]
    pose = np.tile(np.eye(4)[:3, :4].reshape(12), (4, 1))
    bufs = [ctx.upload(pose), ctx.upload(X), ctx.upload(meas), ctx.malloc(4 * 96), ctx.malloc(O), ctx.malloc(O * 8), ctx.malloc(32)]
    d_pose, d_pts, d_mes, d_out, d_inl, d_chi, d_st = bufs
    n = ctypes.c_int64(-1)
    assert lib.slam_index_errors(ctx.handle, ctypes.byref(n)) == 0          # clear
    for table, bad_frames in (([0, 100, 200, 250, 300], 0), ([0, 100, 50, 400, 300], 3), ([-7, 100, 200, 1 << 30, 300], 3)):
        d_off = ctx.upload(np.array(table, np.int32))
        assert lib.slam_pose_optimize_batch_f64(ctx.handle, 4, d_pose.ptr, d_pts.ptr, d_mes.ptr, d_off.ptr, O, 458.654, 457.296,
                                                367.215, 248.375, 4, 10, 5.991 ** 2, 1.0, d_out.ptr, d_inl.ptr, d_chi.ptr,
                                                d_st.ptr) == 0
        ctx.sync()
        assert lib.slam_index_errors(ctx.handle, ctypes.byref(n)) == 0 and n.value == bad_frames, (table, n.value)
        st = d_st.download(np.int32, (4, 2))
        assert st[0, 0] == min(max(table[1], 0), O) - max(table[0], 0)      # frame 0 is sane in every table: all inliers
        d_off.free()
    for b in bufs:
        b.free()
