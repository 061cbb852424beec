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
    assert lib.slam_bf_set_tuning(3, 0) == -1
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
