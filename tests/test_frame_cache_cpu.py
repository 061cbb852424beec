"""CPU suite: host logic of the drop-in matcher that needs no GPU — the frame cache's hit / miss decisions and buffer
rotation (what it passes to slam_bf_match_host), and the lazy MatchList — on a recording stand-in for the library."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))


class _Lib:
    """slam_malloc / slam_free / slam_bf_match_host as far as FrameCache and match_arrays use them."""

    def __init__(self):
        self.calls, self.next_ptr, self.live = [], 0x10000, set()

    def slam_malloc(self, handle, nbytes, out):
        out._obj.value = self.next_ptr
        self.live.add(self.next_ptr)
        self.next_ptr += 0x10000
        return 0

    def slam_free(self, handle, ptr):
        self.live.discard(ptr)
        return 0

    def slam_bf_match_host(self, handle, hq, n, ht, dt, m, keep, mode, param, qi, ti, dist, cnt):
        self.calls.append({"n": n, "m": m, "h_train": ht is not None, "d_train": dt, "keep": keep, "mode": mode, "param": param})
        cnt._obj.value = 0
        return 0


class _Ctx:
    def __init__(self):
        self.lib, self.handle = _Lib(), 1

    def malloc(self, nbytes):
        from slamhip.device import DeviceBuffer

        p = ctypes.c_void_p()
        self.lib.slam_malloc(self.handle, nbytes, ctypes.byref(p))
        return DeviceBuffer(self, p.value, nbytes)


def _rows(n, seed):
    return np.random.default_rng(seed).integers(0, 256, (n, 32), dtype=np.uint8)


def test_frame_cache_serves_the_previous_query_from_the_device():
    from slamhip import matching as m

    ctx = _Ctx()
    cache = m.FrameCache(ctx)
    f = [_rows(200, 1), _rows(180, 2), _rows(200, 3)]
    m.match_arrays(f[0], f[1], None, cache=cache)                     # nothing remembered yet: both matrices go up
    c = ctx.lib.calls[-1]
    assert c["h_train"] and c["d_train"] is None and c["keep"] is not None and c["mode"] == 0
    kept = c["keep"]
    m.match_arrays(f[1].copy(), f[2], 64.0, cache=cache)              # a fresh copy of the last query: recognised by content
    c = ctx.lib.calls[-1]
    assert not c["h_train"] and c["d_train"] == kept and c["keep"] not in (None, kept) and c["mode"] == 1 and c["param"] == 64.0
    kept2 = c["keep"]
    m.match_arrays(f[2], f[0], None, cache=cache)
    c = ctx.lib.calls[-1]
    assert c["d_train"] == kept2 and c["keep"] == kept                 # two buffers alternate
    assert cache.hits == 2 and cache.calls == 3
    changed = f[0].copy()
    changed[7, 3] ^= 1                                                 # one bit differs: not the remembered matrix
    m.match_arrays(changed, f[1], None, cache=cache)
    assert ctx.lib.calls[-1]["h_train"] and ctx.lib.calls[-1]["d_train"] is None and cache.hits == 2
    # an empty current frame forgets; an empty source is not looked up
    m.match_arrays(f[1], np.array([]), None, cache=cache)
    assert ctx.lib.calls[-1]["n"] == 0 and ctx.lib.calls[-1]["keep"] is None
    m.match_arrays(np.array([]), f[1], None, cache=cache)
    assert ctx.lib.calls[-1]["m"] == 0 and not ctx.lib.calls[-1]["h_train"] and ctx.lib.calls[-1]["keep"] is not None
    m.match_arrays(f[1], f[2], None, cache=cache)
    assert ctx.lib.calls[-1]["d_train"] is not None                     # ... but its query rows were kept for the next call
    # growth: a larger frame gets a larger buffer, the old one is freed
    big = _rows(5000, 9)
    m.match_arrays(f[2], big, None, cache=cache)
    m.match_arrays(big, f[0], None, cache=cache)
    assert ctx.lib.calls[-1]["d_train"] is not None and cache.hits >= 4
    # matrices beyond MAX_ROWS bypass the cache entirely
    calls = cache.calls
    huge = np.zeros((m.FrameCache.MAX_ROWS + 1, 32), np.uint8)
    m.match_arrays(huge, f[0], None, ctx=ctx, cache=cache)
    assert cache.calls == calls and ctx.lib.calls[-1]["keep"] is None
    cache.free()
    assert not ctx.lib.live


def test_match_list_is_lazy_and_sequence_like():
    import feature_matchers as fm

    qi, ti = np.array([0, 2, 5], np.int32), np.array([7, 8, 9], np.int32)
    d = np.array([1.0, 2.0, 3.0], np.float32)
    ml = fm.MatchList(qi, ti, d)
    assert len(ml) == 3 and ml._objs is None                           # len() builds nothing
    one = ml[1]
    assert (one.queryIdx, one.trainIdx, one.imgIdx, one.distance) == (2, 8, 0, 2.0) and ml._objs is None
    assert ml[-1].trainIdx == 9
    assert [m.trainIdx for m in ml] == [7, 8, 9] and ml._objs is not None   # first iteration materialises once
    assert ml[0] is ml._objs[0] and [m.queryIdx for m in ml[1:]] == [2, 5]
    assert [m for m in ml if m.distance < 2.5] == ml._objs[:2]          # the reference's own filter idiom (feature_matchers.py:43)
    assert isinstance(ml[0].distance, float) and isinstance(ml[0].queryIdx, int)
    assert fm.MatchList(qi[:0], ti[:0], d[:0]) == [] and not fm.MatchList(qi[:0], ti[:0], d[:0])
    assert list(reversed(ml))[0].queryIdx == 5 and ml.index(ml[2]) == 2
    assert "3 matches" in repr(ml)
