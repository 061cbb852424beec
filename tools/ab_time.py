"""A/B device time of the top-2 search across several builds of libslamhip.so, interleaved in ONE process
(cdna_hip_programming.md rule 24: N variants x M rounds, report the distribution).  Development aid.

    python tools/ab_time.py LIB[:k=v...][,LIB...] [NxM ...] [--rounds R] [--reps K] [--check]

A library may carry tuning knobs of slam_bf_set_tuning, e.g. path/libslamhip.so:R=2:bpc=16 (R, bpc, lead, leadchunk, tail, feed, cold, chunk, queue).

Each library is dlopen'ed privately (RTLD_LOCAL; the builds export the same symbols), gets its own context and its
own copy of the inputs.  Per round and library: K back-to-back searches between two HIP events.  --check compares
every library's table with the first one's (bit-exact).
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))
from slamhip._lib import SIGNATURES  # noqa: E402


def opt(name, default):
    if name in sys.argv:
        i = sys.argv.index(name)
        v = sys.argv[i + 1]
        del sys.argv[i:i + 2]
        return type(default)(v)
    return default


rounds, reps = opt("--rounds", 7), opt("--reps", 0)
check = "--check" in sys.argv
if check:
    sys.argv.remove("--check")
libs = sys.argv[1].split(",")
sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[2:]] or [(65536, 65536), (8192, 65536), (4096, 4096)]


class Lib:
    def __init__(self, spec):
        path, *kv = spec.split(":")
        knobs = dict((k, int(v)) for k, v in (x.split("=") for x in kv))
        self.name = (os.path.basename(path).replace("libslamhip", "").replace(".so", "") or "shipped") + "".join(":" + x for x in kv)
        self.lib = ctypes.CDLL(os.path.abspath(path), mode=os.RTLD_LOCAL | os.RTLD_NOW)
        for n, (res, args) in SIGNATURES.items():
            fn = getattr(self.lib, n, None)
            if fn is not None:
                fn.restype, fn.argtypes = res, args
        self.ctx = ctypes.c_void_p()
        assert self.lib.slam_ctx_create(0, ctypes.byref(self.ctx)) == 0, self.lib.slam_last_error()
        if knobs:
            names = ("R", "bpc", "lead", "leadchunk", "tail", "feed", "cold", "chunk", "queue", "merge")
            nk = 10 if "merge" in knobs else 9 if "queue" in knobs else (8 if ("cold" in knobs or "chunk" in knobs) else 6)      # older builds know six / eight knobs
            k = (ctypes.c_int32 * nk)(*(knobs.get(n, 0) for n in names[:nk]))
            assert self.lib.slam_bf_set_tuning(self.ctx, k, nk) == 0, self.lib.slam_last_error()

    def malloc(self, nbytes):
        p = ctypes.c_void_p()
        assert self.lib.slam_malloc(self.ctx, nbytes, ctypes.byref(p)) == 0
        return p

    def upload(self, a):
        a = np.ascontiguousarray(a)
        p = self.malloc(a.nbytes)
        assert self.lib.slam_upload(self.ctx, p, a.ctypes.data, a.nbytes) == 0
        return p

    def download(self, p, dtype, shape):
        out = np.empty(shape, dtype)
        assert self.lib.slam_download(self.ctx, out.ctypes.data, p, out.nbytes) == 0
        return out


L = [Lib(p) for p in libs]
for n, m in sizes:
    q = np.random.default_rng(228).integers(0, 256, (n, 32), dtype=np.uint8)
    t = np.random.default_rng(229).integers(0, 256, (m, 32), dtype=np.uint8)
    k = reps or (200 if n * m < 1 << 28 else (60 if n * m < 1 << 31 else 30))
    st = []
    for l in L:
        dq, dt, di, dd = l.upload(q), l.upload(t), l.malloc(n * 8), l.malloc(n * 8)
        st.append((dq, dt, di, dd))
    times = [[] for _ in L]
    ms = ctypes.c_float(0)
    for r in range(rounds + 1):                       # round 0 = spin-up, dropped
        for i, l in enumerate(L):
            dq, dt, di, dd = st[i]
            for _ in range(k if r else 3 * k):
                rc = l.lib.slam_bf_knn2_u256(l.ctx, dq, n, dt, m, 0, di, dd)
                assert rc == 0, l.lib.slam_last_error()
            l.lib.slam_sync(l.ctx)
            l.lib.slam_timer_start(l.ctx)
            for _ in range(k):
                l.lib.slam_bf_knn2_u256(l.ctx, dq, n, dt, m, 0, di, dd)
            l.lib.slam_timer_stop(l.ctx, ctypes.byref(ms))
            if r:
                times[i].append(ms.value / k * 1e3)
    ref = None
    for i, l in enumerate(L):
        same = ""
        if check:
            tab = (l.download(st[i][2], np.int32, (n, 2)), l.download(st[i][3], np.int32, (n, 2)))
            if ref is None:
                ref = tab
            same = "  table == first library's: %s" % bool(np.array_equal(tab[0], ref[0]) and np.array_equal(tab[1], ref[1]))
        a = np.array(times[i])
        print(f"{n}x{m} {l.name:>22}: median {np.median(a):9.2f} us  min {a.min():9.2f}  max {a.max():9.2f}  ({rounds} rounds x {k} launches){same}", flush=True)
        for p in st[i]:
            l.lib.slam_free(l.ctx, p)
