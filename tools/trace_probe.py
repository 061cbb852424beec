"""Per-block timeline of one top-2 launch (development aid; needs the experiment build tools/exp/libslamhip_trace.so,
which stamps wall_clock64() at block start / after the prologue / after the scan / at the end).

    python tools/trace_probe.py NxM [lead_rows] [blocks_per_cu] [tail] [queue]        (0 = shipped choice, -1 = off)

Build the experiment library first: tools/build_exp.sh
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))
import slamhip  # noqa: E402
from slamhip import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "tools", "exp", "libslamhip_trace.so")
n, m = (int(v) for v in sys.argv[1].split("x"))
seed = int(sys.argv[2]) if len(sys.argv) > 2 else -1
bpc = int(sys.argv[3]) if len(sys.argv) > 3 else 0
tail = int(sys.argv[4]) if len(sys.argv) > 4 else 0
queue = int(sys.argv[5]) if len(sys.argv) > 5 else 0
ctx = slamhip.Context(0)
lib, h = ctx.lib, ctx.handle
lib.slam_exp_set_trace.argtypes = [ctypes.c_void_p]
ctx.set_tuning(blocks_per_cu=bpc, lead_rows=seed, tail=tail, queue=queue)
plan = (ctypes.c_int32 * 10)()
lib.slam_bf_plan_info(h, n, m, plan)
workers = slamhip.plan_describe(n, m, num_cu=plan[7], blocks_per_cu=bpc, lead_rows=seed, tail=tail, queue=queue)[0]["workers"]
blocks = plan[1] * (workers or plan[3])           # a queue plan launches `workers` blocks per query block
q = slamhip.DeviceDescriptors(ctx, np.random.default_rng(228).integers(0, 256, (n, 32), dtype=np.uint8))
t = slamhip.DeviceDescriptors(ctx, np.random.default_rng(229).integers(0, 256, (m, 32), dtype=np.uint8))
tab = slamhip.Top2Table(ctx, n)
trace = ctx.malloc(blocks * 40)                     # four stamps a block, then one placement word a block
f = lambda: lib.slam_bf_knn2_u256(h, q.buf.ptr, n, t.buf.ptr, m, 0, tab.idx.ptr, tab.dist.ptr)
for _ in range(30):
    f()
ctx.sync()
assert lib.slam_exp_set_trace(trace.ptr) == 0
ctx.timer_start()
f()
ms = ctx.timer_stop()
raw = trace.download(np.uint64, (blocks * 5,))
tr = raw[:blocks * 4].reshape(blocks, 4).astype(np.int64)
where = raw[blocks * 4:]
TICK = 0.01   # wall_clock64 runs at 100 MHz: 10 ns per tick, in us
t0 = tr[:, 0].min()
start, pro, scan, end = ((tr[:, i] - t0) * TICK for i in range(4))
span = end.max()
print(f"{n}x{m} plan chunk={plan[2]} S={plan[3]} workers={workers} qblocks={plan[1]} blocks={blocks} seed={plan[4]}: event time {ms * 1e3:.1f} us, "
      f"first block start -> last block end {span:.1f} us")
dur = end - start
slots = plan[7] * 6                                # resident blocks of the SGPR-fed kernel (6 waves per SIMD)
print(f"block duration: mean {dur.mean():.1f} us, p5 {np.percentile(dur, 5):.1f}, p50 {np.percentile(dur, 50):.1f}, "
      f"p95 {np.percentile(dur, 95):.1f}, max {dur.max():.1f}; prologue mean {np.mean(pro - start):.2f} us, "
      f"scan mean {np.mean(scan - pro):.1f} us, epilogue mean {np.mean(end - scan):.2f} us (max {np.max(end - scan):.1f})")
print(f"occupancy: sum(block time) / (span * {slots} slots) = {dur.sum() / (span * slots):.3f}")
order = np.arange(blocks)
for r in range(0, (blocks + slots - 1) // slots):
    sel = (order >= r * slots) & (order < (r + 1) * slots)
    print(f"  dispatch round {r}: start {start[sel].min():8.1f}..{start[sel].max():8.1f}  end {end[sel].min():8.1f}..{end[sel].max():8.1f}  "
          f"mean duration {dur[sel].mean():7.1f}")
bins = np.linspace(0, span, 41)
act = [(np.minimum(end, b1) - np.maximum(start, b0)).clip(0).sum() / (b1 - b0) for b0, b1 in zip(bins[:-1], bins[1:])]
if workers:
    for name, v in (("start", start), ("prologue end", pro), ("wave 0 out of tickets", scan), ("block end", end)):
        print(f"  {name:>22}: min {v.min():7.1f}  p5 {np.percentile(v, 5):7.1f}  p50 {np.percentile(v, 50):7.1f}  p95 {np.percentile(v, 95):7.1f}  max {v.max():7.1f} us")
if workers:
    qb = plan[1]
    bxs, bys = order % qb, order // qb
    per_q = np.array([scan[bxs == b].max() for b in range(qb)])       # when the queue of a query block ran dry
    print(f"  queue of a query block runs dry: min {per_q.min():.1f} p50 {np.median(per_q):.1f} max {per_q.max():.1f} us; "
          f"by (query block mod 8): " + " ".join(f"{per_q[np.arange(qb) % 8 == x].mean():.1f}" for x in range(8)))
    print("  wave 0 out of tickets, mean by worker index: " + " ".join(f"{scan[bys == k].mean():.1f}" for k in range(0, workers, max(1, workers // 12))))
print("active blocks per 1/40 of the span:", " ".join(f"{a:.0f}" for a in act))
full = np.array(act) >= 0.97 * min(slots, blocks)
print(f"time below 97 % of full residency: head {bins[1:][full][0] - bins[1] if full.any() else span:.1f} us, "
      f"tail {span - bins[1:][full][-1] if full.any() else span:.1f} us")

# Where the blocks ran: HW_REG_HW_ID (gfx9 layout: cu_id bits 11:8, sh_id 12, se_id 15:13) and HW_REG_XCC_ID (bits 3:0).
hw, xcc = (where & 0xFFFFFFFF).astype(np.int64), (where >> 32).astype(np.int64) & 0xF
cu = xcc * 4096 + ((hw >> 13) & 7) * 64 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 15)       # one number per physical CU
ids, cnt = np.unique(cu, return_counts=True)
print(f"placement: {len(ids)} CUs hold blocks; blocks per CU: " + ", ".join(f"{c} on {int((cnt == c).sum())}" for c in sorted(set(cnt))))
print("  blocks per XCD: " + " ".join(str(int((xcc == x).sum())) for x in range(8)))
for c in sorted(set(cnt)):
    sel = np.isin(cu, ids[cnt == c])
    print(f"  CUs with {c} blocks: wave 0 out of work at mean {scan[sel].mean():.1f} us (p5 {np.percentile(scan[sel], 5):.1f}, p95 {np.percentile(scan[sel], 95):.1f}), "
          f"block end mean {end[sel].mean():.1f}, max {end[sel].max():.1f}")
print("  by XCD, mean block end: " + " ".join(f"{end[xcc == x].mean():.1f}" for x in range(8) if (xcc == x).any()))
if workers:
    # a queue's workers by the load of the CU they sit on: does a query block whose workers landed on crowded CUs run dry later?
    load = np.array([cnt[np.searchsorted(ids, c)] for c in cu])
    per_q_load = np.array([load[bxs == b].mean() for b in range(qb)])
    r = np.corrcoef(per_q_load, per_q)[0, 1] if per_q_load.std() > 0 else float("nan")
    print(f"  mean CU load of a query block's workers: {per_q_load.min():.2f} .. {per_q_load.max():.2f}; correlation with the time its queue runs dry: {r:.2f}")
