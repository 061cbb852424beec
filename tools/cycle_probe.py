"""True shader cycles and the in-kernel clock of the top-2 search (development aid; needs tools/exp/libslamhip_cycles.so
from tools/build_exp.sh).  Wave 0 of every block stamps s_memtime (shader cycles) and s_memrealtime (constant 100 MHz)
at block start / after the prologue / after the scan / at the end; the stamps go to a buffer of their own.

    python tools/cycle_probe.py [NxM ...] [--spin SECONDS]

Reports, per size, after SECONDS of back-to-back launches (MI355X_MICROARCH.md "DVFS give-back" item 6):
  * in-kernel clock = delta s_memtime / delta s_memrealtime x 100 MHz, median over blocks;
  * whole-launch cycles per wave-row per SIMD = (HIP-event kernel time x that clock) x (CUs x 4 SIMDs) / wave-rows
    (includes the head, the drain and every stall of the launch);
  * scan-only cycles per wave-row per SIMD = a block's scan cycles / its rows / 8 waves per SIMD, median over the
    blocks of full dispatch rounds (a wave shares its SIMD with 7 others while the chip is full).
The stamp build needs a few more registers than the shipped kernel, so its own time is printed beside the shipped one.
"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))
import slamhip  # noqa: E402
from slamhip import _lib  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
spin = float(sys.argv[sys.argv.index("--spin") + 1]) if "--spin" in sys.argv else 2.0
if "--spin" in sys.argv:
    args = [a for a in args if a != sys.argv[sys.argv.index("--spin") + 1]]
sizes = [tuple(int(v) for v in a.split("x")) for a in args] or [(65536, 65536), (8192, 65536), (4096, 4096)]


def measure(lib_path, n, m, stamped):
    _lib.LIB_PATH = lib_path                    # one library per process: the first call decides (see SHIPPED below)
    ctx = slamhip.Context(0)
    lib, h = ctx.lib, ctx.handle
    q = slamhip.DeviceDescriptors(ctx, np.random.default_rng(228).integers(0, 256, (n, 32), dtype=np.uint8))
    t = slamhip.DeviceDescriptors(ctx, np.random.default_rng(229).integers(0, 256, (m, 32), dtype=np.uint8))
    tab = slamhip.Top2Table(ctx, n)
    plan = ctx.plan_info(n, m)
    plan["workers"] = slamhip.plan_describe(n, m, num_cu=plan["cus"])[0]["workers"]      # > 0: a queue plan (workers x query blocks)
    blocks = plan["qblocks"] * (plan["workers"] or plan["chunks"])
    f = lambda: lib.slam_bf_knn2_u256(h, q.buf.ptr, n, t.buf.ptr, m, 0, tab.idx.ptr, tab.dist.ptr)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < spin:          # back-to-back launches: the clock settles under THIS load
        for _ in range(64):
            f()
        ctx.sync()
    reps = 20
    ctx.timer_start()
    for _ in range(reps):
        f()
    ms = ctx.timer_stop() / reps
    out = {"ms": ms, "plan": plan}
    if stamped:
        lib.slam_exp_set_cycles.argtypes = [ctypes.c_void_p]
        buf = ctx.malloc(blocks * 64)
        assert lib.slam_exp_set_cycles(buf.ptr) == 0
        ctx.timer_start()
        f()
        out["ms_stamped_launch"] = ctx.timer_stop()
        assert lib.slam_exp_set_cycles(None) == 0
        out["stamps"] = buf.download(np.uint64, (blocks, 8)).astype(np.int64)
        buf.free()
    for o in (tab, q, t):
        o.free()
    ctx.close()
    return out


if os.environ.get("CYCLE_PROBE_SHIPPED") == "1":
    # child mode: time the shipped library only (two builds of one library cannot share a process)
    for n, m in sizes:
        print(f"SHIPPED {n} {m} {measure(os.path.join(ROOT, 'slam-experiments_amd', 'lib', 'libslamhip.so'), n, m, False)['ms']:.6f}", flush=True)
    sys.exit(0)

import subprocess  # noqa: E402

child = subprocess.run([sys.executable, os.path.abspath(__file__), *[f"{n}x{m}" for n, m in sizes], "--spin", str(spin)],
                       env=dict(os.environ, CYCLE_PROBE_SHIPPED="1"), capture_output=True, text=True, check=True)
shipped_ms = {(int(a), int(b)): float(c) for _, a, b, c in (ln.split() for ln in child.stdout.splitlines() if ln.startswith("SHIPPED"))}

for n, m in sizes:
    ship = {"ms": shipped_ms[(n, m)]}
    exp = measure(os.path.join(ROOT, "tools", "exp", "libslamhip_cycles.so"), n, m, True)
    s = exp["stamps"]
    plan = exp["plan"]
    c = s[:, 0::2]          # s_memtime at the four points
    r = s[:, 1::2]          # s_memrealtime
    whole = (c[:, 3] - c[:, 0]) / np.maximum(r[:, 3] - r[:, 0], 1) * 100.0
    clock = float(np.median(whole))
    wave_rows = ((n + 63) // 64) * m
    simds = plan["cus"] * 4
    # rows per block from the chunk table: all uniform chunks have plan["chunk"] rows; use the blocks whose scan
    # covers exactly that many rows (the bulk of the grid), identified through the plan's layout: chunk index = block // qblocks
    qb, S = plan["qblocks"], plan["chunks"]
    if plan["workers"]:                      # a queue plan: a block's rows are whatever its waves drew - no per-block figure
        uniform = np.zeros(len(s), bool)
        per_row = np.zeros(0)
    else:
        chunk_of = np.arange(qb * S) // qb
        uniform = (chunk_of >= plan["lead_chunks"]) & (chunk_of < S - plan["tail_chunks"] - 1)
        scan_cyc = (c[:, 2] - c[:, 1])[uniform]
        per_row = scan_cyc / plan["chunk"] / 8.0
    print(f"{n}x{m}: plan {plan}")
    print(f"  shipped kernel {ship['ms'] * 1e3:.1f} us per launch; stamp build {exp['ms'] * 1e3:.1f} us (stamps off), "
          f"{exp['ms_stamped_launch'] * 1e3:.1f} us (the stamped launch)")
    print(f"  in-kernel clock: median {clock:.0f} MHz (p5 {np.percentile(whole, 5):.0f}, p95 {np.percentile(whole, 95):.0f}) over {len(whole)} blocks")
    for label, ms in (("shipped", ship["ms"]), ("stamp build", exp["ms"])):
        cyc = ms * 1e-3 * clock * 1e6 * simds / wave_rows
        print(f"  whole launch, {label}: {cyc:.2f} cycles per wave-row per SIMD at the measured clock "
              f"({ms * 1e-3 * 2.4e9 * simds / wave_rows:.2f} if the clock were 2400 MHz)")
    if uniform.any():
        print(f"  scan only (uniform chunks, {int(uniform.sum())} blocks): median {np.median(per_row):.2f} cycles per wave-row per SIMD "
              f"(p5 {np.percentile(per_row, 5):.2f}, p95 {np.percentile(per_row, 95):.2f}); dual-issue floor 32", flush=True)
