#!/bin/bash
# Builds the two experiment variants of libslamhip.so used by tools/trace_probe.py and tools/exp_probe.py into
# tools/exp/ (development aid; nothing in the product or the tests loads them).  Both are the shipped sources with a
# small patch applied to a temporary copy of bf_hamming.hip:
#   libslamhip_trace.so      every block stamps wall_clock64() at start / after the prologue / after the scan / at the end,
#                            and where it runs (HW_REG_HW_ID, HW_REG_XCC_ID), behind the stamps
#   libslamhip_keepbound.so  the last arriver leaves the final 2nd-best distance in bound[] instead of restoring it
#   libslamhip_count.so      counts, per launch, the 16-row groups and the single rows that take the update path
#   libslamhip_cycles.so     wave 0 of every block stamps s_memtime (shader cycles) AND s_memrealtime (100 MHz) at block
#                            start / after the prologue / after the scan / at the end, into a buffer of its own
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/slam-experiments_amd/csrc"
OUT="$ROOT/tools/exp"
TMP="$(mktemp -d)"
mkdir -p "$OUT"
make -C "$SRC" -j8 all >/dev/null
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I/opt/rocm/include -I$SRC -fvisibility=hidden -DSLAM_BUILD"
OBJS=$(ls "$ROOT"/slam-experiments_amd/lib/obj/*.o | grep -v bf_hamming)

python3 - "$SRC/bf_hamming.hip" "$TMP/bf_trace.hip" "$TMP/bf_keep.hip" "$TMP/bf_count.hip" "$TMP/bf_cycles.hip" <<'EOF'
import sys
src = open(sys.argv[1]).read()
T = '    if (tid == 0 && g_trace) { g_trace[4*(by*(int)gridDim.x+bx)+%d] = wall_clock64(); }\n'
H = ('    if (g_trace) { unsigned hw_, xc_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\\n\\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw_), "=s"(xc_));\n'
     '        if (tid == 0) g_trace[4*(int)gridDim.x*(int)gridDim.y + by*(int)gridDim.x+bx] = ((unsigned long long)xc_ << 32) | hw_; }\n')   # where the block runs
s = src.replace('    const bool leader = !QUEUE && by < lead;\n', T % 0 + H + '    const bool leader = !QUEUE && by < lead;\n', 1)
s = s.replace('        int buf = 0;\n', '    ' + T % 1 + '        int buf = 0;\n', 1)
s = s.replace('        bool fresh = true;', T % 1 + '        bool fresh = true;', 1)
s = s.replace('    // ---- epilogue: merge,', T % 2 + '    // ---- epilogue: merge,', 1)
s = s.replace('    if (!s_last) return;\n', T % 3 + '    if (!s_last) return;\n', 1)
s = s.replace('typedef uint32_t u32;', 'typedef uint32_t u32;\n__device__ unsigned long long* g_trace = nullptr;\n'
              'extern "C" __attribute__((visibility("default"))) int slam_exp_set_trace(void* p) '
              '{ return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &p, sizeof(p)); }', 1)
assert s.count('g_trace[') == 6, "trace hooks did not apply: the kernel source changed"
open(sys.argv[2], 'w').write(s)
old = 'if (!merging) __hip_atomic_store(&st.bound[qi], SLAM_BOUND_IDLE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);'
old2 = 'const unsigned long long v = __hip_atomic_exchange(&st.best[qi], ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);'
assert src.count(old) == 1 and src.count(old2) == 1, "keep-bound hooks did not apply: the kernel source changed"
# bound form: the final 2nd-best distance stays in bound[]; merge form: the final pair stays in best[] (folding the same rows
# again changes nothing) - either way the next search of the same inputs starts every block at its final threshold
k = src.replace(old, 'if (!merging) __hip_atomic_store(&st.bound[qi], k2 == SLAM_KEY_NONE ? SLAM_BOUND_IDLE : '
                     '(k2 >> SLAM_KEY_IDX_BITS), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);')
k = k.replace(old2, old2 + ' if (merging) __hip_atomic_store(&st.best[qi], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);')
open(sys.argv[3], 'w').write(k)
g = '    if (__builtin_expect(__ballot((int)m >= 0) != 0ull, 0)) {\n'
r = '            if (U > 1 && __ballot((int)mu >= 0) == 0ull) continue;\n'
assert src.count(g) == 1 and src.count(r) == 1, "count hooks did not apply: the kernel source changed"
c = src.replace(g, g + '        if (g_fire && (threadIdx.x & 63) == 0) atomicAdd(&g_fire[U > 1 ? 0 : 2], 1ull);\n')
c = c.replace(r, r + '            if (g_fire && (threadIdx.x & 63) == 0) atomicAdd(&g_fire[1], 1ull);\n')
c = c.replace('typedef uint32_t u32;', 'typedef uint32_t u32;\n__device__ unsigned long long* g_fire = nullptr;\n'
              'extern "C" __attribute__((visibility("default"))) int slam_exp_set_fire(void* p) '
              '{ return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fire), &p, sizeof(p)); }', 1)
open(sys.argv[4], 'w').write(c)
# cycle stamps: one asm statement per stamp (s_memtime + s_memrealtime + the wait), wave 0 lane 0 stores both
C = ('    if (g_cyc) { unsigned long long c_, r_; asm volatile("s_memtime %%0\\n\\ts_memrealtime %%1\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(c_), "=s"(r_) :: "memory");\n'
     '        if (tid == 0) { g_cyc[8*(by*(int)gridDim.x+bx)+%d] = c_; g_cyc[8*(by*(int)gridDim.x+bx)+%d] = r_; } }\n')
y = src.replace('    const bool leader = !QUEUE && by < lead;\n', C % (0, 1) + '    const bool leader = !QUEUE && by < lead;\n', 1)
y = y.replace('        int buf = 0;\n', C % (2, 3) + '        int buf = 0;\n', 1)
y = y.replace('        bool fresh = true;', C % (2, 3) + '        bool fresh = true;', 1)
y = y.replace('    // ---- epilogue: merge,', C % (4, 5) + '    // ---- epilogue: merge,', 1)
y = y.replace('    if (!s_last) return;\n', C % (6, 7) + '    if (!s_last) return;\n', 1)
y = y.replace('typedef uint32_t u32;', 'typedef uint32_t u32;\n__device__ unsigned long long* g_cyc = nullptr;\n'
              'extern "C" __attribute__((visibility("default"))) int slam_exp_set_cycles(void* p) '
              '{ return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_cyc), &p, sizeof(p)); }', 1)
assert y.count('g_cyc[') == 10, "cycle hooks did not apply: the kernel source changed"
open(sys.argv[5], 'w').write(y)
EOF
for v in trace keep count cycles; do
    /opt/rocm/bin/hipcc $FLAGS -c "$TMP/bf_$v.hip" -o "$TMP/bf_$v.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libslamhip_trace.so" "$TMP/bf_trace.o" $OBJS -ldl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libslamhip_keepbound.so" "$TMP/bf_keep.o" $OBJS -ldl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libslamhip_count.so" "$TMP/bf_count.o" $OBJS -ldl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libslamhip_cycles.so" "$TMP/bf_cycles.o" $OBJS -ldl
rm -rf "$TMP"
ls -la "$OUT"
