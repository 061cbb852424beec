"""Writes profiles/<round>_isa_handoff_excerpt.txt: the epilogue of bf_top2_kernel<1, true, true> as built (llvm-objdump -d of the
gfx950 code object inside libslamhip.so), from the s_setprio 0 that opens it to the end of the kernel.  Development aid; the
checks themselves are tests/test_isa_handoff_cpu.py.      python tools/isa_excerpt.py [round]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))
from test_isa_handoff_cpu import disassemble  # noqa: E402
from slamhip import _lib  # noqa: E402

rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
with tempfile.TemporaryDirectory() as tmp:
    funcs = disassemble(_lib.LIB_PATH, tmp)
name = next(k for k in funcs if "bf_top2_kernel" in k and "ILi1ELb1ELb1E" in k)
ins = funcs[name]
start = max(k for k, i in enumerate(ins) if i == "s_setprio 0")
head = f"""# Epilogue of {name}
# (bf_top2_kernel<1, true, true>: R = 1, SGPR feed, queue plan) as built by hipcc for gfx950: llvm-objdump -d of the code object inside
# slam-experiments_amd/lib/libslamhip.so, from the s_setprio 0 that opens the epilogue to the end of the kernel (blocks the
# compiler lays out behind s_endpgm - the merge atomics, the polled completion - are reached by the branches and come back).
# tests/test_isa_handoff_cpu.py checks on every build what this excerpt shows once:
#   - every global_atomic_umin (merge into best[], publish into bound[]) carries sc0 = the returning form;
#   - s_waitcnt vmcnt(0), s_barrier, then the arrival ticket global_atomic_add ... sc0 (lane 0);
#   - the last arriver: s_barrier, then global_atomic_swap_x2 ... sc0 takes the result slots (agent-scope atomics on both sides
#     of the hand-off; no agent-scope buffer_wbl2, no load of the slots);
#   - the only write-backs are the system-scope ones (buffer_wbl2 sc0 sc1) of the polled completion: behind the decode, taken only
#     when a host thread polls the completion word (sel.done).
"""
out = os.path.join(ROOT, "profiles", f"{rnd}_isa_handoff_excerpt.txt")
with open(out, "w") as f:
    f.write(head + "\n".join(ins[start:]) + "\n")
print(f"wrote {out}: {len(ins) - start} instructions")
