"""CPU suite: the cross-workgroup hand-off of the top-2 search, pinned on the gfx950 ISA of the built library.

The search's blocks merge their results with agent-scope atomics and hand them to the last arriver of a query block WITHOUT
an agent-scope release fence (bf_hamming.hip, "Arrival ticket").  That is only sound while the compiler keeps emitting what
the source asks for, so this test disassembles the code object inside ``libslamhip.so`` (llvm-objdump cross-disassembles; no
GPU needed) and checks, in every kernel that contains the epilogue:

  1. every ``global_atomic_umin`` is the RETURNING form (``sc0``): the merges into best[] and every write to bound[] - a
     returned value is what tells the wave that the minimum has been taken at the memory side;
  2. the arrival ticket is a returning ``global_atomic_add``; in front of it, in this order and with no vector-memory
     instruction in between: ``s_waitcnt vmcnt(0)`` (every wave has its atomics back), ``s_barrier``;
  3. behind the ticket and a barrier, the last arriver TAKES the result slots with returning 8-byte atomic exchanges
     (``global_atomic_swap_x2 ... sc0``: agent-scope atomics on both sides of the hand-off, no load that a cache could serve;
     the exchange also puts the slot back to idle).

A compiler upgrade that drops a return form or moves a wait fails here instead of racing silently.
The epilogue of bf_top2_kernel<1, true, true> as built for round 4 is kept in profiles/r04_isa_handoff_excerpt.txt."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
VMEM = re.compile(r"^(global_|buffer_(?!inv|wbl2)|flat_|scratch_)")


def disassemble(lib_path, workdir):
    """{kernel name: [instruction, ...]} of every gfx950 code object bundled in the library."""
    local = os.path.join(workdir, "lib.so")
    shutil.copy(lib_path, local)
    subprocess.run([OBJDUMP, "--offloading", local], cwd=workdir, check=True, capture_output=True)
    funcs = {}
    for name in sorted(os.listdir(workdir)):
        if "gfx950" not in name:
            continue
        text = subprocess.run([OBJDUMP, "-d", os.path.join(workdir, name)], check=True, capture_output=True, text=True).stdout
        cur = None
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                label = m.group(1)
                if not re.match(r"^L\w+_\d+$", label):            # the scan's local labels (Lgroup_3 ...) stay inside their kernel
                    cur = funcs.setdefault(label, [])
                continue
            ins = line.split("//")[0].strip()
            if cur is not None and ins and not ins.startswith("."):
                cur.append(ins)
    return funcs


@pytest.fixture(scope="module")
def kernels(built, tmp_path_factory):
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump of the ROCm toolchain not found")
    from slamhip import _lib

    funcs = disassemble(_lib.LIB_PATH, str(tmp_path_factory.mktemp("isa")))
    found = {k: v for k, v in funcs.items() if "bf_top2" in k and "kernel" in k}
    assert len(found) >= 6, sorted(funcs)[:20]          # R = 8, 4, 2; R = 1 through LDS / SGPRs / as a queue; the batch kernel
    return found


def test_every_merge_atomic_returns(kernels):
    for name, ins in kernels.items():
        umin = [i for i in ins if i.startswith("global_atomic_umin")]
        assert len(umin) >= 3, (name, len(umin))
        bad = [i for i in umin if " sc0" not in i]
        assert not bad, f"{name}: atomic minimum without a return value: {bad[:3]}"
        assert not [i for i in ins if i.startswith(("flat_atomic", "global_atomic_cmpswap"))], name


def test_ticket_sits_behind_wait_and_barrier_and_the_last_arriver_reads_with_atomics(kernels):
    checked = 0
    for name, ins in kernels.items():
        swaps = [k for k, i in enumerate(ins) if i.startswith("global_atomic_swap_x2")]
        assert swaps, f"{name}: the last arriver no longer takes the result slots with an atomic exchange"
        # No release fence in the hand-off: round 3's was the agent-scope write-back (buffer_wbl2 sc1).  The only write-backs a
        # kernel may hold are SYSTEM-scope ones (sc0 sc1): the polled completion (sel.done) - every thread of the last arriver
        # makes its result stores visible to the host, one lane releases the completion word - which sits behind the decode
        # (the compiler lays that block out wherever it likes, so the text order says nothing; the scope does).
        wb = [i for i in ins if i.startswith("buffer_wbl2")]
        assert all(i.split() == ["buffer_wbl2", "sc0", "sc1"] for i in wb), f"{name}: an agent-scope release fence is back: {sorted(set(wb))}"
        tickets = set()
        for at in swaps:
            assert " sc0" in ins[at], f"{name}: the exchange must return the slot"
            # backwards from the exchange: a barrier (the block learns whether it is the last arriver), then the ticket
            k = at - 1
            seen_barrier = later_query = False
            while k >= 0 and not ins[k].startswith("global_atomic_add"):
                if ins[k].startswith(("global_atomic_swap_x2", "global_store")):
                    later_query = True                     # several queries per lane: one ticket, R exchanges (each followed by the
                    break                                  # stores of its decode) - the first one is checked, and every kernel has one
                assert not VMEM.match(ins[k]), f"{name}: {ins[k]} between the ticket and the exchange"
                seen_barrier = seen_barrier or ins[k].startswith("s_barrier")
                k -= 1
            if later_query:
                continue
            assert k >= 0 and " sc0" in ins[k] and seen_barrier, f"{name}: no returning ticket + barrier in front of the exchange"
            ticket = k
            assert ticket not in tickets
            tickets.add(ticket)
            # backwards from the ticket: barrier, then the wait, with no vector-memory instruction in between
            k = ticket - 1
            seen_barrier = False
            while k >= 0:
                i = ins[k]
                if i.startswith("s_barrier"):
                    seen_barrier = True
                elif i.startswith("s_waitcnt") and "vmcnt(0)" in i and seen_barrier:
                    break
                else:
                    assert not VMEM.match(i), f"{name}: {i} between the waves' wait and the ticket"
                k -= 1
            assert k >= 0 and seen_barrier, f"{name}: no s_waitcnt vmcnt(0) + s_barrier in front of the ticket"
            checked += 1
        assert tickets, f"{name}: no exchange that follows its ticket directly"
    assert checked >= 7                                  # the batch kernel carries the epilogue twice (both feeds)


def test_queue_tickets_return(kernels):
    """The queue kernel draws its chunk tickets with returning agent-scope adds as well (two sites + the arrival)."""
    name = next(k for k in kernels if "ILi1ELb1ELb1E" in k)
    adds = [i for i in kernels[name] if i.startswith("global_atomic_add")]
    assert len(adds) >= 3 and all(" sc0" in i for i in adds), adds
