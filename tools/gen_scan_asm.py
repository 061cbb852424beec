#!/usr/bin/env python3
"""Writes slam-experiments_amd/csrc/bf_scan_sgpr.h: the scan of the top-2 search with the train rows fed through SGPRs,
as one asm statement (macro SLAM_SCAN_GROUPS_ASM).  The statement is ~460 lines of straight-line gfx950 assembly with a
fixed register plan, so it is generated rather than typed; the output is committed and compiled as is.

    python tools/gen_scan_asm.py            (rewrites the header; `git diff` shows what changed)
    python tools/gen_scan_asm.py PATH       (writes it to PATH: tests/test_abi_and_host_cpu.py holds the committed header to it)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "slam-experiments_amd", "csrc", "bf_scan_sgpr.h")   # (a path argument: write there)
G = 4                                   # rows per SGPR set (two sets alternate)
ACC, TMP, VM, VKEY = 32, 48, 56, 57     # VGPRs: 16 accumulators, 8 temporaries, group mask, key
SA = 36                                 # SGPR set A = s[36 : 36 + 8G), set B right behind it
SB = SA + 8 * G
STMP, SP = 33, 34                       # s33: row index of the update path; s[34:35]: running train pointer


def line(x):
    return f'"{x}\\n\\t"'


def load_set(base, first_row):
    # G rows into s[base : base + 8G): 64 B = two rows per s_load_dwordx16
    return [line(f"s_load_dwordx16 s[{base + 8 * k}:{base + 8 * k + 15}], s[{SP}:{SP + 1}], {hex(32 * (first_row + k))}")
            for k in range(0, G, 2)]


PAIRWISE = os.environ.get("SLAM_SCAN_PAIRWISE") == "1"    # experiment: two temporaries, reused pairwise


def row(r, sbase):
    if PAIRWISE:
        out = []
        for w in range(0, 8, 2):
            out += [line("s_setprio 0"), line(f"v_xor_b32 v{TMP}, s{sbase + w}, %[q{w}]"), line(f"v_xor_b32 v{TMP + 1}, s{sbase + w + 1}, %[q{w + 1}]"),
                    line("s_setprio 2"), line(f"v_bcnt_u32_b32 v{ACC + r}, v{TMP}, " + ("%[init]" if w == 0 else f"v{ACC + r}")),
                    line(f"v_bcnt_u32_b32 v{ACC + r}, v{TMP + 1}, v{ACC + r}")]
        return out
    out = [line("s_setprio 0")]
    out += [line(f"v_xor_b32 v{TMP + w}, s{sbase + w}, %[q{w}]") for w in range(8)]
    out += [line("s_setprio 2"), line(f"v_bcnt_u32_b32 v{ACC + r}, v{TMP}, %[init]")]
    out += [line(f"v_bcnt_u32_b32 v{ACC + r}, v{TMP + w}, v{ACC + r}") for w in range(1, 8)]
    return out


nsub = 16 // G


def prologue():
    # (the loop head on a 64-byte boundary: without it the time of a small search moved by 2-5 % with whatever else changed in
    # the kernel in front of the loop - 2000 x 2000 9.76 / 10.30 us for the same cold loop; profiles/r04_ab_queue.log "fired path")
    return [line(f"s_mov_b64 s[{SP}:{SP + 1}], %[tp]")] + load_set(SA, 0) + [line(".p2align 6"), line("Lgroup_%=:")]


def loads_for(k):
    nxt = SB if k % 2 == 0 else SA
    out = [line("s_waitcnt lgkmcnt(0)")]
    if k < nsub - 1:
        out += load_set(nxt, G * (k + 1))
    else:   # the next group's first rows, unless this is the last group of the call: nothing is read past the requested rows
        out += [line("s_cmp_eq_u32 %[ng], 1"), line("s_cbranch_scc1 Lnopf_%=")] + load_set(nxt, 16) + [line("Lnopf_%=:")]
    return out


def epilogue():
    return [line(f"s_add_u32 s{SP}, s{SP}, 0x200"), line(f"s_addc_u32 s{SP + 1}, s{SP + 1}, 0"),
            line("s_add_u32 %[idx], %[idx], 16"), line("s_sub_u32 %[ng], %[ng], 1"), line("s_cmp_lg_u32 %[ng], 0"),
            '"s_cbranch_scc1 Lgroup_%="']


# ---- the filtered scan -------------------------------------------------------------------------------------------------
body = prologue()
for k in range(nsub):
    body += loads_for(k)
    for i in range(G):
        body += row(G * k + i, (SA if k % 2 == 0 else SB) + 8 * i)
# filter: a row improved some lane <=> its accumulator is below 2^31 there <=> the unsigned minimum of the sixteen is.
# Eight v_min3_u32 / v_min_u32 instead of a tree of fifteen v_and_b32: min3 can only use the v_bcnt issue slot, but the
# scan is bound by the OTHER slot (eight v_xor + the filter against eight v_bcnt), so halving the filter's instructions
# and moving them over measured 1089 -> 1071 us at 64k x 64k, 586 -> 572 us at 32768 x 65536 (profiles/r03_ab_sgpr_feed.log).
# The minimum is taken as five minima of three rows (into the temporaries, which are free once the group's rows are done), then
# over those and row 15: eight instructions like a plain chain - but a group that fires finds its rows by testing the five
# triples first and only the rows of a triple that fired: nine compares where sixteen were for the usual single firing row
# (a fifth of the groups fire at 64k x 64k: SQ counters, profiles/r04_valu_by_exchange_form.log).
body += [line(f"v_min3_u32 v{TMP + t}, v{ACC + 3 * t}, v{ACC + 3 * t + 1}, v{ACC + 3 * t + 2}") for t in range(5)]
MINFORM = os.environ.get("SLAM_SCAN_MINFORM", "8min")    # experiment: how the five triple minima and row 15 are combined
if MINFORM == "8min":        # three more minima: every instruction of the group test in the v_bcnt issue slot
    body += [line(f"v_min3_u32 v{VM}, v{TMP}, v{TMP + 1}, v{TMP + 2}"), line(f"v_min3_u32 v{VM}, v{VM}, v{TMP + 3}, v{TMP + 4}"),
             line(f"v_min_u32 v{VM}, v{VM}, v{ACC + 15}")]
elif MINFORM == "5min5and":  # only the sign bits matter: v_and_b32 on VGPRs can use either issue slot
    body += [line(f"v_and_b32 v{VM}, v{TMP}, v{TMP + 1}"), line(f"v_and_b32 v{TMP + 5}, v{TMP + 2}, v{TMP + 3}"),
             line(f"v_and_b32 v{VM}, v{VM}, v{TMP + 5}"), line(f"v_and_b32 v{TMP + 5}, v{TMP + 4}, v{ACC + 15}"),
             line(f"v_and_b32 v{VM}, v{VM}, v{TMP + 5}")]
elif MINFORM == "6min3and":
    body += [line(f"v_min3_u32 v{TMP + 5}, v{TMP + 3}, v{TMP + 4}, v{ACC + 15}"), line(f"v_and_b32 v{VM}, v{TMP}, v{TMP + 1}"),
             line(f"v_and_b32 v{TMP + 5}, v{TMP + 2}, v{TMP + 5}"), line(f"v_and_b32 v{VM}, v{VM}, v{TMP + 5}")]
else:
    raise SystemExit(f"unknown SLAM_SCAN_MINFORM {MINFORM}")
body += [line(f"v_cmp_lt_i32 vcc, -1, v{VM}"), line("s_cbranch_vccz Lnofire_%=")]


def fired_row(u):   # update path of one row: skipped when its ballot is empty (wave-uniform)
    return [line(f"v_cmp_lt_i32 vcc, -1, v{ACC + u}"), line(f"s_cbranch_vccz Lskip{u}_%="),
            line(f"s_add_u32 s{STMP}, %[idx], {u}"), line(f"v_sub_u32 v{VKEY}, v{ACC + u}, %[init]"),
            line(f"v_lshl_or_b32 v{VKEY}, v{VKEY}, 23, s{STMP}"), line(f"v_med3_u32 %[b2], %[b1], %[b2], v{VKEY}"),
            line(f"v_min_u32 %[b1], %[b1], v{VKEY}"), line(f"Lskip{u}_%=:")]


for t in range(5):
    body += [line(f"v_cmp_lt_i32 vcc, -1, v{TMP + t}"), line(f"s_cbranch_vccz Ltriple{t}_%=")]
    for u in range(3 * t, 3 * t + 3):
        body += fired_row(u)
    body += [line(f"Ltriple{t}_%=:")]
body += fired_row(15)
body += [line(f"v_lshrrev_b32 v{VKEY}, 23, %[b2]"), line(f"v_sub_u32 v{VKEY}, 0x80000000, v{VKEY}"),
         line(f"v_max_u32 %[init], %[init], v{VKEY}"), line("Lnofire_%=:")] + epilogue()


# ---- the unfiltered scan (the first rows of a chunk that starts without a threshold) -----------------------------------
# Every row is folded into (b1, b2): no group test, no ballots, no branches.  A row's accumulator starts at 0, so the key
# is (acc << 23) | index.  The fold of row r is issued behind the v_xor half of row r + 1 (its inputs are long complete),
# in the raised-priority half, where the other one-slot instructions (v_bcnt) are.
def insert(r):
    return [line(f"s_add_u32 s{STMP}, %[idx], {r}"), line(f"v_lshlrev_b32 v{VKEY}, 23, v{ACC + r}"),
            line(f"v_or_b32 v{VKEY}, s{STMP}, v{VKEY}"), line(f"v_med3_u32 %[b2], %[b1], %[b2], v{VKEY}"),
            line(f"v_min_u32 %[b1], %[b1], v{VKEY}")]


def cold_row(r, sbase):
    out = [line("s_setprio 0")]
    out += [line(f"v_xor_b32 v{TMP + w}, s{sbase + w}, %[q{w}]") for w in range(8)]
    out += [line("s_setprio 2")]
    if r:
        out += insert(r - 1)
    out += [line(f"v_bcnt_u32_b32 v{ACC + r}, v{TMP}, 0")]
    out += [line(f"v_bcnt_u32_b32 v{ACC + r}, v{TMP + w}, v{ACC + r}") for w in range(1, 8)]
    return out


cold = prologue()
for k in range(nsub):
    cold += loads_for(k)
    for i in range(G):
        cold += cold_row(G * k + i, (SA if k % 2 == 0 else SB) + 8 * i)
cold += insert(15) + epilogue()

clob = ", ".join([f'"v{i}"' for i in range(ACC, VKEY + 1)] + [f'"s{i}"' for i in range(STMP, SA + 16 * G)] + ['"vcc"', '"scc"', '"memory"'])
ins = ", ".join(f'[q{w}] "v"(Q_[{w}])' for w in range(8)) + ', [tp] "s"(TP_)'
text = f'''// bf_scan_sgpr.h - GENERATED by tools/gen_scan_asm.py; edit the generator, not this file.
//
// SLAM_SCAN_GROUPS_ASM(Q_, TP_, NG_, IDX_, B1_, B2_, INIT_): the scan of NG_ groups of 16 train rows (R = 1, one query per
// lane) as ONE asm statement, the train rows travelling through SGPRs.
//   Q_     u32[8]   the lane's query words (VGPRs, read only)
//   TP_    pointer  first train row of the call (wave-uniform; rows are 32 bytes, 16-byte aligned)
//   NG_    int      groups to scan, >= 1 (wave-uniform, in an SGPR; destroyed)
//   IDX_   u32      train index of the first row (wave-uniform, in an SGPR; advanced by 16 per group)
//   B1_, B2_, INIT_ the lane's best / second-best packed keys and its biased threshold (VGPRs, in/out)
// A train row is the same for every lane of a wave, so it does not have to travel through LDS into vector registers:
// s_load_dwordx16 brings two rows (64 B) from the scalar cache / L2 into SGPRs and v_xor takes its train word as the SGPR
// operand.  Measured (tools/ubench/cycles.hip, profiles/r03_ubench_cycles.log): v_xor with an SGPR operand can use only
// one of the SIMD's two issue slots (4.1 cycles per instruction alone) but still runs BESIDE another wave's v_bcnt (2.18
// together; 2.29 for this row with the priorities below), and the bare loop holds 2.37 GHz where the same loop fed by two
// ds_read_b128 per row drops to 2.13 GHz: the LDS reads and their 1 KiB of vector-register writes per row are what the
// chip throttles on.  So the scan has no LDS tile, no barrier and no staging copy.
// Register plan (fixed; the statement clobbers them): v{ACC}-v{ACC + 15} the group's sixteen biased distances, v{TMP}-v{TMP + 7}
// xor results, v{VM} group minimum, v{VKEY} key; s{STMP} row index in the update path, s[{SP}:{SP + 1}] running train
// pointer, SGPR sets A = s{SA}-s{SB - 1} and B = s{SB}-s{SB + 8 * G - 1} ({G} rows each).  The sets alternate: the loads of the next set
// are issued before the current one is computed; SMEM returns out of order, so the only usable wait is lgkmcnt(0), which
// sits in front of those loads.  The last sub-group of a group loads the next group's first rows unless the call ends
// there: nothing is read past the requested rows.
// Per row: s_setprio 0, eight v_xor (SGPR train word -> temporary), s_setprio 2, eight accumulating v_bcnt (row_acc in
// bf_hamming.hip explains the priorities).  Per group: the unsigned minimum of the sixteen accumulators (five minima of three rows, then over those), one compare, one branch; the
// update block (unlikely) tests the five triples' ballots, then each row's of a triple that fired, and folds a fired row into (b1, b2) by med3 / min on the packed keys
// (distance << 23 | train index), then tightens the threshold: the arithmetic of filter_update, bit for bit.
#pragma once
#define SLAM_SCAN_GROUPS_ASM(Q_, TP_, NG_, IDX_, B1_, B2_, INIT_) \\
    asm volatile( \\
''' + "".join(f"        {l} \\\n" for l in body) + f'''        : [b1] "+v"(B1_), [b2] "+v"(B2_), [init] "+v"(INIT_), [ng] "+s"(NG_), [idx] "+s"(IDX_) \\
        : {ins} \\
        : {clob})
// SLAM_SCAN_GROUPS_COLD_ASM(Q_, TP_, NG_, IDX_, B1_, B2_): the same scan WITHOUT a threshold - every row is folded into
// (B1_, B2_).  For the first rows of a chunk that starts with no bound (nearly every row would take the update path: a wave
// fires when any of its 64 lanes improves, i.e. on about 128 / s of its rows after s rows): four more instructions per
// row instead of a compare and a branch per row.  Accumulators start at 0, the key is (acc << 23) | index.
#define SLAM_SCAN_GROUPS_COLD_ASM(Q_, TP_, NG_, IDX_, B1_, B2_) \\
    asm volatile( \\
''' + "".join(f"        {l} \\\n" for l in cold) + f'''        : [b1] "+v"(B1_), [b2] "+v"(B2_), [ng] "+s"(NG_), [idx] "+s"(IDX_) \\
        : {ins} \\
        : {clob})
'''
with open(OUT, "w") as f:
    f.write(text)
print(f"wrote {OUT}: {len(body)} asm lines")
