"""Summarise rocprofv3 --pmc passes of SQ counters over bench.py for the top-2 kernel (-> profiles/<round>_sq_counters.json).

    python tools/sq_summary.py DIR [DIR ...] > profiles/<round>_sq_counters.json

Each DIR is the output directory of one pass (the counters of one pass must fit the SQ's counter slots; see
tools/profile.sh for the passes)."""
import csv, glob, hashlib, json, os, statistics, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, M = int(os.environ.get("SQ_N", "65536")), int(os.environ.get("SQ_M", "65536"))
raw = {}
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(float)
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if "bf_top2_kernel" not in row["Kernel_Name"]:
                    continue
                acc[(row["Counter_Name"], row["Dispatch_Id"])] += float(row["Counter_Value"])
        per = defaultdict(list)
        for (name, _), v in acc.items():
            per[name].append(v)
        for name, v in per.items():
            raw[name] = {"median": statistics.median(v), "launches": len(v)}
wave_rows = N * M / 64.0
def _sha(name):
    with open(os.path.join(ROOT, "slam-experiments_amd", "csrc", name), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]
out = {"kernel": f"bf_top2_kernel, {N} x {M}, medians over the launches of the profiled command",
       "source_sha": {n: _sha(n) for n in ("bf_hamming.hip", "bf_scan_sgpr.h")}, "wave_rows_per_launch": wave_rows, "per_wave_row": {}, "raw": raw}
for key, name in (("valu_instructions", "SQ_INSTS_VALU"), ("salu_instructions", "SQ_INSTS_SALU"), ("lds_instructions", "SQ_INSTS_LDS"),
                  ("smem_instructions", "SQ_INSTS_SMEM"), ("vmem_read_instructions", "SQ_INSTS_VMEM_RD"),
                  ("vmem_write_instructions", "SQ_INSTS_VMEM_WR")):
    if name in raw:
        out["per_wave_row"][key] = raw[name]["median"] / wave_rows
if "SQ_WAVES" in raw:
    out["waves_per_launch"] = raw["SQ_WAVES"]["median"]
if "GRBM_GUI_ACTIVE" in raw:
    out["grbm_gui_active_per_launch_sum_over_8_xcds"] = raw["GRBM_GUI_ACTIVE"]["median"]
if "SQ_WAVE_CYCLES" in raw and "SQ_WAIT_INST_ANY" in raw:
    out["wait_inst_any_share_of_wave_cycles"] = raw["SQ_WAIT_INST_ANY"]["median"] / raw["SQ_WAVE_CYCLES"]["median"]
print(json.dumps(out, indent=1))
