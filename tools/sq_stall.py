"""Medians per launch of every counter found in rocprofv3 --pmc output directories, for the top-2 kernel (development aid:
where do the wave-cycles of a launch go).    python tools/sq_stall.py LABEL=DIR[,DIR...] [LABEL=DIR...]"""
import csv, glob, os, statistics, sys
from collections import defaultdict

for spec in sys.argv[1:]:
    label, dirs = spec.split("=")
    raw = {}
    for d in dirs.split(","):
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(float)
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    if "bf_top2_kernel" in row["Kernel_Name"]:
                        acc[(row["Counter_Name"], row["Dispatch_Id"])] += float(row["Counter_Value"])
            per = defaultdict(list)
            for (name, _), v in acc.items():
                per[name].append(v)
            for name, v in per.items():
                raw[name] = statistics.median(v)
    print(label)
    wc = raw.get("SQ_WAVE_CYCLES")
    for name in sorted(raw):
        share = f"  ({raw[name] / wc:6.3f} of SQ_WAVE_CYCLES)" if wc and name != "SQ_WAVE_CYCLES" else ""
        print(f"  {name:28s} {raw[name]:16.0f}{share}")
