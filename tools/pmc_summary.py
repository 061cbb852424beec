"""Summarise the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over bench.py into profiles/r01_hbm_counters.json.

Usage (on the GPU box, each pass on its own as MI355X_MICROARCH.md prescribes):
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline
  python tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write > profiles/r01_hbm_counters.json
"""
import csv, glob, json, os, statistics, sys
from collections import defaultdict


def read(dirname, counter):
    per = defaultdict(list)
    files = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {dirname}")
    for f in files:
        with open(f, newline="") as fh:
            acc = defaultdict(float)        # (dispatch id, kernel) -> value summed over counter instances
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                acc[(row["Dispatch_Id"], row["Kernel_Name"].split("(")[0])] += float(row["Counter_Value"])
            for (_, k), v in acc.items():
                per[k].append(v)
    return per


def stats(v):
    return {"median": statistics.median(v), "min": min(v), "max": max(v), "launches": len(v)}


fetch, write = read(sys.argv[1], "FETCH_SIZE"), read(sys.argv[2], "WRITE_SIZE")
raw = {k: {"FETCH_SIZE_KB_per_launch": stats(fetch[k]), "WRITE_SIZE_KB_per_launch": stats(write[k])}
       for k in sorted(set(fetch) & set(write))}


def pick(prefix):
    for k in raw:
        if prefix in k:
            return raw[k]
    return None


out = {
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE  and, separately,  --pmc WRITE_SIZE  -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline",
    "units": "raw counter values are KB per launch",
    "corrections": "gfx950: FETCH_SIZE reports half the bytes of wide (16 B/lane) coalesced reads -> doubled for bf_top2_kernel "
                   "(its bulk global reads are 16 B/lane); WRITE_SIZE exact; the fabric-side counters include Infinity-Cache hits "
                   "(MI355X_MICROARCH.md, HBM section). reproj_rj_kernel reads 4/8/16-B items and its raw FETCH_SIZE already equals "
                   "the algorithmic read bytes, so it is not doubled.",
}
bf, rj = pick("bf_top2_kernel"), pick("reproj_rj_kernel")
if bf:
    f, w = 2 * 1024 * bf["FETCH_SIZE_KB_per_launch"]["median"], 1024 * bf["WRITE_SIZE_KB_per_launch"]["median"]
    out["bf_top2_kernel"] = {"fetch_bytes_corrected": f, "write_bytes": w, "traffic_bytes": f + w, "algorithmic_bytes": 5242880}
if rj:
    f, w = 1024 * rj["FETCH_SIZE_KB_per_launch"]["median"], 1024 * rj["WRITE_SIZE_KB_per_launch"]["median"]
    out["reproj_rj_kernel"] = {"fetch_bytes": f, "write_bytes": w, "traffic_bytes": f + w, "algorithmic_bytes": 1840000000}
out["raw"] = raw
print(json.dumps(out, indent=1))
