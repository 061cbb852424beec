"""Summarise the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over bench.py into profiles/hbm_counters.json.

Usage (on the GPU box, each pass on its own as MI355X_MICROARCH.md prescribes; tools/profile.sh does all of it):
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline
  python tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write > profiles/hbm_counters.json

The summary is stamped with the hash of the kernel sources it was collected on; bench.py prints `traffic: null` with
the reason when a stamp no longer matches the source in the tree.
"""
import csv, glob, hashlib, json, os, statistics, subprocess, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


def sha(name):
    with open(os.path.join(ROOT, "slam-experiments_amd", "csrc", name), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


fetch, write = read(sys.argv[1], "FETCH_SIZE"), read(sys.argv[2], "WRITE_SIZE")
raw = {k: {"FETCH_SIZE_KB_per_launch": stats(fetch[k]), "WRITE_SIZE_KB_per_launch": stats(write[k])}
       for k in sorted(set(fetch) & set(write))}


def pick(prefix):
    for k in raw:
        if prefix in k:
            return raw[k]
    return None


try:
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
except OSError:
    head = ""
out = {
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE  and, separately,  --pmc WRITE_SIZE  -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline",
    "head": head or "unknown (no git on the GPU box: see the commit that added this file)",
    "source_sha": {n: sha(n) for n in ("bf_hamming.hip", "bf_scan_sgpr.h", "reproj.hip")},
    "units": "raw counter values are KB per launch",
    "corrections": "gfx950: FETCH_SIZE = TCC_EA0_RDREQ x 64 B while coalesced streaming reads travel as 128-B requests, so it "
                   "reports half their bytes (MI355X_MICROARCH.md, HBM section: measured for 16 B/lane reads) -> doubled for both "
                   "kernels (bf_top2_kernel stages its rows with 16 B/lane loads; reproj_rj_kernel streams 4-, 8- and 16-B items "
                   "that are just as contiguous per wave: 2 x raw lands on 24 B x 1e7 observations + the pose / point tables, "
                   "raw alone would be half the index and pixel streams the kernel cannot avoid reading). WRITE_SIZE exact. "
                   "The fabric-side counters include Infinity-Cache hits.",
}
bf, rj = pick("bf_top2_kernel"), pick("reproj_rj_kernel")
if bf:
    f, w = 2 * 1024 * bf["FETCH_SIZE_KB_per_launch"]["median"], 1024 * bf["WRITE_SIZE_KB_per_launch"]["median"]
    out["bf_top2_kernel"] = {"fetch_bytes_corrected": f, "write_bytes": w, "traffic_bytes": f + w, "algorithmic_bytes": 5242880}
if rj:
    f, w = 2 * 1024 * rj["FETCH_SIZE_KB_per_launch"]["median"], 1024 * rj["WRITE_SIZE_KB_per_launch"]["median"]
    out["reproj_rj_kernel"] = {"fetch_bytes_corrected": f, "write_bytes": w, "traffic_bytes": f + w,
                               "algorithmic_bytes": 1840000000, "algorithmic_read_bytes": 240000000}
out["raw"] = raw
print(json.dumps(out, indent=1))
