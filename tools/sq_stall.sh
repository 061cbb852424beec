#!/bin/bash
# Where the wave-cycles of the top-2 search go: SQ counters (own rocprofv3 --pmc passes, --kernel-trace only) for a few
# shapes / plans, summarised by tools/sq_stall.py.  Development aid; run from the repo root on the GPU box.
#     tools/sq_stall.sh OUTDIR "label NxM knobs" ...
set -u
R="$(cd "$(dirname "$0")/.." && pwd)"
O="$(mkdir -p "$1" && cd "$1" && pwd)"; shift
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr "\n" " " > "$O/avail.txt"
SPECS=""
for cfg in "$@"; do
    set -- $cfg
    i=0
    DIRS=""
    for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
                "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INSTS_SMEM" \
                "SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH" \
                "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INST_CYCLES_SALU"; do
        i=$((i + 1))
        d="$O/$1_pass$i"
        timeout -k 10 120 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$d" -o p -- python3 "$R/tools/run_search.py" "$2" 12 "$3" > /dev/null 2>> "$O/err.log" \
            || echo "pass failed: $1 $pass" >> "$O/err.log"
        DIRS="$DIRS,$d"
    done
    SPECS="$SPECS $1=${DIRS#,}"
done
cd "$R"
python3 tools/sq_stall.py $SPECS
