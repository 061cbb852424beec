#!/bin/bash
# VALU / SALU instructions per wave-row of the top-2 kernel under tuning knobs (SQ counters; development aid):
#     tools/valu_ab.sh NxM "knobs" ["knobs" ...]      e.g.  tools/valu_ab.sh 65536x65536 "" "merge=1"
set -uo pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
SIZE=$1; shift
O=gpurun_out/r04/valu_ab; mkdir -p $O
i=0
for k in "$@"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/p$i -o sq -- python3 tools/run_search.py $SIZE 12 "$k" > /dev/null 2> $O/p$i.err
    echo "== $SIZE knobs '$k'"
    SQ_N=${SIZE%x*} SQ_M=${SIZE#*x} python3 tools/sq_summary.py $O/p$i | grep -A5 per_wave_row
done
find $O -name "*.db" -delete 2>/dev/null
