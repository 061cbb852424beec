#!/bin/bash
# Samples rocm-smi (shader clock, socket power, temperature) twice a second while a long loop of the 64k x 64k search runs
# (is the gap between the kernel and the 32-cycle issue floor the clock the chip holds under this load?).
#     tools/clock_sample.sh OUT.log [searches]
set -u
R="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$1"; N="${2:-10000}"
( for i in $(seq 1 60); do
    echo "t=$(date +%s.%N | cut -c1-14) $(rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E 'sclk|Socket Power|Average Graphics|Temperature \(Sensor (edge|junction|hotspot)' | sed -E 's/ +/ /g' | tr '\n' '|')"
    sleep 0.5
  done ) > "$OUT.smi" 2>&1 &
SMI=$!
sleep 1.2
python3 "$R/tools/run_search.py" 65536x65536 "$N" > "$OUT.run" 2>&1
sleep 1.0
kill $SMI 2>/dev/null
wait $SMI 2>/dev/null
{ echo "# rocm-smi sampled twice a second around $N back-to-back searches of 65536 x 65536 (tools/run_search.py)"; cat "$OUT.run"; cat "$OUT.smi"; } > "$OUT"
rm -f "$OUT.smi" "$OUT.run"
