#!/bin/bash
# Copies the summaries tools/profile.sh left under gpurun_out/$ROUND/ into profiles/ (tracked), prefixed with the round.
set -euo pipefail
cd "$(dirname "$0")/.."
ROUND="${ROUND:-r04}"
OUT=gpurun_out/$ROUND
for f in bench_n1.json bench_loop_closure_n1.json bench_loop_closure_2ranks_one_gpu.json bench_2ranks_one_gpu.json bench_default_under_rocprof.json \
         bench_under_rocprof.json bench_pipelined.json sq_counters.json sq_counters_shard_8192x65536.json cycles.log sizes.log \
         latency_small_calls.log update_path_share.log ba_timing.log pose_lm_probe.log block_timeline.log perfect_bound_ceiling.log; do
    [ -s "$OUT/$f" ] && cp "$OUT/$f" "profiles/${ROUND}_$f"
done
cp "$OUT/hbm_counters.json" profiles/hbm_counters.json
cp "$OUT/hbm_counters.json" "profiles/${ROUND}_hbm_counters.json"
d=$(find "$OUT/stats_default" -name "*kernel_stats.csv" | head -1); [ -n "$d" ] && cp "$d" "profiles/${ROUND}_bench_default_kernel_stats.csv"
l=$(find "$OUT/stats_long" -name "*kernel_stats.csv" | head -1); [ -n "$l" ] && cp "$l" "profiles/${ROUND}_bench_kernel_stats.csv"
c=$(find "$OUT/stats_4096" -name "*kernel_stats.csv" | head -1); [ -n "$c" ] && cp "$c" "profiles/${ROUND}_search_4096x4096_kernel_stats.csv"
h=$(find "$OUT/stats_shard" -name "*kernel_stats.csv" | head -1); [ -n "$h" ] && cp "$h" "profiles/${ROUND}_search_8192x65536_kernel_stats.csv"
h=$(find "$OUT/stats_shard4" -name "*kernel_stats.csv" | head -1); [ -n "$h" ] && cp "$h" "profiles/${ROUND}_search_16384x65536_kernel_stats.csv"
h=$(find "$OUT/stats_shard2" -name "*kernel_stats.csv" | head -1); [ -n "$h" ] && cp "$h" "profiles/${ROUND}_search_32768x65536_kernel_stats.csv"
ls -la profiles | grep "${ROUND}_\|hbm_counters.json"
