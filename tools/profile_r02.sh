#!/bin/bash
# Collects the round-2 evidence on the GPU box into gpurun_out/r02/ (copy what is to be judged into profiles/):
# kernel-trace stats of the default bench command, the two HBM PMC passes, two SQ passes, the size and latency logs.
# Counters are collected in their own runs with --kernel-trace only (no sys/hip/hsa trace domains next to --pmc).
set -uo pipefail
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02
mkdir -p $OUT
export TMPDIR=/tmp
PY=python3
B="bench.py --steps 5 --warmup 1 --no-cpu-baseline"

$PY bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
echo "bench done: $(cut -c1-160 $OUT/bench_n1.json)"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -o stats -- $PY bench.py > $OUT/bench_default_under_rocprof.json 2> $OUT/stats_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_long -o stats -- $PY bench.py --steps 400 --warmup 20 --no-reproj --no-cpu-baseline > $OUT/bench_long_under_rocprof.json 2> $OUT/stats_long.err
echo "kernel stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- $PY $B > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- $PY $B > /dev/null 2> $OUT/pmc_write.err
$PY tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/hbm_counters.json
echo "hbm counters done"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/sq1 -o sq1 -- $PY $B --no-reproj > /dev/null 2> $OUT/sq1.err
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq2 -o sq2 -- $PY $B --no-reproj > /dev/null 2> $OUT/sq2.err
$PY tools/sq_summary.py $OUT/sq1 $OUT/sq2 > $OUT/sq_counters.json
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/sq_shard -o sq -- $PY tools/run_search.py 8192x65536 30 > /dev/null 2> $OUT/sq_shard.err
SQ_N=8192 SQ_M=65536 $PY tools/sq_summary.py $OUT/sq_shard > $OUT/sq_counters_shard_8192x65536.json
echo "sq counters done"
$PY tools/size_probe.py > $OUT/sizes.log 2>&1
$PY tools/latency.py > $OUT/latency_small_calls.log 2>&1
find $OUT -name "*kernel_stats.csv" | head
# the trace databases are large; keep the csv summaries only
find $OUT -name "*.db" -delete 2>/dev/null
du -sh $OUT
$PY bench.py --pipelined --no-reproj --no-cpu-baseline > $OUT/bench_pipelined.json 2> $OUT/bench_pipelined.err
$PY tools/fire_probe.py > $OUT/update_path_share.log 2>&1
$PY tools/ba_time.py > $OUT/ba_schur_timing.log 2>&1
