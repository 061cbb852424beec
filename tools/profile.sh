#!/bin/bash
# Collects the evidence of a round on the GPU box into gpurun_out/$ROUND/ (default r03) in the order that keeps the kept
# set self-consistent: first the PMC passes (-> profiles/hbm_counters.json, stamped with the kernel source hashes), THEN
# the bench line that reads it (so the kept line carries roofline.traffic), then kernel-trace stats, SQ counters, the
# in-kernel clock, sizes, latencies and the optimiser timings.  Copy what is to be judged into profiles/ afterwards:
#     tools/profile.sh && tools/keep_profiles.sh
# Counters are collected in their own runs with --kernel-trace only (no sys / hip / hsa trace domains next to --pmc), and
# the program itself follows `--` (python3 bench.py ...): no env / bash -c hop between rocprofv3 and the GPU process.
set -uo pipefail
cd "$(dirname "$0")/.."
ROUND="${ROUND:-r04}"
OUT=gpurun_out/$ROUND
mkdir -p $OUT
export TMPDIR=/tmp
PY=python3
B="bench.py --steps 5 --warmup 1 --no-cpu-baseline"

rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- $PY $B > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- $PY $B > /dev/null 2> $OUT/pmc_write.err
$PY tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/hbm_counters.json && cp $OUT/hbm_counters.json profiles/hbm_counters.json
echo "hbm counters done: $(grep -c traffic_bytes $OUT/hbm_counters.json) kernels"

$PY bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
echo "bench done: $(cut -c1-200 $OUT/bench_n1.json)"
# PROFILE_ONLY=hbm: the counters and the line that reads them only (after an edit that touched a kernel source's text)
if [ "${PROFILE_ONLY:-}" = "hbm" ]; then
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -o stats -- $PY bench.py > $OUT/bench_default_under_rocprof.json 2> $OUT/stats_default.err
    exit 0
fi
$PY bench.py --workload loop-closure --steps 10 --warmup 1 > $OUT/bench_loop_closure_n1.json 2> $OUT/bench_loop_closure_n1.err
echo "loop closure done: $(cut -c1-160 $OUT/bench_loop_closure_n1.json)"
SLAM_BENCH_SINGLE_DEVICE=1 SLAM_BENCH_COLLECTIVE=p2p $PY bench.py --gpus 2 --workload loop-closure --steps 3 --warmup 0 > $OUT/bench_loop_closure_2ranks_one_gpu.json 2> $OUT/bench_loop_closure_2ranks_one_gpu.err
SLAM_BENCH_SINGLE_DEVICE=1 SLAM_BENCH_COLLECTIVE=p2p $PY bench.py --gpus 2 --steps 10 --warmup 2 > $OUT/bench_2ranks_one_gpu.json 2> $OUT/bench_2ranks_one_gpu.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -o stats -- $PY bench.py > $OUT/bench_default_under_rocprof.json 2> $OUT/stats_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_long -o stats -- $PY bench.py --steps 400 --warmup 20 --no-reproj --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats_long.err
# BASELINE configs[1] (4096 x 4096) and the 1/8 shard of the headline grid under the same profiler
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_4096 -o stats -- $PY tools/run_search.py 4096x4096 400 > /dev/null 2> $OUT/stats_4096.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_shard -o stats -- $PY tools/run_search.py 8192x65536 200 > /dev/null 2> $OUT/stats_shard.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_shard4 -o stats -- $PY tools/run_search.py 16384x65536 200 > /dev/null 2> $OUT/stats_shard4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_shard2 -o stats -- $PY tools/run_search.py 32768x65536 200 > /dev/null 2> $OUT/stats_shard2.err
echo "kernel stats done"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/sq1 -o sq1 -- $PY $B --no-reproj > /dev/null 2> $OUT/sq1.err
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq2 -o sq2 -- $PY $B --no-reproj > /dev/null 2> $OUT/sq2.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq3 -o sq3 -- $PY $B --no-reproj > /dev/null 2> $OUT/sq3.err
$PY tools/sq_summary.py $OUT/sq1 $OUT/sq2 $OUT/sq3 > $OUT/sq_counters.json
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/sq_shard -o sq -- $PY tools/run_search.py 8192x65536 30 > /dev/null 2> $OUT/sq_shard.err
SQ_N=8192 SQ_M=65536 $PY tools/sq_summary.py $OUT/sq_shard > $OUT/sq_counters_shard_8192x65536.json
echo "sq counters done"
bash tools/build_exp.sh > /dev/null 2>&1
$PY tools/cycle_probe.py 65536x65536 8192x65536 4096x4096 > $OUT/cycles.log 2>&1
( $PY tools/trace_probe.py 8192x65536; $PY tools/trace_probe.py 65536x65536; $PY tools/trace_probe.py 8192x65536 0 0 0 -1 ) > $OUT/block_timeline.log 2>&1
$PY tools/ab_time.py slam-experiments_amd/lib/libslamhip.so,tools/exp/libslamhip_keepbound.so 8192x65536 16384x65536 65536x65536 --rounds 3 > $OUT/perfect_bound_ceiling.log 2>&1
$PY tools/size_probe.py > $OUT/sizes.log 2>&1
$PY tools/latency.py > $OUT/latency_small_calls.log 2>&1
$PY tools/fire_probe.py > $OUT/update_path_share.log 2>&1
$PY tools/ba_time.py > $OUT/ba_timing.log 2>&1
$PY tools/pose_lm_probe.py > $OUT/pose_lm_probe.log 2>&1
$PY bench.py --pipelined --no-reproj --no-cpu-baseline > $OUT/bench_pipelined.json 2> $OUT/bench_pipelined.err
find $OUT -name "*kernel_stats.csv" | head
# the trace databases are large; keep the csv summaries only
find $OUT -name "*.db" -delete 2>/dev/null
du -sh $OUT
