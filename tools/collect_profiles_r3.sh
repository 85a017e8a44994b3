#!/bin/bash
# Run ON THE GPU BOX: round-3 evidence.  bash tools/collect_profiles_r3.sh
#   1. tools/collect_profiles.sh (kernel-trace stats + the PMC passes, each in its own run) for every solver shape bench.py
#      quotes: configs[1] (B = 1024), the same map at B = 16 384, one GPU's shard of configs[3] (8192 x 64, 512^3), f64_fast
#   2. kernel-trace stats of the default bench.py (all extras) and of tools/measure_configs.py (every entry point)
#   3. FETCH_SIZE / WRITE_SIZE / SQ passes of the fp32-I/O ESDF query (config 5b)
# The summaries record vigo_build_id() of the library they were taken from (tools/summarize_pmc.py).
set -o pipefail
export TMPDIR=/tmp
export KERNEL_FILTER=', false, 2, '   # the level instantiation of k_optimize (D = 2)
O=gpurun_out/r3prof
mkdir -p $O
bash tools/collect_profiles.sh r3 > $O/collect_r3.log 2>&1 || echo "collect r3 failed"
bash tools/collect_profiles.sh r3_b16384 --batch 16384 > $O/collect_r3_b16384.log 2>&1 || echo "collect r3_b16384 failed"
bash tools/collect_profiles.sh r3_config4 --workload config4 > $O/collect_r3_config4.log 2>&1 || echo "collect r3_config4 failed"
bash tools/collect_profiles.sh r3_fast --precision f64_fast > $O/collect_r3_fast.log 2>&1 || echo "collect r3_fast failed"
rocprofv3 --kernel-trace --stats -d $O/bench_trace -- python3 bench.py --no-cpu-baseline > $O/bench_traced.json 2> $O/bench_traced.err || echo "bench trace failed"
python3 tools/summarize_pmc.py --kernel-stats $O/bench_trace $O/bench_allkernels_stats.csv > /dev/null
rm -rf $O/bench_trace
rocprofv3 --kernel-trace --stats -d $O/all_trace -- python3 tools/measure_configs.py > $O/configs.out 2> $O/configs.err || echo "configs trace failed"
grep "^{" $O/configs.out > $O/configs.jsonl
python3 tools/summarize_pmc.py --kernel-stats $O/all_trace $O/allkernels_stats.csv > /dev/null
rm -rf $O/all_trace
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/esdf_fetch -- python3 tools/time_esdf.py --f32 > $O/esdf_fetch.json 2> $O/esdf_fetch.err || echo "esdf fetch failed"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/esdf_write -- python3 tools/time_esdf.py --f32 > $O/esdf_write.json 2> $O/esdf_write.err || echo "esdf write failed"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/esdf_sq -- python3 tools/time_esdf.py --f32 > $O/esdf_sq.json 2> $O/esdf_sq.err || echo "esdf sq failed"
python3 tools/summarize_pmc.py --halves $O/pmc_esdf_f32.json k_esdf_query_f32 $O/esdf_fetch $O/esdf_write $O/esdf_sq > /dev/null
rm -rf $O/esdf_fetch $O/esdf_write $O/esdf_sq
python3 bench.py > $O/bench.json 2> $O/bench.err
ls $O
