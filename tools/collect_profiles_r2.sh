#!/bin/bash
# Run ON THE GPU BOX: round-2 evidence.  bash tools/collect_profiles_r2.sh <commit>
#   1. tools/collect_profiles.sh for the default bench (fp64 reference order) and f64_fast: kernel-trace stats + PMC passes
#   2. kernel-trace stats of tools/measure_configs.py (every entry point)
#   3. FETCH_SIZE / WRITE_SIZE / SQ passes of the ESDF query (config 5b) and of the corridor checker (config 3)
set -o pipefail
export VIGO_BUILD=$1
export TMPDIR=/tmp
O=gpurun_out/r2prof
mkdir -p $O
bash tools/collect_profiles.sh r2 > $O/collect_r2.log 2>&1 || echo "collect r2 failed"
bash tools/collect_profiles.sh r2_fast --precision f64_fast > $O/collect_r2_fast.log 2>&1 || echo "collect r2_fast failed"
rocprofv3 --kernel-trace --stats -d $O/all_trace -- python3 tools/measure_configs.py > $O/configs.out 2> $O/configs.err || echo "configs trace failed"
grep "^{" $O/configs.out > $O/configs.jsonl   # (the C++ facade prints its own lines to stdout)
python3 tools/summarize_pmc.py --kernel-stats $O/all_trace $O/allkernels_stats.csv > /dev/null
for k in esdf corridor; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/${k}_fetch -- python3 tools/time_$k.py > $O/${k}_fetch.json 2> $O/${k}_fetch.err || echo "$k fetch failed"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/${k}_write -- python3 tools/time_$k.py > $O/${k}_write.json 2> $O/${k}_write.err || echo "$k write failed"
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/${k}_sq -- python3 tools/time_$k.py > $O/${k}_sq.json 2> $O/${k}_sq.err || echo "$k sq failed"
done
python3 tools/summarize_pmc.py --halves $O/pmc_esdf.json k_esdf_query $O/esdf_fetch $O/esdf_write $O/esdf_sq > /dev/null
python3 tools/summarize_pmc.py $O/pmc_corridor.json k_corridor $O/corridor_fetch $O/corridor_write $O/corridor_sq > /dev/null
python3 bench.py > $O/bench.json 2> $O/bench.err
ls $O gpurun_out/prof_r2 gpurun_out/prof_r2_fast | head -60
