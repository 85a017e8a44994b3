#!/bin/bash
# Run ON THE GPU BOX (gpurun): kernel-trace stats + separate PMC passes of the default bench, written
# under gpurun_out/prof_$1/.  Copy the summaries into profiles/ afterwards (see profiles/README.md).
#   bash tools/collect_profiles.sh TAG [extra bench args]      (KERNEL_FILTER=<substring of the kernel name>, default k_optimize)
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 50 --warmup 5 --no-cpu-baseline --no-extras $@"
rocprofv3 --kernel-trace --stats -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.log || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.log || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.log || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc_sq -- python3 bench.py $ARGS > $OUT/bench_sq.json 2> $OUT/sq.log || true
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS -d $OUT/pmc_sq2 -- python3 bench.py $ARGS > $OUT/bench_sq2.json 2> $OUT/sq2.log || true
python3 tools/summarize_pmc.py $OUT/pmc_summary.json "${KERNEL_FILTER:-k_optimize}" $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_sq2 > /dev/null
python3 tools/summarize_pmc.py --kernel-stats $OUT/trace $OUT/kernel_stats.csv
# the raw rocprofv3 databases are tens of MiB per pass: only the summaries travel back (gpurun_out/ is capped at 64 MiB)
[ -n "$KEEP_RAW" ] || rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_sq2
cat $OUT/pmc_summary.json | head -40
