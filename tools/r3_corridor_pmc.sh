#!/bin/bash
# Run ON THE GPU BOX: counters of the corridor checker's two passes on config 3 (tools/time_corridor.py), then every
# BASELINE config and entry point under the kernel trace (tools/measure_configs.py).
#   bash tools/r3_corridor_pmc.sh <out dir under gpurun_out>
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/$1
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/trace -- python3 tools/time_corridor.py > $O/trace.json 2> $O/trace.err || echo "trace failed"
python3 tools/summarize_pmc.py --kernel-stats $O/trace $O/kernel_stats.csv > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -- python3 tools/time_corridor.py > $O/fetch.json 2> $O/fetch.err || echo "fetch failed"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -- python3 tools/time_corridor.py > $O/write.json 2> $O/write.err || echo "write failed"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/sq -- python3 tools/time_corridor.py > $O/sq.json 2> $O/sq.err || echo "sq failed"
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/sq2 -- python3 tools/time_corridor.py > $O/sq2.json 2> $O/sq2.err || echo "sq2 failed"
rocprofv3 --kernel-trace --stats -d $O/all_trace -- python3 tools/measure_configs.py > $O/configs.out 2> $O/configs.err || echo "configs trace failed"
grep "^{" $O/configs.out > $O/configs.jsonl   # (the C++ facade prints its own lines to stdout)
python3 tools/summarize_pmc.py --kernel-stats $O/all_trace $O/allkernels_stats.csv > /dev/null
rm -rf $O/all_trace
for k in "k_corridor<0" "k_corridor<1"; do
  n=$(echo "$k" | tr -dc '01')
  python3 tools/summarize_pmc.py $O/pmc_corridor_pass$n.json "$k" $O/fetch $O/write $O/sq $O/sq2 > /dev/null || echo "summary $k failed"
done
rm -rf $O/trace $O/fetch $O/write $O/sq $O/sq2
ls $O
