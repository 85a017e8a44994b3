#!/bin/bash
# after tools/r3prof_part1.sh + part2.sh came back through gpurun: copy the summaries into profiles/
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r3prof
cp gpurun_out/prof_r3/pmc_summary.json profiles/pmc_config2_f64.json
cp gpurun_out/prof_r3_b16384/pmc_summary.json profiles/pmc_config2_f64_b16384.json
cp gpurun_out/prof_r3_config4/pmc_summary.json profiles/pmc_config4_f64.json
cp gpurun_out/prof_r3_fast/pmc_summary.json profiles/pmc_config2_f64_fast.json
cp gpurun_out/prof_r3/kernel_stats.csv profiles/r03_kernel_stats.csv
cp gpurun_out/prof_r3_b16384/kernel_stats.csv profiles/r03_b16384_kernel_stats.csv
cp gpurun_out/prof_r3_config4/kernel_stats.csv profiles/r03_config4_kernel_stats.csv
cp gpurun_out/prof_r3_fast/kernel_stats.csv profiles/r03_fast_kernel_stats.csv
cp $O/bench_allkernels_stats.csv profiles/r03_bench_allkernels_stats.csv
cp $O/allkernels_stats.csv profiles/r03_allkernels_stats.csv
cp $O/configs.jsonl profiles/r03_configs.jsonl
cp $O/pmc_esdf_f32.json profiles/r03_pmc_esdf.json
cp $O/facade.log profiles/r03_facade_timing.log
python3 - <<'PY'
import json
for f in ("pmc_config2_f64", "pmc_config2_f64_b16384", "pmc_config4_f64", "pmc_config2_f64_fast"):
    e = json.load(open(f"profiles/{f}.json")); c = e["counters_per_launch"]
    print(f, e["build_id"], e["hbm_bytes_per_launch"], round(c["SQ_INSTS_VALU"]), round(c["SQ_WAIT_INST_ANY"]), round(c["SQ_WAVE_CYCLES"]), round(c["SQ_WAVES"]))
PY
for f in r03_kernel_stats r03_b16384_kernel_stats r03_config4_kernel_stats r03_fast_kernel_stats; do grep k_optimize profiles/$f.csv | cut -c40-160; done
