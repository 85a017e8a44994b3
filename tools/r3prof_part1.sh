set -o pipefail
export TMPDIR=/tmp
# the level instantiation (last template argument D = 2) does the work on these batches; the general one, launched before it,
# exits at once on level waves and is listed in the kernel-stats files
export KERNEL_FILTER=', false, 2, '
O=gpurun_out/r3prof
mkdir -p $O
bash tools/collect_profiles.sh r3 > $O/collect_r3.log 2>&1 || echo "collect r3 failed"
bash tools/collect_profiles.sh r3_b16384 --batch 16384 > $O/collect_r3_b16384.log 2>&1 || echo "collect r3_b16384 failed"
bash tools/collect_profiles.sh r3_config4 --workload config4 > $O/collect_r3_config4.log 2>&1 || echo "collect r3_config4 failed"
ls gpurun_out/prof_r3 gpurun_out/prof_r3_b16384 gpurun_out/prof_r3_config4 | head -50
