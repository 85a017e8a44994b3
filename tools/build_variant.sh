#!/bin/bash
# dev: an alternative build of the solve kernels for A/B timing (tools/exp_solver.py, VIGO_EXP_LIB=trajectory_planner_amd/lib/exp/libvigo_NAME.so)
#   bash tools/build_variant.sh NAME "-DVIGO_RING_TABLE=0 ..."
set -e
cd "$(dirname "$0")/../trajectory_planner_amd/csrc"
NAME=$1; shift
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -mllvm -amdgpu-sched-strategy=max-ilp $@"
mkdir -p build/exp_$NAME ../lib/exp
/opt/rocm/bin/hipcc $FL -mllvm -disable-machine-licm -DVIGO_SOLVER_PART=0 -x hip -c vigo_solver.hip -o build/exp_$NAME/solver.o &
/opt/rocm/bin/hipcc $FL -DVIGO_SOLVER_PART=1 -x hip -c vigo_solver.hip -o build/exp_$NAME/solver_obs.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/exp/libvigo_$NAME.so build/exp_$NAME/solver.o build/exp_$NAME/solver_obs.o \
  build/vigo_api.cpp.o build/vigo_map.hip.o build/vigo_corridor.hip.o build/vigo_fit.hip.o build/vigo_minsnap.hip.o
echo built ../lib/exp/libvigo_$NAME.so
