#!/bin/bash
# Host-side sources (no HIP) under AddressSanitizer + UBSan: the .bt reader on every map given on the command
# line, 300 random min-snap QPs (with and without corridors, feasible and infeasible), B-spline fits/evaluations, A* searches, pwlTraj plans, soft-constraint QPs.
#   bash tools/sanitize_host.sh /root/reference/map/*.bt
# (GPU sanitizers are not available on the pool; the device code is covered by the parity tests and the fuzz sweep.)
set -e
cd "$(dirname "$0")/../trajectory_planner_amd/host"
g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude -o /tmp/vigo_san_test \
    ../../tools/sanitize_host_main.cpp src/octomapBt.cpp src/polyTrajSolver.cpp src/bspline.cpp src/astarOcc.cpp src/piecewiseLinearTraj.cpp
/tmp/vigo_san_test "$@"
