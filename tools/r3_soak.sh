#!/bin/bash
# closing soak of round 3, ON THE GPU BOX: the randomised sweeps on the final library (none is part of the test suite)
O=gpurun_out/r3soak
mkdir -p $O
timeout -k 10 600 python tools/fuzz_solver.py 20000 77 > $O/fuzz_solver.log 2>&1; echo "fuzz_solver rc $?"; tail -1 $O/fuzz_solver.log
timeout -k 10 200 python tools/fuzz_map_gates.py 2000 78 > $O/fuzz_map.log 2>&1; echo "fuzz_map rc $?"; tail -1 $O/fuzz_map.log
timeout -k 10 200 python tools/fuzz_corridor.py 3000 79 > $O/fuzz_corridor.log 2>&1; echo "fuzz_corridor rc $?"; tail -1 $O/fuzz_corridor.log
timeout -k 10 200 python tools/fuzz_minsnap.py 200 80 > $O/fuzz_minsnap.log 2>&1; echo "fuzz_minsnap rc $?"; tail -1 $O/fuzz_minsnap.log
VIGO_FACADE_FUZZ=12 timeout -k 10 400 trajectory_planner_amd/lib/test_facade > $O/facade.log 2>&1; echo "facade rc $?"; grep "^FAIL\|^INFO stress" $O/facade.log | cut -c1-300
