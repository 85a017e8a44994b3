"""-m gpu: the C++ host facades (trajPlanner::bsplineTraj / polyTrajOctomap with the reference's
method names) driven like the reference's nodes drive the originals, through the C ABI."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_facade_program():
    exe = os.path.join(ROOT, "trajectory_planner_amd", "lib", "test_facade")
    assert os.path.exists(exe), "build it with `make -C trajectory_planner_amd/host` (__graft_entry__.build())"
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(p.stdout[-4000:])
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "PASSED" in p.stdout and "FAIL " not in p.stdout
