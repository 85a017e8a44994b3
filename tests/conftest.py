import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def olib():
    import oracle_lib
    oracle_lib.oracle()
    return oracle_lib


@pytest.fixture(scope="session")
def small_world():
    from trajectory_planner_amd import synth
    return synth.make_box_world(synth.SEED_BASE + 2, n=128, n_boxes=60, centre_range=5.5, z_range=2.0)


@pytest.fixture()
def vigo_handle():
    """A GPU handle; fails loudly (no skip) when the HIP library or the GPU is missing."""
    from trajectory_planner_amd.vigo import Vigo
    h = Vigo(0)
    yield h
    h.close()
