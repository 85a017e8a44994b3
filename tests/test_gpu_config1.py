"""-m gpu: BASELINE.json configs[0] — polyTrajOctomap min-snap on map/maze.bt with 8 waypoints, one
makePlan() in corridor mode (cfg/planner_interactive.yaml values).  The map is the committed fixture
tests/golden/maze_config1.npz (made from the reference's maze.bt by tests/golden/make_maze_fixture.py
with this repo's own .bt reader).  The min-snap QP runs on the host, the box sweep of every trajectory
sample on the device; the result is re-checked here with an independent numpy box sweep.
Parity unpinned: the reference holds no expected output for this configuration."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "trajectory_planner_amd", "lib", "libtrajectory_planner_vigo.so")
_dp = C.POINTER(C.c_double)


def load_maze():
    f = np.load(os.path.join(ROOT, "tests", "golden", "maze_config1.npz"))
    nx, ny, nz = (int(v) for v in f["dims"])
    n = nx * ny * nz
    occ = np.unpackbits(f["occ_bits"])[:n].reshape(nx, ny, nz)
    unk = np.unpackbits(f["unk_bits"])[:n].reshape(nx, ny, nz)
    vox = (occ * 4 + unk * 2 + occ).astype(np.uint8)        # no inflation: bit0 = occupied
    return vox, f["origin"].astype(np.float64), float(f["res"][0]), f["waypoints"].astype(np.float64), f


def numpy_box_sweep(vox, origin, res, pts, box, step):
    """PO.cpp:547-589 restated with numpy: float coordinates, floor(coord * (1/res)) keys, unknown or
    outside the grid => occupied"""
    nx, ny, nz = vox.shape
    key0 = np.round(origin / res).astype(np.int64)
    num = [int(box[a] / step) for a in range(3)]
    hit = np.zeros(len(pts), dtype=bool)
    for i in range(num[0] + 1):
        for j in range(num[1] + 1):
            for k in range(num[2] + 1):
                q = np.stack([pts[:, 0] - box[0] / 2 + i * step, pts[:, 1] - box[1] / 2 + j * step,
                              pts[:, 2] - box[2] / 2 + k * step], 1).astype(np.float32)
                idx = np.floor(q.astype(np.float64) * (1.0 / res)).astype(np.int64) - key0
                out = (idx < 0).any(1) | (idx[:, 0] >= nx) | (idx[:, 1] >= ny) | (idx[:, 2] >= nz)
                ic = np.clip(idx, 0, [nx - 1, ny - 1, nz - 1])
                hit |= out | ((vox[ic[:, 0], ic[:, 1], ic[:, 2]] & 6) != 0)
    return hit


def plan(vox, origin, res, wp, cfg):
    L = C.CDLL(LIB)
    L.vigo_host_poly_plan.argtypes = [C.c_int, C.c_int, C.c_int, _dp, C.c_double, C.c_void_p, C.c_int, _dp, _dp, _dp, C.c_int, _dp]
    traj = np.zeros((4096, 3))
    info = np.zeros(8)
    cfg = np.asarray(cfg, dtype=np.float64)
    v = np.ascontiguousarray(vox)
    w = np.ascontiguousarray(wp)
    rc = L.vigo_host_poly_plan(vox.shape[0], vox.shape[1], vox.shape[2], origin.ctypes.data_as(_dp), res, v.ctypes.data_as(C.c_void_p),
                               len(w), w.ctypes.data_as(_dp), cfg.ctypes.data_as(_dp), traj.ctypes.data_as(_dp), len(traj),
                               info.ctypes.data_as(_dp))
    assert rc == 0
    return traj[:int(info[2])], info


def test_config1_makeplan_on_the_maze():
    vox, origin, res, wp, _ = load_maze()
    # cfg/planner_interactive.yaml: box, map_resolution, delT, vel, r0, fs, corridor_res, max iter, timeout, mode=false
    cfg = [0.4, 0.4, 0.2, 0.2, 0.1, 1.0, 0.5, 0.8, 8.0, 100, 0.1, 0.0]
    traj, info = plan(vox, origin, res, wp, cfg)
    valid, iters, nsamp, duration, secs = bool(info[0]), int(info[1]), int(info[2]), info[3], info[4]
    print(f"config 1: valid={valid} corridor iterations={iters} samples={nsamp} duration={duration:.2f}s makePlan={secs * 1e3:.2f} ms")
    assert nsamp == len(traj) and nsamp >= int(duration / 0.1)
    assert np.allclose(traj[0], wp[0], atol=1e-6) and np.allclose(traj[-1], wp[-1], atol=1e-9)
    hits = numpy_box_sweep(vox, origin, res, traj, (0.4, 0.4, 0.2), 0.2)
    if valid:
        assert not hits.any(), f"{hits.sum()} samples of a trajectory reported valid collide"
        legs = np.linalg.norm(np.diff(wp, axis=0), axis=1)
        assert duration == pytest.approx(legs.sum() / 1.0, rel=1e-12)                # avgTimeAllocation, PS.cpp:125-138
    # the straight waypoint legs keep a clearance of >= 0.45 m by construction of the fixture, so the
    # corridor loop (r0 = 0.5, x0.8 per colliding segment) must find a valid trajectory
    assert valid and iters <= 100


def test_config1_adding_waypoint_mode_and_fallback():
    vox, origin, res, wp, _ = load_maze()
    cfg = [0.4, 0.4, 0.2, 0.2, 0.1, 1.0, 0.5, 0.8, 8.0, 100, 0.1, 1.0]      # mode = true (adding waypoints)
    traj, info = plan(vox, origin, res, wp, cfg)
    hits = numpy_box_sweep(vox, origin, res, traj, (0.4, 0.4, 0.2), 0.2)
    if info[0]:
        assert not hits.any()
    # a path through a wall cannot be planned: the planner reports it and falls back to the PWL path
    bad = np.array([wp[0], [wp[0][0] + 40.0, wp[0][1], 1.0]])
    traj, info = plan(vox, origin, res, bad, [0.4, 0.4, 0.2, 0.2, 0.1, 1.0, 0.5, 0.8, 8.0, 20, 0.1, 0.0])
    assert not info[0]
    assert np.allclose(traj[0], bad[0]) and np.allclose(traj[-1], bad[-1])
    assert info[3] == pytest.approx(40.0)                                    # PWL duration at 1 m/s
