"""Generates tests/golden/maze_config1.npz — the data of BASELINE.json configs[0]
("polyTrajOctomap min-snap on map/maze.bt, 8 RRT waypoints").  Run in the authoring container:

    python tests/golden/make_maze_fixture.py

Input: the reference's DATA file map/maze.bt (an OctoMap binary tree), parsed by this repo's own
reader (trajectory_planner_amd/host/src/octomapBt.cpp).  Output (data only, no reference source):
    occ_bits, unk_bits   np.packbits of the dense occupied / unknown voxel planes [nx, ny, nz]
    dims, origin, res    geometry of the dense grid (origin on the octomap key lattice)
    nodes, occupied, free   counters of the tree (nodes == the file's "size" header)
    waypoints            8 free-space waypoints at z = 1.0 (cfg/planner_interactive.yaml env_box),
                         chosen by a seeded (seed 1) straight-line-visibility random walk — the
                         reference takes them from the external global_planner RRT, which is absent
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(ROOT, "trajectory_planner_amd", "lib", "libtrajectory_planner_vigo.so")
BT = "/root/reference/map/maze.bt"
_dp = C.POINTER(C.c_double)


def box_free(vox, origin, res, p, box, step):
    """polyTrajOctomap::checkCollision (PO.cpp:547-568) on the dense grid: unknown or occupied or outside => hit"""
    nx, ny, nz = vox.shape
    lo = [p[a] - box[a] / 2 for a in range(3)]
    num = [int((box[a]) / step) for a in range(3)]
    for i in range(num[0] + 1):
        for j in range(num[1] + 1):
            for k in range(num[2] + 1):
                q = (lo[0] + i * step, lo[1] + j * step, lo[2] + k * step)
                idx = [int(np.floor(np.float32(q[a]) * np.float32(1.0 / res))) - int(round(origin[a] / res)) for a in range(3)]
                if min(idx) < 0 or idx[0] >= nx or idx[1] >= ny or idx[2] >= nz:
                    return False
                if vox[idx[0], idx[1], idx[2]] & 6:
                    return False
    return True


def main():
    L = C.CDLL(LIB)
    L.vigo_host_bt_info.argtypes = [C.c_char_p, C.POINTER(C.c_longlong), _dp, _dp]
    L.vigo_host_bt_load.argtypes = [C.c_char_p, _dp, C.c_int, C.c_void_p, C.c_longlong, C.POINTER(C.c_int), _dp]
    info = (C.c_longlong * 8)()
    origin = (C.c_double * 3)()
    res = C.c_double()
    assert L.vigo_host_bt_info(BT.encode(), info, origin, C.byref(res)) == 0
    assert info[0] == info[1]
    dims = (C.c_int * 3)()
    inflate = (C.c_double * 3)(0, 0, 0)
    buf = np.zeros(int(info[5] * info[6] * info[7]) + 4096, dtype=np.uint8)
    assert L.vigo_host_bt_load(BT.encode(), inflate, 0, buf.ctypes.data_as(C.c_void_p), buf.size, dims, origin) == 0
    nx, ny, nz = dims[0], dims[1], dims[2]
    vox = buf[:nx * ny * nz].reshape(nx, ny, nz)
    org = np.array([origin[0], origin[1], origin[2]])
    r = res.value

    # seeded straight-line-visibility random walk at z = 1.0 with the cfg collision box
    rng = np.random.default_rng(1)
    box, step = (0.4, 0.4, 0.2), 0.2
    # a leg must keep a 0.5 m corridor's worth of clearance so the corridor loop has something to find
    clear = (0.9, 0.9, 0.4)

    def leg_free(a, b):
        n = int(np.ceil(np.linalg.norm(b - a) / 0.05))
        return all(box_free(vox, org, r, a + (b - a) * t / n, clear, step) for t in range(n + 1))

    lo = org + 1.0
    hi = org + np.array([nx, ny, nz]) * r - 1.0
    while True:
        p = np.array([rng.uniform(lo[0], hi[0]), rng.uniform(lo[1], hi[1]), 1.0])
        if box_free(vox, org, r, p, clear, step):
            break
    wps = [p]
    tries = 0
    while len(wps) < 8:
        tries += 1
        assert tries < 200000
        ang = rng.uniform(0, 2 * np.pi)
        d = rng.uniform(1.5, 4.0)
        q = wps[-1] + np.array([d * np.cos(ang), d * np.sin(ang), 0.0])
        if not (lo[0] <= q[0] <= hi[0] and lo[1] <= q[1] <= hi[1]):
            continue
        if any(np.linalg.norm(q - w) < 1.5 for w in wps):
            continue
        if leg_free(wps[-1], q):
            wps.append(q)
    wps = np.array(wps)
    out = os.path.join(HERE, "maze_config1.npz")
    np.savez_compressed(out, occ_bits=np.packbits((vox & 4) != 0), unk_bits=np.packbits((vox & 2) != 0),
                        dims=np.array([nx, ny, nz]), origin=org, res=np.array([r]),
                        nodes=np.array([info[0]]), occupied=np.array([info[3]]), free=np.array([info[4]]), waypoints=wps)
    print("wrote", out, os.path.getsize(out), "bytes; dims", (nx, ny, nz), "origin", org, "waypoints:\n", wps)


if __name__ == "__main__":
    sys.exit(main())
