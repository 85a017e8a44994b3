"""Randomised parity sweep of vigo_corridor_check and vigo_box_collision_points against the oracle (bit for bit):
collision boxes from 0.1 to 1.3 m per axis, map_resolution 0.05-0.45 (fast per-axis path and the generic walk), grid
resolutions 0.1 / 0.05, polynomial degrees 3-9, ragged sample counts, metric bounds inside and outside the grid.
Not part of the test suite; run on the GPU box:  python tools/fuzz_corridor.py [cases] [seed]"""
import ctypes as C, json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import trajectory_planner_amd._lib as L
if os.environ.get("VIGO_EXP_LIB"):                      # dev: an alternative build of the library
    L.LIB_PATH = os.path.join(R, os.environ["VIGO_EXP_LIB"])
import numpy as np, torch
import oracle_lib as ol
from gpu_util import to_dev
from trajectory_planner_amd import synth
from trajectory_planner_amd.vigo import Vigo

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
O = ol.oracle()
bad = 0
t0 = time.time()
for case in range(cases):
    res = float(rng.choice([0.1, 0.05]))
    n = int(rng.choice([48, 96]))
    vox = np.zeros((n, n, 24), dtype=np.uint8)
    for _ in range(int(rng.integers(5, 40))):
        c = rng.integers(2, n - 2, size=2); s = rng.integers(1, 5, size=2)
        vox[max(c[0] - s[0], 0):c[0] + s[0], max(c[1] - s[1], 0):c[1] + s[1], 0:rng.integers(4, 24)] |= 4
    unk = rng.random((n // 8, n // 8, 3)) < 0.08
    vox[np.repeat(np.repeat(np.repeat(unk, 8, 0), 8, 1), 8, 2)] |= 2
    origin = np.array([-n * res / 2, -n * res / 2, -0.5]).round(6)
    origin = np.round(origin / res) * res                                   # octomap key lattice
    world = synth.World(vox, origin, res, np.zeros((0, 6)))
    v = Vigo(0)
    v.set_grid(to_dev(vox, v.device), origin, res)
    g, keep = ol.make_grid(world)
    if rng.random() < 0.3:
        bmin = origin + rng.uniform(0.2, 1.0, 3) * [1, 1, 0.2]; bmax = origin + np.array(vox.shape) * res - rng.uniform(0.2, 1.0, 3) * [1, 1, 0.2]
        v.set_metric_bounds(bmin, bmax); g.bmin[:] = list(bmin); g.bmax[:] = list(bmax)
    box = rng.uniform(0.1, 1.3, size=3) * [1, 1, 0.5]
    map_res = float(rng.choice([0.05, 0.1, 0.2, 0.25, 0.45]))
    if rng.random() < 0.4:
        # box an exact multiple of map_resolution: the lattice count (int)((xmax - xmin) / map_res) then wobbles between
        # two values with the rounding of fx +- box / 2 (the span certificates' nlo != nhi case)
        map_res = float(rng.choice([0.1, 0.2]))
        box = map_res * rng.integers(1, 4, size=3).astype(np.float64)
    deg = int(rng.integers(3, 10)) if rng.random() < 0.6 else 7
    half = n * res / 2
    S = 24
    # sample counts: one sample per lane (<= 512), certified spans of 16 / 64 samples above that
    n_samples = int(rng.choice([int(rng.integers(50, 1500)), int(rng.integers(1500, 6000)), int(rng.integers(6000, 20000))]))
    coeffs, n_samp, delT, dur = synth.make_corridor_segments(int(rng.integers(1 << 30)), S, deg=deg, extent_lo=(-half * 0.9, -half * 0.9, 0.0),
                                                             extent_hi=(half * 0.9, half * 0.9, 1.6), n_samples=n_samples)
    n_samp[:4] = [0, 1, 17, 33]
    n_samp[4:8] = [511, 512, 513, 1025]
    # segments the certificates must refuse or decide as a whole: fast ones (samples further apart than a voxel), one that
    # leaves the map, one that starts far outside, non-finite coefficients, clocks that do not advance or run backwards
    coeffs[8, :, 1:] *= 40.0
    coeffs[9, :, 1] *= 6.0
    coeffs[10, 0, 0] += 3.0 * half
    coeffs[11, 1, 0] = -half + 0.05
    coeffs[11, 1, 1] = -abs(coeffs[11, 1, 1])
    if rng.random() < 0.5:
        coeffs[12, int(rng.integers(0, 3)), int(rng.integers(0, deg + 1))] = float(rng.choice([np.nan, np.inf, -np.inf, 1e300, 1e39, -4e38, 1e20, 3.4028234e38]))
    delT[13] = 0.0
    delT[14] = -delT[14]
    delT[15] = delT[15] * 1e-9
    n_samp[13:15] = np.minimum(n_samp[13:15], 3000)         # (the oracle and the device walk these clocks step by step)
    # boundary huggers: a lattice point of the box rides a voxel face (or the metric bound) while the pose creeps by
    # micrometres, so that the rounding of the float position alone decides keys and counts from sample to sample
    for sgm in range(16, 22):
        a = int(rng.integers(0, 3))
        face = origin[a] + res * float(rng.integers(2, vox.shape[a] - 2))
        off = float(rng.choice([-1.0, 0.0, 1.0])) * box[a] / 2 + float(rng.choice([0.0, map_res, 2 * map_res]))
        coeffs[sgm, a, :] = 0.0
        coeffs[sgm, a, 0] = face - off + float(rng.choice([0.0, 1e-7, -1e-7, 3e-6, -3e-6]))
        coeffs[sgm, a, 1] = float(rng.choice([0.0, 1e-6, -1e-6, 2e-5, -2e-5, 1e-3]))
        if rng.random() < 0.5:
            coeffs[sgm, :, 2:] *= 0.01                       # and slow elsewhere: long certified spans around the flicker
    flag, first, count = (x.cpu().numpy() for x in v.corridor_check(to_dev(coeffs, v.device), to_dev(n_samp, v.device), to_dev(delT, v.device), box, map_res))
    ok = True
    for s in range(S):
        fi, cn = C.c_int(), C.c_int()
        c = np.ascontiguousarray(coeffs[s])
        f = O.vgo_corridor_check_segment(C.byref(g), deg, ol._d(c), int(n_samp[s]), float(delT[s]), ol._d(box), map_res, C.byref(fi), C.byref(cn))
        if (f, fi.value, cn.value) != (flag[s], first[s], count[s]):
            ok = False
            print(json.dumps({"segment": s, "n": int(n_samp[s]), "oracle": [int(f), fi.value, cn.value], "device": [int(flag[s]), int(first[s]), int(count[s])]}), flush=True)
    # the per-pose sweep on random poses
    pts = rng.uniform(-half * 1.1, half * 1.1, size=(400, 3)) * [1, 1, 0.2] + [0, 0, 0.8]
    got = v.box_collision_points(to_dev(pts, v.device), box, map_res).cpu().numpy()
    for i in range(len(pts)):
        ok = ok and got[i] == O.vgo_box_collision(C.byref(g), C.c_float(pts[i, 0]), C.c_float(pts[i, 1]), C.c_float(pts[i, 2]), ol._d(box), C.c_double(map_res))
    if not ok:
        bad += 1
        print(json.dumps({"MISMATCH": case, "res": res, "box": box.tolist(), "map_res": map_res, "deg": deg}), flush=True)
    v.close()
print(json.dumps({"cases": cases, "mismatches": bad, "seconds": time.time() - t0}))
sys.exit(1 if bad else 0)
