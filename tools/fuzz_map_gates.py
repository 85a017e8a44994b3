"""Randomised parity sweep of the map / gate / spline entry points against the CPU oracle (bit for bit): random grid
shapes (odd extents, z not a multiple of 32), origins, resolutions, control-point counts, sample steps, obstacle
lists, fit sizes and inflation radii.  Covers vigo_query_points, vigo_guides_unknown, vigo_traj_collision,
vigo_traj_dynamic_collision, vigo_ctrl_occupancy, vigo_bspline_eval, vigo_bspline_fit (1e-9 vs the oracle's
pivoted QR), vigo_inflate_grid, vigo_pack_grid, vigo_box_collision_points and vigo_esdf_query.  Not part of the test suite; run on the GPU box:
python tools/fuzz_map_gates.py [cases] [seed]"""
import ctypes as C, json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import oracle_lib as ol
from gpu_util import to_dev
from trajectory_planner_amd import sharding, synth
from trajectory_planner_amd.vigo import Vigo, default_params

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 31)
O = ol.oracle()
bad = 0
skipped_box = 0
t0 = time.time()


def fail(case, what, **kw):
    global bad
    bad += 1
    print(json.dumps({"case": case, "what": what, **kw}), flush=True)


for case in range(cases):
    dims = tuple(int(x) for x in rng.integers(3, 90, size=3))
    res = float(rng.choice([0.1, 0.2, 0.05, 0.13]))
    origin = np.round(rng.uniform(-6, 2, size=3) / res) * res if rng.random() < 0.5 else rng.uniform(-6, 2, size=3)
    vox = ((rng.random(dims) < 0.03).astype(np.uint8) * 4 | (rng.random(dims) < 0.1).astype(np.uint8) * 2).astype(np.uint8)
    r = tuple(int(x) for x in rng.integers(0, 4, size=3))
    P = default_params()
    P.ts_ctrl = float(rng.choice([0.1, 0.25, 0.07]))
    v = Vigo(0, P, 0)
    # inflation (bit0 from bit2) against a numpy dilation, then the inflated bytes become the snapshot
    inflated = v.inflate_grid(to_dev(vox, v.device), *r).cpu().numpy()
    ref = (vox & 4) != 0
    for axis, rr in enumerate(r):
        acc = ref.copy()
        for d in range(1, rr + 1):
            a = [slice(None)] * 3; b = [slice(None)] * 3
            a[axis] = slice(d, None); b[axis] = slice(None, -d)
            acc[tuple(a)] |= ref[tuple(b)]
            acc[tuple(b)] |= ref[tuple(a)]
        ref = acc
    if not (np.array_equal((inflated & 1) != 0, ref) and np.array_equal(inflated & 6, vox & 6)):
        fail(case, "inflate", dims=dims, r=r)
    world = synth.World(np.ascontiguousarray(inflated), origin, res, np.zeros((0, 6)))
    v.set_grid(to_dev(world.voxels, v.device), world.origin, world.res)
    if not np.array_equal(v.pack_grid(to_dev(world.voxels, v.device)).cpu().numpy(), sharding.pack_grid_reference(world.voxels)):
        fail(case, "pack", dims=dims)
    g, keep = ol.make_grid(world)
    ext = np.array(dims) * res
    # point queries incl. points outside the map and on voxel faces
    pts = origin + rng.uniform(-0.2, 1.2, size=(4000, 3)) * ext
    pts[:200] = origin + rng.integers(-1, np.array(dims) + 1, size=(200, 3)) * res
    for which in (0, 1):
        if not np.array_equal(v.query_points(to_dev(pts, v.device), which).cpu().numpy(), synth.lookup(world, pts, which)):
            fail(case, "query", which=which, dims=dims)
    if not np.array_equal(v.guides_unknown(to_dev(np.concatenate([pts, pts], 1), v.device)).cpu().numpy(), synth.lookup(world, pts, 1)):
        fail(case, "guides_unknown", dims=dims)
    # gates
    N = int(rng.choice([4, 5, 7, 12, 20, 32, 33, 64, 100]))
    B = int(rng.integers(1, 40))
    dt = float(rng.choice([0.05, 0.025, 0.01, 0.11]))
    start = origin + rng.uniform(0.1, 0.9, size=(B, 3)) * ext
    stepv = rng.normal(size=(B, 3)) * [1, 1, 0.1]
    stepv *= rng.uniform(0.05, 0.3, size=(B, 1)) / np.linalg.norm(stepv, axis=1, keepdims=True)
    ctrl = start[:, None, :] + np.arange(N)[None, :, None] * stepv[:, None, :] + rng.normal(scale=0.03, size=(B, N, 3))
    n_obs = rng.integers(0, 4, size=B)
    ooff = np.concatenate([[0], np.cumsum(n_obs)]).astype(np.int32)
    obs = np.zeros((int(ooff[-1]), 9))
    for i in range(B):
        for j in range(ooff[i], ooff[i + 1]):
            obs[j, 0:3] = ctrl[i, rng.integers(0, N)] + rng.normal(scale=0.6, size=3)
            obs[j, 3:6] = rng.uniform(-1, 1, size=3) * [1, 1, 0]
            obs[j, 6:9] = rng.uniform(0.2, 1.0, size=3)
    dctrl = to_dev(ctrl, v.device)
    flag, first = v.traj_collision(dctrl, dt)
    pt, line = v.ctrl_occupancy(dctrl)
    dyn = v.traj_dynamic_collision(dctrl, dt, to_dev(ooff, v.device), to_dev(obs, v.device) if len(obs) else None) if len(obs) else None
    flag, first, pt, line = (t.cpu().numpy() for t in (flag, first, pt, line))
    dyn = None if dyn is None else dyn.cpu().numpy()
    for i in range(B):
        c = np.ascontiguousarray(ctrl[i])
        fi = C.c_int()
        f = O.vgo_traj_collision(C.byref(g), N, ol._d(c), P.ts_ctrl, dt, C.byref(fi))
        if f != flag[i] or fi.value != first[i]:
            fail(case, "traj_collision", N=N, dt=dt, i=i, ref=[f, fi.value], got=[int(flag[i]), int(first[i])])
        p_ref, l_ref = np.zeros(N, dtype=np.uint8), np.zeros(N, dtype=np.uint8)
        O.vgo_ctrl_occupancy(C.byref(g), N, ol._d(c), ol._u(p_ref), ol._u(l_ref))
        if not (np.array_equal(p_ref, pt[i]) and np.array_equal(l_ref, line[i])):
            fail(case, "ctrl_occupancy", N=N, i=i)
        if dyn is not None:
            o = np.ascontiguousarray(obs[ooff[i]:ooff[i + 1]])
            if O.vgo_traj_dynamic_collision(N, ol._d(c), P.ts_ctrl, dt, len(o), ol._d(o)) != dyn[i]:
                fail(case, "traj_dynamic_collision", N=N, dt=dt, i=i)
    # spline evaluation at clamps, knots and random times
    dur = (N - 3) * P.ts_ctrl
    times = np.concatenate([rng.uniform(-0.1, dur + 0.1, size=40), np.arange(0, N - 2) * P.ts_ctrl, [0.0, dur]])
    Be = min(B, 6)
    for deriv in (0, 1, 2):
        out = v.bspline_eval(to_dev(ctrl[:Be], v.device), to_dev(times, v.device), deriv).cpu().numpy()
        refv = np.zeros_like(out)
        for i in range(Be):
            c = np.ascontiguousarray(ctrl[i])
            for k, t in enumerate(times):
                O.vgo_traj_eval(N, ol._d(c), P.ts_ctrl, deriv, float(t), ol._d(refv[i, k]))
        if not np.array_equal(out, refv):
            fail(case, "bspline_eval", N=N, deriv=deriv)
    # least-squares fit
    K = int(rng.choice([4, 5, 9, 17, 30, 31, 47, 62, 63, 80]))
    Bf = int(rng.integers(1, 70))
    ptsf = np.cumsum(rng.normal(scale=0.2, size=(Bf, K, 3)), axis=1)
    conds = rng.normal(scale=0.5, size=(Bf, 4, 3)) if rng.random() < 0.7 else None
    got = v.bspline_fit(to_dev(ptsf, v.device), None if conds is None else to_dev(conds, v.device), ts=P.ts_ctrl).cpu().numpy()
    want = ol.bspline_fit_batch(ptsf, P.ts_ctrl, conds)
    err = float(np.max(np.abs(got - want)) / max(1.0, float(np.max(np.abs(want)))))
    if not err < 1e-9:
        fail(case, "bspline_fit", K=K, B=Bf, err=err)
    # box sweep of polyTrajOctomap::checkCollision at single points (float positions, PO.cpp:547-589)
    box = rng.choice([0.2, 0.4, 0.6, 0.25], size=3)
    mres = float(rng.choice([0.1, 0.2, res]))
    bp = origin + rng.uniform(-0.1, 1.1, size=(600, 3)) * ext
    from trajectory_planner_amd.vigo import VigoError
    try:
        gotb = v.box_collision_points(to_dev(bp, v.device), box, mres).cpu().numpy()
    except VigoError:      # refused: the origin is not on the octomap key lattice (documented in include/vigo.h)
        gotb = None
        skipped_box += 1
    bx = np.ascontiguousarray(box, dtype=np.float64)
    for i in range(len(bp) if gotb is not None else 0):
        if O.vgo_box_collision(C.byref(g), float(np.float32(bp[i, 0])), float(np.float32(bp[i, 1])), float(np.float32(bp[i, 2])), ol._d(bx), mres) != gotb[i]:
            fail(case, "box_collision_points", i=i, box=box.tolist(), map_res=mres, dims=dims)
            break
    # trilinear ESDF value + gradient on a random float lattice
    ed = tuple(int(x) for x in rng.integers(2, 40, size=3))
    dist = rng.normal(size=ed).astype(np.float32)
    eo = rng.uniform(-2, 2, size=3)
    v.set_esdf(to_dev(dist, v.device), eo, res)
    ep = eo + rng.uniform(-0.2, 1.2, size=(500, 3)) * np.array(ed) * res
    dq, gq = (t.cpu().numpy() for t in v.esdf_query(to_dev(ep, v.device)))
    for i in range(len(ep)):
        dd, gg = C.c_double(), np.zeros(3)
        O.vgo_esdf_query(ed[0], ed[1], ed[2], ol._d(eo), res, dist.ctypes.data_as(C.POINTER(C.c_float)), ol._d(ep[i]), C.byref(dd), ol._d(gg))
        if not (dq[i] == dd.value and np.array_equal(gq[i], gg)):
            fail(case, "esdf_query", i=i, dims=ed)
            break
    # the fp32-I/O entry on the same lattice, bit for bit against its oracle twin (non-finite points included)
    ep32 = ep.astype(np.float32)
    ep32[:3] = [(np.nan, 0, 0), (np.inf, -np.inf, 1e30), (-3e38, 1e-30, 0)]
    got32 = v.esdf_query_f32(to_dev(ep32, v.device)).cpu().numpy()
    ref32 = ol.esdf_query_f32_batch(dist, eo, res, ep32)
    if not np.array_equal(got32.view(np.uint32), ref32.view(np.uint32)):
        fail(case, "esdf_query_f32", dims=ed)
    v.close()
    if (case + 1) % 10 == 0:
        print(f"{case + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(json.dumps({"cases": cases, "mismatches": bad, "box_sweeps_refused_off_lattice": skipped_box, "seconds": time.time() - t0}))
sys.exit(1 if bad else 0)
