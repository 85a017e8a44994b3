"""Measures the BASELINE.json configs 2-5 on one MI355X and prints one JSON line per config
(run on the GPU box; the numbers feed profiles/README.md).  Parity for every kernel timed here is
covered by tests/ -m gpu; this script only times."""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
import trajectory_planner_amd._lib as _L
if os.environ.get("VIGO_EXP_LIB"):   # dev: A/B another build of the library
    _L.LIB_PATH = os.path.join(R, os.environ["VIGO_EXP_LIB"])
from trajectory_planner_amd import synth
from trajectory_planner_amd.vigo import PREC_F32, PREC_F64, Vigo, default_params

dev = torch.device("cuda", 0)
T = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def timeit(fn, reps=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def solve_cfg(name, world, B, N, prec, n_obs=0, reps=20):
    b = synth.make_bspline_batch(world, B, N, 4242 + N + B, start_range=8.0 if N == 32 else 16.0, n_obs=n_obs)
    P = default_params(); P.max_iterations = 50
    v = Vigo(0, P, prec)
    v.set_grid(T(world.voxels), world.origin, world.res)
    ctrl, goff, gpv = T(b.ctrl), T(b.guide_off), T(b.guide_pv)
    ooff, obs = T(b.obs_off), T(b.obs)
    def step():
        gunk = v.guides_unknown(gpv)
        return v.optimize(ctrl, goff, gpv, gunk, ooff, obs)
    dt = timeit(step, reps)
    r = step()
    print(json.dumps({"config": name, "B": B, "N": N, "precision": {PREC_F32: "f32", PREC_F64: "f64", 2: "f64_fast"}[prec], "obstacles_per_traj": n_obs,
                      "ms_per_batch": dt * 1e3, "trajs_per_s": B / dt, "mean_evals": float(r.evals.float().mean()),
                      "mean_iters": float(r.iters.float().mean())}), flush=True)
    v.close()


# config 1: polyTrajOctomap::makePlan on the maze fixture (host QP + device box sweep), via the facade library
import ctypes as C
sys.path.insert(0, os.path.join(R, "tests"))
try:
    import test_gpu_config1 as c1
    vox, origin, res, wp, _ = c1.load_maze()
    cfg = [0.4, 0.4, 0.2, 0.2, 0.1, 1.0, 0.5, 0.8, 8.0, 100, 0.1, 0.0]
    c1.plan(vox, origin, res, wp, cfg)
    ts, cold = [], []
    for k in range(10):
        # a path the previous call did not see (one waypoint moved by millimetres): the QP factorisation kept per
        # host thread does not apply, this is the time of a first makePlan() on a new path
        wp2 = np.array(wp, dtype=np.float64, copy=True)
        wp2[3, 0] += 1e-3 * (k + 1)
        traj, info = c1.plan(vox, origin, res, wp2, cfg)
        cold.append(info[4])
        # the fixture path again: same matrices as two calls ago, not as the last one -> also a fresh factorisation
        traj, info = c1.plan(vox, origin, res, wp, cfg)
        ts.append(info[4])
    rep = []
    for _ in range(10):   # the same path back to back: factorisation reused (what the corridor loop's later rounds see)
        traj, info = c1.plan(vox, origin, res, wp, cfg)
        rep.append(info[4])
    print(json.dumps({"config": "1: polyTrajOctomap makePlan, maze.bt, 8 waypoints (host QP + device sweep)", "valid": bool(info[0]),
                      "corridor_iterations": int(info[1]), "samples": int(info[2]), "makePlan_ms_median": float(np.median(ts)) * 1e3,
                      "makePlan_ms_min": float(np.min(ts)) * 1e3, "makePlan_ms_new_path_median": float(np.median(cold)) * 1e3,
                      "makePlan_ms_same_path_again_median": float(np.median(rep)) * 1e3}), flush=True)
except Exception as e:  # noqa
    print(json.dumps({"config": "1", "error": repr(e)}), flush=True)

# B-spline fit (updatePath batched): config 2 and config 4 shapes
v = Vigo(0)
rng = np.random.default_rng(11)
for (Bf, Kf) in ((1024, 30), (65536, 30), (8192, 62), (65536, 62)):
    pts = T(rng.normal(size=(Bf, Kf, 3)))
    v.bspline_fit(pts)
    dt = timeit(lambda: v.bspline_fit(pts), 50)
    print(json.dumps({"config": f"fit: {Bf} paths x {Kf} waypoints", "ms": dt * 1e3, "paths_per_s": Bf / dt,
                      "algorithmic_GBps": Bf * (2 * Kf + 6) * 24 / dt / 1e9}), flush=True)
v.close()

w256 = synth.make_box_world(synth.SEED_BASE + 2, n=256, n_boxes=200)

# map kernels and rebound-loop gates at the config-2 shape
v = Vigo(0)
vox256 = T(w256.voxels)
dt = timeit(lambda: v.pack_grid(vox256), 50)
print(json.dumps({"config": "map: vigo_pack_grid 256^3 byte grid -> 3 bit planes", "ms": dt * 1e3, "GBps_in": 256 ** 3 / dt / 1e9}), flush=True)
raw = T((w256.voxels & 6))
dt = timeit(lambda: v.inflate_grid(raw, 4, 4, 2), 30)
print(json.dumps({"config": "map: vigo_inflate_grid 256^3, robot half size 0.4 x 0.4 x 0.15 m (r = 4, 4, 2 voxels)", "ms": dt * 1e3,
                  "voxels_per_s": 256 ** 3 / dt, "algorithmic_GBps": 256 ** 3 * 3 / dt / 1e9}), flush=True)
del raw
v.set_grid(vox256, w256.origin, w256.res)
bg = synth.make_bspline_batch(w256, 16384, 32, 99, start_range=8.0, n_obs=2)
ctrl_g, ooff_g, obs_g = T(bg.ctrl), T(bg.obs_off), T(bg.obs)
dtg = w256.res / 2.0 / 2.0          # res / maxVel / 2 with maxVel = 2 (src/bspline_node.cpp:230)
for name, f, unit in (("gate: vigo_traj_collision (hasCollisionTrajectory), 16384 x 32, dt 0.025", lambda: v.traj_collision(ctrl_g, dtg), "trajs_per_s"),
                      ("gate: vigo_traj_dynamic_collision, 2 obstacles each", lambda: v.traj_dynamic_collision(ctrl_g, dtg, ooff_g, obs_g), "trajs_per_s"),
                      ("gate: vigo_ctrl_occupancy (findCollisionSeg queries)", lambda: v.ctrl_occupancy(ctrl_g), "trajs_per_s")):
    f()
    dt = timeit(f, 30)
    print(json.dumps({"config": name, "ms": dt * 1e3, unit: 16384 / dt}), flush=True)
bcg = synth.make_bspline_batch(w256, 65536, 32, 77, start_range=8.0)
dcg = dict(ctrl=T(bcg.ctrl), guide_off=T(bcg.guide_off), guide_pv=T(bcg.guide_pv), guide_unk=T(bcg.guide_unk))
v.cost_grad(**dcg)
dt = timeit(lambda: v.cost_grad(**dcg), 30)
cg_bytes = 65536 * ((6 * 32) * 8 + (3 * 26 + 5) * 8) + bcg.guide_pv.size * 8 + bcg.guide_off.size * 4
print(json.dumps({"config": "vigo_cost_grad alone: 65536 x 32 ctrl pts (one evaluation)", "ms": dt * 1e3, "evals_per_s": 65536 / dt,
                  "algorithmic_GBps": cg_bytes / dt / 1e9}), flush=True)
del bcg, dcg
tq = T(np.linspace(0.0, 5.8, 233))
dt = timeit(lambda: v.bspline_eval(ctrl_g, tq), 30)
print(json.dumps({"config": "bspline::at batched: 16384 x 32 ctrl pts x 233 times", "ms": dt * 1e3, "evals_per_s": 16384 * 233 / dt}), flush=True)
qp = T(np.random.default_rng(1).uniform(-12.7, 12.7, size=(1 << 22, 3)))
dt = timeit(lambda: v.query_points(qp, 0), 30)
print(json.dumps({"config": "map: vigo_query_points 4 M random points (isInflatedOccupied)", "ms": dt * 1e3, "queries_per_s": (1 << 22) / dt}), flush=True)
v.close()
del vox256, bg, ctrl_g, qp

solve_cfg("2: 1024x32, 256^3, 50 it", w256, 1024, 32, PREC_F64)
solve_cfg("2 (f64_fast mode)", w256, 1024, 32, 2)
solve_cfg("2 (fp32 mode)", w256, 1024, 32, PREC_F32)
solve_cfg("2 at B=16384", w256, 16384, 32, PREC_F64, reps=5)
solve_cfg("5a: dynamic-obstacle term, 8 obstacles/traj", w256, 1024, 32, PREC_F64, n_obs=8)

# config 5b: 1 M trilinear ESDF queries per iteration
n = 256
dist, origin = synth.sphere_esdf(n, 0.1, (0.0, 0.0, 0.0), 5.0)
v = Vigo(0)
v.set_esdf(T(dist), origin, 0.1)
rng = np.random.default_rng(5)
pts = T(rng.uniform(-12.7, 12.7, size=(1 << 20, 3)))
dt = timeit(lambda: v.esdf_query(pts), 50)
print(json.dumps({"config": "5b: 1M trilinear ESDF queries (uniform random)", "ms": dt * 1e3, "queries_per_s": (1 << 20) / dt,
                  "algorithmic_GBps": (1 << 20) * 60 / dt / 1e9}), flush=True)
idx = np.lexsort(tuple(np.floor((pts.cpu().numpy()[:, a] + 12.8) / 0.8).astype(int) for a in (2, 1, 0)))
pts_sorted = T(pts.cpu().numpy()[idx])
dt = timeit(lambda: v.esdf_query(pts_sorted), 50)
print(json.dumps({"config": "5b: same queries, brick-sorted", "ms": dt * 1e3, "queries_per_s": (1 << 20) / dt,
                  "algorithmic_GBps": (1 << 20) * 60 / dt / 1e9}), flush=True)
# the same at the fp32 I/O width SURVEY.md §8(d) config 5 states (12 B in, 16 B out): vigo_esdf_query_f32
out32 = torch.empty(1 << 20, 4, dtype=torch.float32, device=dev)
for name, p64 in (("uniform random", pts), ("brick-sorted", pts_sorted)):
    p32 = p64.float().contiguous()
    dt = timeit(lambda: v.esdf_query_f32(p32, out32), 50)
    print(json.dumps({"config": f"5b (fp32 I/O): 1M trilinear ESDF queries, {name}", "ms": dt * 1e3, "queries_per_s": (1 << 20) / dt,
                      "algorithmic_GBps": (1 << 20) * 60 / dt / 1e9, "frac_of_hbm_peak": (1 << 20) * 60 / dt / 8e12}), flush=True)
v.close()

# config 3: 4096 segments x 10 000 samples corridor check on an occupied/free/unknown map
rng = np.random.default_rng(3)
vox = np.zeros((256, 256, 64), dtype=np.uint8)
for _ in range(300):
    c = rng.integers(8, 248, size=2); s = rng.integers(1, 6, size=2)
    vox[c[0] - s[0]:c[0] + s[0], c[1] - s[1]:c[1] + s[1], 0:rng.integers(10, 64)] |= 4
unk = rng.random((32, 32, 8)) < 0.05
vox[np.repeat(np.repeat(np.repeat(unk, 8, 0), 8, 1), 8, 2)] |= 2
world3 = synth.World(vox, np.array([-12.8, -12.8, -1.0]), 0.1, np.zeros((0, 6)))
v = Vigo(0)
v.set_grid(T(world3.voxels), world3.origin, world3.res)
coeffs, n_samp, delT, dur = synth.make_corridor_segments(33, 4096, extent_lo=(-10, -10, 0.5), extent_hi=(10, 10, 2.5), n_samples=10000)
c, ns, dl = T(coeffs), T(n_samp), T(delT)
box = [0.4, 0.4, 0.2]
dt = timeit(lambda: v.corridor_check(c, ns, dl, box, 0.2), 10, 2)
flag, first, count = v.corridor_check(c, ns, dl, box, 0.2)
samples = 4096 * 10000
print(json.dumps({"config": "3: 4096 segments x 10k samples, box [0.4,0.4,0.2] step 0.2", "ms": dt * 1e3, "segments_per_s": 4096 / dt,
                  "samples_per_s": samples / dt, "lattice_points_decided_per_s": samples * 18 / dt, "colliding_segments": int(flag.sum()),
                  "note": "samples and lattice points DECIDED per second: since round 3 most are decided by a span's certificate, not looked up one by one"}), flush=True)
v.close()

# config 3 with real coefficients: batched min-snap QP (586 paths x 7 segments = 4102 segments), corridor 0.5, then the checker
rng = np.random.default_rng(5)
Tn, Wn = 586, 8
wp = np.zeros((Tn, Wn, 3))
wp[:, 0] = rng.uniform(-8, 8, size=(Tn, 3)) * [1, 1, 0.1] + [0, 0, 1.5]
for i in range(1, Wn):
    step = rng.normal(size=(Tn, 3)) * [1, 1, 0.1]
    step *= (rng.uniform(1.0, 3.0, size=(Tn, 1)) / np.linalg.norm(step, axis=1, keepdims=True))
    wp[:, i] = wp[:, i - 1] + step
d_wp, d_cor = T(wp), T(np.full((Tn, Wn - 1), 0.5))
v = Vigo(0)
v.set_grid(T(world3.voxels), world3.origin, world3.res)
dt_free = timeit(lambda: v.minsnap(d_wp), 20, 2)
dt_cor = timeit(lambda: v.minsnap(d_wp, d_cor), 20, 2)
co, kn, st = v.minsnap(d_wp, d_cor)
dur = (kn[:, 1:] - kn[:, :-1]).reshape(-1)
ns3 = torch.full((Tn * (Wn - 1),), 10000, dtype=torch.int32, device=dev)
dl3 = (dur / 10000.0).contiguous()
seg = co.reshape(Tn * (Wn - 1), 3, 8)
dt_chk = timeit(lambda: v.corridor_check(seg, ns3, dl3, box, 0.2), 10, 2)
print(json.dumps({"config": "3 (real coefficients): vigo_minsnap 586 paths x 8 waypoints -> 4102 segments x 10k samples",
                  "minsnap_ms_no_corridor": dt_free * 1e3, "minsnap_ms_corridor_0.5": dt_cor * 1e3, "paths_per_s_corridor": Tn / dt_cor,
                  "solved": int((st == 0).sum()), "infeasible": int((st == -2).sum()), "checker_ms": dt_chk * 1e3,
                  "samples_per_s": Tn * (Wn - 1) * 10000 / dt_chk}), flush=True)
v.close()

# config 4 shard: 8192 x 64 control points, 512^3 grid
w512 = synth.make_box_world(synth.SEED_BASE + 4, n=512, n_boxes=800, centre_range=24.0)
solve_cfg("4 shard: 8192x64, 512^3, 50 it (one of 8 GPUs)", w512, 8192, 64, PREC_F64, reps=5)
solve_cfg("4 shard (f64_fast mode)", w512, 8192, 64, 2, reps=5)
