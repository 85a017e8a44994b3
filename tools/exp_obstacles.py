"""dev experiment: config 5a (dynamic-obstacle term) timing; VIGO_EXP_LIB selects an alternative library build"""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import trajectory_planner_amd._lib as L
if os.environ.get("VIGO_EXP_LIB"):
    L.LIB_PATH = os.path.join(R, os.environ["VIGO_EXP_LIB"])
import numpy as np, torch
from trajectory_planner_amd import synth
from trajectory_planner_amd.vigo import PREC_F64, Vigo, default_params
dev = torch.device("cuda", 0)
T = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
w256 = synth.make_box_world(synth.SEED_BASE + 2, n=256, n_boxes=200)
for n_obs in (0, 1, 8, 24):
    b = synth.make_bspline_batch(w256, 1024, 32, 4242 + 32 + 1024, start_range=8.0, n_obs=n_obs)
    P = default_params(); P.max_iterations = 50
    v = Vigo(0, P, PREC_F64)
    v.set_grid(T(w256.voxels), w256.origin, w256.res)
    ctrl, goff, gpv, ooff, obs = T(b.ctrl), T(b.guide_off), T(b.guide_pv), T(b.obs_off), T(b.obs)
    gunk = v.guides_unknown(gpv)
    f = lambda: v.optimize(ctrl, goff, gpv, gunk, ooff, obs)
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    r = f()
    print(json.dumps({"lib": os.environ.get("VIGO_EXP_LIB", "default"), "n_obs": n_obs, "ms": dt * 1e3, "fx_sum": float(r.fx.sum())}), flush=True)
    v.close()
