"""dev: short trajectories (small N) at batch sizes that fill every SIMD: does a second wave per SIMD pay in fp64?"""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import trajectory_planner_amd._lib as L
if os.environ.get("VIGO_EXP_LIB"):
    L.LIB_PATH = os.path.join(R, os.environ["VIGO_EXP_LIB"])
import numpy as np, torch
from trajectory_planner_amd import synth
from trajectory_planner_amd.vigo import Vigo, default_params
dev = torch.device("cuda", 0)
T = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
w = synth.make_box_world(synth.SEED_BASE + 2, n=256, n_boxes=200)
for (B, N, prec) in ((16384, 16, 0), (16384, 20, 0), (16384, 16, 2), (16384, 16, 1), (16384, 24, 0)):
    b = synth.make_bspline_batch(w, B, N, 77 + N, start_range=8.0)
    P = default_params(); P.max_iterations = 50
    v = Vigo(0, P, prec)
    v.set_grid(T(w.voxels), w.origin, w.res)
    ctrl, goff, gpv = T(b.ctrl), T(b.guide_off), T(b.guide_pv)
    gunk = v.guides_unknown(gpv)
    f = lambda: v.optimize(ctrl, goff, gpv, gunk)
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    r = f()
    print(json.dumps({"lib": os.environ.get("VIGO_EXP_LIB", "default"), "B": B, "N": N, "prec": prec, "ms": round(dt * 1e3, 3),
                      "Mtraj_s": round(B / dt / 1e6, 3), "chk": float(r.ctrl.double().sum())}), flush=True)
    v.close()
