"""dev: where k_optimize's time goes — config 2 with mem_size 1..16 (two-loop length) and iteration caps"""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
from trajectory_planner_amd import synth
from trajectory_planner_amd.vigo import Vigo, default_params
dev = torch.device("cuda", 0)
T = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
w256 = synth.make_box_world(synth.SEED_BASE + 2, n=256, n_boxes=200)
b = synth.make_bspline_batch(w256, 1024, 32, 4242 + 32 + 1024, start_range=8.0)
for (m, it) in ((16, 50), (8, 50), (4, 50), (2, 50), (1, 50), (16, 25), (16, 100), (16, 1)):
    P = default_params(); P.max_iterations = it; P.mem_size = m; P.g_epsilon = 0.0
    v = Vigo(0, P, 0)
    v.set_grid(T(w256.voxels), w256.origin, w256.res)
    ctrl, goff, gpv = T(b.ctrl), T(b.guide_off), T(b.guide_pv)
    gunk = v.guides_unknown(gpv)
    f = lambda: v.optimize(ctrl, goff, gpv, gunk)
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    r = f()
    print(json.dumps({"mem_size": m, "max_it": it, "ms": round(dt * 1e3, 4), "mean_iters": float(r.iters.float().mean()), "mean_evals": float(r.evals.float().mean())}), flush=True)
    v.close()
