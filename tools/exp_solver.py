"""dev: A/B timing of the solve kernel; VIGO_EXP_LIB selects an alternative library build"""
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
w256 = synth.make_box_world(synth.SEED_BASE + 2, n=256, n_boxes=200)
CASES = ((1024, 32, 0), (1024, 32, 2), (16384, 32, 0), (8192, 64, 0)) if not os.environ.get('VIGO_EXP_MATRIX') else tuple((B, N, p) for (B, N) in ((16384, 16), (1024, 32), (16384, 32), (8192, 64), (4096, 128), (2048, 200)) for p in (0, 2, 1))
for (B, N, prec) in CASES:
    b = synth.make_bspline_batch(w256, B, N, 4242 + N + B, start_range=8.0)
    P = default_params(); P.max_iterations = 50
    P.strict_z = 1 if os.environ.get('VIGO_EXP_STRICT') else 0   # 1: the general kernel (no level rule)
    v = Vigo(0, P, prec)
    v.set_grid(T(w256.voxels), w256.origin, w256.res)
    ctrl, goff, gpv = T(b.ctrl), T(b.guide_off), T(b.guide_pv)
    gunk = v.guides_unknown(gpv)
    f = lambda: v.optimize(ctrl, goff, gpv, gunk)
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    r = f()
    print(json.dumps({"lib": os.environ.get("VIGO_EXP_LIB", "default"), "B": B, "N": N, "prec": prec, "ms": round(dt * 1e3, 4), "Mtraj_s": round(B / dt / 1e6, 3),
                      "chk": float(r.ctrl.double().sum())}), flush=True)
    v.close()
