"""dev: timing of vigo_bspline_fit at the config-2 / config-4 shapes (VIGO_EXP_LIB selects an alternative library build)"""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import trajectory_planner_amd._lib as L
if os.environ.get("VIGO_EXP_LIB"):
    L.LIB_PATH = os.path.join(R, os.environ["VIGO_EXP_LIB"])
import numpy as np, torch
from trajectory_planner_amd.vigo import Vigo
dev = torch.device("cuda", 0)
v = Vigo(0)
rng = np.random.default_rng(11)
for (B, K) in ((1024, 30), (65536, 30), (8192, 62), (65536, 62), (1 << 20, 30)):
    pts = torch.from_numpy(rng.normal(size=(B, K, 3))).to(dev)
    v.bspline_fit(pts)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): v.bspline_fit(pts)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print(json.dumps({"lib": os.environ.get("VIGO_EXP_LIB", "default"), "B": B, "K": K, "ms": round(dt * 1e3, 4), "Gpaths_s": round(B / dt / 1e9, 3), "algorithmic_GBps": round(B * (2 * K + 6) * 24 / dt / 1e9, 1)}), flush=True)
