"""dev: timing of vigo_minsnap (586 paths x 8 waypoints, with and without a 0.5 m corridor; and one path alone)"""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
from trajectory_planner_amd.vigo import Vigo
dev = torch.device("cuda", 0)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
rng = np.random.default_rng(5)
for Tn, Wn in ((586, 8), (1, 8), (4096, 8), (512, 11), (2048, 4)):
    wp = np.zeros((Tn, Wn, 3))
    wp[:, 0] = rng.uniform(-8, 8, size=(Tn, 3)) * [1, 1, 0.1] + [0, 0, 1.5]
    for i in range(1, Wn):
        step = rng.normal(size=(Tn, 3)) * [1, 1, 0.1]
        step *= (rng.uniform(1.0, 3.0, size=(Tn, 1)) / np.linalg.norm(step, axis=1, keepdims=True))
        wp[:, i] = wp[:, i - 1] + step
    d_wp, d_cor = T(wp), T(np.full((Tn, Wn - 1), 0.5))
    v = Vigo(0)
    res = {}
    for name, f in (("free", lambda: v.minsnap(d_wp)), ("corridor", lambda: v.minsnap(d_wp, d_cor))):
        for _ in range(2): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): f()
        torch.cuda.synchronize(); res[name] = (time.perf_counter() - t0) / 10 * 1e3
    st = v.minsnap(d_wp, d_cor)[2]
    print(json.dumps({"paths": Tn, "waypoints": Wn, "ms_free": round(res["free"], 4), "ms_corridor": round(res["corridor"], 4),
                      "solved": int((st == 0).sum()), "infeasible": int((st == -2).sum())}), flush=True)
    v.close()
