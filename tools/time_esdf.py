"""Times vigo_esdf_query at the BASELINE configs[4] size (1 048 576 queries, 256^3 lattice): uniform random and
brick-sorted query order; prints one JSON line per case.  Run on the GPU box.  `--f32`: vigo_esdf_query_f32 (float3 in,
float4 out — the I/O width SURVEY.md §8(d) config 5 states) instead of the fp64 entry; `--both`: one after the other."""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
from trajectory_planner_amd import synth
from trajectory_planner_amd.vigo import Vigo
dev = torch.device("cuda", 0)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
n = 256
dist, origin = synth.sphere_esdf(n, 0.1, (0.0, 0.0, 0.0), 5.0)
v = Vigo(0)
dist_d = T(dist)
v.set_esdf(dist_d, origin, 0.1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): v.set_esdf(dist_d, origin, 0.1)
torch.cuda.synchronize()
print(json.dumps({"config": "5b: vigo_set_esdf, 256^3 (row-major lattice -> one 128-B line per 1x3x3 cells, 3.56x the bytes)", "ms": (time.perf_counter() - t0) / 20 * 1e3}))
rng = np.random.default_rng(5)
pts_h = rng.uniform(-12.7, 12.7, size=(1 << 20, 3))
idx = np.lexsort(tuple(np.floor((pts_h[:, a] + 12.8) / 0.4).astype(int) for a in (2, 1, 0)))
modes = ["f32"] if "--f32" in sys.argv else (["f64", "f32"] if "--both" in sys.argv else ["f64"])
for mode in modes:
    for name, p in (("uniform random", pts_h), ("brick-sorted", pts_h[idx])):
        if mode == "f32":
            pts = T(p.astype(np.float32))
            out = torch.empty(pts.shape[0], 4, dtype=torch.float32, device=dev)
            run = lambda: v.esdf_query_f32(pts, out)
        else:
            pts = T(p)
            run = lambda: v.esdf_query(pts)
        for _ in range(5): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): r = run()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        chk = float(r[:, 0].double().sum().item()) if mode == "f32" else float(r[0].sum().item())
        print(json.dumps({"config": f"5b: 1M trilinear ESDF queries, {name}, {mode} I/O", "ms": dt * 1e3, "queries_per_s": (1 << 20) / dt,
                          "algorithmic_GBps": (1 << 20) * 60 / dt / 1e9, "frac_of_hbm_peak": (1 << 20) * 60 / dt / 8e12, "checksum": chk}))
