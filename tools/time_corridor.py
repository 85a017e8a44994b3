"""Times vigo_corridor_check at the BASELINE configs[2] size (4096 segments x 10 000 samples) on the synthetic
256 x 256 x 64 world of tools/measure_configs.py; prints one JSON line.  Run on the GPU box."""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import trajectory_planner_amd._lib as L
if os.environ.get("VIGO_EXP_LIB"):                      # dev: an alternative build of the library
    L.LIB_PATH = os.path.join(R, os.environ["VIGO_EXP_LIB"])
import numpy as np, torch
from trajectory_planner_amd import synth
from trajectory_planner_amd.vigo import Vigo
dev = torch.device("cuda", 0)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
rng = np.random.default_rng(3)
vox = np.zeros((256, 256, 64), dtype=np.uint8)
for _ in range(300):
    c = rng.integers(8, 248, size=2); s = rng.integers(1, 6, size=2)
    vox[c[0] - s[0]:c[0] + s[0], c[1] - s[1]:c[1] + s[1], 0:rng.integers(10, 64)] |= 4
unk = rng.random((32, 32, 8)) < 0.05
vox[np.repeat(np.repeat(np.repeat(unk, 8, 0), 8, 1), 8, 2)] |= 2
v = Vigo(0)
v.set_grid(T(vox), np.array([-12.8, -12.8, -1.0]), 0.1)
coeffs, n_samp, delT, dur = synth.make_corridor_segments(33, 4096, extent_lo=(-10, -10, 0.5), extent_hi=(10, 10, 2.5), n_samples=10000)
c, ns, dl = T(coeffs), T(n_samp), T(delT)
box = [0.4, 0.4, 0.2]
for _ in range(3): v.corridor_check(c, ns, dl, box, 0.2)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): flag, first, count = v.corridor_check(c, ns, dl, box, 0.2)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print(json.dumps({"lib": os.environ.get("VIGO_EXP_LIB", "default"), "config": "3: 4096 segments x 10k samples", "ms": dt * 1e3, "samples_per_s": 4096e4 / dt, "colliding_segments": int(flag.sum())}))
