"""dev tool: where does k_optimize spend its time?  times the config-2 batch for several
mem_size / max_iterations settings (run on the GPU box)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
from trajectory_planner_amd import synth
from trajectory_planner_amd.vigo import Vigo, default_params

world = synth.make_box_world(synth.SEED_BASE + 2, n=256, n_boxes=200)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 32
b = synth.make_bspline_batch(world, B, N, synth.SEED_BASE + 1002)
v = Vigo(0)
dev = v.device
T = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
ctrl, goff, gpv, gunk = T(b.ctrl), T(b.guide_off), T(b.guide_pv), T(b.guide_unk)
for (m, it) in ((16, 50), (8, 50), (4, 50), (1, 50), (16, 25), (16, 1)):
    P = default_params(); P.mem_size = m; P.max_iterations = it; v.set_params(P)
    for _ in range(3): r = v.optimize(ctrl, goff, gpv, gunk)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): r = v.optimize(ctrl, goff, gpv, gunk)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"B={B} N={N} m={m:2d} iters={it:3d}: {dt*1e6:8.1f} us  evals/traj {r.evals.float().mean().item():.1f} iters {r.iters.float().mean().item():.1f}")
