"""Randomised parity sweep of vigo_optimize / vigo_cost_grad against the emulation-mode oracle (bit for bit) over
control-point counts, batch sizes, history lengths, iteration caps, obstacle counts and both fp64 arithmetic modes.
Not part of the test suite (minutes of oracle time); run on the GPU box:  python tools/fuzz_solver.py [cases] [seed]"""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import oracle_lib as ol
from gpu_util import batch_to_dev
from trajectory_planner_amd import synth
from trajectory_planner_amd.vigo import Vigo, default_params

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
world = synth.make_box_world(synth.SEED_BASE + 2, n=128, n_boxes=60, centre_range=5.5, z_range=2.0)
bad = 0
t0 = time.time()
for case in range(cases):
    N = int(rng.choice([7, 8, 9, 12, 16, 20, 31, 32, 33, 40, 63, 64, 65, 90, 128, 129, 160, 200]))
    B = int(rng.integers(1, 40 if N <= 64 else 8))
    P = default_params()
    P.mem_size = int(rng.choice([1, 2, 3, 5, 8, 15, 16]))
    P.max_iterations = int(rng.choice([1, 2, 5, 17, 18, 30, 50, 80]))
    P.g_epsilon = float(rng.choice([0.0, 0.01, 0.3]))
    P.max_linesearch = int(rng.choice([2, 5, 40]))
    P.plan_in_z = int(rng.integers(0, 2))
    P.uncertain_factor = float(rng.choice([1.0, 1.5]))
    n_obs = int(rng.choice([0, 0, 1, 3, 17]))
    P.pred_horizon = float(rng.choice([2.0, 2.0, 0.5, 3.0, 12.0]))   # 11 / 3 / 16 / 61 predicted steps per obstacle
    fast = bool(rng.integers(0, 2))
    if rng.random() < 0.04:
        # a batch with more waves than the chip has SIMDs: the register-capped instantiations (two waves per SIMD; the
        # level kernels with 4 / 5 history pairs in registers)
        N = int(rng.choice([16, 21, 22, 27, 32, 33, 48, 64]))
        B = int(rng.integers(2100, 2700)) if N <= 32 else int(rng.integers(1050, 1400))
        n_obs = int(rng.choice([0, 0, 0, 1]))
        P.max_iterations = int(rng.choice([5, 18, 30]))
    # level trajectories, trajectories with vertical jitter, or both in one batch (the level rule and its D = 2 kernel);
    # some of the level ones level to the bit, some with a vertical offset of a few ulps (inside / around the 2^-40 band)
    zj = float(rng.choice([0.0, 0.0, 0.03]))
    b = synth.make_bspline_batch(world, B, N, int(rng.integers(1 << 30)), start_range=3.0, n_obs=n_obs, z_jitter=zj,
                                 z_share=float(rng.choice([0.3, 0.5, 1.0])))
    for i in range(B):
        kind = int(rng.integers(0, 4))
        if np.ptp(b.ctrl[i, :, 2]) < 1e-6:
            if kind == 0:
                b.ctrl[i, :, 2] = float(rng.choice([1.0, 0.0, -2.5, 1e-300, 1234.5]))
            elif kind == 1:
                b.ctrl[i, int(rng.integers(0, N)), 2] += float(rng.choice([1e-13, 5e-13, 9.0e-13, 9.2e-13, 2e-12, 1e-9]))
    w = np.ones((B, 4)) * rng.choice([1.0, 2.0, 4.0], size=(B, 4))
    v = Vigo(0, P, 2 if fast else 0)
    v.set_grid(torch.from_numpy(world.voxels).cuda(), world.origin, world.res)
    d = batch_to_dev(b, v.device, w)
    r = v.optimize(**d)
    c, g, t = v.cost_grad(**d)
    G, PPL = ol.emulation_shape(N)
    ol.set_emulation(G, PPL)
    ol.oracle().vgo_set_emulation_fast(1 if fast else 0)
    try:
        e = ol.optimize_batch(P, b, w)
        ce, ge, te = ol.cost_grad_batch(P, b, w)
    finally:
        ol.oracle().vgo_set_emulation_fast(0)
        ol.set_emulation(0)
    ok = all(np.array_equal(getattr(r, k).cpu().numpy(), e[k], equal_nan=True) for k in ("status", "iters", "evals", "x", "ctrl", "fx"))
    ok = ok and np.array_equal(c.cpu().numpy(), ce, equal_nan=True) and np.array_equal(g.cpu().numpy(), ge, equal_nan=True)
    if not ok:
        bad += 1
        if bad <= int(os.environ.get("VIGO_FUZZ_DETAIL", "0")):
            for k in ("status", "iters", "evals", "fx", "x", "ctrl"):
                a, bb = getattr(r, k).cpu().numpy(), e[k]
                rows = np.nonzero(~np.isclose(a, bb, rtol=0, atol=0, equal_nan=True).reshape(B, -1).all(1))[0]
                print("   ", k, "differs in trajectories", rows.tolist()[:12], flush=True)
            zs = b.ctrl[:, :, 2]
            print("    z spread per trajectory", np.ptp(zs, axis=1).tolist(), "z0", zs[:, 0].tolist(), flush=True)
            print("    cost equal", np.array_equal(c.cpu().numpy(), ce, equal_nan=True), "grad equal", np.array_equal(g.cpu().numpy(), ge, equal_nan=True), flush=True)
            a, bb = r.ctrl.cpu().numpy(), e["ctrl"]
            i = int(np.nonzero(~(a == bb).reshape(B, -1).all(1))[0][0]) if not np.array_equal(a, bb, equal_nan=True) else 0
            print("    first differing trajectory", i, "status gpu/oracle", int(r.status[i]), int(e["status"][i]), "iters", int(r.iters[i]), int(e["iters"][i]),
                  "evals", int(r.evals[i]), int(e["evals"][i]), "max |diff| ctrl", float(np.nanmax(np.abs(a[i] - bb[i]))), "fx", float(r.fx[i]), float(e["fx"][i]), flush=True)
        print(json.dumps({"MISMATCH": case, "N": N, "B": B, "m": int(P.mem_size), "it": int(P.max_iterations), "obs": n_obs, "fast": fast,
                          "geps": float(P.g_epsilon), "ls": int(P.max_linesearch), "z": int(P.plan_in_z)}), flush=True)
    v.close()
    if case % 20 == 19:
        print(f"{case + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(json.dumps({"cases": cases, "mismatches": bad, "seconds": time.time() - t0}))
sys.exit(1 if bad else 0)
