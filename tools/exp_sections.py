"""dev: shader-clock breakdown of one L-BFGS iteration of k_optimize (library built with -DVIGO_PROFILE_SECTIONS=1,
selected through VIGO_EXP_LIB); prints mean cycles per iteration for: cost/gradient evaluation, line-search logic,
update (norms, s/y, ys/yy reduction), two-loop, tail (history shift, g.d reduction)."""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import trajectory_planner_amd._lib as L
L.LIB_PATH = os.path.join(R, os.environ.get("VIGO_EXP_LIB", "trajectory_planner_amd/lib/libvigo_prof.so"))
import numpy as np, torch
from trajectory_planner_amd import synth
from trajectory_planner_amd.vigo import Vigo, default_params
dev = torch.device("cuda", 0)
T = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
w256 = synth.make_box_world(synth.SEED_BASE + 2, n=256, n_boxes=200)
for (B, N, prec, n_obs) in ((1024, 32, 0, 0), (1024, 32, 2, 0), (8192, 64, 0, 0), (1024, 32, 0, 8)):
    b = synth.make_bspline_batch(w256, B, N, 4242 + N + B, start_range=8.0, n_obs=n_obs)
    P = default_params(); P.max_iterations = 50
    v = Vigo(0, P, prec)
    v.set_grid(T(w256.voxels), w256.origin, w256.res)
    ctrl, goff, gpv = T(b.ctrl), T(b.guide_off), T(b.guide_pv)
    gunk = v.guides_unknown(gpv)
    kw = dict(obs_off=T(b.obs_off), obs=T(b.obs)) if n_obs else {}
    r = v.optimize(ctrl, goff, gpv, gunk, **kw)
    raw = r.x.reshape(B, -1)[:, :10].cpu().numpy(); x = raw[:, :9]; probe = raw[:, 9]
    x = np.concatenate([x[:, :5], x[:, 7:9], x[:, 5:7]], axis=1)   # eval, ls, upd, two, tail, pre, trial, k, evals
    it = x[:, 7].mean()
    names = ["eval", "ls_tests", "update", "two_loop", "tail", "ls_setup_xupdate", "trial_interval"]
    tot = x[:, :7].sum(1).mean()
    print(json.dumps({"B": B, "N": N, "prec": prec, "n_obs": n_obs, "iters": round(float(it), 1), "evals": round(float(x[:, 8].mean()), 1),
                      "cycles_per_iter": {n: int(x[:, i].mean() / it) for i, n in enumerate(names)},
                      "share": {n: round(float(x[:, i].mean() / tot), 3) for i, n in enumerate(names)}, "total_cycles": int(tot), "probe_cycles": int(probe.mean() / it)}), flush=True)
    v.close()
