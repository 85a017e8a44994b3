import sys, time
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np, torch
import oracle_lib as ol
from trajectory_planner_amd import synth
from trajectory_planner_amd.vigo import Vigo, default_params
world = synth.make_box_world(synth.SEED_BASE+2, n=256, n_boxes=200)
for N in (32, 20, 64, 40):
    b = synth.make_bspline_batch(world, 256, N, 123+N, n_obs=2 if N in (20,64) else 0)
    P = default_params(); P.max_iterations=50
    v = Vigo(0, P)
    dev = v.device
    T = lambda a, dt=None: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ctrl=T(b.ctrl); goff=T(b.guide_off); gpv=T(b.guide_pv); gunk=T(b.guide_unk); ooff=T(b.obs_off); obs=T(b.obs)
    cost, grad, terms = v.cost_grad(ctrl, goff, gpv, gunk, ooff, obs)
    torch.cuda.synchronize()
    grp = ol.emulation_group(N)
    ol.set_emulation(grp)
    c_e, g_e, t_e = ol.cost_grad_batch(P, b)
    ol.set_emulation(0)
    c_r, g_r, t_r = ol.cost_grad_batch(P, b)
    print(f"N={N} pairs={len(b.guide_pv)} cost_grad: emu exact cost={np.array_equal(cost.cpu().numpy(), c_e)} grad={np.array_equal(grad.cpu().numpy(), g_e)} terms={np.array_equal(terms.cpu().numpy(), t_e)}; vs ref max rel cost {np.max(np.abs(cost.cpu().numpy()-c_r)/np.abs(c_r)):.3g} grad {np.max(np.abs(grad.cpu().numpy()-g_r))/np.max(np.abs(g_r)):.3g}")
    if not np.array_equal(cost.cpu().numpy(), c_e):
        d = np.abs(cost.cpu().numpy()-c_e); i=np.argmax(d); print("  worst cost", i, cost[i].item(), c_e[i], terms[i].cpu().numpy(), t_e[i])
    if not np.array_equal(grad.cpu().numpy(), g_e):
        d = np.abs(grad.cpu().numpy()-g_e); print("  grad max abs diff", d.max(), "count", (d>0).sum())
    t0=time.time()
    r = v.optimize(ctrl, goff, gpv, gunk, ooff, obs)
    torch.cuda.synchronize(); t1=time.time()
    ol.set_emulation(grp); re = ol.optimize_batch(P, b); ol.set_emulation(0)
    rr = ol.optimize_batch(P, b)
    gc = r.ctrl.cpu().numpy()
    print(f"  optimize first call {t1-t0:.4f}s: emu exact ctrl={np.array_equal(gc, re['ctrl'])} x={np.array_equal(r.x.cpu().numpy(), re['x'])} status={np.array_equal(r.status.cpu().numpy(), re['status'])} iters={np.array_equal(r.iters.cpu().numpy(), re['iters'])} evals={np.array_equal(r.evals.cpu().numpy(), re['evals'])} fx={np.array_equal(r.fx.cpu().numpy(), re['fx'])}")
    rel = np.abs(gc-rr['ctrl']).reshape(b.B,-1).max(1)/np.abs(rr['ctrl']).reshape(b.B,-1).max(1)
    print(f"  vs ref-order oracle: median {np.median(rel):.3g} p99 {np.quantile(rel,.99):.3g} max {rel.max():.3g} frac<=1e-4 {(rel<=1e-4).mean():.4f}; status {np.unique(r.status.cpu().numpy(), return_counts=True)}")
    if not np.array_equal(gc, re['ctrl']):
        dd = np.abs(gc-re['ctrl']).reshape(b.B,-1).max(1); print("  #traj differing from emu:", (dd>0).sum(), "max", dd.max(), "iters gpu/emu", r.iters[:8].cpu().numpy(), re['iters'][:8], "evals", r.evals[:8].cpu().numpy(), re['evals'][:8])
    # timing
    for _ in range(3): v.optimize(ctrl, goff, gpv, gunk, ooff, obs, inplace=False)
    torch.cuda.synchronize(); t0=time.time()
    for _ in range(10): v.optimize(ctrl, goff, gpv, gunk, ooff, obs)
    torch.cuda.synchronize(); dt=(time.time()-t0)/10
    print(f"  B={b.B} solve {dt*1e6:.1f} us -> {b.B/dt:.3g} traj/s")
    v.close()
# config-2 timing at B=1024
b = synth.make_bspline_batch(world, 1024, 32, 5)
P = default_params(); P.max_iterations=50
v = Vigo(0, P); dev=v.device
T = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
ctrl=T(b.ctrl); goff=T(b.guide_off); gpv=T(b.guide_pv); gunk=T(b.guide_unk)
for B in (1024, 4096, 16384):
    reps = B//1024
    c=ctrl.repeat(reps,1,1).contiguous(); 
    go = torch.cat([goff[:-1].repeat(reps), goff[-1:]]) if reps>1 else goff  # same pairs reused (offsets repeat)
    for _ in range(2): v.optimize(c, go, gpv, gunk)
    torch.cuda.synchronize(); t0=time.time()
    for _ in range(5): v.optimize(c, go, gpv, gunk)
    torch.cuda.synchronize(); dt=(time.time()-t0)/5
    print(f"B={B}: {dt*1e3:.3f} ms/solve -> {B/dt:.4g} traj/s")
