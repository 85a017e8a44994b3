"""-m gpu: BASELINE configs[1] (1024 x 32, 256^3, 50 iterations) on guide pairs produced by the planner's OWN host
pipeline — findCollisionSeg -> A* -> assignGuidePointsSemiCircle (bsplineTraj.cpp:403-571, bsplineTraj.h:206-304) through
libtrajectory_planner_vigo.so, product code calling product code — in a world dense enough that over half of the
trajectories cross an inflated box, and again after one and two re-guides (bsplineTraj.cpp:640-648 replayed on the
optimizer's own output: pairs are APPENDED, so control points that stay in collision carry three and more pairs — past the
two the solve kernel keeps in registers).  At every stage, all 1024 trajectories are bit-exact against the emulation-mode
oracle (status, iteration and evaluation counts, x, control points, objective).

Against the reference-order oracle the 1e-4 bar is held by 99.9 % of the trajectories before the first re-guide and by
97 - 98.5 % after it — and that is the problem's own sensitivity, not the kernel's: 50 unconverged L-BFGS iterations on
these collision-heavy trajectories amplify a last-bit difference past 1e-4 in a few per cent of the cases.
The test measures that floor itself: the reference-order oracle against ITSELF with one coordinate of one control point
moved by one ulp.  The kernel must be at least as close to the reference as the reference is to its one-ulp neighbour
(measured on the CPU: 94.8 % / 94.9 % within 1e-4 at the two re-guided stages against the kernel's 96.9 % / 98.5 %), the
median must stay at rounding level and the objective reached must agree."""
import numpy as np
import pytest
import torch

import oracle_lib as ol
from gpu_util import batch_to_dev, emulation, rel_err_per_traj, to_dev
from trajectory_planner_amd import synth
from trajectory_planner_amd.vigo import PREC_F64, PREC_F64_FAST, default_params

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _solve_and_compare(v, P, b, label, fast=False):
    d = batch_to_dev(b, v.device)
    gunk = v.guides_unknown(d["guide_pv"])
    assert np.array_equal(gunk.cpu().numpy(), b.guide_unk)
    d["guide_unk"] = gunk
    r = v.optimize(**d)
    g = {k: getattr(r, k).cpu().numpy() for k in ("ctrl", "x", "status", "fx", "iters", "evals")}
    if fast:
        ol.oracle().vgo_set_emulation_fast(1)
    try:
        with emulation(b.N):
            e = ol.optimize_batch(P, b)
    finally:
        if fast:
            ol.oracle().vgo_set_emulation_fast(0)
    for k in ("status", "iters", "evals", "x", "ctrl", "fx"):
        assert np.array_equal(g[k], e[k]), f"{label}: {k} differs from the emulation-mode oracle"
    ref = ol.optimize_batch(P, b)
    rel = rel_err_per_traj(g["ctrl"], ref["ctrl"])
    # the reference's own sensitivity on this batch: one coordinate of one free control point moved by one ulp
    c2 = b.ctrl.copy()
    c2[:, b.N // 2 - 1, 0] = np.nextafter(c2[:, b.N // 2 - 1, 0], np.inf)
    nudged = ol.optimize_batch(P, synth.Batch(c2, b.guide_off, b.guide_pv, b.guide_unk))
    rel_floor = rel_err_per_traj(nudged["ctrl"], ref["ctrl"])
    fx_rel = np.abs(g["fx"] - ref["fx"]) / np.maximum(np.abs(ref["fx"]), 1e-300)
    hist, per_traj, share = synth.pairs_histogram(b)
    print(f"\n[{label}] guide pairs {len(b.guide_pv)} ({per_traj.mean():.2f} per trajectory, max {per_traj.max()}; {share * 100:.1f} % of trajectories "
          f"guided); pairs per free control point {hist.tolist()}; mean iterations {g['iters'].mean():.1f}, evaluations {g['evals'].mean():.1f}; "
          f"vs reference-order oracle: median {np.median(rel):.2e} p99 {np.quantile(rel, .99):.2e} max {rel.max():.2e}, "
          f"within 1e-4: {(rel <= TOL).mean() * 100:.2f} % (the reference against its one-ulp neighbour: {(rel_floor <= TOL).mean() * 100:.2f} %, "
          f"median {np.median(rel_floor):.2e}); objective reached: median rel. difference {np.median(fx_rel):.2e}, max {fx_rel.max():.2e}")
    # as close to the reference as the reference is to itself one ulp away (0.5 % slack for the two samples' scatter)
    assert (rel <= TOL).mean() >= (rel_floor <= TOL).mean() - 0.005, label
    assert np.median(rel) < 1e-8 and np.median(fx_rel) < 1e-9 and fx_rel.max() < 5e-3, label
    return g, rel, hist, share


def test_config2_on_pipeline_guides_at_full_size(vigo_handle):
    v = vigo_handle
    world = synth.make_pipeline_world()
    b = synth.make_pipeline_batch(world, 1024, 32, synth.SEED_BASE + 2 + 2000)
    P = default_params()
    P.max_iterations = 50
    v.set_params(P)
    v.set_grid(to_dev(world.voxels, v.device), world.origin, world.res)
    shares, maxpairs = [], []
    for stage in range(3):
        g, rel, hist, share = _solve_and_compare(v, P, b, f"pipeline guides, {stage} re-guide(s)")
        assert (rel <= TOL).mean() >= (0.995, 0.96, 0.97)[stage], f"stage {stage}: {(rel > TOL).sum()} trajectories outside 1e-4 (max {rel.max():.3e})"
        assert np.array_equal(g["ctrl"][:, :3], b.ctrl[:, :3]) and np.array_equal(g["ctrl"][:, -3:], b.ctrl[:, -3:])
        shares.append(share)
        maxpairs.append(len(hist) - 1)
        if stage < 2:
            b = synth.reguide_batch(world, b, g["ctrl"])
    assert shares[0] >= 0.30                      # >= 30 % of the trajectories have a collision segment to begin with
    assert maxpairs[2] >= 3                       # after two re-guides control points carry three pairs and more
    # the throughput mode on the heaviest stage: bit-exact vs its own emulation, the same floor
    v.set_precision(PREC_F64_FAST)
    try:
        g, rel, _, _ = _solve_and_compare(v, P, b, "pipeline guides, 2 re-guides, f64_fast", fast=True)
        assert (rel <= TOL).mean() >= 0.94, f"f64_fast: {(rel > TOL).sum()} trajectories outside 1e-4 (max {rel.max():.3e})"
    finally:
        v.set_precision(PREC_F64)
