"""-m gpu: parity of the HIP cost/gradient and whole-solve kernels, called through the C ABI
(include/vigo.h), against the CPU oracle on the same seeded inputs.

Two comparisons per case:
  exact   oracle in device-emulation mode (same formulas, lane-tree sums): BIT FOR BIT —
          cost, gradient, control points, x, status, iteration and evaluation counts;
  1e-4    oracle in reference order (what the reference computes): control points within
          1e-4 relative (north_star's tolerance), the distribution printed.
"""
import os

import numpy as np
import pytest
import torch

import oracle_lib as ol
from gpu_util import batch_to_dev, emulation, rel_err_per_traj, to_dev
from trajectory_planner_amd import synth
from trajectory_planner_amd.vigo import PREC_F32, PREC_F64, Vigo, default_params

pytestmark = pytest.mark.gpu
TOL = 1e-4  # BASELINE.json north_star: control points within 1e-4 relative


def solve_both(v, P, b, weights=None):
    d = batch_to_dev(b, v.device, weights)
    r = v.optimize(**d)
    torch.cuda.synchronize()
    return r, d


@pytest.mark.parametrize("N,B,n_obs,iters", [(32, 257, 0, 50), (32, 64, 2, 200), (7, 33, 0, 50), (12, 50, 1, 50),
                                             (33, 40, 0, 50), (64, 65, 2, 50), (50, 31, 0, 50),
                                             (65, 9, 0, 50), (82, 20, 1, 50), (128, 6, 0, 50), (129, 5, 1, 50), (200, 3, 0, 50)])
def test_optimize_matches_oracle(vigo_handle, small_world, N, B, n_obs, iters):
    v = vigo_handle
    P = default_params()
    P.max_iterations = iters
    v.set_params(P)
    b = synth.make_bspline_batch(small_world, B, N, 1000 + N + B, start_range=3.0, n_obs=n_obs)
    r, d = solve_both(v, P, b)
    with emulation(N):
        e = ol.optimize_batch(P, b)
    g = {k: getattr(r, k).cpu().numpy() for k in ("ctrl", "x", "status", "fx", "iters", "evals")}
    for k in ("status", "iters", "evals", "x", "ctrl", "fx"):
        assert np.array_equal(g[k], e[k]), f"{k} differs from the emulation-mode oracle"
    ref = ol.optimize_batch(P, b)
    rel = rel_err_per_traj(g["ctrl"], ref["ctrl"])
    print(f"\n[N={N} B={B} obs={n_obs} it={iters}] vs reference-order oracle: median {np.median(rel):.2e} "
          f"p99 {np.quantile(rel, .99):.2e} max {rel.max():.2e}; status {dict(zip(*np.unique(g['status'], return_counts=True)))}")
    if iters <= 50:
        # measured: EVERY trajectory of every case inside 1e-4 (worst 8.3e-7) — demanded, not a quantile
        assert (rel <= TOL).all() and np.median(rel) < 1e-8, f"{(rel > TOL).sum()} trajectories outside 1e-4, max {rel.max():.3e}"
        print(f"    equal status codes: {(g['status'] == ref['status']).mean() * 100:.1f} %")
        assert (g["status"] == ref["status"]).mean() >= 0.97
    else:
        # 200 unconverged iterations amplify a last-bit difference (summation order, pow vs x*x)
        # past 1e-4 — SURVEY.md §9 measured the same on the CPU alone; the reference's own result
        # is then libm/compiler dependent.  Parity at this setting = the bit-exact match with the
        # emulation-mode oracle asserted above, plus agreement of the objective reached.
        frel = np.abs(g["fx"] - ref["fx"]) / np.abs(ref["fx"])
        print(f"    objective: median rel diff {np.median(frel):.2e} max {frel.max():.2e}")
        assert np.median(frel) < 1e-3


def test_cost_grad_matches_oracle_all_terms(vigo_handle, small_world):
    v = vigo_handle
    for (N, n_obs, planz, unc) in [(32, 2, 0, 1.0), (20, 3, 1, 2.0), (64, 1, 0, 2.0), (9, 0, 1, 1.0), (100, 2, 1, 2.0), (256, 1, 0, 1.0)]:
        P = default_params()
        P.plan_in_z, P.uncertain_factor = planz, unc
        v.set_params(P)
        b = synth.make_bspline_batch(small_world, 130, N, 40 + N, start_range=3.0, n_obs=n_obs)
        rng = np.random.default_rng(N)
        b.guide_pv[:, 5] = rng.normal(0, 0.3, size=len(b.guide_pv))   # v_z != 0 exercises planInZ
        w = rng.uniform(0.5, 4.0, size=(b.B, 4))
        d = batch_to_dev(b, v.device, w)
        cost, grad, terms = v.cost_grad(**d)
        with emulation(N):
            ce, ge, te = ol.cost_grad_batch(P, b, w)
        assert np.array_equal(cost.cpu().numpy(), ce) and np.array_equal(grad.cpu().numpy(), ge)
        assert np.array_equal(terms.cpu().numpy(), te)
        cr, gr, tr = ol.cost_grad_batch(P, b, w)
        assert np.max(np.abs(cost.cpu().numpy() - cr) / np.abs(cr)) < 1e-13
        assert np.max(np.abs(grad.cpu().numpy() - gr)) <= 1e-12 * np.max(np.abs(gr))


def test_golden_reference_solves(vigo_handle):
    """tests/golden/lbfgs_ref.npz: solves produced by the reference's own lbfgs_optimize."""
    v = vigo_handle
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lbfgs_ref.npz"))
    worst = 0.0
    for k, (N, iters, status, evals) in enumerate(g["meta"]):
        P = default_params()
        P.max_iterations = int(iters)
        v.set_params(P)
        N = int(N)
        goff = np.zeros(N + 1, dtype=np.int32)
        goff[:] = g[f"c{k}_goff"]
        obs = g[f"c{k}_obs"]
        r = v.optimize(to_dev(g[f"c{k}_ctrl0"][None], v.device), to_dev(goff, v.device),
                       to_dev(g[f"c{k}_gpv"], v.device) if len(g[f"c{k}_gpv"]) else None,
                       to_dev(g[f"c{k}_gunk"], v.device) if len(g[f"c{k}_gunk"]) else None,
                       None, to_dev(obs, v.device) if len(obs) else None, to_dev(g[f"c{k}_w"][None], v.device))
        ctrl = r.ctrl.cpu().numpy()[0]
        rel = np.abs(ctrl - g[f"c{k}_ctrl"]).max() / np.abs(g[f"c{k}_ctrl"]).max()
        if iters <= 50:      # 200-iteration cases: see test_optimize_matches_oracle
            worst = max(worst, rel)
            assert rel <= TOL, (k, rel)
        else:
            assert abs(float(r.fx.cpu()[0]) - g[f"c{k}_fx"][0]) <= 2e-2 * abs(g[f"c{k}_fx"][0]), k
    print(f"\nworst relative control-point error over the golden reference solves: {worst:.2e}")


def test_edge_cases(vigo_handle, small_world):
    v = vigo_handle
    P = default_params()
    P.max_iterations = 50
    v.set_params(P)
    dev = v.device
    # empty batch
    r = v.optimize(torch.zeros(0, 32, 3, dtype=torch.float64, device=dev), torch.zeros(1, dtype=torch.int32, device=dev))
    assert r.status.numel() == 0
    # no guides at all (guide_off == NULL) and odd batch size; an already-minimal straight slow line
    line = np.zeros((3, 16, 3))
    line[:, :, 0] = np.arange(16) * 0.15
    line[:, :, 2] = 1.0
    r = v.optimize(to_dev(line, dev))
    assert (r.status.cpu().numpy() == 2).all() and (r.iters.cpu().numpy() == 0).all()   # LBFGS_ALREADY_MINIMIZED
    assert np.array_equal(r.ctrl.cpu().numpy(), line)
    # unsupported N is refused loudly
    from trajectory_planner_amd.vigo import VigoError
    with pytest.raises(VigoError):
        v.optimize(torch.zeros(1, 6, 3, dtype=torch.float64, device=dev))
    with pytest.raises(VigoError):
        v.optimize(torch.zeros(1, 257, 3, dtype=torch.float64, device=dev))     # > VIGO_MAX_CTRL_POINTS
    with pytest.raises(VigoError):
        v.optimize(torch.zeros(1, 230, 3, dtype=torch.float64, device=dev))     # history does not fit 160 KiB of LDS
    # the unbounded setting of lbfgs.hpp (max_iterations = 0) is refused: every wave must reach an exit
    P0 = default_params()
    P0.max_iterations = 0
    with pytest.raises(VigoError):
        v.set_params(P0)
    # max_iterations = 1 and mem_size = 3 (history ring wraps many times)
    for (iters, mem) in ((1, 16), (50, 3), (50, 1)):
        P2 = default_params()
        P2.max_iterations, P2.mem_size = iters, mem
        v.set_params(P2)
        b = synth.make_bspline_batch(small_world, 21, 32, 5 + mem, start_range=3.0)
        r = v.optimize(**batch_to_dev(b, dev))
        with emulation(32):
            e = ol.optimize_batch(P2, b)
        assert np.array_equal(r.ctrl.cpu().numpy(), e["ctrl"]) and np.array_equal(r.status.cpu().numpy(), e["status"])


def test_line_search_failure_keeps_last_trial_point(vigo_handle, small_world):
    """BT.cpp:803 / LB:1192: on ls < 0, ctrl holds the last trial while x is reverted."""
    v = vigo_handle
    P = default_params()
    P.max_iterations = 200
    v.set_params(P)
    b = synth.make_bspline_batch(small_world, 512, 20, 4242, start_range=3.0, n_obs=2)
    r = v.optimize(**batch_to_dev(b, v.device))
    st = r.status.cpu().numpy()
    fail = np.nonzero((st < 0) & (st != -1004))[0]
    assert len(fail) > 0, "the family no longer produces a line-search failure; pick another seed"
    ctrl, x = r.ctrl.cpu().numpy(), r.x.cpu().numpy()
    differs = [not np.array_equal(ctrl[i, 3:-3], x[i]) for i in fail]
    assert any(differs)
    with emulation(20):
        e = ol.optimize_batch(P, b)
    assert np.array_equal(ctrl, e["ctrl"]) and np.array_equal(x, e["x"]) and np.array_equal(st, e["status"])


def test_config2_at_the_references_iteration_cap(vigo_handle):
    """BT.cpp:698 runs the solver with max_iterations = 200; BASELINE config 2 quotes 50.  The control-point statement
    at the reference's own cap, on the config-2 batch (1024 x 32, 256^3): bit-identical to the emulation-mode oracle (the
    parity gate proper), and against the reference-order oracle what 200 unconverged iterations leave of a last-bit
    difference in summation order — measured 70 % of the trajectories within 1e-4, median 4.4e-8, objective reached
    equal to 2e-9 (median), 97.8 % equal status codes; SURVEY.md §9 measured the same amplification on the CPU alone
    (an injected relative noise of 1e-15 per evaluation: 1e-3 after 50 iterations in the worst case), so the
    reference's own result at this setting depends on its compiler and libm to that extent."""
    v = vigo_handle
    P = default_params()
    P.max_iterations = 200
    v.set_params(P)
    world, b = synth.config2()
    v.set_grid(torch.from_numpy(world.voxels).to(v.device), world.origin, world.res)
    r = v.optimize(**batch_to_dev(b, v.device))
    g = {k: getattr(r, k).cpu().numpy() for k in ("ctrl", "x", "status", "fx", "iters", "evals")}
    with emulation(b.N):
        e = ol.optimize_batch(P, b)
    for k in ("status", "iters", "evals", "x", "ctrl", "fx"):
        assert np.array_equal(g[k], e[k]), f"{k} differs from the emulation-mode oracle at 200 iterations"
    ref = ol.optimize_batch(P, b)
    rel = rel_err_per_traj(g["ctrl"], ref["ctrl"])
    frel = np.abs(g["fx"] - ref["fx"]) / np.abs(ref["fx"])
    print(f"\n[config 2 at 200 iterations] vs reference-order oracle: within 1e-4 {(rel <= TOL).mean():.3f}, median {np.median(rel):.2e}, "
          f"p90 {np.quantile(rel, .9):.2e}, max {rel.max():.2e}; objective median rel diff {np.median(frel):.2e}; "
          f"equal status {(g['status'] == ref['status']).mean():.3f}; mean iterations {g['iters'].mean():.1f}")
    assert np.median(rel) < 1e-6 and (rel <= TOL).mean() >= 0.6
    assert np.median(frel) < 1e-7 and (g["status"] == ref["status"]).mean() >= 0.95


def test_determinism_permutation_and_subbatch_invariance(vigo_handle, small_world):
    """size-independent properties at the BASELINE config-2 size (1024 x 32, 50 iterations):
    re-running gives identical bits; a trajectory's result does not depend on its batch slot."""
    v = vigo_handle
    P = default_params()
    P.max_iterations = 50
    v.set_params(P)
    world, b = synth.config2()
    d = batch_to_dev(b, v.device)
    r1 = v.optimize(**d)
    r2 = v.optimize(**d)
    assert torch.equal(r1.ctrl, r2.ctrl) and torch.equal(r1.status, r2.status) and torch.equal(r1.fx, r2.fx)
    # permute the trajectories (rebuild the CSR accordingly)
    perm = np.random.default_rng(0).permutation(b.B)
    N = b.N
    counts = np.diff(b.guide_off).reshape(b.B, N)[perm]
    starts = b.guide_off[:-1].reshape(b.B, N)[perm]
    new_off = np.zeros(b.B * N + 1, dtype=np.int32)
    new_off[1:] = np.cumsum(counts.reshape(-1))
    idx = np.concatenate([np.arange(s, s + c) for s, c in zip(starts.reshape(-1), counts.reshape(-1))]) if counts.sum() else np.zeros(0, int)
    pb = synth.Batch(b.ctrl[perm], new_off, b.guide_pv[idx.astype(int)], b.guide_unk[idx.astype(int)])
    rp = v.optimize(**batch_to_dev(pb, v.device))
    assert np.array_equal(rp.ctrl.cpu().numpy(), r1.ctrl.cpu().numpy()[perm])
    assert np.array_equal(rp.evals.cpu().numpy(), r1.evals.cpu().numpy()[perm])
    # objective never increases; statuses are legal; full-size parity against the oracle
    c0, _, _ = v.cost_grad(**d)
    assert (r1.fx <= c0 + 1e-12).all()
    st = r1.status.cpu().numpy()
    assert set(np.unique(st)) <= {0, 2, -1004, -1008, -1005, -1001, -1007, -1006, -1003, -1010, -1009}
    ref = ol.optimize_batch(P, b)
    rel = rel_err_per_traj(r1.ctrl.cpu().numpy(), ref["ctrl"])
    print(f"\nconfig 2 (1024x32, 50 it) vs reference-order oracle: median {np.median(rel):.2e} p99 {np.quantile(rel, .99):.2e} "
          f"max {rel.max():.2e} within 1e-4: {(rel <= TOL).mean():.4f}")
    assert (rel <= TOL).all()          # every trajectory of the BASELINE batch (measured: max 1.2e-6)


def test_fp32_mode_cost_and_statistics(vigo_handle, small_world):
    """VIGO_PREC_F32 (throughput mode): cost/gradient within 1e-5 relative of the fp64 oracle; the
    50-iteration end point is reported, not gated at 1e-4 (fp32 state cannot hold it)."""
    v = vigo_handle
    P = default_params()
    P.max_iterations = 50
    v.set_params(P)
    v.set_precision(PREC_F32)
    b = synth.make_bspline_batch(small_world, 256, 32, 99, start_range=3.0, n_obs=1)
    d = batch_to_dev(b, v.device)
    cost, grad, _ = v.cost_grad(**d)
    cr, gr, _ = ol.cost_grad_batch(P, b)
    assert np.max(np.abs(cost.cpu().numpy() - cr) / np.abs(cr)) < 1e-5
    assert np.max(np.abs(grad.cpu().numpy() - gr)) <= 1e-5 * np.max(np.abs(gr))
    r = v.optimize(**d)
    ref = ol.optimize_batch(P, b)
    rel = rel_err_per_traj(r.ctrl.cpu().numpy(), ref["ctrl"])
    print(f"\nfp32 mode end-point error vs fp64 oracle: median {np.median(rel):.2e} p90 {np.quantile(rel, .9):.2e} max {rel.max():.2e}")
    assert np.median(rel) < 5e-2 and np.isfinite(r.ctrl.cpu().numpy()).all()
    assert (r.fx.cpu().numpy() <= cr * (1 + 1e-5) + 1e-6).all()
    v.set_precision(PREC_F64)


@pytest.mark.parametrize("N,B,prec,n_obs,mem,iters", [
    (32, 4100, PREC_F32, 1, 16, 30), (64, 2100, PREC_F32, 1, 16, 30), (16, 4100, PREC_F64, 1, 16, 30), (20, 4100, 2, 1, 16, 30),
    (32, 4100, PREC_F32, 0, 16, 30), (64, 2100, PREC_F32, 0, 16, 30), (16, 4100, PREC_F64, 0, 16, 30), (20, 4100, 2, 0, 16, 30),
    # fp64 at 21 < N <= 64 without obstacles: the level instantiations that keep 4 / 5 history pairs in registers (eight
    # waves per CU) — full history (the steady-state two-loop after the 16th iteration) and a history of 15 (the general
    # two-loop throughout, ring of 10 / 9 slots)
    (32, 4200, PREC_F64, 0, 16, 40), (32, 4200, 2, 0, 16, 40), (32, 2300, PREC_F64, 0, 15, 30), (64, 2100, PREC_F64, 0, 16, 40),
    (50, 1300, 2, 0, 16, 30), (64, 1300, PREC_F64, 0, 15, 25)])
def test_large_batches_use_the_two_wave_kernel_with_identical_results(vigo_handle, small_world, N, B, prec, n_obs, mem, iters):
    """Batches with more wavefronts than the chip has SIMDs whose history leaves room for eight waves per CU (fp32
    state, or short fp64 trajectories) run the register-capped instantiation (two waves per SIMD, scratch
    spills): the same arithmetic — every trajectory bit-identical to the same batch solved in slices small
    enough for the one-wave instantiation.  Without obstacles the batch (level trajectories with some vertically
    jittered ones among them) goes through the register-capped LEVEL instantiation and the general one."""
    v = vigo_handle
    P = default_params()
    P.max_iterations = iters
    P.mem_size = mem
    v.set_params(P)
    v.set_precision(prec)
    b = synth.make_bspline_batch(small_world, B, N, 1234 + N, start_range=3.0, n_obs=n_obs, z_jitter=0.0 if n_obs else 0.02, z_share=0.1)
    d = batch_to_dev(b, v.device)
    full = v.optimize(**d)
    step = 500
    for lo in range(0, B, step):
        hi = min(B, lo + step)
        goff = b.guide_off[lo * N:hi * N + 1]
        ooff = obs = None
        if n_obs:
            ooff = b.obs_off[lo:hi + 1]
            obs = b.obs[ooff[0]:ooff[-1]]
            ooff = ooff - ooff[0]
        sl = synth.Batch(b.ctrl[lo:hi], goff - goff[0], b.guide_pv[goff[0]:goff[-1]], b.guide_unk[goff[0]:goff[-1]], ooff, obs)
        part = v.optimize(**batch_to_dev(sl, v.device))
        for k in ("ctrl", "x", "status", "fx", "iters", "evals"):
            assert torch.equal(getattr(part, k), getattr(full, k)[lo:hi]), (k, lo)
    assert bool(torch.isfinite(full.ctrl).all())
    if prec != PREC_F32 and not n_obs:       # and against the emulation oracle (which has no fp32 mode)
        fast = prec == 2
        ol.oracle().vgo_set_emulation_fast(1 if fast else 0)
        try:
            with emulation(N):
                e = ol.optimize_batch(P, b)
        finally:
            ol.oracle().vgo_set_emulation_fast(0)
        for k in ("status", "iters", "evals", "x", "ctrl", "fx"):
            assert np.array_equal(getattr(full, k).cpu().numpy(), e[k]), f"{k} differs from the emulation-mode oracle"
    v.set_precision(PREC_F64)


@pytest.mark.parametrize("N,B,n_obs", [(32, 300, 0), (20, 64, 2), (64, 40, 1), (100, 12, 0)])
def test_fast_mode_matches_its_emulation_and_the_reference_gate(vigo_handle, small_world, N, B, n_obs):
    """VIGO_PREC_F64_FAST (explicit fma + one reciprocal per history pair): bit-exact against the
    oracle's fast emulation, and inside the same 1e-4 gate against the reference-order oracle."""
    from trajectory_planner_amd.vigo import PREC_F64_FAST
    v = vigo_handle
    P = default_params()
    P.max_iterations = 50
    v.set_params(P)
    v.set_precision(PREC_F64_FAST)
    b = synth.make_bspline_batch(small_world, B, N, 777 + N, start_range=3.0, n_obs=n_obs)
    d = batch_to_dev(b, v.device)
    cost, grad, terms = v.cost_grad(**d)
    r = v.optimize(**d)
    ol.oracle().vgo_set_emulation_fast(1)
    try:
        with emulation(N):
            ce, ge, te = ol.cost_grad_batch(P, b)
            e = ol.optimize_batch(P, b)
    finally:
        ol.oracle().vgo_set_emulation_fast(0)
    assert np.array_equal(cost.cpu().numpy(), ce) and np.array_equal(grad.cpu().numpy(), ge)
    for k in ("status", "iters", "evals", "x", "ctrl", "fx"):
        assert np.array_equal(getattr(r, k).cpu().numpy(), e[k]), f"{k} differs from the fast-emulation oracle"
    ref = ol.optimize_batch(P, b)
    rel = rel_err_per_traj(r.ctrl.cpu().numpy(), ref["ctrl"])
    print(f"\n[fast N={N}] vs reference-order oracle: median {np.median(rel):.2e} p99 {np.quantile(rel, .99):.2e} max {rel.max():.2e}")
    assert (rel <= TOL).all() and np.median(rel) < 1e-8          # measured: every trajectory, worst 1.1e-6
    v.set_precision(PREC_F64)


@pytest.mark.parametrize("mode", ["f64", "f64_fast", "f64_strict_z"])
@pytest.mark.parametrize("name,n,n_boxes,centre,B,N,start", [("configs[1]", 256, 200, 12.0, 1024, 32, 8.0),
                                                            ("configs[3] shard", 512, 800, 24.0, 8192, 64, 16.0)])
def test_baseline_sizes_in_full(vigo_handle, name, n, n_boxes, centre, B, N, start, mode):
    """BASELINE.json configs[1] (1024 x 32, 256^3) and one GPU's shard of configs[3] (8192 x 64, 512^3) at FULL
    size, the bench.py workloads themselves: every trajectory bit-exact vs the emulation-mode oracle and
    within 1e-4 of the reference-order oracle (the oracle needs ~0.1 s / ~5 s for them) — in the reference-order
    arithmetic (`value` of bench.py) AND in the f64_fast mode bench.py quotes as `other_mode` (explicit fma, one
    reciprocal per history pair): bit-exact against its own emulation, the same 1e-4 bar on every trajectory.
    These batches are level (paths at z = 1.0), so they run the level instantiations; `f64_strict_z`
    (vigo_params_t.strict_z = 1: no level rule) sends the same batches through the GENERAL kernel, as round 2 did."""
    from trajectory_planner_amd.vigo import PREC_F64, PREC_F64_FAST
    v = vigo_handle
    cfg = 2 if N == 32 else 4
    world = synth.make_box_world(synth.SEED_BASE + cfg, n=n, n_boxes=n_boxes, centre_range=centre)
    b = synth.make_bspline_batch(world, B, N, synth.SEED_BASE + cfg + 1000, start_range=start)
    P = default_params()
    P.max_iterations = 50
    P.strict_z = 1 if mode == "f64_strict_z" else 0
    v.set_params(P)
    v.set_grid(to_dev(world.voxels, v.device), world.origin, world.res)
    d = batch_to_dev(b, v.device)
    gunk = v.guides_unknown(d["guide_pv"])
    assert np.array_equal(gunk.cpu().numpy(), b.guide_unk)
    d["guide_unk"] = gunk
    fast = mode == "f64_fast"
    v.set_precision(PREC_F64_FAST if fast else PREC_F64)
    try:
        r = v.optimize(**d)
        g = {k: getattr(r, k).cpu().numpy() for k in ("ctrl", "x", "status", "fx", "iters", "evals")}
    finally:
        v.set_precision(PREC_F64)
    ol.oracle().vgo_set_emulation_fast(1 if fast else 0)
    try:
        with emulation(N):
            e = ol.optimize_batch(P, b)
    finally:
        ol.oracle().vgo_set_emulation_fast(0)
    for k in ("status", "iters", "evals", "x", "ctrl", "fx"):
        assert np.array_equal(g[k], e[k]), f"{name} ({mode}): {k} differs from the emulation-mode oracle"
    ref = ol.optimize_batch(P, b)
    rel = rel_err_per_traj(g["ctrl"], ref["ctrl"])
    print(f"\n[{name}: {B} x {N}, {mode}] vs reference-order oracle: median {np.median(rel):.2e} p99 {np.quantile(rel, .99):.2e} "
          f"max {rel.max():.2e}; within 1e-4: {(rel <= TOL).mean() * 100:.2f} %")
    assert (rel <= TOL).all() and np.median(rel) < 1e-8     # 100 % (measured in f64: max 1.2e-6 / 2.1e-6)
    # fixed boundary control points never move (BT.cpp:690-691)
    assert np.array_equal(g["ctrl"][:, :3], b.ctrl[:, :3]) and np.array_equal(g["ctrl"][:, -3:], b.ctrl[:, -3:])



@pytest.mark.parametrize("N,B,n_obs,prec", [(32, 301, 0, "f64"), (32, 301, 0, "f64_fast"), (20, 90, 0, "f64"), (64, 75, 0, "f64"), (32, 120, 2, "f64"), (100, 20, 0, "f64")])
def test_level_rule_on_mixed_batches(vigo_handle, small_world, N, B, n_obs, prec):
    """THE LEVEL RULE (include/vigo.h): a trajectory whose control points share one height to 2^-40 (and no z planning)
    keeps its z fixed — the z terms of smoothness and feasibility, rounding noise on such input, are taken as exactly
    zero — and calls without obstacles solve waves of such trajectories with a kernel that carries x and y only.  Here:
    batches that MIX exactly level trajectories (z = 1.0 to the bit), level-to-rounding ones (as the fit leaves them) and
    trajectories with centimetres of vertical jitter, so that level waves, general waves and waves that pair one of
    each all occur.  (a) every trajectory bit-identical to the emulation-mode oracle, which applies the same rule per
    trajectory; (b) nothing depends on the pairing: the batch in another order gives the same result per trajectory;
    (c) against the reference-order oracle, which knows no such rule, every trajectory stays within 1e-4 and a level
    trajectory's z within 1e-12 of what the reference computes."""
    from trajectory_planner_amd.vigo import PREC_F64, PREC_F64_FAST
    v = vigo_handle
    P = default_params()
    P.max_iterations = 50
    v.set_params(P)
    b = synth.make_bspline_batch(small_world, B, N, 4100 + N + B, start_range=3.0, n_obs=n_obs, z_jitter=0.03, z_share=0.4)
    spread = np.ptp(b.ctrl[:, :, 2], axis=1)
    wavy = spread > 1e-6
    exact = (~wavy) & (np.arange(B) % 3 == 0)
    b.ctrl[exact, :, 2] = 1.0                                  # exactly level
    # ... and some a hair inside / outside the rule's band of 2^-40 (9.09e-13): the decision is made ONCE per solve from the
    # points it starts with — a trajectory just outside that the smoothing pulls inside must not change sides on the way
    # (the two launches of a solve each look at the control points: the level kernel runs first for that reason)
    edge = np.nonzero((~wavy) & (np.arange(B) % 3 == 1))[0]
    for j, i in enumerate(edge):
        b.ctrl[i, 3 + j % (N - 6), 2] += (8.9e-13, 9.3e-13, 9.6e-13, 2e-12)[j % 4]
    zmin, zmax = b.ctrl[:, :, 2].min(1), b.ctrl[:, :, 2].max(1)
    level = (zmax - zmin) <= 2.0 ** -40 * np.maximum(1.0, np.maximum(np.abs(zmin), np.abs(zmax)))     # the rule (include/vigo.h)
    nearly = (~wavy) & ~level
    assert wavy.sum() > B // 5 and level.sum() > B // 5 and exact.sum() > 0 and (level & ~exact).sum() > 0 and nearly.sum() > 0
    fast = prec == "f64_fast"
    v.set_precision(PREC_F64_FAST if fast else PREC_F64)
    ol.oracle().vgo_set_emulation_fast(1 if fast else 0)
    try:
        r, d = solve_both(v, P, b)
        cost, grad, terms = v.cost_grad(**d)
        with emulation(N):
            e = ol.optimize_batch(P, b)
            ce, ge, te = ol.cost_grad_batch(P, b)
        g = {k: getattr(r, k).cpu().numpy() for k in ("ctrl", "x", "status", "fx", "iters", "evals")}
        assert np.array_equal(cost.cpu().numpy(), ce) and np.array_equal(grad.cpu().numpy(), ge) and np.array_equal(terms.cpu().numpy(), te)
        for k in ("status", "iters", "evals", "x", "ctrl", "fx"):
            assert np.array_equal(g[k], e[k]), f"{k} differs from the emulation-mode oracle"
        # (b) another order: other neighbours in the wave, other waves level / general
        perm = np.random.default_rng(N + B).permutation(B)
        cnt = np.diff(b.guide_off).reshape(B, N)[perm]
        goff = np.concatenate([[0], np.cumsum(cnt.reshape(-1))]).astype(np.int32)
        starts = b.guide_off[:-1].reshape(B, N)[perm].reshape(-1)
        idx = np.concatenate([np.arange(s, s + c) for s, c in zip(starts, cnt.reshape(-1))]) if goff[-1] else np.zeros(0, dtype=np.int64)
        obs_off = obs = None
        if b.obs is not None:
            per = np.diff(b.obs_off)[perm]
            obs_off = np.concatenate([[0], np.cumsum(per)]).astype(np.int32)
            obs = np.concatenate([b.obs[b.obs_off[i]:b.obs_off[i + 1]] for i in perm])
        bp = synth.Batch(np.ascontiguousarray(b.ctrl[perm]), goff, np.ascontiguousarray(b.guide_pv[idx.astype(np.int64)]),
                         np.ascontiguousarray(b.guide_unk[idx.astype(np.int64)]), obs_off, obs)
        rp, _ = solve_both(v, P, bp)
        for k in ("status", "iters", "evals", "x", "ctrl", "fx"):
            assert np.array_equal(getattr(rp, k).cpu().numpy(), g[k][perm]), f"{k} depends on the order of the batch"
    finally:
        v.set_precision(PREC_F64)
        ol.oracle().vgo_set_emulation_fast(0)
    # the rule at work: a level trajectory's z has not moved at all, a wavy one's has (and one a hair outside the band is
    # not level: its z follows the reference's dynamics)
    assert np.array_equal(g["ctrl"][level, :, 2], b.ctrl[level, :, 2])
    assert not np.array_equal(g["ctrl"][nearly, :, 2], b.ctrl[nearly, :, 2])
    assert (np.abs(g["ctrl"][wavy, 3:-3, 2] - b.ctrl[wavy, 3:-3, 2]).max(1) > 1e-6).all()
    ref = ol.optimize_batch(P, b)                              # reference order: no level rule
    rel = rel_err_per_traj(g["ctrl"], ref["ctrl"])
    zdev = np.abs(g["ctrl"][level, :, 2] - ref["ctrl"][level, :, 2]).max()
    print(f"\n[level rule N={N} B={B} obs={n_obs} {prec}] level {level.sum()} (exactly {exact.sum()}), a hair outside the band {nearly.sum()}, wavy {wavy.sum()}; vs reference-order oracle: "
          f"median {np.median(rel):.2e} max {rel.max():.2e}; largest z difference on a level trajectory {zdev:.2e}")
    assert (rel <= TOL).all() and zdev < 1e-12



@pytest.mark.parametrize("N,B", [(32, 200), (64, 60)])
def test_strict_z_switches_the_level_rule_off(vigo_handle, small_world, N, B):
    """vigo_params_t.strict_z = 1: no level rule — the reference's arithmetic on the z axis whatever the input (the general
    kernel alone, as before round 3).  Bit-identical to the emulation oracle under the same switch; the z of a trajectory
    that is level only to rounding now drifts by that rounding noise, exactly as in the reference-order oracle's run."""
    v = vigo_handle
    P = default_params()
    P.max_iterations = 50
    P.strict_z = 1
    v.set_params(P)
    b = synth.make_bspline_batch(small_world, B, N, 990 + N, start_range=3.0)
    r, d = solve_both(v, P, b)
    cost, grad, terms = v.cost_grad(**d)
    with emulation(N):
        e = ol.optimize_batch(P, b)
        ce, ge, te = ol.cost_grad_batch(P, b)
    g = {k: getattr(r, k).cpu().numpy() for k in ("ctrl", "x", "status", "fx", "iters", "evals")}
    assert np.array_equal(cost.cpu().numpy(), ce) and np.array_equal(grad.cpu().numpy(), ge)
    for k in ("status", "iters", "evals", "x", "ctrl", "fx"):
        assert np.array_equal(g[k], e[k]), f"{k} differs from the emulation-mode oracle"
    drift = np.abs(g["ctrl"][:, 3:-3, 2] - b.ctrl[:, 3:-3, 2]).max(1)
    assert (drift > 0).mean() > 0.5 and drift.max() < 1e-11          # z moves, by rounding noise
    P.strict_z = 0
    v.set_params(P)
    r2, _ = solve_both(v, P, b)
    assert np.array_equal(r2.ctrl.cpu().numpy()[:, :, 2], b.ctrl[:, :, 2])     # the rule back on: z untouched
    rel = rel_err_per_traj(r2.ctrl.cpu().numpy(), g["ctrl"])
    print(f"\n[strict_z N={N}] z drift without the rule: max {drift.max():.2e}; with vs without the rule: max rel. difference of control points {rel.max():.2e}")
    assert rel.max() < 1e-6


def test_parameter_variations_stay_bit_exact(vigo_handle, small_world):
    """Paths of the kernel the default batch does not reach: more obstacles than the LDS cache holds (20 > 16:
    the rest is read from HBM/L2), three or more guide pairs on a control point (beyond the two kept in
    registers), plan_in_z with the height term and its reproduced quirks, an unknown-space factor != 1,
    per-trajectory weights as the rebound loop produces them, a short history and early convergence."""
    v = vigo_handle
    rng = np.random.default_rng(77)
    P = default_params()
    P.max_iterations = 40
    P.mem_size = 5
    P.plan_in_z = 1
    P.uncertain_factor = 1.7
    P.dthresh = 0.6
    P.dist_thresh_dynamic = 0.7
    P.g_epsilon = 0.05
    v.set_params(P)
    b = synth.make_bspline_batch(small_world, 96, 32, 321, start_range=3.0, n_obs=20, guide2_prob=0.9)
    # pile extra guide pairs onto the points that already have two (CSR rebuilt)
    cnt = np.diff(b.guide_off)
    new_pv, new_unk, new_off = [], [], [0]
    for i, c in enumerate(cnt):
        pv = b.guide_pv[b.guide_off[i]:b.guide_off[i + 1]]
        unk = b.guide_unk[b.guide_off[i]:b.guide_off[i + 1]]
        if c == 2:
            extra = pv[[0, 1, 0]] + rng.normal(0, 0.05, size=(3, 6)) * [1, 1, 1, 0, 0, 0]
            pv = np.concatenate([pv, extra])
            unk = np.concatenate([unk, [1, 0, 1]]).astype(np.uint8)
        new_pv.append(pv)
        new_unk.append(unk)
        new_off.append(new_off[-1] + len(pv))
    b2 = synth.Batch(b.ctrl + rng.normal(0, 0.02, size=b.ctrl.shape) * [0, 0, 1], np.array(new_off, dtype=np.int32), np.concatenate(new_pv),
                     np.concatenate(new_unk).astype(np.uint8), b.obs_off, b.obs)
    assert np.diff(b2.guide_off).max() >= 5 and np.diff(b2.obs_off).max() == 20
    w = np.ones((b2.B, 4)) * rng.choice([1.0, 2.0, 4.0, 8.0], size=(b2.B, 4))
    d = batch_to_dev(b2, v.device, w)
    r = v.optimize(**d)
    g = {k: getattr(r, k).cpu().numpy() for k in ("ctrl", "x", "status", "fx", "iters", "evals")}
    with emulation(32):
        e = ol.optimize_batch(P, b2, w)
    for k in ("status", "iters", "evals", "x", "ctrl", "fx"):
        assert np.array_equal(g[k], e[k], equal_nan=True), f"{k} differs from the emulation-mode oracle"
    cost, grad, terms = v.cost_grad(**d)
    with emulation(32):
        ce, ge, te = ol.cost_grad_batch(P, b2, w)
    assert np.array_equal(cost.cpu().numpy(), ce) and np.array_equal(grad.cpu().numpy(), ge) and np.array_equal(terms.cpu().numpy(), te)
    assert (terms.cpu().numpy()[:, 3] > 0).any() and len(np.unique(g["status"])) >= 2


@pytest.mark.parametrize("scale", [1e-25, 1e-40, 1e-80, 1e-120, float("nan")])
def test_two_loop_division_fallback_is_exact(vigo_handle, small_world, scale):
    """The steady-state two-loop divides by ys with Markstein's exact reciprocal sequence, proven for operands
    within 2^+-500; beyond that (or on a NaN) it repeats the recursion with true divisions.  Control points
    scaled by 1e-25 / 1e-40 keep the dividends (~ scale^2) inside that range (the reciprocal path runs), 1e-80 /
    1e-120 put them far below 2^-500 (the fallback runs) while some of the 40 solves still run all 40
    iterations, i.e. through the steady-state path with a full history; a NaN control point covers the other
    trigger.  Results stay bit-identical to the emulation-mode oracle, which always divides."""
    v = vigo_handle
    P = default_params()
    P.max_iterations = 40
    P.g_epsilon = 0.0
    v.set_params(P)
    b = synth.make_bspline_batch(small_world, 40, 32, 4711, start_range=3.0)
    rng = np.random.default_rng(3)
    ctrl = b.ctrl + rng.normal(0, 0.05, size=b.ctrl.shape)
    if np.isnan(scale):
        ctrl[7, 10, 1] = np.nan
        nb = synth.Batch(ctrl, b.guide_off, b.guide_pv, b.guide_unk)
    else:
        ctrl = ctrl * scale                                                  # no guides: keeps the problem scale-free
        nb = synth.Batch(ctrl, np.zeros(b.B * b.N + 1, dtype=np.int32), np.zeros((0, 6)), np.zeros(0, dtype=np.uint8))
    r = v.optimize(**batch_to_dev(nb, v.device))
    g = {k: getattr(r, k).cpu().numpy() for k in ("ctrl", "x", "status", "fx", "iters", "evals")}
    with emulation(32):
        e = ol.optimize_batch(P, nb)
    for k in ("status", "iters", "evals", "x", "ctrl", "fx"):
        assert np.array_equal(g[k], e[k], equal_nan=True), f"{k} differs from the emulation-mode oracle at scale {scale}"
    assert (g["iters"] > 20).any()                                            # the history did fill up: the steady path ran
