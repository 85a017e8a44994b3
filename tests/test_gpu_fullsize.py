"""-m gpu: the BASELINE.json configs that round 1 only covered at reduced size, at their STATED sizes.

  configs[2]  4096 segments x 10 000 samples through vigo_corridor_check on the maze fixture (map/maze.bt as
              tests/golden/maze_config1.npz), box [0.4, 0.4, 0.2], step 0.2: flag / first colliding sample / count
              against the C oracle on 256 of the segments at full sample count, determinism and count consistency
              on all 4096;
  sampler     >= 1e6 (segment, t) pairs of vigo_poly_sample (the sampler inside the checker, PS.cpp:1026-1056)
              bit for bit against the oracle;
  configs[4]  1 048 576 trilinear ESDF queries against a 256^3 lattice (uniform and brick-sorted), every one
              compared with the oracle; vigo_optimize 1024 x 32 with 8 dynamic obstacles per trajectory against both
              oracle modes.
"""
import os

import numpy as np
import pytest
import torch

import oracle_lib as ol
from gpu_util import batch_to_dev, emulation, rel_err_per_traj, to_dev
from trajectory_planner_amd import synth
from trajectory_planner_amd.vigo import default_params

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BOX = np.array([0.4, 0.4, 0.2])     # cfg/planner_interactive.yaml collision_box
STEP = 0.2                          # map_resolution


def maze_world():
    f = np.load(os.path.join(ROOT, "tests", "golden", "maze_config1.npz"))
    nx, ny, nz = (int(v) for v in f["dims"])
    n = nx * ny * nz
    occ = np.unpackbits(f["occ_bits"])[:n].reshape(nx, ny, nz)
    unk = np.unpackbits(f["unk_bits"])[:n].reshape(nx, ny, nz)
    vox = (occ * 5 + unk * 2).astype(np.uint8)
    return synth.World(vox, f["origin"].astype(np.float64), float(f["res"][0]), np.zeros((0, 6)))


def test_config3_corridor_checker_at_full_size(vigo_handle):
    """BASELINE configs[2]: 4096 segments x 10 000 samples on the maze map."""
    v = vigo_handle
    w = maze_world()
    v.set_grid(to_dev(w.voxels, v.device), w.origin, w.res)
    S, NS = 4096, 10000
    # segments inside the maze's extents (x [-16.9, 5.0], y [-16.7, 3.8], z [-1.1, 3.0])
    coeffs, n_samp, delT, dur = synth.make_corridor_segments(303, S, extent_lo=(-15.0, -15.0, 0.2), extent_hi=(3.5, 2.5, 2.2), n_samples=NS)
    d_c, d_n, d_t = to_dev(coeffs, v.device), to_dev(n_samp, v.device), to_dev(delT, v.device)
    flag, first, count = (x.cpu().numpy() for x in v.corridor_check(d_c, d_n, d_t, BOX, STEP))
    flag2, first2, count2 = (x.cpu().numpy() for x in v.corridor_check(d_c, d_n, d_t, BOX, STEP))
    assert np.array_equal(flag, flag2) and np.array_equal(first, first2) and np.array_equal(count, count2)
    # consistency of the three outputs on every segment
    assert np.array_equal(flag != 0, count > 0)
    assert np.array_equal(first >= 0, count > 0) and (first < NS).all() and (count <= NS).all()
    assert (first[count > 0] + count[count > 0] <= NS).all()
    assert 0.05 < flag.mean() < 0.95, flag.mean()
    # the oracle (PO.cpp:547-589, :634-656 + PS.cpp:1026-1056) on 256 segments, all 10 000 samples each (2.56 M box sweeps)
    pick = np.sort(np.random.default_rng(1).choice(S, 256, replace=False))
    g, keep = ol.make_grid(w)
    with ol.pow_mode(True):       # pow(t, d) as the correctly rounded power: what the device implements
        f_o, fi_o, cn_o = ol.corridor_check_batch(g, coeffs[pick], n_samp[pick], delT[pick], BOX, STEP)
    assert np.array_equal(f_o, flag[pick]) and np.array_equal(fi_o, first[pick]) and np.array_equal(cn_o, count[pick])
    with ol.pow_mode(False):      # libm's pow (the reference as built on this host): same indices on these inputs
        f_l, fi_l, cn_l = ol.corridor_check_batch(g, coeffs[pick[:64]], n_samp[pick[:64]], delT[pick[:64]], BOX, STEP)
    assert np.array_equal(f_l, flag[pick[:64]]) and np.array_equal(fi_l, first[pick[:64]]) and np.array_equal(cn_l, count[pick[:64]])
    print(f"\nconfig 3: {S} segments x {NS} samples, {int(flag.sum())} colliding; 256 segments == oracle (flag, first, count)")


def test_sampler_positions_bit_exact_on_a_million_pairs(vigo_handle):
    """vigo_poly_sample == oracle on 128 x 8192 = 1 048 576 (segment, t) pairs: fp64 positions bit for bit with the
    correctly-rounded-pow oracle, and the float positions (what the box sweep consumes) also with the libm-pow oracle"""
    v = vigo_handle
    S, NS = 128, 8192
    coeffs, n_samp, delT, dur = synth.make_corridor_segments(77, S, extent_lo=(-15.0, -15.0, 0.2), extent_hi=(3.5, 2.5, 2.2), n_samples=NS)
    delT[:16] = 0.1 / 16                                   # the reference's sample_delta_time scale, long clocks
    coeffs[16:24] *= 37.0                                  # larger magnitudes
    # strongly cancelling terms (|x| << sum |c_d| t^d around the zero crossing): the float filter of the kernels cannot
    # certify many of these samples and falls back to the exact-power chain
    coeffs[24:32, :, 0] = 3000.0
    coeffs[24:32, :, 1] = -6000.0 / dur[24:32, None]
    coeffs[32:34, 0, 3] = np.inf                           # non-finite coefficients: nothing certifies
    coeffs[34:36, 1, 2] = np.nan
    p64, p32 = v.poly_sample(to_dev(coeffs, v.device), to_dev(n_samp, v.device), to_dev(delT, v.device), NS, want_f64=True, want_f32=True)
    p64, p32 = p64.cpu().numpy(), p32.cpu().numpy()
    with ol.pow_mode(True):
        ref = ol.poly_sample(coeffs, n_samp, delT, NS)
    assert np.array_equal(p64, ref, equal_nan=True)
    assert np.array_equal(p32, ref.astype(np.float32), equal_nan=True)
    with ol.pow_mode(False):
        ref_libm = ol.poly_sample(coeffs, n_samp, delT, NS)
    fin = np.isfinite(ref).all(axis=2)
    differ = (ref_libm != ref).any(axis=2)[fin].mean()
    print(f"\nsampler: {S * NS} positions bit-identical to the exact-pow oracle; libm-pow positions differ in the last bit for "
          f"{differ * 100:.3f} % of them (glibc's own rounding)")
    assert differ < 0.05
    assert np.array_equal(p32, ref_libm.astype(np.float32), equal_nan=True)
    # other degrees go through the generic path
    for deg in (3, 9, 15):
        c2, n2, t2, _ = synth.make_corridor_segments(80 + deg, 16, deg=deg, n_samples=2048)
        q64, _ = v.poly_sample(to_dev(c2, v.device), to_dev(n2, v.device), to_dev(t2, v.device), 2048)
        with ol.pow_mode(True):
            assert np.array_equal(q64.cpu().numpy(), ol.poly_sample(c2, n2, t2, 2048)), deg


def test_config5_esdf_queries_at_full_size(vigo_handle):
    """BASELINE configs[4], second half: 1 048 576 trilinear queries against the 256^3 fp32 ESDF of the config-2
    occupancy (exact EDT), uniform and brick-sorted, every query bit for bit against the oracle."""
    v = vigo_handle
    world = synth.make_box_world(synth.SEED_BASE + 2, n=256, n_boxes=200)
    dist, origin = synth.edt_esdf(world)
    v.set_esdf(to_dev(dist, v.device), origin, world.res)
    rng = np.random.default_rng(5)
    Q = 1 << 20
    pts = rng.uniform(-12.9, 12.9, size=(Q, 3))          # a little beyond the lattice: clamped queries included
    d_ref, g_ref = ol.esdf_query_batch(dist, origin, world.res, pts)
    d, g = v.esdf_query(to_dev(pts, v.device))
    assert np.array_equal(d.cpu().numpy(), d_ref) and np.array_equal(g.cpu().numpy(), g_ref)
    brick = np.floor((pts - origin) / (4 * world.res)).astype(np.int64)
    order = np.lexsort((brick[:, 2], brick[:, 1], brick[:, 0]))
    ds, gs = v.esdf_query(to_dev(pts[order], v.device))
    assert np.array_equal(ds.cpu().numpy(), d_ref[order]) and np.array_equal(gs.cpu().numpy(), g_ref[order])
    assert np.isfinite(d_ref).all() and np.isfinite(g_ref).all()
    far = d_ref > 1.0                                    # away from the surfaces the EDT is smooth: |grad| ~ 1
    assert abs(np.median(np.linalg.norm(g_ref[far], axis=1)) - 1.0) < 0.05
    # the same 1 048 576 queries at the fp32 I/O width of SURVEY.md §8(d) config 5 (12 B in, 16 B out; vigo_esdf_query_f32):
    # every value and gradient bit for bit against the oracle's fp32 twin, both query orders; and fp32 rounding is all
    # that separates the two entries
    p32 = pts.astype(np.float32)
    ref32 = ol.esdf_query_f32_batch(dist, origin, world.res, p32)
    got32 = v.esdf_query_f32(to_dev(p32, v.device)).cpu().numpy()
    assert got32.dtype == np.float32 and np.array_equal(got32.view(np.uint32), ref32.view(np.uint32))
    got32s = v.esdf_query_f32(to_dev(p32[order], v.device)).cpu().numpy()
    assert np.array_equal(got32s.view(np.uint32), ref32[order].view(np.uint32))
    # (the value is continuous across cell borders: every query agrees; the gradient of a trilinear interpolant is not, so
    # the few points that fp32 rounds into the neighbouring cell see that cell's slope)
    gdiff = np.abs(got32[:, 1:] - g_ref).max(1)
    print(f"\nfp32 entry vs fp64 entry: value max diff {np.abs(got32[:, 0] - d_ref).max():.2e}; gradient diff median {np.median(gdiff):.2e}, "
          f"p99.9 {np.quantile(gdiff, 0.999):.2e}, {(gdiff > 1e-3).sum()} of {Q} queries beyond 1e-3 (cell-border points)")
    assert np.abs(got32[:, 0] - d_ref).max() < 2e-4 and np.quantile(gdiff, 0.999) < 1e-3 and (gdiff > 1e-3).mean() < 1e-4


def test_config5_dynamic_obstacle_term_at_full_size(vigo_handle):
    """BASELINE configs[4], first half: the dynamic-obstacle term (BT.cpp:1001-1064) with O = 8 obstacles per
    trajectory on the config-2 batch (1024 x 32, 256^3, 50 iterations): bit-exact vs the emulation-mode oracle,
    every trajectory within 1e-4 of the reference-order oracle."""
    v = vigo_handle
    world = synth.make_box_world(synth.SEED_BASE + 2, n=256, n_boxes=200)
    b = synth.make_bspline_batch(world, 1024, 32, synth.SEED_BASE + 5 + 1000, n_obs=8)
    P = default_params()
    P.max_iterations = 50
    v.set_params(P)
    v.set_grid(to_dev(world.voxels, v.device), world.origin, world.res)
    d = batch_to_dev(b, v.device)
    c, gr, terms = v.cost_grad(**d)
    with emulation(32):
        ce, ge, te = ol.cost_grad_batch(P, b)
        e = ol.optimize_batch(P, b)
    assert np.array_equal(c.cpu().numpy(), ce) and np.array_equal(gr.cpu().numpy(), ge) and np.array_equal(terms.cpu().numpy(), te)
    assert (te[:, 3] > 0).mean() > 0.5                  # the obstacle term is active on most trajectories
    r = v.optimize(**d)
    got = {k: getattr(r, k).cpu().numpy() for k in ("ctrl", "x", "status", "fx", "iters", "evals")}
    for k in ("status", "iters", "evals", "x", "ctrl", "fx"):
        assert np.array_equal(got[k], e[k]), f"{k} differs from the emulation-mode oracle"
    ref = ol.optimize_batch(P, b)
    rel = rel_err_per_traj(got["ctrl"], ref["ctrl"])
    print(f"\nconfig 5a (1024 x 32, 8 obstacles): vs reference-order oracle median {np.median(rel):.2e} p99 {np.quantile(rel, .99):.2e} "
          f"max {rel.max():.2e}; within 1e-4: {(rel <= 1e-4).mean() * 100:.2f} %")
    # measured: 100.00 % of the 1024 trajectories inside 1e-4 (worst 2.1e-6) — demanded, not a quantile
    assert (rel <= 1e-4).all() and np.median(rel) < 1e-8, f"{(rel > 1e-4).sum()} trajectories outside 1e-4, max {rel.max():.3e}"
