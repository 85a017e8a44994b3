"""-m gpu: the C ABI's error behaviour, through raw ctypes (the Python wrapper's own checks bypassed): a NULL
handle, NULL required pointers, negative or zero counts and out-of-range enumerators must come back as a
negative vigo_status_t — never a crash, never a launch on garbage — and must leave the handle usable.
The reference's classes signal errors by `false` / early return (bsplineTraj.cpp:333-343, polyTrajOctomap.cpp:
226-238); the facades translate these codes into that behaviour."""
import ctypes as C

import numpy as np
import pytest
import torch

from trajectory_planner_amd import _lib

pytestmark = pytest.mark.gpu
NULL = C.c_void_p(0)


@pytest.fixture(scope="module")
def raw():
    L = _lib.load()
    h = C.c_void_p()
    assert L.vigo_create(C.byref(h), 0) == 0
    yield L, h
    assert L.vigo_destroy(h) == 0


def dptr(t):
    return C.c_void_p(t.data_ptr())


def test_null_handle_is_refused_everywhere(raw):
    L, _ = raw
    d = torch.zeros(64, dtype=torch.float64, device="cuda")
    o = torch.zeros(64, dtype=torch.uint8, device="cuda")
    three = (C.c_double * 3)(0, 0, 0)
    calls = [
        lambda: L.vigo_set_stream(NULL, NULL),
        lambda: L.vigo_set_params(NULL, None),
        lambda: L.vigo_get_params(NULL, None),
        lambda: L.vigo_set_precision(NULL, 0),
        lambda: L.vigo_set_grid(NULL, 4, 4, 4, three, 0.1, dptr(o)),
        lambda: L.vigo_inflate_grid(NULL, 4, 4, 4, dptr(o), 1, 1, 1),
        lambda: L.vigo_pack_grid(NULL, 4, 4, 4, dptr(o), dptr(d)),
        lambda: L.vigo_set_metric_bounds(NULL, three, three),
        lambda: L.vigo_query_points(NULL, 0, 1, dptr(d), dptr(o)),
        lambda: L.vigo_guides_unknown(NULL, 1, dptr(d), dptr(o)),
        lambda: L.vigo_bspline_fit(NULL, 1, 8, 0.1, dptr(d), NULL, dptr(d)),
        lambda: L.vigo_ctrl_occupancy(NULL, 1, 8, dptr(d), dptr(o), dptr(o)),
        lambda: L.vigo_box_collision_points(NULL, 1, dptr(d), three, 0.1, dptr(o)),
        lambda: L.vigo_esdf_query(NULL, 1, dptr(d), dptr(d), dptr(d)),
        lambda: L.vigo_poly_sample(NULL, 1, 7, dptr(d), dptr(d), dptr(d), 4, dptr(d), NULL),
        lambda: L.vigo_rebound_rounds(NULL, 1, 8, dptr(d), NULL, NULL, NULL, NULL, NULL, 0, dptr(d), 0.05, 0.0, 1, dptr(d)),
        lambda: L.vigo_destroy(NULL),
    ]
    for i, f in enumerate(calls):
        assert f() < 0, i
    L.vigo_last_error(NULL)                            # a static message or NULL, not a crash


def test_new_entry_points_refuse_bad_arguments(raw):
    """vigo_poly_sample / vigo_rebound_rounds: negative counts, NULL arrays, out-of-range rounds and ratios are
    refused on the host (negative status, nothing launched); the host utilities answer NaN outside their domain"""
    import math
    L, h = raw
    d = torch.zeros(4096, dtype=torch.float64, device="cuda")
    i32 = torch.zeros(4096, dtype=torch.int32, device="cuda")
    assert L.vigo_poly_sample(h, -1, 7, dptr(d), dptr(i32), dptr(d), 4, dptr(d), NULL) < 0
    assert L.vigo_poly_sample(h, 2, 16, dptr(d), dptr(i32), dptr(d), 4, dptr(d), NULL) < 0         # degree > 15
    assert L.vigo_poly_sample(h, 2, 7, NULL, dptr(i32), dptr(d), 4, dptr(d), NULL) < 0
    assert L.vigo_poly_sample(h, 2, 7, dptr(d), dptr(i32), dptr(d), 4, NULL, NULL) < 0              # no output at all
    assert L.vigo_poly_sample(h, 0, 7, NULL, NULL, NULL, 0, NULL, NULL) == 0                        # empty is fine
    three = (C.c_double * 3)(0, 0, 0)
    vox = torch.zeros(8 * 8 * 40, dtype=torch.uint8, device="cuda")
    assert L.vigo_set_grid(h, 8, 8, 40, three, 0.1, dptr(vox)) == 0
    args = lambda **kw: [kw.get("B", 2), kw.get("N", 32), dptr(d), NULL, NULL, NULL, NULL, NULL, 0, kw.get("w", dptr(d)),
                         kw.get("dt", 0.05), kw.get("ncr", 0.0), kw.get("rounds", 2), kw.get("state", dptr(i32))]
    assert L.vigo_rebound_rounds(h, *args(w=NULL)) < 0
    assert L.vigo_rebound_rounds(h, *args(state=NULL)) < 0
    assert L.vigo_rebound_rounds(h, *args(rounds=-1)) < 0 and L.vigo_rebound_rounds(h, *args(rounds=65)) < 0
    assert L.vigo_rebound_rounds(h, *args(ncr=1.0)) < 0 and L.vigo_rebound_rounds(h, *args(ncr=float("nan"))) < 0
    assert L.vigo_rebound_rounds(h, *args(dt=0.0)) < 0 and L.vigo_rebound_rounds(h, *args(dt=float("nan"))) < 0
    assert L.vigo_rebound_rounds(h, *args(N=6)) < 0 and L.vigo_rebound_rounds(h, *args(B=-1)) < 0
    assert L.vigo_rebound_rounds(h, *args(B=0)) == 0
    # state entries with a status the call does not know, a negative failCount or a segment count beyond the array:
    # treated as "not active" / handed to the host, never indexed out of range
    st = torch.zeros(4, 104, dtype=torch.int32, device="cuda")
    st[0, 0] = 7
    st[1, 7] = 1000
    st[2, 2] = -5
    st[:, 1] = 1
    ctrl = torch.randn(4, 32, 3, dtype=torch.float64, device="cuda")
    w = torch.ones(4, 4, dtype=torch.float64, device="cuda")
    assert L.vigo_rebound_rounds(h, 4, 32, dptr(ctrl), NULL, NULL, NULL, NULL, NULL, 0, dptr(w), 0.05, 0.0, 3, dptr(st)) == 0
    torch.cuda.synchronize()
    assert int(st[0, 0]) == 7 and bool(torch.isfinite(w).all())
    assert math.isnan(L.vigo_exact_pow(2.0, -1)) and math.isnan(L.vigo_exact_pow(2.0, 16)) and L.vigo_exact_pow(2.0, 10) == 1024.0


def test_bad_arguments_return_codes_and_keep_the_handle_usable(raw):
    L, h = raw
    dev = torch.device("cuda", 0)
    vox = torch.zeros(8, 8, 40, dtype=torch.uint8, device=dev)
    origin = (C.c_double * 3)(0, 0, 0)
    d = torch.zeros(4096, dtype=torch.float64, device=dev)
    o = torch.zeros(4096, dtype=torch.uint8, device=dev)
    i32 = torch.zeros(4096, dtype=torch.int32, device=dev)
    # map entry points: before a grid, with bad dims, with NULL data
    assert L.vigo_set_grid(h, 0, 8, 8, origin, 0.1, dptr(vox)) < 0
    assert L.vigo_set_grid(h, 8, 8, -1, origin, 0.1, dptr(vox)) < 0
    assert L.vigo_set_grid(h, 8, 8, 40, origin, 0.0, dptr(vox)) < 0
    assert L.vigo_set_grid(h, 8, 8, 40, origin, float("nan"), dptr(vox)) < 0
    assert L.vigo_set_grid(h, 8, 8, 40, origin, 0.1, NULL) < 0
    assert L.vigo_set_grid(h, 8, 8, 40, None, 0.1, dptr(vox)) < 0
    assert L.vigo_inflate_grid(h, 8, 8, 40, NULL, 1, 1, 1) < 0
    assert L.vigo_inflate_grid(h, 8, 8, 40, dptr(vox), -1, 0, 0) < 0
    assert L.vigo_pack_grid(h, 8, 8, 40, NULL, dptr(d)) < 0
    assert L.vigo_pack_grid(h, 8, 8, 40, dptr(vox), NULL) < 0
    assert L.vigo_set_grid(h, 8, 8, 40, origin, 0.1, dptr(vox)) == 0
    assert L.vigo_query_points(h, 0, 4, NULL, dptr(o)) < 0
    assert L.vigo_query_points(h, 0, 4, dptr(d), NULL) < 0
    assert L.vigo_query_points(h, 7, 4, dptr(d), dptr(o)) < 0          # no such plane
    assert L.vigo_query_points(h, 0, -4, dptr(d), dptr(o)) < 0
    assert L.vigo_query_points(h, 0, 0, NULL, NULL) == 0                 # empty is fine
    assert L.vigo_guides_unknown(h, 3, NULL, dptr(o)) < 0
    # solver entry points
    for B, N, ctrl in ((1, 32, NULL), (-1, 32, dptr(d)), (1, 6, dptr(d)), (1, 100000, dptr(d))):
        assert L.vigo_optimize(h, B, N, ctrl, NULL, NULL, NULL, NULL, NULL, 0, NULL, NULL, NULL, NULL, NULL, NULL) < 0, (B, N)
        assert L.vigo_cost_grad(h, B, N, ctrl, NULL, NULL, NULL, NULL, NULL, 0, NULL, dptr(d), dptr(d), NULL) < 0, (B, N)
    # offsets (or a shared count) without the list they index mean "none": the kernels must not touch the
    # missing list.  The offsets here claim 5 pairs / obstacles per entry, so a dereference would fault.
    i32[:] = torch.arange(4096, dtype=torch.int32, device=dev) * 5
    ok_ctrl = torch.zeros(1, 32, 3, dtype=torch.float64, device=dev)
    ok_ctrl[0, :, 0] = torch.arange(32, dtype=torch.float64, device=dev) * 0.2
    base = ok_ctrl.clone()
    assert L.vigo_optimize(h, 1, 32, dptr(base), NULL, NULL, NULL, NULL, NULL, 0, NULL, NULL, NULL, NULL, NULL, NULL) == 0
    for goff, ooff, ns in ((dptr(i32), NULL, 0), (NULL, dptr(i32), 0), (NULL, NULL, 3), (dptr(i32), dptr(i32), 7)):
        c = ok_ctrl.clone()
        assert L.vigo_optimize(h, 1, 32, dptr(c), goff, NULL, NULL, ooff, NULL, ns, NULL, NULL, NULL, NULL, NULL, NULL) == 0
        assert torch.equal(c, base)                      # same result as with no lists at all
        assert L.vigo_cost_grad(h, 1, 32, dptr(c), goff, NULL, NULL, ooff, NULL, ns, NULL, dptr(d), dptr(d), NULL) == 0
    assert L.vigo_optimize(h, 1, 32, dptr(d), NULL, NULL, NULL, NULL, dptr(d), -3, NULL, NULL, NULL, NULL, NULL, NULL) < 0
    assert L.vigo_traj_dynamic_collision(h, 1, 8, dptr(d), 0.05, dptr(i32), NULL, 0, dptr(o)) == 0
    assert L.vigo_traj_dynamic_collision(h, 1, 8, dptr(d), 0.05, NULL, NULL, 2, dptr(o)) == 0
    torch.cuda.synchronize()
    i32.zero_()
    assert L.vigo_optimize(h, 0, 32, NULL, NULL, NULL, NULL, NULL, NULL, 0, NULL, NULL, NULL, NULL, NULL, NULL) == 0   # empty batch
    from trajectory_planner_amd.vigo import default_params
    for field, val in (("ts_ctrl", float("inf")), ("ts", float("inf")), ("ts_ctrl", float("nan")), ("max_linesearch", 2**31 - 1),
                       ("max_iterations", 2**31 - 1), ("mem_size", 0), ("mem_size", 17), ("past", 1), ("pred_horizon", 0.0)):
        Pb = default_params()
        setattr(Pb, field, val)
        assert L.vigo_set_params(h, C.byref(Pb)) < 0, field      # refused: the handle keeps its previous parameters
    assert L.vigo_set_precision(h, 17) < 0
    assert L.vigo_set_params(h, None) < 0
    # spline / gates
    assert L.vigo_bspline_fit(h, 1, 1, 0.1, dptr(d), NULL, dptr(d)) < 0                 # fewer than 2 waypoints
    assert L.vigo_bspline_fit(h, 1, 8, 0.0, dptr(d), NULL, dptr(d)) < 0
    assert L.vigo_bspline_fit(h, 1, 8, 0.1, NULL, NULL, dptr(d)) < 0
    assert L.vigo_bspline_fit(h, 1, 8, 0.1, dptr(d), NULL, NULL) < 0
    assert L.vigo_bspline_eval(h, 1, 2, dptr(d), 0, 4, dptr(d), dptr(d)) < 0            # N < degree + 1
    assert L.vigo_bspline_eval(h, 1, 8, dptr(d), 5, 4, dptr(d), dptr(d)) < 0            # no such derivative
    assert L.vigo_bspline_eval(h, 1, 8, NULL, 0, 4, dptr(d), dptr(d)) < 0
    assert L.vigo_traj_collision(h, 1, 8, dptr(d), 0.0, dptr(o), NULL) < 0               # dt must be positive
    assert L.vigo_traj_collision(h, 1, 8, dptr(d), -1.0, dptr(o), NULL) < 0
    assert L.vigo_traj_collision(h, 1, 8, dptr(d), float("nan"), dptr(o), NULL) < 0
    assert L.vigo_traj_collision(h, 1, 8, NULL, 0.05, dptr(o), NULL) < 0
    assert L.vigo_traj_collision(h, 1, 8, dptr(d), 0.05, NULL, NULL) < 0
    assert L.vigo_traj_dynamic_collision(h, 1, 8, dptr(d), 0.05, NULL, dptr(d), -1, dptr(o)) < 0
    assert L.vigo_ctrl_occupancy(h, 1, 8, NULL, dptr(o), dptr(o)) < 0
    # min-snap / corridor
    assert L.vigo_minsnap(h, 1, 1, 7, 4, 4, 1.0, 8.0, dptr(d), NULL, NULL, dptr(d), dptr(d), dptr(i32)) < 0     # one waypoint
    assert L.vigo_minsnap(h, 1, 4, 7, 4, 4, 0.0, 8.0, dptr(d), NULL, NULL, dptr(d), dptr(d), dptr(i32)) < 0     # zero speed
    assert L.vigo_minsnap(h, 1, 4, 7, 4, 4, 1.0, 8.0, NULL, NULL, NULL, dptr(d), dptr(d), dptr(i32)) < 0
    assert L.vigo_minsnap(h, 1, 4, 7, 4, 4, 1.0, 8.0, dptr(d), NULL, NULL, NULL, dptr(d), dptr(i32)) < 0
    box = (C.c_double * 3)(0.4, 0.4, 0.2)
    assert L.vigo_corridor_check(h, 1, 7, NULL, dptr(i32), dptr(d), box, 0.1, dptr(o), NULL, NULL) < 0
    assert L.vigo_corridor_check(h, 1, 99, dptr(d), dptr(i32), dptr(d), box, 0.1, dptr(o), NULL, NULL) < 0
    assert L.vigo_corridor_check(h, 1, 7, dptr(d), dptr(i32), dptr(d), box, 0.0, dptr(o), NULL, NULL) < 0
    assert L.vigo_corridor_check(h, 1, 7, dptr(d), dptr(i32), dptr(d), None, 0.1, dptr(o), NULL, NULL) < 0
    for bad_box in ((float("nan"), 0.4, 0.2), (float("inf"), 0.4, 0.2), (1e9, 1e9, 1e9), (100.0, 100.0, 100.0)):
        bb = (C.c_double * 3)(*bad_box)                      # an unbounded (or absurd) lattice per pose is refused
        assert L.vigo_corridor_check(h, 1, 7, dptr(d), dptr(i32), dptr(d), bb, 0.1, dptr(o), NULL, NULL) < 0
        assert L.vigo_box_collision_points(h, 4, dptr(d), bb, 0.1, dptr(o)) < 0
    assert L.vigo_box_collision_points(h, 4, NULL, box, 0.1, dptr(o)) < 0
    assert L.vigo_box_collision_points(h, 4, dptr(d), box, -0.1, dptr(o)) < 0
    # ESDF
    assert L.vigo_esdf_query(h, 4, dptr(d), dptr(d), dptr(d)) < 0                        # before vigo_set_esdf
    f32 = torch.zeros(64, dtype=torch.float32, device=dev)
    assert L.vigo_set_esdf(h, 1, 4, 4, origin, 0.1, dptr(f32)) < 0                       # a trilinear cell needs 2 samples per axis
    assert L.vigo_set_esdf(h, 4, 4, 4, origin, 0.1, NULL) < 0
    assert L.vigo_set_esdf(h, 4, 4, 4, origin, 0.1, dptr(f32)) == 0
    assert L.vigo_esdf_query(h, 4, NULL, dptr(d), dptr(d)) < 0
    assert L.vigo_esdf_query_f32(h, 4, NULL, dptr(f32)) < 0 and L.vigo_esdf_query_f32(h, 4, dptr(f32), NULL) < 0
    assert L.vigo_esdf_query_f32(h, -1, dptr(f32), dptr(f32)) < 0
    assert L.vigo_esdf_query_f32(h, 1, dptr(f32), C.c_void_p(f32.data_ptr() + 4)) < 0           # the 16-byte store needs its alignment
    torch.cuda.synchronize()
    # every refusal left a message, and the handle still works
    assert len(L.vigo_last_error(h)) > 0
    pts = torch.zeros(4, 3, dtype=torch.float64, device=dev)
    out = torch.full((4,), 9, dtype=torch.uint8, device=dev)
    assert L.vigo_query_points(h, 0, 4, dptr(pts), dptr(out)) == 0
    assert out.cpu().tolist() == [0, 0, 0, 0]
    ctrl = torch.zeros(2, 32, 3, dtype=torch.float64, device=dev)
    ctrl[:, :, 0] = torch.arange(32, dtype=torch.float64, device=dev) * 0.01
    st = torch.full((2,), 77, dtype=torch.int32, device=dev)
    assert L.vigo_optimize(h, 2, 32, dptr(ctrl), NULL, NULL, NULL, NULL, NULL, 0, NULL, NULL, dptr(st), NULL, NULL, NULL) == 0
    torch.cuda.synchronize()
    assert (st.cpu().numpy() != 77).all() and bool(torch.isfinite(ctrl).all())


def test_extreme_values_in_device_data_do_not_fault(raw):
    """NaN, +-inf, +-1e300 and denormals inside the DEVICE arrays (which the host cannot validate): every kernel
    must run to completion with in-range indexing — positions that are not finite or not in the map count as
    outside (occupied / unknown, the contract of include/vigo.h), solver statuses are lbfgs error codes."""
    L, h = raw
    dev = torch.device("cuda", 0)
    vox = torch.zeros(16, 16, 40, dtype=torch.uint8, device=dev)
    origin = (C.c_double * 3)(0, 0, 0)
    assert L.vigo_set_grid(h, 16, 16, 40, origin, 0.1, dptr(vox)) == 0
    bad = [float("nan"), float("inf"), -float("inf"), 1e300, -1e300, 5e-324, -2.5e9, 2147483648.0 * 0.1, 0.5]
    pts = torch.tensor([[a, b, c] for a in bad for b in (0.5, bad[0], bad[3]) for c in (0.5, bad[1])], dtype=torch.float64, device=dev)
    Q = pts.shape[0]
    out = torch.full((Q,), 9, dtype=torch.uint8, device=dev)
    for which in (0, 1):
        assert L.vigo_query_points(h, which, Q, dptr(pts), dptr(out)) == 0
        torch.cuda.synchronize()
        o = out.cpu().numpy()
        assert set(o.tolist()) <= {0, 1}
        inside = ((pts >= 0) & (pts < torch.tensor([1.6, 1.6, 4.0], device=dev))).all(dim=1).cpu().numpy()
        assert (o[~inside] == 1).all() and (o[inside] == 0).all()
    # ESDF sampler: clamps to the lattice, never reads outside it
    f32 = torch.rand(8, 8, 8, dtype=torch.float32, device=dev)
    assert L.vigo_set_esdf(h, 8, 8, 8, origin, 0.1, dptr(f32)) == 0
    dq = torch.zeros(Q, dtype=torch.float64, device=dev)
    gq = torch.zeros(Q, 3, dtype=torch.float64, device=dev)
    assert L.vigo_esdf_query(h, Q, dptr(pts), dptr(dq), dptr(gq)) == 0
    # box sweep at the same positions
    box = (C.c_double * 3)(0.4, 0.4, 0.2)
    assert L.vigo_box_collision_points(h, Q, dptr(pts), box, 0.1, dptr(out)) == 0
    torch.cuda.synchronize()
    assert set(out.cpu().numpy().tolist()) <= {0, 1}
    # gates, spline evaluation, cost/gradient and the solver on control points seeded with the same values
    B, N = len(bad), 32
    ctrl = torch.zeros(B, N, 3, dtype=torch.float64, device=dev)
    ctrl[:, :, 0] = torch.arange(N, dtype=torch.float64, device=dev) * 0.04 + 0.1
    ctrl[:, :, 1] = 0.8
    ctrl[:, :, 2] = 1.0
    for b, v in enumerate(bad):
        ctrl[b, 10 + b, b % 3] = v
    flag = torch.full((B,), 9, dtype=torch.uint8, device=dev)
    first = torch.zeros(B, dtype=torch.int32, device=dev)
    assert L.vigo_traj_collision(h, B, N, dptr(ctrl), 0.025, dptr(flag), dptr(first)) == 0
    pt = torch.zeros(B, N, dtype=torch.uint8, device=dev)
    ln = torch.zeros(B, N, dtype=torch.uint8, device=dev)
    assert L.vigo_ctrl_occupancy(h, B, N, dptr(ctrl), dptr(pt), dptr(ln)) == 0
    obs = torch.tensor([[0.5, 0.8, 1.0, float("nan"), 1e300, 0.0, float("inf"), 0.5, 0.5],
                        [float("nan"), 0.8, 1.0, 0.1, 0.0, 0.0, 0.5, 0.5, 0.5]], dtype=torch.float64, device=dev)
    assert L.vigo_traj_dynamic_collision(h, B, N, dptr(ctrl), 0.025, NULL, dptr(obs), 2, dptr(flag)) == 0
    times = torch.tensor([0.0, 0.3, float("nan"), float("inf"), -1e300, 1e300], dtype=torch.float64, device=dev)
    ev = torch.zeros(B, 6, 3, dtype=torch.float64, device=dev)
    for deriv in (0, 1, 2):
        assert L.vigo_bspline_eval(h, B, N, dptr(ctrl), deriv, 6, dptr(times), dptr(ev)) == 0
    goff = (torch.arange(B * N + 1, dtype=torch.int32, device=dev) // 4).contiguous()   # a guide pair every 4th point
    G = int(goff[-1].item())
    gpv = torch.rand(G, 6, dtype=torch.float64, device=dev)
    gpv[::3, 0] = float("nan")
    gpv[1::3, 4] = float("inf")
    gunk = torch.zeros(G, dtype=torch.uint8, device=dev)
    assert L.vigo_guides_unknown(h, G, dptr(gpv), dptr(gunk)) == 0
    cost = torch.zeros(B, dtype=torch.float64, device=dev)
    grad = torch.zeros(B, N - 6, 3, dtype=torch.float64, device=dev)
    assert L.vigo_cost_grad(h, B, N, dptr(ctrl), dptr(goff), dptr(gpv), dptr(gunk), NULL, dptr(obs), 2, NULL, dptr(cost), dptr(grad), NULL) == 0
    st = torch.full((B,), 77, dtype=torch.int32, device=dev)
    work = ctrl.clone()
    assert L.vigo_optimize(h, B, N, dptr(work), dptr(goff), dptr(gpv), dptr(gunk), NULL, dptr(obs), 2, NULL, NULL, dptr(st), NULL, NULL, NULL) == 0
    torch.cuda.synchronize()
    assert (st.cpu().numpy() != 77).all()
    # fit, min-snap and the corridor checker
    fit_in = torch.rand(4, 12, 3, dtype=torch.float64, device=dev)
    fit_in[0, 3, 1] = float("nan"); fit_in[1, 0, 0] = float("inf"); fit_in[2, 5, 2] = 1e300
    fit_out = torch.zeros(4, 14, 3, dtype=torch.float64, device=dev)
    assert L.vigo_bspline_fit(h, 4, 12, 0.1, dptr(fit_in), NULL, dptr(fit_out)) == 0
    wp = torch.rand(5, 6, 3, dtype=torch.float64, device=dev) * 4
    wp[0, 2, 0] = float("nan"); wp[1, 1, 1] = float("inf"); wp[2, 3, 2] = 1e300; wp[3, 4] = wp[3, 3]
    cor = torch.full((5, 5), 0.3, dtype=torch.float64, device=dev)
    cor[4, 1] = float("nan")
    co = torch.zeros(5, 5, 3, 8, dtype=torch.float64, device=dev)
    kn = torch.zeros(5, 6, dtype=torch.float64, device=dev)
    ms = torch.full((5,), 77, dtype=torch.int32, device=dev)
    assert L.vigo_minsnap(h, 5, 6, 7, 4, 4, 1.0, 8.0, dptr(wp), dptr(cor), NULL, dptr(co), dptr(kn), dptr(ms)) == 0
    torch.cuda.synchronize()
    assert (ms[:3].cpu().numpy() < 0).all()                 # non-finite waypoints are reported, not solved
    seg = torch.zeros(6, 3, 8, dtype=torch.float64, device=dev)
    seg[:, :, 0] = 0.8
    seg[0, 0, 1] = float("nan"); seg[1, 1, 7] = float("inf"); seg[2, 2, 3] = 1e300; seg[3, 0, 1] = -1e300; seg[4, 0, 1] = 5e-324
    ns = torch.tensor([100, 100, 100, 100, 100, 0], dtype=torch.int32, device=dev)
    dT = torch.tensor([0.01, 0.01, 0.01, float("nan"), 1e300, 0.01], dtype=torch.float64, device=dev)
    cf = torch.full((6,), 9, dtype=torch.uint8, device=dev)
    c1 = torch.zeros(6, dtype=torch.int32, device=dev)
    c2 = torch.zeros(6, dtype=torch.int32, device=dev)
    assert L.vigo_corridor_check(h, 6, 7, dptr(seg), dptr(ns), dptr(dT), box, 0.1, dptr(cf), dptr(c1), dptr(c2)) == 0
    torch.cuda.synchronize()
    assert set(cf.cpu().numpy().tolist()) <= {0, 1}


def test_check_lists_finds_inconsistent_offsets():
    """vigo_check_lists: the integration-time validator for the CSR lists the solve kernels index"""
    from trajectory_planner_amd.vigo import Vigo
    v = Vigo(0)
    dev = v.device
    B, N = 5, 32
    goff = (torch.arange(B * N + 1, dtype=torch.int32, device=dev) // 3).contiguous()
    G = int(goff[-1].item())
    ooff = torch.tensor([0, 2, 2, 5, 5, 6], dtype=torch.int32, device=dev)
    assert v.check_lists(B, N, goff, G, ooff, 6) == 0
    assert v.check_lists(B, N, None, 0, None, 0) == 0
    assert v.check_lists(B, N, goff, G - 1, ooff, 6) >= 1            # the last offsets point past the pairs
    assert v.check_lists(B, N, goff, G, ooff, 5) == 1                # past the obstacles
    bad = goff.clone(); bad[40] = bad[39] - 1                        # decreasing
    assert v.check_lists(B, N, bad, G) >= 1
    bad = goff.clone(); bad[0] = 1                                   # does not start at 0 (and decreases right after)
    assert v.check_lists(B, N, bad, G) >= 1
    bad = ooff.clone(); bad[2] = -4
    assert v.check_lists(B, N, None, 0, bad, 6) >= 1
    with pytest.raises(ValueError):
        v.check_lists(B, N, goff[:-1].contiguous(), G)
    v.close()


def test_python_wrapper_refuses_mis_shaped_tensors(vigo_handle):
    """the kernels index by the extents the wrapper derives: a tensor of another shape must never reach them"""
    v = vigo_handle
    dev = v.device
    z = lambda *shape, dtype=torch.float64: torch.zeros(*shape, dtype=dtype, device=dev)
    v.set_grid(z(8, 8, 40, dtype=torch.uint8), np.zeros(3), 0.1)
    for call in (lambda: v.query_points(z(5, 2)),
                 lambda: v.guides_unknown(z(5, 3)),
                 lambda: v.optimize(z(2, 32, 3), z(2 * 32 + 1, dtype=torch.int32), z(4, 5)),
                 lambda: v.optimize(z(2, 32, 3), obs=z(3, 8)),
                 lambda: v.optimize(z(2, 32, 3), z(2 * 32 + 1, dtype=torch.int32), z(4, 6), z(3, dtype=torch.uint8)),
                 lambda: v.bspline_fit(z(2, 8, 3), z(2, 3, 3)),
                 lambda: v.bspline_eval(z(2, 8, 2), z(4)),
                 lambda: v.traj_collision(z(2, 8), 0.05),
                 lambda: v.traj_dynamic_collision(z(2, 8, 3), 0.05, z(2, dtype=torch.int32), z(1, 9)),
                 lambda: v.ctrl_occupancy(z(2, 8, 4)),
                 lambda: v.minsnap(z(2, 4, 3), z(2, 4)),
                 lambda: v.minsnap(z(2, 4, 3), conds=z(2, 4, 2)),
                 lambda: v.corridor_check(z(3, 3, 8), z(2, dtype=torch.int32), z(3), [0.4, 0.4, 0.2], 0.2),
                 lambda: v.corridor_check(z(3, 2, 8), z(3, dtype=torch.int32), z(3), [0.4, 0.4, 0.2], 0.2),
                 lambda: v.box_collision_points(z(5, 4), [0.4, 0.4, 0.2], 0.2),
                 lambda: v.esdf_query(z(5))):
        with pytest.raises((ValueError, TypeError)):
            call()
    # a caller-owned SolveResult is written by the kernel: extents, dtype and placement are checked like the inputs'
    from trajectory_planner_amd.vigo import SolveResult, VigoError
    good = lambda B, N: SolveResult(None, z(B, N - 6, 3), z(B, dtype=torch.int32), z(B), z(B, dtype=torch.int32), z(B, dtype=torch.int32))
    v.optimize(z(2, 32, 3), out=good(2, 32))
    for bad in (good(1, 32), good(2, 20)):
        with pytest.raises(ValueError):
            v.optimize(z(2, 32, 3), out=bad)
    o = good(2, 32)
    o.status = z(2, dtype=torch.int64)
    with pytest.raises(TypeError):
        v.optimize(z(2, 32, 3), out=o)
    o = good(2, 32)
    o.fx = torch.zeros(2, dtype=torch.float64)
    with pytest.raises(ValueError):
        v.optimize(z(2, 32, 3), out=o)
    with pytest.raises(VigoError):                     # N < 7 is refused before anything is allocated
        v.optimize(z(2, 5, 3))
    # dtype and placement are checked too
    with pytest.raises(TypeError):
        v.query_points(z(5, 3, dtype=torch.float32))
    with pytest.raises(ValueError):
        v.query_points(torch.zeros(5, 3, dtype=torch.float64))
