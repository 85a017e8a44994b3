"""-m gpu: voxel snapshot, point queries, B-spline evaluation, the rebound-loop gates, the
corridor checker and the ESDF sampler against the CPU oracle (bit-exact for flags/indices and
for the fp64 spline values; ESDF to 1e-12)."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle_lib as ol
from gpu_util import to_dev
from trajectory_planner_amd import synth
from trajectory_planner_amd.vigo import Vigo, default_params

pytestmark = pytest.mark.gpu


def set_world(v, world):
    v.set_grid(to_dev(world.voxels, v.device), world.origin, world.res)


def test_point_queries_and_packing(vigo_handle, small_world):
    v = vigo_handle
    set_world(v, small_world)
    rng = np.random.default_rng(0)
    pts = rng.uniform(-7, 7, size=(20000, 3))
    pts[:100] = small_world.origin + rng.integers(0, 128, size=(100, 3)) * small_world.res   # on voxel faces
    for which in (0, 1):
        out = v.query_points(to_dev(pts, v.device), which).cpu().numpy()
        assert np.array_equal(out, synth.lookup(small_world, pts, which))
    # non-multiple-of-32 z extent and odd dims go through the tail path of the packer
    w2 = synth.World(np.ascontiguousarray(small_world.voxels[:77, :45, :41]), small_world.origin, 0.1, small_world.boxes)
    set_world(v, w2)
    out = v.query_points(to_dev(pts, v.device), 0).cpu().numpy()
    assert np.array_equal(out, synth.lookup(w2, pts, 0))
    # packed snapshot round trip (the RCCL broadcast path); the numpy statement of the format
    # (used by the CPU gloo test) is pinned to the kernel here
    packed = v.pack_grid(to_dev(small_world.voxels, v.device))
    from trajectory_planner_amd import sharding
    assert np.array_equal(packed.cpu().numpy(), sharding.pack_grid_reference(small_world.voxels))
    odd = np.ascontiguousarray(small_world.voxels[:33, :17, :41])
    assert np.array_equal(v.pack_grid(to_dev(odd, v.device)).cpu().numpy(), sharding.pack_grid_reference(odd))
    v2 = Vigo(0)
    v2.set_grid_packed(packed, small_world.voxels.shape, small_world.origin, small_world.res)
    assert np.array_equal(v2.query_points(to_dev(pts, v.device), 1).cpu().numpy(), synth.lookup(small_world, pts, 1))
    g = v2.guides_unknown(to_dev(np.concatenate([pts, pts], axis=1), v.device)).cpu().numpy()
    assert np.array_equal(g, synth.lookup(small_world, pts, 1))
    v2.close()


def test_bspline_eval_is_bit_exact(vigo_handle, small_world):
    v = vigo_handle
    b = synth.make_bspline_batch(small_world, 17, 32, 3, start_range=3.0)
    P = default_params()
    dur = (b.N - 3) * P.ts_ctrl
    n = ol.oracle().vgo_sample_times(dur, 0.05, None, 0)
    times = np.zeros(n)
    ol.oracle().vgo_sample_times(dur, 0.05, ol._d(times), n)
    times = np.concatenate([times, [-0.3, dur, dur + 1.0, 0.2, 0.4]])   # clamps and exact knots
    for deriv in (0, 1, 2):
        out = v.bspline_eval(to_dev(b.ctrl, v.device), to_dev(times, v.device), deriv).cpu().numpy()
        ref = np.zeros_like(out)
        for i in range(b.B):
            c = np.ascontiguousarray(b.ctrl[i])
            for k, t in enumerate(times):
                ol.oracle().vgo_traj_eval(b.N, ol._d(c), P.ts_ctrl, deriv, float(t), ol._d(ref[i, k]))
        assert np.array_equal(out, ref), deriv


def test_gates_match_oracle(vigo_handle, small_world):
    v = vigo_handle
    set_world(v, small_world)
    g, keep = ol.make_grid(small_world)
    O = ol.oracle()
    P = default_params()
    for N, dt in ((32, 0.05), (20, 0.025), (64, 0.05)):
        b = synth.make_bspline_batch(small_world, 300, N, 8 + N, start_range=4.0, n_obs=2)
        ctrl = to_dev(b.ctrl, v.device)
        flag, first = v.traj_collision(ctrl, dt)
        pt, line = v.ctrl_occupancy(ctrl)
        dyn = v.traj_dynamic_collision(ctrl, dt, to_dev(b.obs_off, v.device), to_dev(b.obs, v.device))
        flag, first, pt, line, dyn = (t.cpu().numpy() for t in (flag, first, pt, line, dyn))
        for i in range(b.B):
            c = np.ascontiguousarray(b.ctrl[i])
            fi = C.c_int()
            f = O.vgo_traj_collision(C.byref(g), N, ol._d(c), P.ts_ctrl, dt, C.byref(fi))
            assert f == flag[i] and fi.value == first[i], (i, f, flag[i], fi.value, first[i])
            p_ref, l_ref = np.zeros(N, dtype=np.uint8), np.zeros(N, dtype=np.uint8)
            O.vgo_ctrl_occupancy(C.byref(g), N, ol._d(c), ol._u(p_ref), ol._u(l_ref))
            assert np.array_equal(p_ref, pt[i]) and np.array_equal(l_ref, line[i])
            o = np.ascontiguousarray(b.obs[b.obs_off[i]:b.obs_off[i + 1]])
            assert O.vgo_traj_dynamic_collision(N, ol._d(c), P.ts_ctrl, dt, len(o), ol._d(o)) == dyn[i]
        if N <= 20:   # longer paths leave this 12.8 m world (out of map counts as occupied)
            assert 0 < flag.mean() < 1
        assert line.any()


def maze_like_world(n=96, res=0.1, seed=5):
    """occupied / free / unknown voxels with an origin on the octomap key lattice"""
    rng = np.random.default_rng(seed)
    vox = np.zeros((n, n, 40), dtype=np.uint8)
    for _ in range(40):
        c = rng.integers(5, n - 5, size=2)
        s = rng.integers(1, 5, size=2)
        vox[c[0] - s[0]:c[0] + s[0], c[1] - s[1]:c[1] + s[1], 0:rng.integers(10, 40)] |= 4
    unk = rng.random((n // 8, n // 8, 5)) < 0.08
    vox[np.repeat(np.repeat(np.repeat(unk, 8, 0), 8, 1), 8, 2)] |= 2
    return synth.World(vox, np.array([-4.8, -4.8, -0.5]), res, np.zeros((0, 6)))


def test_corridor_checker_matches_oracle(vigo_handle):
    v = vigo_handle
    w = maze_like_world()
    set_world(v, w)
    g, keep = ol.make_grid(w)
    O = ol.oracle()
    box = np.array([0.4, 0.4, 0.2])
    coeffs, n_samp, delT, dur = synth.make_corridor_segments(11, 96, extent_lo=(-4, -4, 0.6), extent_hi=(4, 4, 1.6), n_samples=3000)
    n_samp[:8] = [0, 1, 2, 15, 16, 17, 255, 4097]          # ragged sample counts
    delT[8:16] = 0.1                                       # the reference's sample_delta_time
    n_samp[8:16] = (dur[8:16] / 0.1).astype(np.int32) + 1
    coeffs[16, :, 0] = [4.6, 0.0, 1.0]                     # leaves the metric bounds
    coeffs[17, 0, 1] = 3.0                                 # long fast segment: tile too big for LDS
    coeffs[17, 1, 1] = 2.5
    for map_res in (0.2, 0.1):
        flag, first, count = v.corridor_check(to_dev(coeffs, v.device), to_dev(n_samp, v.device), to_dev(delT, v.device), box, map_res)
        flag, first, count = flag.cpu().numpy(), first.cpu().numpy(), count.cpu().numpy()
        for s in range(len(coeffs)):
            fi, cn = C.c_int(), C.c_int()
            c = np.ascontiguousarray(coeffs[s])
            f = O.vgo_corridor_check_segment(C.byref(g), 7, ol._d(c), int(n_samp[s]), float(delT[s]), ol._d(box), map_res,
                                             C.byref(fi), C.byref(cn))
            assert (f, fi.value, cn.value) == (flag[s], first[s], count[s]), (s, map_res, f, fi.value, cn.value, flag[s], first[s], count[s])
        assert 0 < flag.mean() < 1
    # tighter metric bounds make everything near the rim collide (octomap getMetricMin/Max)
    v.set_metric_bounds([-1, -1, 0.0], [1, 1, 3.0])
    flag2, _, _ = v.corridor_check(to_dev(coeffs, v.device), to_dev(n_samp, v.device), to_dev(delT, v.device), box, 0.2)
    g.bmin[:] = [-1, -1, 0.0]
    g.bmax[:] = [1, 1, 3.0]
    for s in range(len(coeffs)):
        c = np.ascontiguousarray(coeffs[s])
        f = O.vgo_corridor_check_segment(C.byref(g), 7, ol._d(c), int(n_samp[s]), float(delT[s]), ol._d(box), 0.2, None, None)
        assert f == flag2.cpu().numpy()[s]


@pytest.mark.parametrize("box,map_res", [((0.4, 0.4, 0.2), 0.2), ((0.6, 0.3, 0.2), 0.1), ((0.55, 0.47, 0.23), 0.2), ((1.0, 0.9, 0.3), 0.2)])
def test_corridor_span_certificates_and_what_they_refuse(vigo_handle, box, map_res):
    """The checker decides spans of 32 / 16 samples at once where it can prove every sample's verdict (vigo_corridor.hip,
    SpanConst) and hands the rest to the per-sample walk: flags, first indices and counts must be the reference walk's
    (PO.cpp:547-589, :634-656) whatever route a segment takes — long slow segments (certificates), a box that is an
    exact multiple of map_resolution (the lattice count then changes from pose to pose with the rounding of
    fx +- box / 2: the verdict per choice of counts), sample counts either side of the one-sample-per-lane limit, fast
    segments, segments that leave the map or start outside it, non-finite coefficients (a pose at infinity does NOT
    collide: x86's conversion of the NaN count), clocks that stand still, run backwards or barely move, a box of more
    than three cells per axis (the last parameter set)."""
    v = vigo_handle
    w = maze_like_world()
    set_world(v, w)
    g, keep = ol.make_grid(w)
    O = ol.oracle()
    box = np.array(box)
    coeffs, n_samp, delT, dur = synth.make_corridor_segments(51, 40, extent_lo=(-4, -4, 0.6), extent_hi=(4, 4, 1.6), n_samples=9000)
    n_samp[:8] = [511, 512, 513, 1023, 1025, 4096, 8191, 33]
    coeffs[8, :, 1:] *= 40.0                               # samples further apart than a voxel
    coeffs[9, :, 1] *= 6.0
    coeffs[10, 0, 0] += 9.0                                # outside the map for good
    coeffs[11, 1, 0], coeffs[11, 1, 1] = -4.7, -0.4        # walks out of it
    coeffs[12, 0, 3] = np.inf
    coeffs[13, 2, 0] = np.nan
    coeffs[14, 1, 5] = 1e300
    delT[15], delT[16], delT[17] = 0.0, -delT[16], delT[17] * 1e-9
    n_samp[15:17] = 2500
    coeffs[18, :, 1:] = 0.0                                # a segment that does not move
    coeffs[19, 0, 0], coeffs[20, 1, 0], coeffs[21, 2, 0] = 1e39, -3.4028234e38, 1e20   # finite doubles beyond / at / far below the floats' range
    flag, first, count = (x.cpu().numpy() for x in v.corridor_check(to_dev(coeffs, v.device), to_dev(n_samp, v.device), to_dev(delT, v.device), box, map_res))
    for s in range(len(coeffs)):
        fi, cn = C.c_int(), C.c_int()
        c = np.ascontiguousarray(coeffs[s])
        f = O.vgo_corridor_check_segment(C.byref(g), 7, ol._d(c), int(n_samp[s]), float(delT[s]), ol._d(box), map_res, C.byref(fi), C.byref(cn))
        assert (f, fi.value, cn.value) == (flag[s], first[s], count[s]), (s, f, fi.value, cn.value, flag[s], first[s], count[s])
    assert 0 < flag.mean() < 1


def test_corridor_checker_on_more_segments_than_the_clock_workspace_holds(vigo_handle):
    """Up to 16384 segments the sample-clock tables come from a kernel of their own (k_corridor_clocks, a thread per
    segment); above that every workgroup writes its own: same results either way, and across the boundary"""
    v = vigo_handle
    w = maze_like_world()
    set_world(v, w)
    g, keep = ol.make_grid(w)
    box = np.array([0.4, 0.4, 0.2])
    S = 16400
    coeffs, n_samp, delT, dur = synth.make_corridor_segments(61, S, extent_lo=(-4, -4, 0.6), extent_hi=(4, 4, 1.6), n_samples=40)
    n_samp[::97] = 700
    delT[::97] = dur[::97] / 700
    dc, dn, dt = to_dev(coeffs, v.device), to_dev(n_samp, v.device), to_dev(delT, v.device)
    big = [x.cpu().numpy() for x in v.corridor_check(dc, dn, dt, box, 0.2)]
    small = [x.cpu().numpy() for x in v.corridor_check(dc[:16384], dn[:16384], dt[:16384], box, 0.2)]
    for a, b in zip(big, small):
        assert np.array_equal(a[:16384], b)
    pick = np.sort(np.concatenate([np.arange(0, S, 97)[:60], np.random.default_rng(2).choice(S, 240, replace=False)]))
    f_o, fi_o, cn_o = ol.corridor_check_batch(g, coeffs[pick], n_samp[pick], delT[pick], box, 0.2)
    assert np.array_equal(f_o, big[0][pick]) and np.array_equal(fi_o, big[1][pick]) and np.array_equal(cn_o, big[2][pick])
    assert 0 < big[0].mean() < 1


def test_inflate_grid_matches_numpy_dilation(vigo_handle):
    """vigo_inflate_grid: bit0 = box dilation of bit2 (integer/byte work: bit-exact), other bits kept, in place"""
    v = vigo_handle
    rng = np.random.default_rng(9)
    for shape, r in (((37, 50, 19), (2, 1, 0)), ((64, 64, 64), (4, 4, 2)), ((5, 3, 70), (0, 0, 3)), ((16, 16, 16), (0, 0, 0)), ((9, 130, 11), (8, 3, 20))):
        vox = (rng.random(shape) < 0.02).astype(np.uint8) * 4 + (rng.random(shape) < 0.3).astype(np.uint8) * 2 + (rng.random(shape) < 0.5).astype(np.uint8)
        occ = (vox & 4) != 0
        ref = occ.copy()
        for axis, rr in enumerate(r):
            acc = ref.copy()
            for d in range(1, rr + 1):
                sl_to = [slice(None)] * 3; sl_from = [slice(None)] * 3
                sl_to[axis] = slice(d, None); sl_from[axis] = slice(None, -d)
                acc[tuple(sl_to)] |= ref[tuple(sl_from)]
                acc[tuple(sl_from)] |= ref[tuple(sl_to)]
            ref = acc
        got = v.inflate_grid(to_dev(vox, v.device), *r).cpu().numpy()
        assert np.array_equal((got & 1) != 0, ref), (shape, r)
        assert np.array_equal(got & 6, vox & 6)


@pytest.mark.parametrize("deg", [3, 5, 9])
def test_corridor_checker_other_polynomial_degrees(vigo_handle, deg):
    """polynomial_degree other than the cfg's 7 (the kernel keeps degree-7 coefficients in registers and
    walks any other degree from LDS): same bit-exact agreement with the oracle"""
    v = vigo_handle
    w = maze_like_world()
    set_world(v, w)
    g, keep = ol.make_grid(w)
    O = ol.oracle()
    box = np.array([0.4, 0.4, 0.2])
    coeffs, n_samp, delT, dur = synth.make_corridor_segments(20 + deg, 40, deg=deg, extent_lo=(-4, -4, 0.6), extent_hi=(4, 4, 1.6), n_samples=2000)
    flag, first, count = (x.cpu().numpy() for x in v.corridor_check(to_dev(coeffs, v.device), to_dev(n_samp, v.device), to_dev(delT, v.device), box, 0.2))
    for s in range(len(coeffs)):
        fi, cn = C.c_int(), C.c_int()
        c = np.ascontiguousarray(coeffs[s])
        f = O.vgo_corridor_check_segment(C.byref(g), deg, ol._d(c), int(n_samp[s]), float(delT[s]), ol._d(box), 0.2, C.byref(fi), C.byref(cn))
        assert (f, fi.value, cn.value) == (flag[s], first[s], count[s]), (deg, s)
    assert 0 < flag.mean() < 1


def test_esdf_query_matches_oracle_and_sphere(vigo_handle):
    v = vigo_handle
    n, res = 64, 0.1
    dist, origin = synth.sphere_esdf(n, res, (0.3, -0.2, 0.1), 1.0)
    v.set_esdf(to_dev(dist, v.device), origin, res)
    rng = np.random.default_rng(2)
    pts = rng.uniform(-3.6, 3.6, size=(5000, 3))       # includes points outside the lattice (clamped)
    d, g = v.esdf_query(to_dev(pts, v.device))
    d, g = d.cpu().numpy(), g.cpu().numpy()
    for i in range(0, 5000, 7):
        dd, gg = C.c_double(), np.zeros(3)
        ol.oracle().vgo_esdf_query(n, n, n, ol._d(origin), res, dist.ctypes.data_as(C.POINTER(C.c_float)), ol._d(pts[i]),
                                   C.byref(dd), ol._d(gg))
        assert d[i] == dd.value and np.array_equal(g[i], gg)
    inside = np.abs(pts).max(1) < 3.0
    r = pts - np.array([0.3, -0.2, 0.1])
    far = inside & (np.linalg.norm(r, axis=1) > 0.3)
    assert np.max(np.abs(d[far] - (np.linalg.norm(r[far], axis=1) - 1.0))) < 1e-2


@pytest.mark.parametrize("shape", [(2, 2, 2), (2, 3, 4), (5, 6, 7), (17, 9, 31), (64, 65, 66), (33, 2, 100)])
def test_esdf_query_ragged_lattices(vigo_handle, shape):
    """Every residue of the lattice size modulo the brick step (overlapping bricks hold 3 cells per axis), the minimum
    size, non-cubic lattices; random sample values so that a wrong corner cannot go unnoticed; every query compared."""
    v = vigo_handle
    rng = np.random.default_rng(sum(shape))
    dist = rng.normal(size=shape).astype(np.float32)
    origin, res = np.array([-0.35, 0.2, 1.0]), 0.25
    v.set_esdf(to_dev(dist, v.device), origin, res)
    hi = origin + res * np.array(shape)
    pts = rng.uniform(origin - 0.6, hi + 0.6, size=(4000, 3))
    # exact cell borders, the first and the last sample of every axis
    pts[:50] = origin + res * (rng.integers(0, np.array(shape) + 1, size=(50, 3)) + 0.5)
    pts[50] = origin + 0.5 * res
    pts[51] = hi - 0.5 * res
    d, g = v.esdf_query(to_dev(pts, v.device))
    d_ref, g_ref = ol.esdf_query_batch(dist, origin, res, pts)
    assert np.array_equal(d.cpu().numpy(), d_ref) and np.array_equal(g.cpu().numpy(), g_ref)
    # the fp32 entry on the same lattice: bit for bit against its oracle twin, non-finite and far-away points included
    p32 = pts.astype(np.float32)
    p32[60] = (np.nan, 0.0, 0.0)
    p32[61] = (np.inf, -np.inf, 1e30)
    p32[62] = (-1e30, 1e-30, 3e38)
    got = v.esdf_query_f32(to_dev(p32, v.device)).cpu().numpy()
    ref = ol.esdf_query_f32_batch(dist, origin, res, p32)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_stale_hip_error_of_another_library_is_not_reported(vigo_handle):
    """The launchers report hipGetLastError(); an error another library of the process left in that slot
    (found by tools/fuzz_map_gates.py: torch had left one before the first vigo call of a fresh process)
    must not surface as a failure of ours."""
    v = vigo_handle
    path = None
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            path = line.split()[-1]
            break
    assert path, "HIP runtime not mapped"
    hip = C.CDLL(path)
    hip.hipFree.argtypes = [C.c_void_p]
    vox = np.zeros((8, 8, 40), dtype=np.uint8)
    vox[4, 4, 20] = 4
    v.set_grid(to_dev(vox, v.device), np.zeros(3), 0.1)
    for call in (lambda: v.inflate_grid(to_dev(vox, v.device), 1, 1, 1),
                 lambda: v.pack_grid(to_dev(vox, v.device)),
                 lambda: v.query_points(to_dev(np.zeros((5, 3)), v.device), 0)):
        assert hip.hipFree(C.c_void_p(0x1234)) != 0          # leaves hipErrorInvalidValue behind
        call()                                                # must not raise


def test_bspline_eval_beyond_the_grid_y_limit(vigo_handle):
    """more trajectories than gridDim.y allows in one launch (65535): sliced launches, same values"""
    v = vigo_handle
    rng = np.random.default_rng(4)
    B, N = 70001, 8
    ctrl = to_dev(rng.normal(size=(B, N, 3)), v.device)
    times = to_dev(np.array([0.0, 0.13, 0.5, 0.99]), v.device)
    out = v.bspline_eval(ctrl, times, 0)
    ref = v.bspline_eval(ctrl[65000:], times, 0)
    assert torch.equal(out[65000:], ref) and bool(torch.isfinite(out).all())
    one = np.zeros(3)
    c = np.ascontiguousarray(ctrl[70000].cpu().numpy())
    ol.oracle().vgo_traj_eval(N, ol._d(c), default_params().ts_ctrl, 0, 0.5, ol._d(one))
    assert np.array_equal(out[70000, 2].cpu().numpy(), one)
