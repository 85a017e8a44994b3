"""The workload generator that puts the planner's own host pipeline under the headline configuration
(trajectory_planner_amd/synth.py: make_pipeline_world / make_pipeline_batch / reguide_batch over
libtrajectory_planner_vigo.so's vigo_host_bspline_guides_batch — no GPU involved).  The batch entry point must give what
the one-path entry point (tests/test_prologue_restatement.py holds that one to a restatement of bsplineTraj.cpp:403-571)
gives path by path; the CSR lists must be well formed; the dense world must produce the share of guided trajectories
the full-size GPU test and bench.py rely on."""
import ctypes as C

import numpy as np

from trajectory_planner_amd import synth


def _one_path_prologue(world, path):
    from test_prologue_restatement import _host
    host = _host()
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    cap = 200000
    ctrl, nctrl = np.zeros(cap), C.c_int()
    seg, nseg = np.zeros(cap, dtype=np.int32), C.c_int()
    goff, gout = np.zeros(cap, dtype=np.int32), np.zeros(cap)
    poff, pout = np.zeros(cap, dtype=np.int32), np.zeros(cap)
    vox = np.ascontiguousarray(world.voxels)
    dims = (C.c_int * 3)(*vox.shape)
    origin = np.ascontiguousarray(world.origin)
    cfg = np.ascontiguousarray(synth.PIPELINE_CFG)
    rc = host.vigo_host_bspline_prologue(vox.ctypes.data_as(C.c_void_p), dims, origin.ctypes.data_as(dp), world.res, len(path),
                                         np.ascontiguousarray(path).ctypes.data_as(dp), cfg.ctypes.data_as(dp), ctrl.ctypes.data_as(dp),
                                         C.byref(nctrl), seg.ctypes.data_as(ip), C.byref(nseg), goff.ctypes.data_as(ip),
                                         gout.ctypes.data_as(dp), poff.ctypes.data_as(ip), pout.ctypes.data_as(dp), cap)
    n = nctrl.value
    return rc, nseg.value, ctrl[:3 * n].reshape(n, 3).copy(), goff[:n + 1].copy(), gout[:6 * goff[n]].reshape(-1, 6).copy()


def test_batch_prologue_equals_the_one_path_prologue():
    world = synth.make_box_world(synth.SEED_BASE + 2, n=128, n_boxes=40, centre_range=5.5, z_range=2.0)
    rng = np.random.default_rng(3)
    K, N = 22, 24
    s = np.arange(K) * synth.CTRL_SPACING
    start = np.concatenate([rng.uniform(-3.5, 3.5, size=(48, 2)), np.full((48, 1), 1.0)], axis=1)
    heading = rng.uniform(0, 2 * np.pi, size=48)
    dirv = np.stack([np.cos(heading), np.sin(heading), np.zeros(48)], axis=1)
    paths = start[:, None, :] + s[None, :, None] * dirv[:, None, :]
    ctrl, status, n_seg, goff, gpv = synth.host_guides(world, N, paths=paths)
    guided = 0
    for t in range(48):
        rc, nseg1, c1, goff1, gpv1 = _one_path_prologue(world, paths[t])
        if rc == -1:
            assert status[t] == -1
            continue
        assert rc == 0
        if nseg1 == -1:
            assert status[t] == -2 and goff[t * N] == goff[(t + 1) * N]
            continue
        assert status[t] == 0 and n_seg[t] == nseg1 and np.array_equal(ctrl[t], c1)
        assert np.array_equal(goff[t * N:(t + 1) * N + 1] - goff[t * N], goff1)
        assert np.array_equal(gpv[goff[t * N]:goff[(t + 1) * N]], gpv1)
        guided += len(gpv1) > 0
    assert guided >= 5 and (status == 0).sum() >= 20
    # the re-guide entry on the same (unmoved) control points appends exactly the same pairs again
    ok = status == 0
    _, st2, _, goff2, gpv2 = synth.host_guides(world, N, ctrl=ctrl[ok])
    sub = np.nonzero(ok)[0]
    for k, t in enumerate(sub):
        assert st2[k] == 0
        assert np.array_equal(gpv2[goff2[k * N]:goff2[(k + 1) * N]], gpv[goff[t * N]:goff[(t + 1) * N]])


def test_append_guides_keeps_push_order():
    rng = np.random.default_rng(0)
    ca, cb = rng.integers(0, 3, size=40), rng.integers(0, 3, size=40)
    goff_a, goff_b = np.concatenate([[0], np.cumsum(ca)]).astype(np.int32), np.concatenate([[0], np.cumsum(cb)]).astype(np.int32)
    a, b = rng.normal(size=(goff_a[-1], 6)), rng.normal(size=(goff_b[-1], 6))
    goff, pv = synth.append_guides(goff_a, a, goff_b, b)
    for i in range(40):
        want = np.concatenate([a[goff_a[i]:goff_a[i + 1]], b[goff_b[i]:goff_b[i + 1]]])
        assert np.array_equal(pv[goff[i]:goff[i + 1]], want)


def test_dense_world_gives_the_share_of_guided_trajectories_the_headline_extra_needs():
    world = synth.make_pipeline_world()
    b = synth.make_pipeline_batch(world, 1024, 32, synth.SEED_BASE + 2 + 2000)
    hist, per_traj, share = synth.pairs_histogram(b)
    assert b.ctrl.shape == (1024, 32, 3) and b.guide_off[-1] == len(b.guide_pv) == len(b.guide_unk)
    assert np.all(np.diff(b.guide_off) >= 0) and share >= 0.30 and per_traj.mean() >= 3.0
    cnt = np.diff(b.guide_off).reshape(1024, 32)
    # (a segment that runs to the goal — findCollisionSeg's `i == endIdx - 1` corner, BT.cpp:426-430 — also gives pairs to the
    # fixed end points; getDistanceCost only walks the free ones, BT.cpp:830, and so does the kernel)
    assert cnt[:, :2].sum() == 0
    v = b.guide_pv[:, 3:]
    assert np.allclose(np.linalg.norm(v, axis=1), 1.0, atol=1e-12)  # guide directions are unit vectors (BT.cpp:532-533)
    assert np.isfinite(b.guide_pv).all()
