"""-m gpu: handles are independent — several host threads, each with its own handle and HIP stream, solving and
running gates concurrently give bit-identical results to the same calls made one after another (include/vigo.h:
"a handle owns its device buffers and HIP stream binding"; the facades keep one handle per planner thread)."""
import threading

import numpy as np
import pytest
import torch

from gpu_util import batch_to_dev
from trajectory_planner_amd import synth
from trajectory_planner_amd.vigo import Vigo, default_params

pytestmark = pytest.mark.gpu


def test_concurrent_handles_on_their_own_streams(small_world):
    dev = torch.device("cuda", 0)
    P = default_params()
    P.max_iterations = 30
    jobs = []
    for k, (B, N, prec) in enumerate(((64, 32, 0), (33, 64, 0), (17, 20, 2), (9, 100, 0))):
        b = synth.make_bspline_batch(small_world, B, N, 900 + k, start_range=3.0, n_obs=k % 2)
        jobs.append((b, prec))
    vox = torch.from_numpy(small_world.voxels).to(dev)

    def run(job, stream, out, idx, rounds):
        b, prec = job
        with torch.cuda.stream(stream):
            v = Vigo(0, P, prec)
            v.use_current_stream()
            v.set_grid(vox, small_world.origin, small_world.res)
            d = batch_to_dev(b, dev)
            res = []
            for _ in range(rounds):
                r = v.optimize(**d)
                flag, first = v.traj_collision(r.ctrl, 0.05)
                c, g, t = v.cost_grad(**d)
                res.append((r.ctrl.clone(), r.status.clone(), r.fx.clone(), flag.clone(), first.clone(), c.clone(), g.clone()))
            stream.synchronize()
            v.close()
        out[idx] = res

    serial = [None] * len(jobs)
    for i, job in enumerate(jobs):
        run(job, torch.cuda.Stream(dev), serial, i, 1)
    conc = [None] * len(jobs)
    threads = [threading.Thread(target=run, args=(job, torch.cuda.Stream(dev), conc, i, 6)) for i, job in enumerate(jobs)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
        assert not t.is_alive()
    for i in range(len(jobs)):
        assert conc[i] is not None and len(conc[i]) == 6
        for rnd in conc[i]:
            for a, b in zip(rnd, serial[i][0]):
                assert torch.equal(a, b) or (torch.isnan(a) & torch.isnan(b) | (a == b)).all(), i


def test_set_params_is_ordered_with_the_bound_stream(small_world):
    """vigo_set_params right behind an asynchronous vigo_optimize on a side stream: the running solve keeps the
    constants it was launched with (the kernels read them through device memory for the whole solve), the next
    one sees the new ones — same bits as the same calls with a synchronisation in between."""
    dev = torch.device("cuda", 0)
    b = synth.make_bspline_batch(small_world, 2048, 32, 4321, start_range=3.0)
    d = batch_to_dev(b, dev)
    P1 = default_params()
    P1.max_iterations = 50
    P1.g_epsilon = 0.0                 # every trajectory runs all iterations: a long solve
    P2 = default_params()
    P2.max_iterations = 7
    P2.w_smoothness = 3.0
    P2.dthresh = 0.7

    def run(sync_between):
        st = torch.cuda.Stream(dev)
        with torch.cuda.stream(st):
            v = Vigo(0, P1)
            v.use_current_stream()
            r1 = v.optimize(**d)
            if sync_between:
                st.synchronize()
            v.set_params(P2)           # must not reach the solve already queued
            r2 = v.optimize(**d)
            v.set_params(P1)
            r3 = v.optimize(**d)
            st.synchronize()
            v.close()
        return [(r.ctrl.clone(), r.iters.clone(), r.fx.clone()) for r in (r1, r2, r3)]

    serial = run(True)
    for _ in range(3):
        got = run(False)
        for a, s_ in zip(got, serial):
            for x, y in zip(a, s_):
                assert torch.equal(x, y)
    assert int(serial[0][1].max()) > 7 and int(serial[1][1].max()) <= 8 and torch.equal(serial[0][0], serial[2][0])


def test_recreated_handles_raise_their_own_lds_limit(small_world):
    """the dynamic-LDS attribute of the solve kernels is per-handle launch state (no function statics): a handle created
    after another was destroyed solves N = 128 and N = 200 (> 64 KiB of dynamic LDS) like the first one did"""
    import oracle_lib as ol
    from gpu_util import emulation
    dev = torch.device("cuda", 0)
    P = default_params()
    P.max_iterations = 20
    for round_ in range(2):
        v = Vigo(0, P)
        for N, B in ((128, 5), (200, 3), (32, 40)):
            b = synth.make_bspline_batch(small_world, B, N, 50 + N, start_range=3.0)
            r = v.optimize(**batch_to_dev(b, dev))
            with emulation(N):
                e = ol.optimize_batch(P, b)
            assert np.array_equal(r.ctrl.cpu().numpy(), e["ctrl"]) and np.array_equal(r.status.cpu().numpy(), e["status"]), (round_, N)
        v.close()
