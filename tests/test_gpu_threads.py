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
