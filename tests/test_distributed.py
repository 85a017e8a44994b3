"""-m "not gpu": the N > 1 path on CPU with world_size-2 gloo — batch sharding, the snapshot
broadcast and the max-over-ranks timing reduction that bench.py uses.  (The kernels themselves
need a GPU; here each rank runs the oracle on its shard, which is exactly what makes this a
test of the sharding/broadcast logic and not of the kernels.)"""
import os
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, json
    sys.path.insert(0, os.environ["VIGO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["VIGO_ROOT"], "tests"))
    import numpy as np, torch, torch.distributed as dist
    import oracle_lib as ol
    from trajectory_planner_amd import synth, sharding
    dist.init_process_group("gloo")
    rank, ws = dist.get_rank(), dist.get_world_size()
    world = synth.make_box_world(synth.SEED_BASE + 2, n=64, n_boxes=20, centre_range=2.5, z_range=1.0)
    full = synth.make_bspline_batch(world, 37, 20, 5, start_range=1.5)           # 37: ragged over 2 ranks
    # rank 0 owns the map: pack (numpy stand-in of vigo_pack_grid) and broadcast the snapshot
    packed = torch.from_numpy(sharding.pack_grid_reference(world.voxels)) if rank == 0 else torch.empty(
        sharding.packed_words(*world.voxels.shape), dtype=torch.int32)
    dist.broadcast(packed, src=0)
    assert np.array_equal(packed.numpy(), sharding.pack_grid_reference(world.voxels))
    lo, hi = sharding.shard_range(full.B, rank, ws)
    mine = sharding.slice_batch(full, lo, hi)
    P = ol.default_params(); P.max_iterations = 20
    r = ol.optimize_batch(P, mine)
    # gather the per-rank control points on rank 0 (disjoint slices, no reduction)
    out = [None] * ws
    dist.all_gather_object(out, (lo, hi, r["ctrl"]))
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                                        # bench.py's timing rule
    if rank == 0:
        merged = np.zeros_like(full.ctrl)
        for (a, b, c) in out: merged[a:b] = c
        ref = ol.optimize_batch(P, full)["ctrl"]
        print(json.dumps({"ok": bool(np.array_equal(merged, ref)), "tmax": float(t.item()),
                          "ranges": [(a, b) for (a, b, _) in out]}))
    dist.destroy_process_group()
''')


def test_two_rank_gloo_sharding_broadcast_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, VIGO_ROOT=ROOT, MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    import json
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["ok"] and abs(res["tmax"] - 0.2) < 1e-12
    assert res["ranges"] == [[0, 19], [19, 37]]


def test_shard_ranges_cover_and_are_disjoint():
    from trajectory_planner_amd import sharding
    for B in (0, 1, 7, 8, 1024, 65536, 65537):
        for ws in (1, 2, 3, 8):
            r = [sharding.shard_range(B, k, ws) for k in range(ws)]
            assert r[0][0] == 0 and r[-1][1] == B
            assert all(r[k][1] == r[k + 1][0] for k in range(ws - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_slice_batch_rebases_csr():
    from trajectory_planner_amd import sharding, synth
    world = synth.make_box_world(synth.SEED_BASE + 2, n=64, n_boxes=20, centre_range=2.5, z_range=1.0)
    full = synth.make_bspline_batch(world, 11, 16, 3, start_range=1.5, n_obs=2)
    s = sharding.slice_batch(full, 4, 9)
    assert s.B == 5 and s.guide_off[0] == 0 and s.guide_off[-1] == len(s.guide_pv)
    assert s.obs_off[0] == 0 and s.obs_off[-1] == len(s.obs) == 10
    lo = full.guide_off[4 * 16]
    assert np.array_equal(s.guide_pv, full.guide_pv[lo:lo + len(s.guide_pv)])
