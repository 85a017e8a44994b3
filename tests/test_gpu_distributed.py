"""-m gpu: the N > 1 rank logic of bench.py rehearsed with two ranks on the one GPU of the box (every rank on
cuda:0, collectives over gloo): sharded batches, the map broadcast, the max-over-ranks timing and the configs[3]
shard extra all run; no scaling number is claimed from it (the printed line is marked as a rehearsal).  The real
1 -> 8 GPU runs over RCCL are the driver's."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_rehearsal_of_bench_on_one_gpu():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--rehearse-on-one-gpu"]
    # a fresh child process (never a re-exec of this one, which has initialised the GPU)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]               # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and "REHEARSAL" in out["data"]
    assert out["map_bcast_ms"] > 0 and out["map_snapshot_identical_on_all_ranks"] is True
    assert out["value"] > 0 and abs(out["value"] - 2 * out["config"]["trajs_per_gpu"] * out["steps"] / (out["ms_per_step"] * out["steps"] / 1e3)) < 1e-6 * out["value"]
    assert out["roofline"]["kernel_ms"] > 0
    shard = out["config4_shard"]                             # BASELINE configs[3]: 8192 x 64 per GPU, on every rank count
    assert shard["n_gpus"] == 2 and shard["map_bcast_ms"] > 0 and shard["value"] > 0 and "8192" in shard["workload"]
    strong = out["config4_strong"]                           # ONE 65 536 x 64 batch cut in two contiguous slices
    assert strong["n_gpus"] == 2 and strong["slice_bounds"] == [[0, 32768], [32768, 65536]] and strong["value"] > 0
    assert out["collectives"]["backend"] == "gloo"


def test_rccl_first_contact_with_one_rank():
    """RCCL itself, on this box's one GPU: a fresh child (torch.distributed.run, one rank) runs bench.py with
    --force-collectives, which creates the `nccl` (= RCCL) process group with device_id and then makes exactly the calls
    the N > 1 path makes — barrier, broadcast of the int32 packed snapshot (256^3 and 512^3 maps), all_reduce MIN / MAX
    on fp64 device tensors (the snapshot signatures, the max-over-ranks clock).  Library load, communicator creation,
    dtype and op support are what this proves; a world of one moves no bytes over xGMI and no scaling is claimed."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29543", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "2",
           "--force-collectives", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    col = out["collectives"]
    print("\n" + json.dumps(col))
    assert col["backend"] == "nccl" and col["world_size"] == 1 and col["forced_at_world_size_1"] is True and col["rccl_version"]
    assert out["n_gpus"] == 1 and out["map_bcast_ms"] > 0 and out["map_snapshot_identical_on_all_ranks"] is True
    assert out["value"] > 0 and out["config4_shard"]["map_bcast_ms"] > 0
    strong = out["config4_strong"]                          # ONE 65 536 x 64 batch; at N = 1 the whole of it on this GPU
    assert strong["scaling"] == "strong" and strong["slice_bounds"] == [[0, 65536]] and strong["value"] > 0
