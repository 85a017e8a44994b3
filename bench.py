#!/usr/bin/env python3
"""bench.py — B-spline trajectories/s of the batched ViGO solve on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one batch that is already resident in HBM:
restore the initial control points (D2D), look up isUnknown for every guide point in the voxel
snapshot (bsplineTraj.cpp:841, hoisted), and run the whole 50-iteration L-BFGS solve for all
trajectories in ONE kernel launch (vigo_optimize).  N > 1: every rank owns an independent batch
of the same size (weak scaling); the only collective is the one-off RCCL broadcast of the packed
voxel snapshot from rank 0 before the timed region (reported as map_bcast_ms).

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     achieved = ALGORITHMIC bytes of the solve kernel per launch (SURVEY.md §8(d)
               streaming model, fp64, evaluated from the iteration/evaluation counts the kernel
               reports) / its average duration measured with HIP events on the launch stream;
               peak = 8 TB/s HBM3E.  traffic = HBM bytes per launch from rocprofv3 PMC
               (profiles/pmc_<workload>_<precision>.json, when that summary exists) else null.
               The kernel keeps its state in LDS/VGPRs, so the roofline that BINDS it is VALU issue:
               issue_frac = SQ_INSTS_VALU x 4 cycles / (SIMDs x kernel cycles at the stated clock),
               simd_occupancy = resident solver waves / SIMDs, measured_hbm_frac = traffic / kernel
               time / peak — all three next to the streaming-model frac.
  cpu_baseline the CPU oracle (fp64 restatement, oracle/) timed on this box's host cores on a
               bounded sample of the same workload: kind "port", 1 thread (plus an all-core
               figure in cpu_baseline_allcores).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s measured copy rate
SHADER_CLOCK_GHZ = 2.4  # MI355X peak engine clock (MI355X_MICROARCH.md); issue_frac is quoted against it
VALU_ISSUE_CYCLES = 4   # a wave64 VALU instruction occupies its SIMD's issue port for 4 cycles (16 lanes per cycle)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="config2", choices=["config2", "config4"])
    ap.add_argument("--batch", type=int, default=0, help="override trajectories per GPU")
    ap.add_argument("--iters", type=int, default=50, help="L-BFGS max_iterations (BASELINE: 50)")
    ap.add_argument("--precision", default="f64", choices=["f64", "f32", "f64_fast"],
                    help="f64: reference expression order (default); f64_fast: explicit fma + reciprocal "
                         "two-loop (VIGO_PREC_F64_FAST, same 1e-4 parity gate); f32: fp32 state")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the PCIe-inclusive and other-arithmetic-mode side measurements")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--force-collectives", action="store_true",
                    help="N = 1 only: create the RCCL process group anyway (world size 1) and make exactly the collective calls "
                         "the N > 1 path makes — broadcast of the packed snapshot, all_reduce MIN/MAX on fp64 device tensors, barrier — "
                         "so RCCL's library load, dtypes and ops are exercised on a one-GPU box")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="rehearsal of the N > 1 rank logic on a one-GPU box: every rank uses cuda:0 and the collectives "
                         "go over gloo (host-staged); timings are meaningless, the printed line is marked")
    return ap.parse_args()


def workload(args, rank, world_size):
    from trajectory_planner_amd import synth
    if args.workload == "config2":
        B = args.batch or 1024
        world = synth.make_box_world(synth.SEED_BASE + 2, n=256, n_boxes=200)
        batch = synth.make_bspline_batch(world, B, 32, synth.SEED_BASE + 2 + 1000 + rank)
        name = f"configs[1]: {B} B-spline trajs x 32 ctrl pts, 256^3 voxel grid, {args.iters} L-BFGS iters, per GPU"
    else:
        B = args.batch or 8192
        world = synth.make_box_world(synth.SEED_BASE + 4, n=512, n_boxes=800, centre_range=24.0)
        batch = synth.make_bspline_batch(world, B, 64, synth.SEED_BASE + 4 + 1000 + rank, start_range=16.0)
        name = f"configs[3] shard: {B} B-spline trajs x 64 ctrl pts, 512^3 voxel grid, {args.iters} L-BFGS iters, per GPU"
    return world, batch, name


def algorithmic_bytes(n, m, iters, evals, gpairs, elem=8):
    """SURVEY.md §8(d) streaming model (every BLAS-1 pass of the CPU reference touches memory):
    per iteration k: 4*n*min(m,k) (two-loop reads s_j,y_j twice) + 12*n (xp,gp save; s,y; norms),
    per evaluation: 7*n + 7*G; elem = 8 bytes in the fp64 mode, 4 in fp32."""
    import numpy as np
    it = np.maximum(iters.astype(np.int64) - 1, 0)          # two-loops executed
    ramp = np.where(it <= m, it * (it + 1) // 2, m * (m + 1) // 2 + (it - m) * m)
    return elem * (4 * n * ramp + 12 * n * iters.astype(np.int64) + evals.astype(np.int64) * (7 * n + 7 * gpairs))


def level_count(batch):
    """trajectories the kernels' level rule applies to (plan_in_z off in every bench configuration): all control points at one
    height to 2^-40 relative — the synthetic paths of SURVEY.md §8(d) fly at z = 1.0"""
    import numpy as np
    z = batch.ctrl[:, :, 2]
    zmin, zmax = z.min(1), z.max(1)
    return int(((zmax - zmin) <= 2.0 ** -40 * np.maximum(1.0, np.maximum(np.abs(zmin), np.abs(zmax)))).sum())


def cpu_baseline(args, seconds, threads):
    """time the CPU oracle (reference order) on repeats of the config batch; returns traj/s"""
    import numpy as np
    import oracle_lib as ol
    world, batch, _ = workload(args, 0, 1)
    P = ol.default_params()
    P.max_iterations = args.iters
    sub = min(batch.B, 1024)          # the batch itself (config 2), or its first 1024 trajectories
    from trajectory_planner_amd import synth
    piece = synth.Batch(batch.ctrl[:sub], np.ascontiguousarray(batch.guide_off[:sub * batch.N + 1]), batch.guide_pv,
                        batch.guide_unk)
    if threads == 1:
        done, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            ol.optimize_batch(P, piece)
            done += sub
        return done / (time.perf_counter() - t0), done
    import multiprocessing as mp

    def work(q, secs):
        d, t = 0, time.perf_counter()
        while time.perf_counter() - t < secs:
            ol.optimize_batch(P, piece)
            d += sub
        q.put((d, time.perf_counter() - t))

    q = mp.Queue()
    ps = [mp.Process(target=work, args=(q, seconds)) for _ in range(threads)]
    t0 = time.perf_counter()
    [p.start() for p in ps]
    res = [q.get() for _ in ps]
    [p.join() for p in ps]
    wall = time.perf_counter() - t0
    return sum(d for d, _ in res) / wall, sum(d for d, _ in res)


def pmc_view(precision, name, kern_ms, waves, simds, lib_build_id, suffix=""):
    """Counter evidence for one quoted shape: profiles/pmc_<name>_<precision><suffix>.json (rocprofv3 --pmc passes of this
    very command, summarised by tools/summarize_pmc.py).  The counters are a committed record, not measured in this run:
    `pmc_stale` says whether the solve kernels' sources have changed since they were taken (vigo_build_id's solver hash
    against the file's)."""
    path = os.path.join(ROOT, "profiles", f"pmc_{name}_{precision}{suffix}.json")
    if not os.path.exists(path):
        return {"pmc_source": None}
    try:
        pmc = json.load(open(path))
    except Exception:
        return {"pmc_source": None}
    cnt = pmc.get("counters_per_launch", {})
    valu = cnt.get("SQ_INSTS_VALU")
    kernel_cycles = kern_ms * 1e-3 * SHADER_CLOCK_GHZ * 1e9
    traffic = pmc.get("hbm_bytes_per_launch")
    solver_now = lib_build_id.split()[0] if lib_build_id else None
    solver_then = str(pmc.get("build_id", "")).split()[0] if pmc.get("build_id") else None
    out = {"traffic": traffic,
           "issue_frac": (valu * VALU_ISSUE_CYCLES / (simds * kernel_cycles)) if valu else None,
           "issue_frac_of_occupied_simds": (valu * VALU_ISSUE_CYCLES / (min(waves, simds) * kernel_cycles)) if valu else None,
           "valu_insts_per_wave": (valu / cnt["SQ_WAVES"]) if valu and cnt.get("SQ_WAVES") else None,
           "measured_hbm_frac": (traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
           "pmc_source": os.path.relpath(path, ROOT) + " (" + str(pmc.get("build_id") or pmc.get("build", "build not recorded")) + ")",
           "pmc_stale": solver_then != solver_now}
    if cnt.get("SQ_ACTIVE_INST_VALU") and cnt.get("SQ_WAVE_CYCLES"):
        # share of the waves' lifetime during which a VALU instruction of theirs was executing
        out["valu_busy_of_wave_lifetime"] = cnt["SQ_ACTIVE_INST_VALU"] / cnt["SQ_WAVE_CYCLES"]
    if cnt.get("SQ_WAIT_INST_ANY") and cnt.get("SQ_WAVE_CYCLES"):
        out["memory_wait_of_wave_lifetime"] = cnt["SQ_WAIT_INST_ANY"] / cnt["SQ_WAVE_CYCLES"]
        out["wait_inst_any_wave_cycles"] = cnt["SQ_WAIT_INST_ANY"]
    return out


def main():
    args = parse()
    if args.cpu_baseline_only:
        # child process: never touches the GPU
        # the GPU box gives one GPU a 16-core CPU share; never start more workers than that
        threads = max(1, min(len(os.sched_getaffinity(0)), 16))
        one, n1 = cpu_baseline(args, args.cpu_seconds, 1)
        allc, na = cpu_baseline(args, max(4.0, args.cpu_seconds / 2), threads)
        print(json.dumps({"single": one, "single_n": n1, "all": allc, "all_n": na, "threads": threads}))
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size != args.gpus and world_size > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}")

    # CPU baseline first, in a child that never initialises HIP (rank 0, N = 1 only)
    cpu = None
    if rank == 0 and world_size == 1 and not args.no_cpu_baseline:
        child = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--workload", args.workload,
                                "--iters", str(args.iters), "--cpu-seconds", str(args.cpu_seconds)] +
                               (["--batch", str(args.batch)] if args.batch else []),
                               capture_output=True, text=True, check=True)
        cpu = json.loads(child.stdout.strip().splitlines()[-1])

    import numpy as np
    import torch
    import torch.distributed as dist
    from trajectory_planner_amd.vigo import PREC_F32, PREC_F64, Vigo, default_params

    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world_size > 1 or args.force_collectives
    if use_dist:
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            if world_size == 1:      # --force-collectives outside torch.distributed.run: a rendezvous of one
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29577")
                os.environ.setdefault("RANK", "0")
                os.environ.setdefault("WORLD_SIZE", "1")
            dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI

    # Rehearsal (gloo, every rank on cuda:0) takes the SAME tensor placement as the real runs — device tensors handed
    # to the collectives — so the 2-rank test on a one-GPU box exercises the code path RCCL will see; host staging is
    # only the fallback for a gloo build without device-tensor support (VIGO_REHEARSE_HOST_STAGING=1).
    host_staging = args.rehearse_on_one_gpu and os.environ.get("VIGO_REHEARSE_HOST_STAGING") == "1"

    def bcast(t):
        if host_staging:
            h = t.cpu()
            dist.broadcast(h, src=0)
            t.copy_(h)
        else:
            dist.broadcast(t, src=0)

    def max_over_ranks(x):
        te = torch.tensor([x], dtype=torch.float64, device="cpu" if host_staging else dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        return float(te.item())

    world, batch, wname = workload(args, rank, world_size)
    P = default_params()
    P.max_iterations = args.iters
    prec_code = {"f32": PREC_F32, "f64": PREC_F64, "f64_fast": 2}
    v = Vigo(local_rank, P, prec_code[args.precision])
    v.use_current_stream()
    from trajectory_planner_amd.vigo import SolveResult
    T = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    def install_map(vv, world_):
        """voxel snapshot: rank 0 packs, ONE RCCL broadcast over xGMI, every rank adopts it; returns (packed, ms)"""
        dims_ = world_.voxels.shape
        nwords = vv._lib.vigo_grid_packed_bytes(*dims_) // 4
        if rank == 0:
            packed_ = vv.pack_grid(torch.from_numpy(world_.voxels).to(dev))
        else:
            packed_ = torch.empty(nwords, dtype=torch.int32, device=dev)
        ms = 0.0
        if use_dist:
            torch.cuda.synchronize()
            dist.barrier()
            t_b0 = time.perf_counter()
            bcast(packed_)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t_b0) * 1e3
        vv.set_grid_packed(packed_, dims_, world_.origin, world_.res)
        return packed_, ms

    def timed_solves(vv, batch_, steps, warmup):
        """`steps` passes of the hot path over a resident batch (restore the control points D2D, isUnknown of the
        guide points, the whole solve in one launch), barrier + synchronize on both sides, MAX over ranks;
        HIP events around the solve launch on its stream.  Returns (elapsed s, kernel ms, SolveResult, buffers)."""
        ctrl0_, goff_, gpv_ = T(batch_.ctrl), T(batch_.guide_off), T(batch_.guide_pv)
        work_ = ctrl0_.clone()
        B_, N_ = batch_.B, batch_.N
        res_ = SolveResult(work_, torch.empty(B_, N_ - 6, 3, dtype=torch.float64, device=dev),
                           torch.empty(B_, dtype=torch.int32, device=dev), torch.empty(B_, dtype=torch.float64, device=dev),
                           torch.empty(B_, dtype=torch.int32, device=dev), torch.empty(B_, dtype=torch.int32, device=dev))
        e0 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        e1 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]

        def one(i=None):
            work_.copy_(ctrl0_)
            gunk = vv.guides_unknown(gpv_) if gpv_.shape[0] else None
            if i is not None:
                e0[i].record()
            vv.optimize(work_, goff_, gpv_ if gpv_.shape[0] else None, gunk, inplace=True, out=res_)
            if i is not None:
                e1[i].record()

        for _ in range(warmup):
            one()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        t_0 = time.perf_counter()
        for i in range(steps):
            one(i)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        el = time.perf_counter() - t_0
        if use_dist:
            el = max_over_ranks(el)
        return el, float(np.mean([x.elapsed_time(y) for x, y in zip(e0, e1)])), res_, (ctrl0_, goff_, gpv_, work_, one)

    packed, bcast_ms = install_map(v, world)
    dims = world.voxels.shape
    snapshot_ok = None
    if use_dist:
        # every rank answers the same seeded point queries from the snapshot it adopted; the answers must agree
        qp = T(np.random.default_rng(99).uniform(world.origin.min() - 0.5, -world.origin.min() + 0.5, size=(4096, 3)))
        sig = torch.stack([v.query_points(qp, w).to(torch.float64) @ torch.arange(1, 4097, dtype=torch.float64, device=dev)
                           for w in (0, 1)])
        if host_staging:
            sig = sig.cpu()
        lo, hi = sig.clone(), sig.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        snapshot_ok = bool(torch.equal(lo, hi) and float(sig.sum()) > 0)
    B, N = batch.B, batch.N
    n = 3 * (N - 6)
    elapsed, kern_ms, res, (ctrl0, goff, gpv, work, step) = timed_solves(v, batch, args.steps, args.warmup)
    iters = res.iters.cpu().numpy()      # of the timed configuration (the side measurements below overwrite res)
    evals = res.evals.cpu().numpy()

    simds = 4 * torch.cuda.get_device_properties(dev).multi_processor_count
    build_id = v._lib.vigo_build_id().decode()
    # ---- side measurements (outside the timed region above; rank 0 of a single-GPU run only) ----
    extra = {}
    if rank == 0 and world_size == 1 and not args.no_extras:
        k2 = max(10, args.steps // 4)
        # (a) the boundary handing over HOST buffers: pinned H2D of control points + guide pairs,
        #     solve, D2H of control points and status — the PCIe-inclusive rate (never `value`)
        h_ctrl = torch.from_numpy(np.ascontiguousarray(batch.ctrl)).pin_memory()
        h_gpv = torch.from_numpy(np.ascontiguousarray(batch.guide_pv)).pin_memory()
        h_goff = torch.from_numpy(np.ascontiguousarray(batch.guide_off)).pin_memory()
        h_out = torch.empty_like(h_ctrl).pin_memory()
        h_st = torch.empty(B, dtype=torch.int32).pin_memory()

        def host_step():
            work.copy_(h_ctrl, non_blocking=True)
            gpv.copy_(h_gpv, non_blocking=True)
            goff.copy_(h_goff, non_blocking=True)
            gunk = v.guides_unknown(gpv) if gpv.shape[0] else None
            v.optimize(work, goff, gpv if gpv.shape[0] else None, gunk, inplace=True, out=res)
            h_out.copy_(work, non_blocking=True)
            h_st.copy_(res.status, non_blocking=True)
            torch.cuda.synchronize()

        for _ in range(3):
            host_step()
        t1 = time.perf_counter()
        for _ in range(k2):
            host_step()
        dt = (time.perf_counter() - t1) / k2
        extra["pcie_inclusive"] = {"value": B / dt, "unit": "trajectories/s", "ms_per_step": dt * 1e3,
                                   "note": "pinned H2D of ctrl/guides + solve + D2H of ctrl/status, synchronous per batch"}
        # (b) the other fp64 arithmetic mode on the same batch (same 1e-4 parity gate, tests/test_gpu_solver.py)
        other = "f64_fast" if args.precision == "f64" else "f64"
        v.set_precision({"f32": PREC_F32, "f64": PREC_F64, "f64_fast": 2}[other])
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(k2):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / k2
        extra["other_mode"] = {"arithmetic": other, "value": B / dt, "unit": "trajectories/s", "ms_per_step": dt * 1e3}
        v.set_precision({"f32": PREC_F32, "f64": PREC_F64, "f64_fast": 2}[args.precision])
        # (b2) the level rule switched off (vigo_params_t.strict_z = 1): the general kernel alone, the reference's arithmetic on
        #      the z axis too — what a batch of trajectories with vertical structure costs, and what this batch cost before round 3
        P.strict_z = 1
        v.set_params(P)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(k2):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / k2
        extra["level_rule_off"] = {"value": B / dt, "unit": "trajectories/s", "ms_per_step": dt * 1e3,
                                   "note": "strict_z = 1: every trajectory through the general (x, y, z) kernel; never `value`"}
        P.strict_z = 0
        v.set_params(P)
        # (c) fixed work: g_epsilon = 0 switches the convergence exit off, every trajectory runs all iterations
        #     (SURVEY.md §8(d) asks for both; `value` above is the reference-faithful g_epsilon = 0.01)
        geps = float(P.g_epsilon)
        P.g_epsilon = 0.0
        v.set_params(P)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(k2):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / k2
        extra["fixed_work_g_epsilon_0"] = {"value": B / dt, "unit": "trajectories/s", "ms_per_step": dt * 1e3,
                                            "mean_iters": float(res.iters.float().mean().item())}
        P.g_epsilon = geps
        v.set_params(P)
        # (d) two batches in flight: a 1024-trajectory launch is 512 wavefronts on a chip of 1024 SIMDs, so a
        #     planner service keeps a second, independent batch running on another HIP stream (own handle)
        lanes = []
        for _ in range(2):
            st = torch.cuda.Stream(dev)
            with torch.cuda.stream(st):
                vv = Vigo(local_rank, P, {"f32": PREC_F32, "f64": PREC_F64, "f64_fast": 2}[args.precision])
                vv.use_current_stream()
                vv.set_grid_packed(packed, dims, world.origin, world.res)
                wk = ctrl0.clone()
                rs = SolveResult(wk, torch.empty_like(res.x), torch.empty_like(res.status), torch.empty_like(res.fx),
                                 torch.empty_like(res.iters), torch.empty_like(res.evals))
            lanes.append((st, vv, wk, rs))
        torch.cuda.synchronize()

        def lane_step(i):
            st, vv, wk, rs = lanes[i % 2]
            with torch.cuda.stream(st):
                wk.copy_(ctrl0)
                gunk = vv.guides_unknown(gpv) if gpv.shape[0] else None
                vv.optimize(wk, goff, gpv if gpv.shape[0] else None, gunk, inplace=True, out=rs)

        for i in range(6):
            lane_step(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(2 * k2):
            lane_step(i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / (2 * k2)
        same = bool(torch.equal(lanes[0][3].fx, lanes[1][3].fx) and torch.equal(lanes[0][2], lanes[1][2]))
        extra["two_batches_in_flight"] = {"value": B / dt, "unit": "trajectories/s", "ms_per_step": dt * 1e3,
                                          "streams": 2, "results_identical": same,
                                          "note": "independent batches alternating over two HIP streams; never `value`"}
        for _, vv, _, _ in lanes:
            vv.close()
        # (e) the same build on a batch that fills the chip: 16 384 x 32 on the same map (8192 waves on 1024 SIMDs)
        if args.workload == "config2":
            from trajectory_planner_amd import synth
            big = synth.make_bspline_batch(world, 16384, 32, synth.SEED_BASE + 2 + 5000)
            kb = max(5, args.steps // 20)
            el, kms, r_big, _ = timed_solves(v, big, kb, 2)
            extra["config2_at_16384"] = {"value": 16384 * kb / el, "unit": "trajectories/s", "kernel_ms": kms,
                                         "simd_occupancy": min(1.0, (16384 // 2) / simds), "note": "same map, B = 16384 x 32; never `value`"}
            extra["config2_at_16384"].update(pmc_view(args.precision, "config2", kms, 16384 // 2, simds, build_id, "_b16384"))
            del big, r_big
            # (e2) configs[1] on guides from the planner's OWN host pipeline (findCollisionSeg -> A* -> assignGuidePointsSemiCircle,
            #      bsplineTraj.cpp:403-571, through libtrajectory_planner_vigo.so — product code) in a world dense enough that
            #      over half the trajectories cross an obstacle; then the same trajectories two re-guides later
            #      (bsplineTraj.cpp:640-648 replayed on the optimizer's output: pairs are APPENDED, the next solve starts
            #      from the moved control points).  Never `value`.
            try:
                pw = synth.make_pipeline_world()
                vp = Vigo(local_rank, P, prec_code[args.precision])
                vp.use_current_stream()
                vp.set_grid(T(pw.voxels), pw.origin, pw.res)
                pb = synth.make_pipeline_batch(pw, 1024, 32, synth.SEED_BASE + 2 + 2000)
                rec = {"workload": "configs[1] shape (1024 x 32, 256^3, %d iterations) in the dense world, guides from the host pipeline" % args.iters,
                       "stages": []}
                for stage in range(3):
                    hist, per_traj, share = synth.pairs_histogram(pb)
                    elp, kmsp, rp, _ = timed_solves(vp, pb, kb * 2, 2)
                    rec["stages"].append({"reguides": stage, "guide_pairs": int(pb.guide_pv.shape[0]),
                                          "pairs_per_trajectory_mean": float(per_traj.mean()), "pairs_per_trajectory_max": int(per_traj.max()),
                                          "trajectories_with_guides": share, "max_pairs_per_point": int(len(hist) - 1),
                                          "pairs_per_free_point_histogram": [int(x) for x in hist],
                                          "kernel_ms": kmsp, "value": 1024 * kb * 2 / elp, "unit": "trajectories/s",
                                          "mean_iters": float(rp.iters.float().mean().item()), "mean_evals": float(rp.evals.float().mean().item()),
                                          # the launch lasts as long as its slowest wave: the tail of the line-search evaluations
                                          "max_evals": int(rp.evals.max().item()), "evals_p99": float(torch.quantile(rp.evals.float(), 0.99).item())})
                    if stage < 2:
                        pb = synth.reguide_batch(pw, pb, rp.ctrl.cpu().numpy())
                rec["kernel_ms"] = rec["stages"][0]["kernel_ms"]
                rec["guide_pairs"] = rec["stages"][0]["guide_pairs"]
                extra["config2_pipeline_guides"] = rec
                vp.close()
                del pw, pb
            except (RuntimeError, OSError) as e:      # the host facade library is not built: say so, do not invent numbers
                extra["config2_pipeline_guides"] = {"error": str(e)}
            # (e3) BASELINE configs[2]: the min-snap corridor checker, 4096 segments x 10 000 samples on a 256 x 256 x 64 world
            #      (tools/time_corridor.py's input; parity at this size: tests/test_gpu_fullsize.py).  Never `value`.
            try:
                rng3 = np.random.default_rng(3)
                vox3 = np.zeros((256, 256, 64), dtype=np.uint8)
                for _ in range(300):
                    c3 = rng3.integers(8, 248, size=2); s3 = rng3.integers(1, 6, size=2)
                    vox3[c3[0] - s3[0]:c3[0] + s3[0], c3[1] - s3[1]:c3[1] + s3[1], 0:rng3.integers(10, 64)] |= 4
                unk3 = rng3.random((32, 32, 8)) < 0.05
                vox3[np.repeat(np.repeat(np.repeat(unk3, 8, 0), 8, 1), 8, 2)] |= 2
                v3 = Vigo(local_rank, P, prec_code[args.precision])
                v3.use_current_stream()
                v3.set_grid(T(vox3), np.array([-12.8, -12.8, -1.0]), 0.1)
                co3, ns3, dl3, _ = synth.make_corridor_segments(33, 4096, extent_lo=(-10, -10, 0.5), extent_hi=(10, 10, 2.5), n_samples=10000)
                dco, dns, ddl = T(co3), T(ns3), T(dl3)
                for _ in range(3):
                    fl3, _, _ = v3.corridor_check(dco, dns, ddl, [0.4, 0.4, 0.2], 0.2)
                torch.cuda.synchronize()
                t3 = time.perf_counter()
                for _ in range(20):
                    fl3, _, _ = v3.corridor_check(dco, dns, ddl, [0.4, 0.4, 0.2], 0.2)
                torch.cuda.synchronize()
                d3 = (time.perf_counter() - t3) / 20
                extra["config3_corridor"] = {"workload": "configs[2]: 4096 min-snap segments x 10 000 samples, box 0.4 / 0.4 / 0.2, map_resolution 0.2, 256 x 256 x 64 voxels",
                                             "ms": d3 * 1e3, "value": 4096e4 / d3, "unit": "samples/s", "colliding_segments": int(fl3.sum().item()),
                                             "note": "three launches per call (per-segment clock tables, certified spans, the walk for what they refuse); "
                                                     "counters: profiles/r03_pmc_corridor_pass0.json; never `value`"}
                v3.close()
                del vox3, dco
            except (RuntimeError, OSError) as e:
                extra["config3_corridor"] = {"error": str(e)}
    # (f) one GPU's shard of BASELINE configs[3] (8192 x 64 control points, 512^3 map) on every rank count, so the
    #     1 -> 8 record covers the configuration BASELINE names for 8 GPUs; never `value`
    if args.workload == "config2" and not args.no_extras:
        from trajectory_planner_amd import sharding, synth
        args4 = argparse.Namespace(**vars(args))
        args4.workload, args4.batch = "config4", 0
        world4, batch4, name4 = workload(args4, rank, world_size)
        v4 = Vigo(local_rank, P, prec_code[args.precision])
        v4.use_current_stream()
        _, bcast4_ms = install_map(v4, world4)
        k4 = max(5, args.steps // 20)
        el4, kms4, r4, _ = timed_solves(v4, batch4, k4, 2)
        extra["config4_shard"] = {"workload": name4, "value": batch4.B * world_size * k4 / el4, "unit": "trajectories/s", "n_gpus": world_size,
                                  "ms_per_step": el4 / k4 * 1e3, "kernel_ms": kms4, "map_bcast_ms": bcast4_ms, "steps": k4,
                                  "simd_occupancy": min(1.0, batch4.B / simds), "mean_iters": float(r4.iters.float().mean().item()),
                                  "note": "whole-job aggregate over all ranks (max-over-ranks time); never `value`"}
        extra["config4_shard"].update(pmc_view(args.precision, "config4", kms4, batch4.B, simds, build_id))
        del batch4, r4
        # (g) BASELINE configs[3] as ONE batch: 65 536 x 64 trajectories seeded once (every rank generates the same batch),
        #     cut into contiguous slices with sharding.shard_range / slice_batch, rank r solves slice r — strong scaling of
        #     the named configuration; at N = 1 the whole batch runs on the one GPU.  Max-over-ranks time, aggregate rate.
        B4 = 65536
        full4 = synth.make_bspline_batch(world4, B4, 64, synth.SEED_BASE + 4 + 7000, start_range=16.0)
        lo4, hi4 = sharding.shard_range(B4, rank, world_size)
        mine4 = sharding.slice_batch(full4, lo4, hi4)
        del full4
        ks = max(3, args.steps // 40)
        els, kmss, rs4, _ = timed_solves(v4, mine4, ks, 1)
        bounds = [list(sharding.shard_range(B4, r, world_size)) for r in range(world_size)]
        extra["config4_strong"] = {"workload": f"configs[3]: ONE batch of {B4} B-spline trajs x 64 ctrl pts, 512^3 voxel grid, {args.iters} L-BFGS iters, "
                                               f"cut into {world_size} contiguous slice(s)",
                                   "value": B4 * ks / els, "unit": "trajectories/s", "n_gpus": world_size, "scaling": "strong",
                                   "ms_per_step": els / ks * 1e3, "kernel_ms_rank0": kmss, "steps": ks, "slice_bounds": bounds,
                                   "slice_rank0": [lo4, hi4], "mean_iters_rank0": float(rs4.iters.float().mean().item()),
                                   "note": "whole-job aggregate (max-over-ranks time); one seeded batch sliced by rank; never `value`"}
        v4.close()
        del world4, mine4, rs4
    gp = np.diff(batch.guide_off).reshape(B, N).sum(1)
    elem = 4 if args.precision == "f32" else 8
    alg_bytes = float(algorithmic_bytes(n, P.mem_size, iters, evals, gp, elem).sum())
    compulsory = float(B * (2 * 3 * N * 8 + 3 * (N - 6) * 8 + 24) + gpv.numel() * 8 + gpv.shape[0])

    if rank == 0:
        total = B * world_size * args.steps
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        tpb = 2 if N <= 32 else 1                       # trajectories per solver wavefront (vigo_solver.hip)
        waves = (B + tpb - 1) // tpb
        pv = pmc_view(args.precision, args.workload, kern_ms, waves, simds, build_id) if not args.batch else {"pmc_source": None}
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": pv.get("traffic"),
                "kernel": "vigo::k_optimize (two launches per solve since round 3: first the level instantiation <..., D = 2>, "
                          "which does the work on this batch, then the general instantiation, whose waves of level trajectories "
                          "exit at once; kernel_ms = HIP events around both)",
                "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": alg_bytes,
                "byte_model": "SURVEY.md §8(d) streaming model x fp64 (state streamed per BLAS-1 pass, as the CPU "
                              "reference does); the kernel keeps that state in LDS/VGPRs, so HBM sees only the "
                              "compulsory bytes below and `frac` is a model figure, not what binds the kernel",
                "compulsory_bytes_per_launch": compulsory,
                "compulsory_frac": compulsory / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                # the roofline that binds: VALU issue.  One wave64 VALU instruction holds its SIMD's issue port for
                # 4 cycles, so the chip retires at most SIMDs / 4 wave-instructions per cycle.
                "binding": "valu_issue",
                "simds": simds, "waves_per_launch": waves, "simd_occupancy": min(1.0, waves / simds),
                "shader_clock_ghz": SHADER_CLOCK_GHZ, "library_build_id": build_id}
        roof.update({k: x for k, x in pv.items() if k != "traffic"})
        out = {
            "metric": "B-spline trajs/s (32 ctrl pts, 256^3 grid, 50 iters) @1 GPU; % HBM roofline"
            if args.workload == "config2" else "B-spline trajs/s (64 ctrl pts, 512^3 grid, 50 iters)",
            "value": total / elapsed,
            "unit": "trajectories/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.precision == "f32" else "f64",
            "arithmetic": args.precision,
            "data": "synthetic (seeded straight paths + box world, SURVEY.md §8d; no dataset exists for this path)"
                    + (" — REHEARSAL of the rank logic on one GPU, not a measurement" if args.rehearse_on_one_gpu else ""),
            "config": {"workload": wname, "trajs_per_gpu": B, "ctrl_pts": N, "lbfgs_iters": args.iters,
                       "mem_size": int(P.mem_size), "g_epsilon": float(P.g_epsilon), "grid": list(dims),
                       "guide_pairs_per_gpu": int(gpv.shape[0]), "sharding": f"batch-dp{world_size}, no data-path collective",
                       "mean_iters": float(iters.mean()), "mean_evals": float(evals.mean()),
                       # how many of the batch's trajectories the level rule (include/vigo.h) applies to: those run the D = 2 kernel
                       "level_trajectories_per_gpu": int(level_count(batch))},
            "map_bcast_ms": bcast_ms,
            "map_snapshot_identical_on_all_ranks": snapshot_ok,
            "collectives": None if not use_dist else {
                "backend": dist.get_backend(), "world_size": dist.get_world_size(), "forced_at_world_size_1": bool(args.force_collectives and world_size == 1),
                "calls": ["barrier", "broadcast(int32 packed snapshot, %d words)" % packed.numel(), "all_reduce(MIN, f64)", "all_reduce(MAX, f64)"],
                "rccl_version": ".".join(str(x) for x in torch.cuda.nccl.version()) if dist.get_backend() == "nccl" else None},
            "roofline": roof,
        }
        out.update(extra)
        if cpu is not None:
            out["cpu_baseline"] = {"value": cpu["single"], "unit": "trajectories/s", "cores": 1, "kind": "port",
                                   "sample": f"{cpu['single_n']} solves: repeats of the batch itself (its first 1024 trajectories), "
                                             f"oracle/vigo_oracle.c (fp64, reference order, gcc -O3) on 1 host thread.  Caveat: the port "
                                             f"does not make the reference's five 3xN heap allocations per evaluation "
                                             f"(bsplineTraj.cpp:807-810,817) nor its per-call L-BFGS mallocs (lbfgs.hpp:1107-1123), so it "
                                             f"is a slightly optimistic stand-in for the reference's optimize()"}
            out["cpu_baseline_allcores"] = {"value": cpu["all"], "unit": "trajectories/s", "cores": cpu["threads"],
                                            "kind": "port", "sample": f"{cpu['all_n']} solves, one process per host core"}
        print(json.dumps(out))
    v.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
