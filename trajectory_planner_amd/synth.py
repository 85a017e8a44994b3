"""Synthetic workloads of the shapes BASELINE.json / SURVEY.md §8(d) name (numpy only).

Everything is seeded (numpy PCG64, seed = 0x5EED0000 + config id) and uses the parameter
values of the reference's cfg files.  There is no dataset to download: the reference's hot path
consumes control points, guide pairs, obstacles and a voxel map, all of which are generated
here in the layouts include/vigo.h defines.
"""
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

SEED_BASE = 0x5EED0000

# cfg/bspline_interactive/occupancy_map.yaml:9 robot size [0.8,0.8,0.3] -> inflation half-sizes
ROBOT_HALF = np.array([0.4, 0.4, 0.15])
CTRL_SPACING = 0.25   # controlPointDistance_, bsplineTraj.h:46
CTRL_TS = 0.2         # controlPointsTs_, bsplineTraj.h:47


def fit_matrix(K: int, ts: float) -> np.ndarray:
    """The (K+4)x(K+2) system of bspline::parameterizeToBspline (bspline.cpp:95-110)."""
    A = np.zeros((K + 4, K + 2))
    for i in range(K):
        A[i, i:i + 3] = np.array([1.0, 4.0, 1.0]) / 6.0
    A[K, 0:3] = np.array([-1.0, 0.0, 1.0]) / 2.0 / ts
    A[K + 1, K - 1:K + 2] = np.array([-1.0, 0.0, 1.0]) / 2.0 / ts
    A[K + 2, 0:3] = np.array([1.0, -2.0, 1.0]) / ts / ts
    A[K + 3, K - 1:K + 2] = np.array([1.0, -2.0, 1.0]) / ts / ts
    return A


def fit_control_points(points: np.ndarray, ts: float = CTRL_TS, cond: Optional[np.ndarray] = None) -> np.ndarray:
    """Least-squares fit of waypoints [B,K,3] (+ start/end vel/acc [B,4,3], default 0) to
    control points [B,K+2,3] (bspline.cpp:74-138; numpy lstsq stands in for Eigen's QR)."""
    B, K, _ = points.shape
    A = fit_matrix(K, ts)
    rhs = np.zeros((B, K + 4, 3))
    rhs[:, :K] = points
    if cond is not None:
        rhs[:, K:] = cond
    pinv = np.linalg.pinv(A)
    return np.ascontiguousarray(np.einsum("ij,bjk->bik", pinv, rhs))


@dataclass
class World:
    voxels: np.ndarray          # uint8 [nx,ny,nz]: bit0 inflated-occ, bit1 unknown, bit2 occupied
    origin: np.ndarray          # [3]
    res: float
    boxes: np.ndarray           # [nb,6] centre xyz, half-extent xyz (un-inflated)
    box_id: Optional[np.ndarray] = None  # int16 [nx,ny,nz] index of an inflated box covering the voxel, -1 none


def make_box_world(seed: int, n: int = 256, res: float = 0.1, n_boxes: int = 200, centre_range: float = 12.0,
                   unknown_frac: float = 0.10, z_range: Optional[float] = None, keep_ids: bool = True) -> World:
    """n^3 grid centred on the origin with axis-aligned boxes: centres U(-r,r)^3 (z spread
    z_range, default r), half-extents U(0.2,1.0)^3, inflated by the robot half size; 10 % of the
    8^3 bricks unknown."""
    rng = np.random.default_rng(seed)
    origin = np.array([-n * res / 2, -n * res / 2, -n * res / 2])
    vox = np.zeros((n, n, n), dtype=np.uint8)
    ids = np.full((n, n, n), -1, dtype=np.int16) if keep_ids else None
    c = rng.uniform(-centre_range, centre_range, size=(n_boxes, 3))
    if z_range is not None:
        c[:, 2] = rng.uniform(-z_range, z_range, size=n_boxes)
    h = rng.uniform(0.2, 1.0, size=(n_boxes, 3))

    def idx_range(lo, hi):
        i0 = np.clip(np.ceil((lo - origin) / res - 0.5).astype(int), 0, n)       # voxel centres inside
        i1 = np.clip(np.floor((hi - origin) / res - 0.5).astype(int) + 1, 0, n)
        return i0, i1

    for b in range(n_boxes):
        i0, i1 = idx_range(c[b] - h[b], c[b] + h[b])
        vox[i0[0]:i1[0], i0[1]:i1[1], i0[2]:i1[2]] |= 4
        j0, j1 = idx_range(c[b] - h[b] - ROBOT_HALF, c[b] + h[b] + ROBOT_HALF)
        vox[j0[0]:j1[0], j0[1]:j1[1], j0[2]:j1[2]] |= 1
        if ids is not None:
            ids[j0[0]:j1[0], j0[1]:j1[1], j0[2]:j1[2]] = b
    # unknown: a fraction of the 8^3 bricks
    nb = n // 8
    mask = rng.random((nb, nb, nb)) < unknown_frac
    unk = np.repeat(np.repeat(np.repeat(mask, 8, 0), 8, 1), 8, 2)
    vox[unk] |= 2
    return World(vox, origin, res, np.concatenate([c, h], axis=1), ids)


@dataclass
class Batch:
    ctrl: np.ndarray                 # [B,N,3] f64
    guide_off: np.ndarray            # [B*N+1] i32
    guide_pv: np.ndarray             # [G,6] f64
    guide_unk: np.ndarray            # [G] u8
    obs_off: Optional[np.ndarray] = None   # [B+1] i32
    obs: Optional[np.ndarray] = None       # [O,9] f64
    weights: Optional[np.ndarray] = None   # [B,4]
    meta: dict = field(default_factory=dict)

    @property
    def B(self):
        return self.ctrl.shape[0]

    @property
    def N(self):
        return self.ctrl.shape[1]


def voxel_index(world: World, pts: np.ndarray) -> np.ndarray:
    return np.floor((pts - world.origin) / world.res).astype(np.int64)


def lookup(world: World, pts: np.ndarray, bit: int) -> np.ndarray:
    """bit of the voxel byte at pts [...,3]; outside the box -> 1 (contract of include/vigo.h)."""
    idx = voxel_index(world, pts)
    n = np.array(world.voxels.shape)
    inside = np.all((idx >= 0) & (idx < n), axis=-1)
    ic = np.clip(idx, 0, n - 1)
    v = (world.voxels[ic[..., 0], ic[..., 1], ic[..., 2]] >> bit) & 1
    return np.where(inside, v, 1).astype(np.uint8)


def make_bspline_batch(world: World, B: int, N: int, seed: int, start_range: float = 8.0, jitter: float = 0.05,
                       n_obs: int = 0, guide2_prob: float = 0.3, z_jitter: float = 0.0, z_share: float = 0.5) -> Batch:
    """Straight 0.25 m-spaced paths with lateral jitter, fitted to N control points; guide pairs
    for the free control points that start inside an inflated obstacle.  z_jitter > 0: a share
    z_share of the trajectories (chosen by a second generator, so the rest of the batch is unchanged) also gets
    vertical jitter N(0, z_jitter^2) per path point — trajectories that are NOT level (the kernels' level rule)."""
    rng = np.random.default_rng(seed)
    K = N - 2
    start = np.concatenate([rng.uniform(-start_range, start_range, size=(B, 2)), np.full((B, 1), 1.0)], axis=1)
    heading = rng.uniform(0.0, 2 * np.pi, size=B)
    dirv = np.stack([np.cos(heading), np.sin(heading), np.zeros(B)], axis=1)
    lat = np.stack([-np.sin(heading), np.cos(heading), np.zeros(B)], axis=1)
    s = np.arange(K) * CTRL_SPACING
    pts = start[:, None, :] + s[None, :, None] * dirv[:, None, :]
    pts = pts + rng.normal(0.0, jitter, size=(B, K, 1)) * lat[:, None, :]
    if z_jitter > 0.0:
        rz = np.random.default_rng(seed + 77)
        wavy = rz.random(B) < z_share
        pts[:, :, 2] += rz.normal(0.0, z_jitter, size=(B, K)) * wavy[:, None]
    ctrl = fit_control_points(pts, CTRL_TS)

    # guide pairs: p = c + (penetration + 0.3) * u, v = u (bsplineTraj.cpp:532: direction points
    # from the control point to its guide point), u = nearest xy face normal of the inflated box
    interior = np.zeros((B, N), dtype=bool)
    interior[:, 3:N - 3] = True
    occ = lookup(world, ctrl, 0).astype(bool) & interior
    counts = np.zeros((B, N), dtype=np.int64)
    bi, pi = np.nonzero(occ)
    pv_list = []
    if bi.size:
        c = ctrl[bi, pi]
        idx = np.clip(voxel_index(world, c), 0, np.array(world.voxels.shape) - 1)
        box = world.box_id[idx[:, 0], idx[:, 1], idx[:, 2]]
        ok = box >= 0
        bi, pi, c, box = bi[ok], pi[ok], c[ok], box[ok]
        bc = world.boxes[box, :3]
        bh = world.boxes[box, 3:] + ROBOT_HALF
        pen = np.stack([c[:, 0] - (bc[:, 0] - bh[:, 0]), (bc[:, 0] + bh[:, 0]) - c[:, 0],
                        c[:, 1] - (bc[:, 1] - bh[:, 1]), (bc[:, 1] + bh[:, 1]) - c[:, 1]], axis=1)
        face = np.argmin(pen, axis=1)
        normals = np.array([[-1.0, 0, 0], [1.0, 0, 0], [0, -1.0, 0], [0, 1.0, 0]])
        u = normals[face]
        depth = pen[np.arange(len(face)), face]
        p1 = c + (depth + 0.3)[:, None] * u
        second = rng.random(len(face)) < guide2_prob
        ang = rng.uniform(-0.5, 0.5, size=len(face))
        u2 = np.stack([u[:, 0] * np.cos(ang) - u[:, 1] * np.sin(ang),
                       u[:, 0] * np.sin(ang) + u[:, 1] * np.cos(ang), np.zeros(len(face))], axis=1)
        p2 = c + (depth + 0.4)[:, None] * u2
        counts[bi, pi] = 1 + second.astype(np.int64)
        # emit in (b, i, j) order
        order = np.lexsort((pi, bi))
        for o in order:
            pv_list.append(np.concatenate([p1[o], u[o]]))
            if second[o]:
                pv_list.append(np.concatenate([p2[o], u2[o]]))
    guide_off = np.zeros(B * N + 1, dtype=np.int32)
    guide_off[1:] = np.cumsum(counts.reshape(-1))
    guide_pv = np.array(pv_list, dtype=np.float64).reshape(-1, 6) if pv_list else np.zeros((0, 6))
    guide_unk = lookup(world, guide_pv[:, :3], 1) if len(guide_pv) else np.zeros(0, dtype=np.uint8)

    obs = obs_off = None
    if n_obs > 0:
        # obstacles near each path (config 5: pos in map, vel U(-1,1)^2, size U(0.3,1.0)^2 x 1.7)
        mid = ctrl[:, N // 2]
        pos = mid[:, None, :] + rng.uniform(-2.0, 2.0, size=(B, n_obs, 3))
        pos[..., 2] = 1.0
        vel = np.concatenate([rng.uniform(-1.0, 1.0, size=(B, n_obs, 2)), np.zeros((B, n_obs, 1))], axis=2)
        size = np.concatenate([rng.uniform(0.3, 1.0, size=(B, n_obs, 2)), np.full((B, n_obs, 1), 1.7)], axis=2)
        obs = np.ascontiguousarray(np.concatenate([pos, vel, size], axis=2).reshape(B * n_obs, 9))
        obs_off = (np.arange(B + 1) * n_obs).astype(np.int32)
    return Batch(np.ascontiguousarray(ctrl), guide_off, np.ascontiguousarray(guide_pv), guide_unk.astype(np.uint8),
                 obs_off, obs, None, {"seed": seed, "K": K})


def config2(B: int = 1024, N: int = 32, grid: int = 256, n_boxes: int = 200, world: Optional[World] = None,
            n_obs: int = 0):
    """BASELINE.json configs[1]: 1024 trajectories x 32 control points, 256^3 voxel grid."""
    if world is None:
        world = make_box_world(SEED_BASE + 2, n=grid, n_boxes=n_boxes)
    return world, make_bspline_batch(world, B, N, SEED_BASE + 2 + 1000, n_obs=n_obs)


def config4(B: int = 65536, N: int = 64, grid: int = 512, n_boxes: int = 800, rank: int = 0, world_size: int = 1,
            world: Optional[World] = None):
    """BASELINE.json configs[3]: 65 536 x 64 control points, 512^3 grid; rank r gets the
    contiguous slice [r*B/ws, (r+1)*B/ws) (generated independently per rank, seed + rank)."""
    if world is None:
        world = make_box_world(SEED_BASE + 4, n=grid, n_boxes=n_boxes, centre_range=24.0)
    per = B // world_size
    return world, make_bspline_batch(world, per, N, SEED_BASE + 4 + 1000 + rank, start_range=16.0)


def make_corridor_segments(seed: int, S: int, deg: int = 7, extent_lo=(-8.0, -8.0, 0.5), extent_hi=(8.0, 8.0, 1.5),
                           n_samples: int = 10000):
    """Config 3: S smooth degree-`deg` segments (a straight chord <= 5 m plus a bounded random
    bend), duration U(1,5) s, n_samples samples each (delT = dur / n_samples)."""
    rng = np.random.default_rng(seed)
    lo, hi = np.array(extent_lo), np.array(extent_hi)
    p0 = rng.uniform(lo, hi, size=(S, 3))
    dirn = rng.normal(size=(S, 3))
    dirn[:, 2] *= 0.1
    dirn /= np.linalg.norm(dirn, axis=1, keepdims=True)
    length = rng.uniform(0.5, 5.0, size=S)
    dur = rng.uniform(1.0, 5.0, size=S)
    coeffs = np.zeros((S, 3, deg + 1))
    coeffs[:, :, 0] = p0
    coeffs[:, :, 1] = dirn * (length / dur)[:, None]
    # bend: bounded higher-order terms, scaled so each contributes <= ~0.3 m over the duration
    for d in range(2, deg + 1):
        amp = rng.uniform(-0.3, 0.3, size=(S, 3)) / (d - 1)
        amp[:, 2] *= 0.2
        coeffs[:, :, d] = amp / (dur[:, None] ** d)
    n_samp = np.full(S, n_samples, dtype=np.int32)
    delT = dur / n_samples
    return np.ascontiguousarray(coeffs), n_samp, np.ascontiguousarray(delT), dur


def sphere_esdf(n: int, res: float, centre, radius: float):
    """Analytic signed distance to a sphere sampled at voxel centres (float32 [n,n,n])."""
    origin = np.array([-n * res / 2] * 3)
    ax = origin[0] + (np.arange(n) + 0.5) * res
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    d = np.sqrt((X - centre[0]) ** 2 + (Y - centre[1]) ** 2 + (Z - centre[2]) ** 2) - radius
    return d.astype(np.float32), origin


def edt_esdf(world: World):
    """Config 5: fp32 signed distance field of a world's occupied voxels (bit2) by the exact Euclidean distance
    transform (scipy), positive outside, negative inside, in metres; samples at voxel centres.  Returns
    (float32 [nx,ny,nz], origin)."""
    from scipy import ndimage
    occ = (world.voxels & 4) != 0
    d = (ndimage.distance_transform_edt(~occ) - ndimage.distance_transform_edt(occ)) * world.res
    return np.ascontiguousarray(d.astype(np.float32)), np.array(world.origin, dtype=np.float64)


# ---- guides from the planner's own host pipeline (product code: libtrajectory_planner_vigo.so) ---------------------------
# cfg/bspline_interactive/bspline_planner_param.yaml: distance_threshold, min_height, max_height, max_obstacle_size
PIPELINE_CFG = np.array([0.5, 0.7, 1.3, 5.0, 5.0, 3.0])


def _host_lib():
    import ctypes as C
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libtrajectory_planner_vigo.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: build the host facade (make -C trajectory_planner_amd/host)")
    lib = C.CDLL(path)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    lib.vigo_host_bspline_guides_batch.restype = C.c_int
    lib.vigo_host_bspline_guides_batch.argtypes = [C.c_void_p, ip, dp, C.c_double, C.c_int, C.c_int, dp, dp, C.c_int, dp, dp, ip, ip, ip, dp,
                                                   C.c_longlong]
    return lib


def host_guides(world: World, N: int, paths: Optional[np.ndarray] = None, ctrl: Optional[np.ndarray] = None, cfg=PIPELINE_CFG):
    """bsplineTraj's host pipeline for a batch on one map (host/src/cabi_host.cpp: vigo_host_bspline_guides_batch).
    paths [n,K,3]: updatePath (fit) + makePlan()'s prologue findCollisionSeg -> A* -> assignGuidePointsSemiCircle
    (bsplineTraj.cpp:333-350, :403-571).  ctrl [n,N,3] instead: the re-guide step the rebound loop takes on its CURRENT
    control points (bsplineTraj.cpp:640-648), returning the pairs that step appends.
    Returns (ctrl [n,N,3], status [n], n_seg [n], guide_off [n*N+1], guide_pv [G,6])."""
    import ctypes as C
    lib = _host_lib()
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    vox = np.ascontiguousarray(world.voxels)
    dims = (C.c_int * 3)(*vox.shape)
    origin = np.ascontiguousarray(world.origin, dtype=np.float64)
    cfg = np.ascontiguousarray(cfg, dtype=np.float64)
    src = paths if ctrl is None else ctrl
    n = src.shape[0]
    src = np.ascontiguousarray(src, dtype=np.float64)
    ctrl_out = np.zeros((n, N, 3))
    status, n_seg = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
    goff = np.zeros(n * N + 1, dtype=np.int32)
    null = C.cast(None, dp)
    cap = 16 * n + 1024            # guide pairs the output can take; grown when the library says it does not fit (-2)
    while True:
        gpv = np.zeros((cap, 6))
        rc = lib.vigo_host_bspline_guides_batch(vox.ctypes.data_as(C.c_void_p), dims, origin.ctypes.data_as(dp), float(world.res), n,
                                                src.shape[1] if ctrl is None else 0, src.ctypes.data_as(dp) if ctrl is None else null,
                                                src.ctypes.data_as(dp) if ctrl is not None else null, N, cfg.ctypes.data_as(dp),
                                                ctrl_out.ctypes.data_as(dp), status.ctypes.data_as(ip), n_seg.ctypes.data_as(ip),
                                                goff.ctypes.data_as(ip), gpv.ctypes.data_as(dp), cap)
        if rc == -2 and cap < 64 * n * N:
            cap *= 4
            continue
        break
    if rc != 0:
        raise RuntimeError(f"vigo_host_bspline_guides_batch failed ({rc})")
    return ctrl_out, status, n_seg, goff, np.ascontiguousarray(gpv[:goff[-1]])


def append_guides(goff_a, gpv_a, goff_b, gpv_b):
    """per control point: the pairs of list a, then the pairs of list b (assignGuidePointsSemiCircle push_backs)"""
    ca, cb = np.diff(goff_a), np.diff(goff_b)
    goff = np.zeros(len(goff_a), dtype=np.int32)
    goff[1:] = np.cumsum(ca + cb)
    gpv = np.zeros((int(goff[-1]), 6))
    # destination rows of a's and b's pairs
    ia = np.repeat(goff[:-1], ca) + (np.arange(len(gpv_a)) - np.repeat(goff_a[:-1], ca))
    ib = np.repeat(goff[:-1] + ca, cb) + (np.arange(len(gpv_b)) - np.repeat(goff_b[:-1], cb))
    gpv[ia] = gpv_a
    gpv[ib] = gpv_b
    return goff, gpv


def make_pipeline_world(seed: int = SEED_BASE + 2, n: int = 256, n_boxes: int = 110, z_range: float = 3.0) -> World:
    """config 2's box world with the boxes gathered around the flight height (z centres U(-z_range, z_range) instead of
    U(-12, 12)): dense enough that a third or more of 7 m straight paths at z = 1 cross an inflated box"""
    return make_box_world(seed, n=n, n_boxes=n_boxes, z_range=z_range)


def make_pipeline_batch(world: World, B: int, N: int, seed: int, start_range: float = 8.0, jitter: float = 0.05) -> Batch:
    """B plannable trajectories with the guide pairs the planner's own prologue gives them: straight jittered paths as
    in make_bspline_batch, whose start and goal are free; a candidate the pipeline refuses (goal occupied) or whose A*
    fails is replaced by the next one (the reference's makePlan() returns false there and never optimises)."""
    rng = np.random.default_rng(seed)
    K = N - 2
    s = np.arange(K) * CTRL_SPACING
    acc_ctrl, acc_cnt, acc_pv, acc_seg = [], [], [], []
    have, tried = 0, 0
    while have < B:
        M = max(256, 2 * (B - have))
        start = np.concatenate([rng.uniform(-start_range, start_range, size=(M, 2)), np.full((M, 1), 1.0)], axis=1)
        heading = rng.uniform(0.0, 2 * np.pi, size=M)
        dirv = np.stack([np.cos(heading), np.sin(heading), np.zeros(M)], axis=1)
        lat = np.stack([-np.sin(heading), np.cos(heading), np.zeros(M)], axis=1)
        pts = start[:, None, :] + s[None, :, None] * dirv[:, None, :]
        pts = pts + rng.normal(0.0, jitter, size=(M, K, 1)) * lat[:, None, :]
        free = (lookup(world, pts[:, 0], 0) == 0) & (lookup(world, pts[:, -1], 0) == 0)
        pts = pts[free]
        tried += M
        if len(pts) == 0:
            continue
        ctrl, status, n_seg, goff, gpv = host_guides(world, N, paths=pts)
        ok = np.nonzero(status == 0)[0][:B - have]
        cnt = np.diff(goff).reshape(-1, N)
        for t in ok:
            acc_ctrl.append(ctrl[t])
            acc_cnt.append(cnt[t])
            acc_pv.append(gpv[goff[t * N]:goff[(t + 1) * N]])
            acc_seg.append(n_seg[t])
        have += len(ok)
    ctrl = np.ascontiguousarray(np.stack(acc_ctrl))
    guide_off = np.zeros(B * N + 1, dtype=np.int32)
    guide_off[1:] = np.cumsum(np.concatenate(acc_cnt))
    guide_pv = np.ascontiguousarray(np.concatenate(acc_pv)) if guide_off[-1] else np.zeros((0, 6))
    guide_unk = lookup(world, guide_pv[:, :3], 1).astype(np.uint8) if len(guide_pv) else np.zeros(0, dtype=np.uint8)
    return Batch(ctrl, guide_off, guide_pv, guide_unk, None, None, None,
                 {"seed": seed, "K": K, "n_seg": np.array(acc_seg), "candidates": tried, "guides": "host pipeline"})


def reguide_batch(world: World, b: Batch, ctrl_now: np.ndarray) -> Batch:
    """the batch as the rebound loop sees it after one more re-guide on the control points ctrl_now (the optimizer's
    output so far): findCollisionSeg -> A* -> assignGuidePointsSemiCircle on them (bsplineTraj.cpp:640-648) APPENDS pairs
    to the lists; the next optimize() starts from ctrl_now with the longer lists.  A* failures append nothing."""
    _, status, n_seg, goff2, gpv2 = host_guides(world, b.N, ctrl=ctrl_now)
    goff, gpv = append_guides(b.guide_off, b.guide_pv, goff2, gpv2)
    unk = lookup(world, gpv[:, :3], 1).astype(np.uint8) if len(gpv) else np.zeros(0, dtype=np.uint8)
    meta = dict(b.meta)
    meta["reguides"] = meta.get("reguides", 0) + 1
    meta["n_seg_reguide"] = n_seg
    return Batch(np.ascontiguousarray(ctrl_now), goff, np.ascontiguousarray(gpv), unk, b.obs_off, b.obs, b.weights, meta)


def pairs_histogram(b: Batch):
    """(pairs per free control point histogram, guide pairs per trajectory, share of trajectories with any pair)"""
    cnt = np.diff(b.guide_off).reshape(b.B, b.N)
    per_traj = cnt.sum(1)
    return np.bincount(cnt[:, 3:b.N - 3].reshape(-1)), per_traj, float((per_traj > 0).mean())
