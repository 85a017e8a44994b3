"""ctypes bindings of the CPU oracle (oracle/libvigo_oracle.so) and of the verbatim-reference
L-BFGS shim (oracle/_ref/libref_lbfgs.so).  TEST INFRASTRUCTURE: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libvigo_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libref_lbfgs.so")

import sys
sys.path.insert(0, ROOT)
from trajectory_planner_amd._lib import VigoParams  # noqa: E402  (struct layout only)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_up = C.POINTER(C.c_uint8)

EVAL_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, _dp, _dp, C.c_int)
TRACE_FN = C.CFUNCTYPE(None, C.c_void_p, _dp, _dp, C.c_double, C.c_double, C.c_int)


class Grid(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int), ("origin", C.c_double * 3),
                ("res", C.c_double), ("vox", C.c_void_p), ("bmin", C.c_double * 3), ("bmax", C.c_double * 3)]


def build():
    subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)


_oracle = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            build()
        L = C.CDLL(ORACLE_SO)
        L.vgo_default_params.argtypes = [C.POINTER(VigoParams)]
        L.vgo_set_emulation.argtypes = [C.c_int]
        L.vgo_set_emulation2.argtypes = [C.c_int, C.c_int]
        L.vgo_get_emulation.restype = C.c_int
        L.vgo_cost_grad.restype = C.c_double
        L.vgo_cost_grad.argtypes = [C.POINTER(VigoParams), C.c_int, _dp, _ip, _dp, _up, C.c_int, _dp, _dp, _dp, _dp, _dp]
        L.vgo_lbfgs.restype = C.c_int
        L.vgo_lbfgs.argtypes = [C.c_int, _dp, _dp, EVAL_FN, C.c_void_p, C.POINTER(VigoParams), C.POINTER(C.c_int),
                                C.POINTER(C.c_int), TRACE_FN, C.c_void_p]
        L.vgo_optimize.restype = C.c_int
        L.vgo_cost_grad_batch.argtypes = [C.POINTER(VigoParams), C.c_int, C.c_int, _dp, _ip, _dp, _up, _ip, _dp, C.c_int,
                                          _dp, _dp, _dp, _dp]
        L.vgo_optimize_batch.argtypes = [C.POINTER(VigoParams), C.c_int, C.c_int, _dp, _ip, _dp, _up, _ip, _dp, C.c_int,
                                         _dp, _dp, _ip, _dp, _ip, _ip]
        L.vgo_bspline_at.argtypes = [C.c_int, C.c_int, _dp, C.c_double, C.c_double, _dp]
        L.vgo_traj_eval.argtypes = [C.c_int, _dp, C.c_double, C.c_int, C.c_double, _dp]
        L.vgo_sample_times.restype = C.c_int
        L.vgo_sample_times.argtypes = [C.c_double, C.c_double, _dp, C.c_int]
        L.vgo_grid_init.argtypes = [C.POINTER(Grid), C.c_int, C.c_int, C.c_int, _dp, C.c_double, C.c_void_p]
        for name in ("vgo_is_inflated_occupied", "vgo_is_unknown"):
            getattr(L, name).restype = C.c_int
            getattr(L, name).argtypes = [C.POINTER(Grid), _dp]
        L.vgo_is_inflated_occupied_line.restype = C.c_int
        L.vgo_is_inflated_occupied_line.argtypes = [C.POINTER(Grid), _dp, _dp]
        L.vgo_traj_collision.restype = C.c_int
        L.vgo_traj_collision.argtypes = [C.POINTER(Grid), C.c_int, _dp, C.c_double, C.c_double, C.POINTER(C.c_int)]
        L.vgo_traj_dynamic_collision.restype = C.c_int
        L.vgo_traj_dynamic_collision.argtypes = [C.c_int, _dp, C.c_double, C.c_double, C.c_int, _dp]
        L.vgo_ctrl_occupancy.argtypes = [C.POINTER(Grid), C.c_int, _dp, _up, _up]
        L.vgo_box_collision.restype = C.c_int
        L.vgo_box_collision.argtypes = [C.POINTER(Grid), C.c_float, C.c_float, C.c_float, _dp, C.c_double]
        L.vgo_poly_pos.argtypes = [C.c_int, _dp, _dp, _dp, C.c_double, _dp]
        L.vgo_corridor_check_segment.restype = C.c_int
        L.vgo_corridor_check_segment.argtypes = [C.POINTER(Grid), C.c_int, _dp, C.c_int, C.c_double, _dp, C.c_double,
                                                 C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.vgo_esdf_query.argtypes = [C.c_int, C.c_int, C.c_int, _dp, C.c_double, C.POINTER(C.c_float), _dp, _dp, _dp]
        L.vgo_esdf_query_batch.argtypes = [C.c_int, C.c_int, C.c_int, _dp, C.c_double, C.POINTER(C.c_float), C.c_int64, _dp, _dp, _dp]
        L.vgo_find_collision_seg.restype = C.c_int
        L.vgo_find_collision_seg.argtypes = [C.POINTER(Grid), C.c_int, _dp, C.c_double, _ip, C.c_int]
        L.vgo_rebound_decide.restype = C.c_int
        L.vgo_rebound_decide.argtypes = [C.POINTER(VigoParams), C.POINTER(Grid), C.c_int, _dp, _ip, _dp, C.c_int, _dp, C.c_double,
                                         C.c_double, _dp, _ip]
        L.vgo_set_pow_mode.argtypes = [C.c_int]
        L.vgo_pow_exact.restype = C.c_double
        L.vgo_pow_exact.argtypes = [C.c_double, C.c_int]
        L.vgo_poly_sample.argtypes = [C.c_int, C.c_int, _dp, _ip, _dp, C.c_int, _dp, C.POINTER(C.c_float)]
        L.vgo_corridor_check_batch.argtypes = [C.POINTER(Grid), C.c_int, C.c_int, _dp, _ip, _dp, _dp, C.c_double, _up, _ip, _ip]
        _oracle = L
    return _oracle


def ref():
    """The verbatim reference lbfgs.hpp shim, or None when it was never built."""
    global _ref
    if _ref is None:
        if not os.path.exists(REF_SO):
            if os.path.exists("/root/reference/include/trajectory_planner/solver/lbfgs.hpp"):
                build()
            if not os.path.exists(REF_SO):
                return None
        L = C.CDLL(REF_SO)
        L.ref_lbfgs_optimize.restype = C.c_int
        L.ref_lbfgs_optimize.argtypes = [C.c_int, _dp, _dp, EVAL_FN, C.c_void_p, _ip, _dp, C.POINTER(C.c_int), TRACE_FN,
                                         C.c_void_p]
        _ref = L
    return _ref


def default_params() -> VigoParams:
    p = VigoParams()
    oracle().vgo_default_params(C.byref(p))
    return p


def set_emulation(group: int, ppl: int = 1):
    oracle().vgo_set_emulation2(int(group), int(ppl))


def emulation_group(N: int) -> int:
    return 32 if N <= 32 else 64


def emulation_shape(N: int):
    """(lanes per trajectory, control points per lane) the HIP kernels use for N control points"""
    if N <= 32:
        return 32, 1
    if N <= 64:
        return 64, 1
    return (64, 2) if N <= 128 else (64, 4)


def _d(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _i(a):
    return a.ctypes.data_as(_ip) if a is not None else None


def _u(a):
    return a.ctypes.data_as(_up) if a is not None else None


def _c(a, dtype):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


def cost_grad_batch(P, batch, weights=None):
    """oracle vgo_cost_grad_batch over a synth.Batch -> (cost[B], grad[B,N-6,3], terms[B,4])"""
    B, N = batch.B, batch.N
    ctrl = _c(batch.ctrl, np.float64)
    goff = _c(batch.guide_off, np.int32)
    gpv = _c(batch.guide_pv, np.float64)
    gunk = _c(batch.guide_unk, np.uint8)
    ooff = _c(batch.obs_off, np.int32)
    obs = _c(batch.obs, np.float64)
    w = _c(weights if weights is not None else batch.weights, np.float64)
    ns = 0 if (obs is None or ooff is not None) else obs.shape[0]
    cost = np.zeros(B)
    grad = np.zeros((B, N - 6, 3))
    terms = np.zeros((B, 4))
    oracle().vgo_cost_grad_batch(C.byref(P), B, N, _d(ctrl), _i(goff), _d(gpv), _u(gunk), _i(ooff), _d(obs), ns, _d(w),
                                 _d(cost), _d(grad), _d(terms))
    return cost, grad, terms


def optimize_batch(P, batch, weights=None):
    """oracle vgo_optimize_batch -> dict(ctrl, x, status, fx, iters, evals)"""
    B, N = batch.B, batch.N
    ctrl = np.array(batch.ctrl, dtype=np.float64, order="C", copy=True)
    goff = _c(batch.guide_off, np.int32)
    gpv = _c(batch.guide_pv, np.float64)
    gunk = _c(batch.guide_unk, np.uint8)
    ooff = _c(batch.obs_off, np.int32)
    obs = _c(batch.obs, np.float64)
    w = _c(weights if weights is not None else batch.weights, np.float64)
    ns = 0 if (obs is None or ooff is not None) else obs.shape[0]
    x = np.zeros((B, N - 6, 3))
    status = np.zeros(B, dtype=np.int32)
    fx = np.zeros(B)
    iters = np.zeros(B, dtype=np.int32)
    evals = np.zeros(B, dtype=np.int32)
    oracle().vgo_optimize_batch(C.byref(P), B, N, _d(ctrl), _i(goff), _d(gpv), _u(gunk), _i(ooff), _d(obs), ns, _d(w),
                                _d(x), _i(status), _d(fx), _i(iters), _i(evals))
    return dict(ctrl=ctrl, x=x, status=status, fx=fx, iters=iters, evals=evals)


def bspline_fit_batch(points, ts, conds=None):
    """oracle vgo_bspline_fit_batch: points [B,K,3] (+ conds [B,4,3]) -> control points [B,K+2,3]"""
    pts = _c(points, np.float64)
    B, K, _ = pts.shape
    cd = _c(conds, np.float64)
    out = np.zeros((B, K + 2, 3))
    O = oracle()
    O.vgo_bspline_fit_batch.argtypes = [C.c_int, C.c_int, C.c_double, _dp, _dp, _dp]
    O.vgo_bspline_fit_batch.restype = None
    O.vgo_bspline_fit_batch(B, K, float(ts), _d(pts), _d(cd), _d(out))
    return out


class pow_mode:
    """context manager: pow(t, d) of the polynomial sampler as the correctly rounded power (True) or libm's (False)"""

    def __init__(self, exact=True):
        self.exact = exact

    def __enter__(self):
        self.prev = oracle().vgo_get_pow_mode()
        oracle().vgo_set_pow_mode(1 if self.exact else 0)

    def __exit__(self, *a):
        oracle().vgo_set_pow_mode(self.prev)


def poly_sample(coeffs, n_samp, delT, stride, f32=False):
    """oracle vgo_poly_sample: positions [S,stride,3] (float64, or float32 after pose2Octomap's cast)"""
    co = _c(coeffs, np.float64)
    S, _, d1 = co.shape
    ns, dl = _c(n_samp, np.int32), _c(delT, np.float64)
    out = np.zeros((S, stride, 3), dtype=np.float32 if f32 else np.float64)
    oracle().vgo_poly_sample(S, d1 - 1, _d(co), _i(ns), _d(dl), int(stride), None if f32 else _d(out),
                             out.ctypes.data_as(C.POINTER(C.c_float)) if f32 else None)
    return out


def corridor_check_batch(grid, coeffs, n_samp, delT, box, map_res):
    """oracle checker over S segments -> (flag u8[S], first i32[S], count i32[S])"""
    co = _c(coeffs, np.float64)
    S, _, d1 = co.shape
    ns, dl, bx = _c(n_samp, np.int32), _c(delT, np.float64), _c(box, np.float64)
    flag, first, count = np.zeros(S, dtype=np.uint8), np.zeros(S, dtype=np.int32), np.zeros(S, dtype=np.int32)
    oracle().vgo_corridor_check_batch(C.byref(grid), S, d1 - 1, _d(co), _i(ns), _d(dl), _d(bx), float(map_res), _u(flag),
                                      _i(first), _i(count))
    return flag, first, count


def esdf_query_batch(dist, origin, res, pts):
    """oracle trilinear ESDF over Q points -> (d [Q], grad [Q,3])"""
    ds = np.ascontiguousarray(dist, dtype=np.float32)
    pt = _c(pts, np.float64)
    org = _c(origin, np.float64)
    Q = pt.shape[0]
    d, g = np.zeros(Q), np.zeros((Q, 3))
    oracle().vgo_esdf_query_batch(ds.shape[0], ds.shape[1], ds.shape[2], _d(org), float(res),
                                  ds.ctypes.data_as(C.POINTER(C.c_float)), Q, _d(pt), _d(d), _d(g))
    return d, g


def esdf_query_f32_batch(dist, origin, res, pts):
    """oracle fp32 trilinear ESDF over Q float32 points -> float32 [Q,4] = (d, grad)"""
    ds = np.ascontiguousarray(dist, dtype=np.float32)
    pt = np.ascontiguousarray(pts, dtype=np.float32)
    org = _c(origin, np.float64)
    Q = pt.shape[0]
    out = np.zeros((Q, 4), dtype=np.float32)
    fp = C.POINTER(C.c_float)
    O = oracle()
    O.vgo_esdf_query_f32_batch.argtypes = [C.c_int, C.c_int, C.c_int, _dp, C.c_double, fp, C.c_int64, fp, fp]
    O.vgo_esdf_query_f32_batch.restype = None
    O.vgo_esdf_query_f32_batch(ds.shape[0], ds.shape[1], ds.shape[2], _d(org), float(res), ds.ctypes.data_as(fp), Q,
                               pt.ctypes.data_as(fp), out.ctypes.data_as(fp))
    return out


def make_grid(world):
    """vgo_grid_t over a synth.World (keeps the numpy array alive via the returned tuple)."""
    vox = np.ascontiguousarray(world.voxels, dtype=np.uint8)
    g = Grid()
    origin = np.ascontiguousarray(world.origin, dtype=np.float64)
    oracle().vgo_grid_init(C.byref(g), vox.shape[0], vox.shape[1], vox.shape[2], _d(origin), float(world.res),
                           vox.ctypes.data_as(C.c_void_p))
    return g, vox
