"""Thin Python host side over the C ABI (include/vigo.h).

PyTorch is used for device memory, streams and torch.distributed only; every computation on
the hot path happens inside libvigo_hip.so.  All tensor arguments must be contiguous CUDA
(= HIP) tensors of the dtype the header states; nothing here touches the CPU oracle.
"""
import ctypes as C
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from ._lib import VigoParams

PREC_F64 = 0
PREC_F32 = 1
PREC_F64_FAST = 2

# lbfgs.hpp:20-80 status codes worth naming
LBFGS_CONVERGENCE = 0
LBFGS_STOP = 1
LBFGS_ALREADY_MINIMIZED = 2
_LB_BASE = -1024
_LB_NAMES = [
    "LBFGSERR_UNKNOWNERROR", "LBFGSERR_LOGICERROR", "LBFGSERR_CANCELED", "LBFGSERR_INVALID_N",
    "LBFGSERR_INVALID_MEMSIZE", "LBFGSERR_INVALID_GEPSILON", "LBFGSERR_INVALID_TESTPERIOD",
    "LBFGSERR_INVALID_DELTA", "LBFGSERR_INVALID_MINSTEP", "LBFGSERR_INVALID_MAXSTEP",
    "LBFGSERR_INVALID_FDECCOEFF", "LBFGSERR_INVALID_SCURVCOEFF", "LBFGSERR_INVALID_XTOL",
    "LBFGSERR_INVALID_MAXLINESEARCH", "LBFGSERR_OUTOFINTERVAL", "LBFGSERR_INCORRECT_TMINMAX",
    "LBFGSERR_ROUNDING_ERROR", "LBFGSERR_MINIMUMSTEP", "LBFGSERR_MAXIMUMSTEP",
    "LBFGSERR_MAXIMUMLINESEARCH", "LBFGSERR_MAXIMUMITERATION", "LBFGSERR_WIDTHTOOSMALL",
    "LBFGSERR_INVALIDPARAMETERS", "LBFGSERR_INCREASEGRADIENT",
]
LBFGS_CODES = {name: _LB_BASE + i for i, name in enumerate(_LB_NAMES)}
LBFGSERR_ROUNDING_ERROR = LBFGS_CODES["LBFGSERR_ROUNDING_ERROR"]
LBFGSERR_MAXIMUMLINESEARCH = LBFGS_CODES["LBFGSERR_MAXIMUMLINESEARCH"]
LBFGSERR_MAXIMUMITERATION = LBFGS_CODES["LBFGSERR_MAXIMUMITERATION"]


class VigoError(RuntimeError):
    pass


def default_params() -> VigoParams:
    p = VigoParams()
    _lib.load().vigo_default_params(C.byref(p))
    return p


def _ptr(t: Optional[torch.Tensor], dtype, name: str, device=None):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if t.dtype != dtype:
        raise TypeError(f"{name}: dtype {t.dtype}, expected {dtype}")
    if not t.is_cuda:
        raise ValueError(f"{name}: must live on the GPU (the C ABI takes device pointers)")
    if device is not None and t.device != device:
        raise ValueError(f"{name}: on {t.device}, handle is bound to {device}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    return C.c_void_p(t.data_ptr())


def _shape(t: Optional[torch.Tensor], want, name: str):
    """the kernels index by these extents: a tensor of another shape is an out-of-range read on the device"""
    if t is None:
        return
    got = tuple(t.shape)
    if len(got) != len(want) or any(w is not None and g != w for g, w in zip(got, want)):
        raise ValueError(f"{name}: shape {got}, expected {tuple('*' if w is None else w for w in want)}")


@dataclass
class SolveResult:
    ctrl: torch.Tensor      # [B,N,3] optData_.controlPoints after optimize() (last evaluated point)
    x: torch.Tensor         # [B,N-6,3] the vector lbfgs_optimize returns
    status: torch.Tensor    # [B] int32, lbfgs.hpp:20-80
    fx: torch.Tensor        # [B]
    iters: torch.Tensor     # [B] int32
    evals: torch.Tensor     # [B] int32


class Vigo:
    """One vigo_handle_t bound to one GPU (one per process/rank)."""

    def __init__(self, device: int = 0, params: Optional[VigoParams] = None, precision: int = PREC_F64):
        self._lib = _lib.load()
        if not torch.cuda.is_available():
            raise VigoError("no HIP device visible: the ViGO hot path has no CPU fallback")
        self.device = torch.device("cuda", device)
        h = C.c_void_p()
        rc = self._lib.vigo_create(C.byref(h), device)
        if rc != 0:
            raise VigoError(f"vigo_create failed with {rc}")
        self._h = h
        self.params = params if params is not None else default_params()
        self.set_params(self.params)
        self.set_precision(precision)
        self._grid_meta = None

    # ---- lifecycle -------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.vigo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            msg = self._lib.vigo_last_error(self._h)
            raise VigoError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def set_params(self, p: VigoParams):
        self._check(self._lib.vigo_set_params(self._h, C.byref(p)), "vigo_set_params")
        self.params = p

    def set_precision(self, precision: int):
        self._check(self._lib.vigo_set_precision(self._h, precision), "vigo_set_precision")
        self.precision = precision

    def use_current_stream(self):
        """Bind launches to torch's current stream on this device."""
        s = torch.cuda.current_stream(self.device)
        self._check(self._lib.vigo_set_stream(self._h, C.c_void_p(s.cuda_stream)), "vigo_set_stream")

    # ---- voxel map ---------------------------------------------------------------------
    def set_grid(self, voxels: torch.Tensor, origin, res: float):
        """voxels: uint8 [nx,ny,nz] on the GPU (bit0 inflated-occupied, bit1 unknown, bit2 occupied)."""
        nx, ny, nz = voxels.shape
        o = (C.c_double * 3)(*origin)
        self._check(self._lib.vigo_set_grid(self._h, nx, ny, nz, o, float(res),
                                            _ptr(voxels, torch.uint8, "voxels", self.device)), "vigo_set_grid")
        self._grid_meta = (nx, ny, nz, tuple(float(v) for v in origin), float(res))

    def inflate_grid(self, voxels: torch.Tensor, rx: int, ry: int, rz: int) -> torch.Tensor:
        """in place: bit0 := OR of bit2 over the (2rx+1)(2ry+1)(2rz+1) voxel box"""
        nx, ny, nz = voxels.shape
        self._check(self._lib.vigo_inflate_grid(self._h, nx, ny, nz, _ptr(voxels, torch.uint8, "voxels", self.device), rx, ry, rz),
                    "vigo_inflate_grid")
        return voxels

    def pack_grid(self, voxels: torch.Tensor) -> torch.Tensor:
        """Pack a byte grid into the snapshot format (int32 tensor) for an RCCL broadcast."""
        nx, ny, nz = voxels.shape
        nbytes = self._lib.vigo_grid_packed_bytes(nx, ny, nz)
        packed = torch.empty(nbytes // 4, dtype=torch.int32, device=self.device)
        self._check(self._lib.vigo_pack_grid(self._h, nx, ny, nz, _ptr(voxels, torch.uint8, "voxels", self.device),
                                             C.c_void_p(packed.data_ptr())), "vigo_pack_grid")
        return packed

    def set_grid_packed(self, packed: torch.Tensor, dims, origin, res: float):
        nx, ny, nz = dims
        if packed.numel() * 4 != self._lib.vigo_grid_packed_bytes(nx, ny, nz):
            raise ValueError("packed grid has the wrong size for these dims")
        o = (C.c_double * 3)(*origin)
        self._check(self._lib.vigo_set_grid_packed(self._h, nx, ny, nz, o, float(res),
                                                   _ptr(packed, torch.int32, "packed", self.device)),
                    "vigo_set_grid_packed")
        self._grid_meta = (nx, ny, nz, tuple(float(v) for v in origin), float(res))

    def set_metric_bounds(self, bmin, bmax):
        self._check(self._lib.vigo_set_metric_bounds(self._h, (C.c_double * 3)(*bmin), (C.c_double * 3)(*bmax)),
                    "vigo_set_metric_bounds")

    def query_points(self, pts: torch.Tensor, which: int = 0) -> torch.Tensor:
        _shape(pts, (None, 3), "pts")
        q = pts.shape[0]
        out = torch.empty(q, dtype=torch.uint8, device=self.device)
        self._check(self._lib.vigo_query_points(self._h, which, q, _ptr(pts, torch.float64, "pts", self.device),
                                                C.c_void_p(out.data_ptr())), "vigo_query_points")
        return out

    def guides_unknown(self, guide_pv: torch.Tensor) -> torch.Tensor:
        _shape(guide_pv, (None, 6), "guide_pv")
        g = guide_pv.shape[0]
        out = torch.empty(g, dtype=torch.uint8, device=self.device)
        if g:
            self._check(self._lib.vigo_guides_unknown(self._h, g, _ptr(guide_pv, torch.float64, "guide_pv", self.device),
                                                      C.c_void_p(out.data_ptr())), "vigo_guides_unknown")
        return out

    # ---- cost / solve ------------------------------------------------------------------
    def _solve_ptrs(self, ctrl, guide_off, guide_pv, guide_unk, obs_off, obs, weights):
        B, N, three = ctrl.shape
        if three != 3:
            raise ValueError("ctrl must be [B,N,3]")
        if guide_off is not None and guide_off.numel() != B * N + 1:
            raise ValueError("guide_off must have B*N+1 entries")
        if obs_off is not None and obs_off.numel() != B + 1:
            raise ValueError("obs_off must have B+1 entries")
        if weights is not None and tuple(weights.shape) != (B, 4):
            raise ValueError("weights must be [B,4]")
        _shape(guide_pv, (None, 6), "guide_pv")
        _shape(obs, (None, 9), "obs")
        if guide_unk is not None and guide_pv is not None and guide_unk.numel() != guide_pv.shape[0]:
            raise ValueError("guide_unk must have one entry per guide pair")
        n_shared = 0
        if obs is not None and obs_off is None:
            n_shared = obs.shape[0]
        d = self.device
        return (B, N, _ptr(ctrl, torch.float64, "ctrl", d), _ptr(guide_off, torch.int32, "guide_off", d),
                _ptr(guide_pv, torch.float64, "guide_pv", d), _ptr(guide_unk, torch.uint8, "guide_unk", d),
                _ptr(obs_off, torch.int32, "obs_off", d), _ptr(obs, torch.float64, "obs", d), n_shared,
                _ptr(weights, torch.float64, "weights", d))

    def check_lists(self, B: int, N: int, guide_off=None, n_pairs: int = 0, obs_off=None, n_obs: int = 0) -> int:
        """vigo_check_lists: number of CSR violations in guide_off [B*N+1] / obs_off [B+1] (0 = safe to pass)."""
        if guide_off is not None and guide_off.numel() != B * N + 1:
            raise ValueError("guide_off must have B*N+1 entries")
        if obs_off is not None and obs_off.numel() != B + 1:
            raise ValueError("obs_off must have B+1 entries")
        rc = self._lib.vigo_check_lists(self._h, B, N, _ptr(guide_off, torch.int32, "guide_off", self.device), int(n_pairs),
                                        _ptr(obs_off, torch.int32, "obs_off", self.device), int(n_obs))
        if rc < 0:
            self._check(rc, "vigo_check_lists")
        return rc

    def cost_grad(self, ctrl, guide_off=None, guide_pv=None, guide_unk=None, obs_off=None, obs=None,
                  weights=None, want_terms=True):
        """vigo_cost_grad: returns (cost[B], grad[B,N-6,3], terms[B,4] or None)."""
        B, N, pc, po, ppv, pu, poo, pob, ns, pw = self._solve_ptrs(ctrl, guide_off, guide_pv, guide_unk,
                                                                    obs_off, obs, weights)
        cost = torch.empty(B, dtype=torch.float64, device=self.device)
        grad = torch.empty(B, max(N - 6, 0), 3, dtype=torch.float64, device=self.device)
        terms = torch.empty(B, 4, dtype=torch.float64, device=self.device) if want_terms else None
        self._check(self._lib.vigo_cost_grad(self._h, B, N, pc, po, ppv, pu, poo, pob, ns, pw,
                                             C.c_void_p(cost.data_ptr()), C.c_void_p(grad.data_ptr()),
                                             C.c_void_p(terms.data_ptr()) if want_terms else None), "vigo_cost_grad")
        return cost, grad, terms

    def optimize(self, ctrl, guide_off=None, guide_pv=None, guide_unk=None, obs_off=None, obs=None,
                 weights=None, inplace=False, out: Optional[SolveResult] = None) -> SolveResult:
        """vigo_optimize.  ctrl is cloned unless inplace=True (the C ABI updates it in place)."""
        _shape(ctrl, (None, None, 3), "ctrl")
        if ctrl.shape[1] < 7:
            raise VigoError(f"vigo_optimize needs N >= 7 control points (got {ctrl.shape[1]}): VIGO_ERR_UNSUPPORTED_N")
        work = ctrl if inplace else ctrl.clone()
        B, N, pc, po, ppv, pu, poo, pob, ns, pw = self._solve_ptrs(work, guide_off, guide_pv, guide_unk,
                                                                    obs_off, obs, weights)
        d = self.device
        if out is not None:
            # a caller-owned result is written by the kernel: wrong extents are out-of-range device writes
            _shape(out.x, (B, N - 6, 3), "out.x")
            for name in ("status", "iters", "evals", "fx"):
                _shape(getattr(out, name), (B,), f"out.{name}")
            _ptr(out.x, torch.float64, "out.x", d)
            _ptr(out.fx, torch.float64, "out.fx", d)
            for name in ("status", "iters", "evals"):
                _ptr(getattr(out, name), torch.int32, f"out.{name}", d)
        if out is None:
            out = SolveResult(
                ctrl=work,
                x=torch.empty(B, N - 6, 3, dtype=torch.float64, device=self.device),
                status=torch.empty(B, dtype=torch.int32, device=self.device),
                fx=torch.empty(B, dtype=torch.float64, device=self.device),
                iters=torch.empty(B, dtype=torch.int32, device=self.device),
                evals=torch.empty(B, dtype=torch.int32, device=self.device),
            )
        else:
            out.ctrl = work
        self._check(self._lib.vigo_optimize(self._h, B, N, pc, po, ppv, pu, poo, pob, ns, pw,
                                            C.c_void_p(out.x.data_ptr()), C.c_void_p(out.status.data_ptr()),
                                            C.c_void_p(out.fx.data_ptr()), C.c_void_p(out.iters.data_ptr()),
                                            C.c_void_p(out.evals.data_ptr())), "vigo_optimize")
        return out

    # ---- the rebound loop between two A* calls ---------------------------------------------
    # vigo_rebound_state_t as int32 columns: status, solve_first, fail_count, gate_static, gate_dynamic, rounds,
    # lbfgs_status, n_seg, then 2 * VIGO_MAX_COLLISION_SEGS segment indices
    REBOUND_MAX_SEGS = 48
    REBOUND_STATE_INTS = 8 + 2 * 48
    RB_ACTIVE, RB_DONE, RB_NEEDS_HOST = 0, 1, 2

    def rebound_rounds(self, ctrl, guide_off, guide_pv, guide_unk, obs_off, obs, weights, gate_dt, state,
                       max_rounds=4, not_check_ratio=0.0):
        """vigo_rebound_rounds: ctrl [B,N,3], weights [B,4] and state [B, REBOUND_STATE_INTS] (int32) are updated in place."""
        B, N, pc, po, ppv, pu, poo, pob, ns, pw = self._solve_ptrs(ctrl, guide_off, guide_pv, guide_unk, obs_off, obs, weights)
        if weights is None:
            raise ValueError("weights [B,4] are required (the loop doubles them per trajectory)")
        _shape(state, (B, self.REBOUND_STATE_INTS), "state")
        self._check(self._lib.vigo_rebound_rounds(self._h, B, N, pc, po, ppv, pu, poo, pob, ns, pw, float(gate_dt),
                                                  float(not_check_ratio), int(max_rounds),
                                                  _ptr(state, torch.int32, "state", self.device)), "vigo_rebound_rounds")

    # ---- spline fit, evaluation and gates -----------------------------------------------
    def bspline_fit(self, points, conds=None, ts=None):
        """bspline::parameterizeToBspline for a batch: points [B,K,3] (+ conds [B,4,3]) -> ctrl [B,K+2,3]"""
        _shape(points, (None, None, 3), "points")
        B, K, _ = points.shape
        _shape(conds, (B, 4, 3), "conds")
        ts = float(self.params.ts_ctrl if ts is None else ts)
        out = torch.empty(B, K + 2, 3, dtype=torch.float64, device=self.device)
        self._check(self._lib.vigo_bspline_fit(self._h, B, K, ts, _ptr(points, torch.float64, "points", self.device),
                                               _ptr(conds, torch.float64, "conds", self.device),
                                               C.c_void_p(out.data_ptr())), "vigo_bspline_fit")
        return out

    def bspline_eval(self, ctrl, times, deriv=0):
        _shape(ctrl, (None, None, 3), "ctrl")
        B, N, _ = ctrl.shape
        T = times.numel()
        out = torch.empty(B, T, 3, dtype=torch.float64, device=self.device)
        self._check(self._lib.vigo_bspline_eval(self._h, B, N, _ptr(ctrl, torch.float64, "ctrl", self.device), deriv, T,
                                                _ptr(times, torch.float64, "times", self.device),
                                                C.c_void_p(out.data_ptr())), "vigo_bspline_eval")
        return out

    def traj_collision(self, ctrl, dt):
        _shape(ctrl, (None, None, 3), "ctrl")
        B, N, _ = ctrl.shape
        flag = torch.empty(B, dtype=torch.uint8, device=self.device)
        first = torch.empty(B, dtype=torch.int32, device=self.device)
        self._check(self._lib.vigo_traj_collision(self._h, B, N, _ptr(ctrl, torch.float64, "ctrl", self.device),
                                                  float(dt), C.c_void_p(flag.data_ptr()),
                                                  C.c_void_p(first.data_ptr())), "vigo_traj_collision")
        return flag, first

    def traj_dynamic_collision(self, ctrl, dt, obs_off=None, obs=None):
        _shape(ctrl, (None, None, 3), "ctrl")
        B, N, _ = ctrl.shape
        _shape(obs, (None, 9), "obs")
        if obs_off is not None and obs_off.numel() != B + 1:
            raise ValueError("obs_off must have B+1 entries")
        flag = torch.empty(B, dtype=torch.uint8, device=self.device)
        ns = obs.shape[0] if (obs is not None and obs_off is None) else 0
        self._check(self._lib.vigo_traj_dynamic_collision(
            self._h, B, N, _ptr(ctrl, torch.float64, "ctrl", self.device), float(dt),
            _ptr(obs_off, torch.int32, "obs_off", self.device), _ptr(obs, torch.float64, "obs", self.device), ns,
            C.c_void_p(flag.data_ptr())), "vigo_traj_dynamic_collision")
        return flag

    def ctrl_occupancy(self, ctrl):
        _shape(ctrl, (None, None, 3), "ctrl")
        B, N, _ = ctrl.shape
        pt = torch.empty(B, N, dtype=torch.uint8, device=self.device)
        line = torch.empty(B, N, dtype=torch.uint8, device=self.device)
        self._check(self._lib.vigo_ctrl_occupancy(self._h, B, N, _ptr(ctrl, torch.float64, "ctrl", self.device),
                                                  C.c_void_p(pt.data_ptr()), C.c_void_p(line.data_ptr())),
                    "vigo_ctrl_occupancy")
        return pt, line

    # ---- corridor checker ---------------------------------------------------------------
    def minsnap(self, waypoints, corridor=None, conds=None, deg=7, diff=4, cont=4, vel=1.0, corridor_res=8.0):
        """polyTrajSolver::solve for a batch of paths: waypoints [T,W,3] (+ corridor [T,W-1], conds [T,4,3])
        -> (coeffs [T,W-1,3,deg+1], knots [T,W], status [T])"""
        _shape(waypoints, (None, None, 3), "waypoints")
        T, W, _ = waypoints.shape
        _shape(corridor, (T, W - 1), "corridor")
        _shape(conds, (T, 4, 3), "conds")
        coeffs = torch.zeros(T, W - 1, 3, deg + 1, dtype=torch.float64, device=self.device)
        knots = torch.zeros(T, W, dtype=torch.float64, device=self.device)
        status = torch.full((T,), -99, dtype=torch.int32, device=self.device)
        self._check(self._lib.vigo_minsnap(self._h, T, W, deg, diff, cont, float(vel), float(corridor_res),
                                           _ptr(waypoints, torch.float64, "waypoints", self.device),
                                           _ptr(corridor, torch.float64, "corridor", self.device),
                                           _ptr(conds, torch.float64, "conds", self.device),
                                           C.c_void_p(coeffs.data_ptr()), C.c_void_p(knots.data_ptr()),
                                           C.c_void_p(status.data_ptr())), "vigo_minsnap")
        return coeffs, knots, status

    def corridor_check(self, coeffs, n_samp, delT, box, map_res):
        """coeffs [S,3,deg+1] f64, n_samp [S] i32, delT [S] f64 -> (flag u8[S], first i32[S], count i32[S])."""
        _shape(coeffs, (None, 3, None), "coeffs")
        S, three, d1 = coeffs.shape
        _shape(n_samp, (S,), "n_samp")
        _shape(delT, (S,), "delT")
        flag = torch.empty(S, dtype=torch.uint8, device=self.device)
        first = torch.empty(S, dtype=torch.int32, device=self.device)
        count = torch.empty(S, dtype=torch.int32, device=self.device)
        self._check(self._lib.vigo_corridor_check(
            self._h, S, d1 - 1, _ptr(coeffs, torch.float64, "coeffs", self.device),
            _ptr(n_samp, torch.int32, "n_samp", self.device), _ptr(delT, torch.float64, "delT", self.device),
            (C.c_double * 3)(*box), float(map_res), C.c_void_p(flag.data_ptr()), C.c_void_p(first.data_ptr()),
            C.c_void_p(count.data_ptr())), "vigo_corridor_check")
        return flag, first, count

    def poly_sample(self, coeffs, n_samp, delT, stride, want_f64=True, want_f32=False):
        """vigo_poly_sample: positions of polyTrajSolver::getTrajectory for S segments ->
        (pos f64 [S,stride,3] or None, pos f32 [S,stride,3] or None); rows k >= n_samp[s] are left untouched (zero)."""
        _shape(coeffs, (None, 3, None), "coeffs")
        S, _, d1 = coeffs.shape
        _shape(n_samp, (S,), "n_samp")
        _shape(delT, (S,), "delT")
        p64 = torch.zeros(S, stride, 3, dtype=torch.float64, device=self.device) if want_f64 else None
        p32 = torch.zeros(S, stride, 3, dtype=torch.float32, device=self.device) if want_f32 else None
        self._check(self._lib.vigo_poly_sample(
            self._h, S, d1 - 1, _ptr(coeffs, torch.float64, "coeffs", self.device),
            _ptr(n_samp, torch.int32, "n_samp", self.device), _ptr(delT, torch.float64, "delT", self.device), int(stride),
            C.c_void_p(p64.data_ptr()) if want_f64 else None, C.c_void_p(p32.data_ptr()) if want_f32 else None), "vigo_poly_sample")
        return p64, p32

    def box_collision_points(self, pts, box, map_res):
        """vigo_box_collision_points: pts [M,3] f64 -> uint8 [M]"""
        _shape(pts, (None, 3), "pts")
        M = pts.shape[0]
        out = torch.empty(M, dtype=torch.uint8, device=self.device)
        self._check(self._lib.vigo_box_collision_points(self._h, M, _ptr(pts, torch.float64, "pts", self.device),
                                                        (C.c_double * 3)(*box), float(map_res),
                                                        C.c_void_p(out.data_ptr())), "vigo_box_collision_points")
        return out

    # ---- ESDF ---------------------------------------------------------------------------
    def set_esdf(self, dist: torch.Tensor, origin, res: float):
        nx, ny, nz = dist.shape
        self._check(self._lib.vigo_set_esdf(self._h, nx, ny, nz, (C.c_double * 3)(*origin), float(res),
                                            _ptr(dist, torch.float32, "dist", self.device)), "vigo_set_esdf")

    def esdf_query(self, pts: torch.Tensor):
        _shape(pts, (None, 3), "pts")
        q = pts.shape[0]
        d = torch.empty(q, dtype=torch.float64, device=self.device)
        g = torch.empty(q, 3, dtype=torch.float64, device=self.device)
        self._check(self._lib.vigo_esdf_query(self._h, q, _ptr(pts, torch.float64, "pts", self.device),
                                              C.c_void_p(d.data_ptr()), C.c_void_p(g.data_ptr())), "vigo_esdf_query")
        return d, g

    def esdf_query_f32(self, pts: torch.Tensor, out: Optional[torch.Tensor] = None):
        """vigo_esdf_query_f32: pts float32 [Q,3] -> float32 [Q,4] = (distance, gradient)"""
        _shape(pts, (None, 3), "pts")
        q = pts.shape[0]
        if out is None:
            out = torch.empty(q, 4, dtype=torch.float32, device=self.device)
        _shape(out, (q, 4), "out")
        self._check(self._lib.vigo_esdf_query_f32(self._h, q, _ptr(pts, torch.float32, "pts", self.device),
                                                  _ptr(out, torch.float32, "out", self.device)), "vigo_esdf_query_f32")
        return out
