"""ctypes binding of the C ABI declared in include/vigo.h.

The library is built in-tree by trajectory_planner_amd/csrc/Makefile (see
__graft_entry__.build()).  There is no CPU fallback: if the shared object is missing or does
not load, importing the product path raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libvigo_hip.so")


class VigoParams(C.Structure):
    """vigo_params_t (include/vigo.h)."""

    _fields_ = [
        ("dthresh", C.c_double),
        ("dist_thresh_dynamic", C.c_double),
        ("ts_ctrl", C.c_double),
        ("ts", C.c_double),
        ("pred_horizon", C.c_double),
        ("uncertain_factor", C.c_double),
        ("w_distance", C.c_double),
        ("w_smoothness", C.c_double),
        ("w_feasibility", C.c_double),
        ("w_dynamic", C.c_double),
        ("min_height", C.c_double),
        ("max_height", C.c_double),
        ("plan_in_z", C.c_int32),
        ("mem_size", C.c_int32),
        ("max_iterations", C.c_int32),
        ("max_linesearch", C.c_int32),
        ("past", C.c_int32),
        ("strict_z", C.c_int32),
        ("g_epsilon", C.c_double),
        ("delta", C.c_double),
        ("min_step", C.c_double),
        ("max_step", C.c_double),
        ("f_dec_coeff", C.c_double),
        ("s_curv_coeff", C.c_double),
        ("xtol", C.c_double),
    ]


_vp = C.c_void_p
_i = C.c_int
_i64 = C.c_int64
_d = C.c_double
_d3 = C.POINTER(C.c_double)

# name -> (restype, argtypes); every symbol include/vigo.h declares
PROTOTYPES = {
    "vigo_create": (_i, [C.POINTER(_vp), _i]),
    "vigo_destroy": (_i, [_vp]),
    "vigo_set_stream": (_i, [_vp, _vp]),
    "vigo_default_params": (None, [C.POINTER(VigoParams)]),
    "vigo_set_params": (_i, [_vp, C.POINTER(VigoParams)]),
    "vigo_get_params": (_i, [_vp, C.POINTER(VigoParams)]),
    "vigo_set_precision": (_i, [_vp, _i]),
    "vigo_last_error": (C.c_char_p, [_vp]),
    "vigo_abi_version": (_i, []),
    "vigo_build_arch": (C.c_char_p, []),
    "vigo_build_id": (C.c_char_p, []),
    "vigo_set_grid": (_i, [_vp, _i, _i, _i, _d3, _d, _vp]),
    "vigo_set_grid_host": (_i, [_vp, _i, _i, _i, _d3, _d, _vp]),
    "vigo_inflate_grid": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i]),
    "vigo_grid_packed_bytes": (C.c_size_t, [_i, _i, _i]),
    "vigo_pack_grid": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "vigo_set_grid_packed": (_i, [_vp, _i, _i, _i, _d3, _d, _vp]),
    "vigo_set_metric_bounds": (_i, [_vp, _d3, _d3]),
    "vigo_query_points": (_i, [_vp, _i, _i64, _vp, _vp]),
    "vigo_guides_unknown": (_i, [_vp, _i64, _vp, _vp]),
    "vigo_check_lists": (_i, [_vp, _i, _i, _vp, _i64, _vp, _i64]),
    "vigo_cost_grad": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "vigo_optimize": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vigo_rebound_rounds": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _d, _d, _i, _vp]),
    "vigo_bspline_fit": (_i, [_vp, _i, _i, _d, _vp, _vp, _vp]),
    "vigo_bspline_eval": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp]),
    "vigo_traj_collision": (_i, [_vp, _i, _i, _vp, _d, _vp, _vp]),
    "vigo_traj_dynamic_collision": (_i, [_vp, _i, _i, _vp, _d, _vp, _vp, _i, _vp]),
    "vigo_ctrl_occupancy": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "vigo_minsnap": (_i, [_vp, _i, _i, _i, _i, _i, _d, _d, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vigo_corridor_check": (_i, [_vp, _i, _i, _vp, _vp, _vp, _d3, _d, _vp, _vp, _vp]),
    "vigo_box_collision_points": (_i, [_vp, _i64, _vp, _d3, _d, _vp]),
    "vigo_poly_sample": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "vigo_accumulated_time": (_d, [_d, _i64]),
    "vigo_clock_table_time": (_d, [_d, _i64, _i64]),
    "vigo_exact_pow": (_d, [_d, _i]),
    "vigo_exact_pow_dd": (_d, [_d, _i, C.POINTER(_i)]),
    "vigo_exact_pow_integer": (_d, [_d, _i]),
    "vigo_set_esdf": (_i, [_vp, _i, _i, _i, _d3, _d, _vp]),
    "vigo_esdf_query": (_i, [_vp, _i64, _vp, _vp, _vp]),
    "vigo_esdf_query_f32": (_i, [_vp, _i64, _vp, _vp]),
}

_lib = None


def load():
    """Load libvigo_hip.so (once) and attach prototypes.  Raises if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C trajectory_planner_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback for the hot path"
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
