"""-m "not gpu": the C-ABI library loads here (no GPU) and exports every symbol include/vigo.h
declares; host-only entry points behave."""
import ctypes as C
import os
import re

import numpy as np

from trajectory_planner_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vigo.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vigo_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"libvigo_hip.so does not export {n}"
    # and the python prototype table covers the header
    assert set(names) == set(_lib.PROTOTYPES), set(names) ^ set(_lib.PROTOTYPES)


def test_default_params_match_reference_cfg():
    p = _lib.VigoParams()
    _lib.load().vigo_default_params(C.byref(p))
    # cfg/bspline_interactive/bspline_planner_param.yaml:4-19, bsplineTraj.cpp:697-699, lbfgs.hpp:942-954
    assert (p.dthresh, p.dist_thresh_dynamic, p.ts, p.ts_ctrl) == (0.5, 0.5, 0.1, 0.2)
    assert (p.w_distance, p.w_smoothness, p.w_feasibility, p.w_dynamic) == (1.0, 1.0, 1.0, 1.0)
    assert p.plan_in_z == 0 and p.uncertain_factor == 1.0 and p.pred_horizon == 2.0
    assert (p.mem_size, p.max_iterations, p.max_linesearch, p.past) == (16, 200, 40, 0)
    assert (p.g_epsilon, p.f_dec_coeff, p.s_curv_coeff, p.xtol) == (0.01, 1e-4, 0.9, 1e-16)
    assert (p.min_step, p.max_step) == (1e-20, 1e20)


def test_params_struct_layout_matches_oracle(olib):
    a = _lib.VigoParams()
    _lib.load().vigo_default_params(C.byref(a))
    b = olib.default_params()
    assert bytes(a) == bytes(b)


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        return
    h = C.c_void_p()
    rc = _lib.load().vigo_create(C.byref(h), 0)
    assert rc == -2 and not h.value  # VIGO_ERR_NO_DEVICE, no CPU fallback


def test_packed_bytes():
    lib = _lib.load()
    assert lib.vigo_grid_packed_bytes(256, 256, 256) == 3 * 256 * 256 * 8 * 4
    assert lib.vigo_grid_packed_bytes(219, 205, 41) == 3 * 219 * 205 * 2 * 4
    assert lib.vigo_grid_packed_bytes(0, 1, 1) == 0


def test_accumulated_time_equals_the_reference_loop():
    """vigo_accumulated_time(d, k) == k-fold `t += d` (polyTrajSolver.cpp:1129), bit for bit."""
    lib = _lib.load()
    rng = np.random.default_rng(7)
    ds = list(rng.uniform(1e-5, 2.0, 40)) + [0.1, 0.05, 0.025, 1.0 / 3, 2.0 ** -7, 3 * 2.0 ** -9, 1e-4, 4.56e-4]
    for d in ds:
        d = float(d)
        ks = sorted(set(int(k) for k in rng.integers(0, 20000, 6)) | {0, 1, 2, 3, 10000})
        t, k_done = 0.0, 0
        for k in ks:
            for _ in range(k - k_done):
                t = t + d
            k_done = k
            assert lib.vigo_accumulated_time(d, k) == t, (d, k)
    assert lib.vigo_accumulated_time(0.0, 5) == 0.0


def test_header_is_plain_c(tmp_path):
    """include/vigo.h must compile as C99 and as C++14 (the reference's standard, CMakeLists.txt:6) on its own:
    no C++ or torch types at the boundary"""
    import subprocess
    src = tmp_path / "use_vigo.c"
    src.write_text('#include "vigo.h"\nint main(void) { vigo_params_t p; vigo_default_params(&p); return (int)sizeof(vigo_handle_t) == 0; }\n')
    inc = os.path.join(ROOT, "include")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I", inc, str(src)], check=True)
    subprocess.run(["g++", "-std=c++14", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", "-I", inc, str(src)], check=True)
    # and it links: every declared entry point resolves against the built library (no GPU needed to link)
    lib_dir = os.path.join(ROOT, "trajectory_planner_amd", "lib")
    subprocess.run(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(tmp_path / "use_vigo"), "-L", lib_dir, "-lvigo_hip",
                    "-Wl,-rpath," + lib_dir], check=True)
    assert subprocess.run([str(tmp_path / "use_vigo")]).returncode == 0
